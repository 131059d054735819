// Small dense complex eigen-algebra on the host and the Krylov-Schur outer iteration built on it: what SLEPc's DS object
// and EPSSolve_KrylovSchur_Default do behind the reference's single call  eps.solve()  (Solver/utils.py:268-270).
//
// Everything here is O(ncv^3) work on the projected (ncv x ncv) matrix; the O(n) parts of the iteration (operator applies,
// orthogonalisation, basis updates) are the device kernels behind lsa_krylov_* (solver.hip).  In tree, no LAPACK: a consumer
// of the C-ABI gets eigenpairs from ONE call (lsa_eigs_sinvert) with nothing but this library loaded, and the result does
// not depend on which BLAS a process happens to carry.
//
//   schur_hessenberg   Householder reduction to Hessenberg form, then the single-shift QR algorithm with Wilkinson shifts,
//                      Schur vectors accumulated (complex arithmetic: the Schur form is upper triangular)
//   schur_swap/reorder adjacent swaps by one rotation each; wanted eigenvalues to the leading block, order kept
//   tri_eigenvectors   back substitution on the triangular factor
//   krylov_schur       Stewart's Krylov-Schur restart: expand to ncv vectors, Schur form with the wanted Ritz values first,
//                      residual estimates |b^H y|, relative convergence test (EPS_CONV_REL), truncate to nconv + (m - nconv)/2
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lsa_internal.h"

namespace {

struct Z {
    double re, im;
};
inline Z operator+(Z a, Z b) { return {a.re + b.re, a.im + b.im}; }
inline Z operator-(Z a, Z b) { return {a.re - b.re, a.im - b.im}; }
inline Z operator-(Z a) { return {-a.re, -a.im}; }
inline Z operator*(Z a, Z b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Z operator*(double a, Z b) { return {a * b.re, a * b.im}; }
inline Z conj(Z a) { return {a.re, -a.im}; }
inline double abs1(Z a) { return std::fabs(a.re) + std::fabs(a.im); }
inline double abs2(Z a) { return a.re * a.re + a.im * a.im; }
inline double zabs(Z a) { return std::hypot(a.re, a.im); }
inline Z zdiv(Z a, Z b) {  // Smith
    if (std::fabs(b.re) >= std::fabs(b.im)) {
        const double r = b.im / b.re, d = b.re + b.im * r;
        return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
    }
    const double r = b.re / b.im, d = b.re * r + b.im;
    return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
inline Z zsqrt(Z a) {
    const double m = zabs(a);
    if (m == 0.0) return {0.0, 0.0};
    double re = std::sqrt(0.5 * (m + std::fabs(a.re))), im = 0.5 * a.im / re;
    if (a.re < 0.0) {
        const double t = re;
        re = std::fabs(im);
        im = a.im < 0.0 ? -t : t;
    }
    return {re, im};
}

// rotation R = [c s; -conj(s) c] (c real) with R [f; g] = [r; 0]
inline void givens(Z f, Z g, double& c, Z& s, Z& r) {
    // both entries of ordinary size (LAPACK 3.10's zlartg, unscaled branch): one square root and one division on the chain
    // from one rotation of a QR sweep to the next
    const double f2 = abs2(f), g2 = abs2(g);
    if (f2 > 1e-200 && f2 < 1e200 && g2 > 1e-200 && g2 < 1e200) {
        const double h2 = f2 + g2, d = std::sqrt(f2 * h2), id = 1.0 / d;
        c = f2 * id;
        const Z fd = id * f;
        s = fd * conj(g);
        r = (h2 * id) * f;
        return;
    }
    const double ng = zabs(g);
    if (ng == 0.0) {
        c = 1.0;
        s = {0.0, 0.0};
        r = f;
        return;
    }
    const double nf = zabs(f);
    if (nf == 0.0) {
        c = 0.0;
        s = (1.0 / ng) * conj(g);
        r = {ng, 0.0};
        return;
    }
    const double nrm = std::hypot(nf, ng);
    const Z ph = (1.0 / nf) * f;
    c = nf / nrm;
    s = (1.0 / nrm) * (ph * conj(g));
    r = nrm * ph;
}

// column-major n x n views
struct Mat {
    Z* a;
    int ld;
    inline Z& operator()(int i, int j) const { return a[(size_t)j * ld + i]; }
};

// rows (p, q) of A, columns [j0, j1): [a_p; a_q] <- R [a_p; a_q]
inline void rot_rows(const Mat& A, int p, int q, int j0, int j1, double c, Z s) {
    const Z ms = -conj(s);
    for (int j = j0; j < j1; ++j) {
        const Z x = A(p, j), y = A(q, j);
        A(p, j) = c * x + s * y;
        A(q, j) = ms * x + c * y;
    }
}
// columns (p, q) of A, rows [i0, i1): [a_p a_q] <- [a_p a_q] R^H
inline void rot_cols(const Mat& A, int p, int q, int i0, int i1, double c, Z s) {
    Z* xp = &A(0, p);
    Z* xq = &A(0, q);
    const Z cs = conj(s);
    for (int i = i0; i < i1; ++i) {
        const Z x = xp[i], y = xq[i];
        xp[i] = c * x + cs * y;
        xq[i] = c * y - s * x;
    }
}

// A <- P^H A P (upper Hessenberg on return), Q <- Q P;  Q must hold a unitary matrix on entry (the identity, or a basis to carry along).
// Row-oriented, from the bottom: row r keeps only its entry in column r - 1 of those left of the diagonal, by a Householder
// reflector on the coordinates 0 .. r - 1.  Rows that conform already cost nothing, so the matrix of a Krylov-Schur restart
// (triangle of k rows, one full row, Hessenberg below) costs the reduction of a k x k matrix, not of the whole.
__attribute__((always_inline)) inline void hessenberg_reduce_body(int n, const Mat& A, const Mat& Q) {
    std::vector<Z> v((size_t)n), w((size_t)n);
    for (int r = n - 1; r >= 2; --r) {
        // x = A[r, 0:r]: wanted x^T P = gamma e_{r-1}^T with P = I - tp v v^H on the coordinates 0 .. r - 1
        double xnorm2 = 0.0;
        for (int j = 0; j < r - 1; ++j) xnorm2 += abs2(A(r, j));
        if (xnorm2 == 0.0) continue;
        const Z alpha = A(r, r - 1);
        const double nrm = std::sqrt(abs2(alpha) + xnorm2);
        const double beta = alpha.re >= 0.0 ? -nrm : nrm;
        // LAPACK's zlarfg on (alpha; x): G^H (alpha; x) = beta e, G = I - tau u u^H, u = (1; x / (alpha - beta)).  Here P^T = G^H,
        // i.e. v = conj(u) and tp = conj(tau).
        const Z tau = {(beta - alpha.re) / beta, -alpha.im / beta};
        const Z scale = zdiv({1.0, 0.0}, alpha - Z{beta, 0.0});
        const Z tp = conj(tau);
        v[(size_t)r - 1] = {1.0, 0.0};
        for (int j = 0; j < r - 1; ++j) v[(size_t)j] = conj(scale * A(r, j));
        // right: B[0:r+1, 0:r] -= tp (B v) v^H  (rows below r have no entries in these columns), and Q[:, 0:r] likewise
        for (int pass = 0; pass < 2; ++pass) {
            const Mat& B = pass == 0 ? A : Q;
            const int rows = pass == 0 ? r + 1 : n;
            for (int i = 0; i < rows; ++i) w[(size_t)i] = {0.0, 0.0};
            for (int j = 0; j < r; ++j) {
                const Z vj = v[(size_t)j];
                const Z* col = &B(0, j);
                for (int i = 0; i < rows; ++i) w[(size_t)i] = w[(size_t)i] + col[i] * vj;
            }
            for (int j = 0; j < r; ++j) {
                const Z f = tp * conj(v[(size_t)j]);
                Z* col = &B(0, j);
                for (int i = 0; i < rows; ++i) col[i] = col[i] - w[(size_t)i] * f;
            }
        }
        A(r, r - 1) = {beta, 0.0};
        for (int j = 0; j < r - 1; ++j) A(r, j) = {0.0, 0.0};
        // left: A[0:r, :] -= conj(tp) v (v^H A[0:r, :])
        const Z ctp = conj(tp);
        for (int j = 0; j < n; ++j) {
            Z d = {0.0, 0.0};
            Z* col = &A(0, j);
            for (int i = 0; i < r; ++i) d = d + conj(v[(size_t)i]) * col[i];
            d = ctp * d;
            for (int i = 0; i < r; ++i) col[i] = col[i] - v[(size_t)i] * d;
        }
    }
}
// the same body built for the host's vector width (the loops are complex axpys and dot products down contiguous columns)
void hessenberg_reduce_generic(int n, const Mat& A, const Mat& Q) { hessenberg_reduce_body(n, A, Q); }
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2,fma"))) void hessenberg_reduce_avx2(int n, const Mat& A, const Mat& Q) { hessenberg_reduce_body(n, A, Q); }
__attribute__((target("avx512f,avx512dq,fma"), min_vector_width(512))) void hessenberg_reduce_avx512(int n, const Mat& A, const Mat& Q) {
    hessenberg_reduce_body(n, A, Q);
}
#endif
void hessenberg_reduce(int n, const Mat& A, const Mat& Q) {
#if !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    static const char* isa = getenv("LSA_DENSE_ISA");
    static const bool want512 = !isa || !strcmp(isa, "avx512"), want256 = want512 || !strcmp(isa, "avx2");
    static const int level = (want512 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("fma")) ? 2
                             : (want256 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"))                                     ? 1
                                                                                                                                                 : 0;
    if (level == 2) return hessenberg_reduce_avx512(n, A, Q);
    if (level == 1) return hessenberg_reduce_avx2(n, A, Q);
#endif
    hessenberg_reduce_generic(n, A, Q);
}

// [x; y] <- R [x; y] element-wise on split (real / imaginary) contiguous arrays: the inner loops of the QR sweeps.
// R = [c s; -conj(s) c].  The arrays start on 64-byte lines and `len` is a multiple of 8 (rows and columns of the work
// layout are padded to that).  Two builds of the same body: the portable one and one for AVX2 + FMA hosts, picked at run time.
#define LSA_ROT_BODY                                                     \
    xr = (double*)__builtin_assume_aligned(xr, 64);                      \
    xi = (double*)__builtin_assume_aligned(xi, 64);                      \
    yr = (double*)__builtin_assume_aligned(yr, 64);                      \
    yi = (double*)__builtin_assume_aligned(yi, 64);                      \
    len &= ~7; /* callers pass whole 64-byte lines: no peeled head, no scalar tail */ \
    for (int j = 0; j < len; ++j) {                                      \
        const double ar = xr[j], ai = xi[j], br = yr[j], bi = yi[j];     \
        xr[j] = c * ar + (sr * br - si * bi);                            \
        xi[j] = c * ai + (sr * bi + si * br);                            \
        yr[j] = c * br - (sr * ar + si * ai);                            \
        yi[j] = c * bi - (sr * ai - si * ar);                            \
    }
void rot_pair_generic(int len, double* __restrict__ xr, double* __restrict__ xi, double* __restrict__ yr, double* __restrict__ yi, double c, double sr,
                      double si) {
    LSA_ROT_BODY
}
#if !defined(__HIP_DEVICE_COMPILE__)  // (this file is host code; the device pass of the compiler knows neither the target nor the builtins)
__attribute__((target("avx2,fma"))) void rot_pair_avx2(int len, double* __restrict__ xr, double* __restrict__ xi, double* __restrict__ yr,
                                                       double* __restrict__ yi, double c, double sr, double si) {
    LSA_ROT_BODY
}
__attribute__((target("avx512f,avx512dq,fma"), min_vector_width(512))) void rot_pair_avx512(int len, double* __restrict__ xr, double* __restrict__ xi,
                                                                                             double* __restrict__ yr, double* __restrict__ yi, double c,
                                                                                             double sr, double si) {
    LSA_ROT_BODY
}
#endif
#undef LSA_ROT_BODY
typedef void (*rot_pair_fn)(int, double*, double*, double*, double*, double, double, double);

// columns k, k + 1 of one row of the split planes (pr, pi point at column k): [a b] <- [a b] R^H, R^H = [c -s; conj(s) c], i.e.
// a' = c a + conj(s) b, b' = c b - s a.  The two columns are neighbours in memory: one 16-byte vector per plane, with
// cc = {c, c}, srn = {Re s, -Re s}, sii = {Im s, Im s}.
typedef double v2d __attribute__((ext_vector_type(2)));
inline void rot_cols2(double* pr, double* pi, v2d cc, v2d srn, v2d sii) {
    v2d vr, vi;
    memcpy(&vr, pr, sizeof vr);
    memcpy(&vi, pi, sizeof vi);
    const v2d wr = __builtin_shufflevector(vr, vr, 1, 0), wi = __builtin_shufflevector(vi, vi, 1, 0);
    const v2d nr = cc * vr + (srn * wr + sii * wi);
    const v2d ni = cc * vi + (srn * wi - sii * wr);
    memcpy(pr, &nr, sizeof nr);
    memcpy(pi, &ni, sizeof ni);
}
rot_pair_fn pick_rot_pair() {
#if !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    const char* isa = getenv("LSA_DENSE_ISA");  // "avx2" / "generic": measurement aid
    const bool want512 = !isa || !strcmp(isa, "avx512"), want256 = want512 || !strcmp(isa, "avx2");
    if (want512 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("fma")) return rot_pair_avx512;
    if (want256 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return rot_pair_avx2;
#endif
    return rot_pair_generic;
}

// Schur form of an upper Hessenberg H (in place: upper triangular T on return), Q <- Q U with H = U T U^H.
// Single-shift QR with Wilkinson shifts and the standard small-subdiagonal deflation test; returns false if an eigenvalue
// fails to converge in 30 sweeps per eigenvalue (does not happen for the Rayleigh quotients of an Arnoldi process).
//
// Work layout (80 x 80 is 2.9 ms with complex scalars in column-major storage, four times per 30 k-unknown eigen-solve, the GPU
// idle meanwhile): H is kept ROW-major and Q column-major, both split into real and imaginary planes, so that the two long
// updates of every rotation -- rows k, k + 1 of H to the right of the bulge, columns k, k + 1 of Q -- are contiguous real
// loops the compiler vectorises.  The column update of H inside the active window (a few rows around the bulge) stays
// strided; its part ABOVE the window (rows that only collect the sweep's rotations) is applied after the sweep, row by row,
// four rows at a time.
bool hessenberg_qr(int n, const Mat& H, const Mat& Q) {
    static const rot_pair_fn rot_pair = pick_rot_pair();
    const double ulp = 2.220446049250313e-16, smlnum = 2.2250738585072014e-308 * (n / ulp);
    const int ld = (n + 7) & ~7;  // whole 64-byte lines per row of H / column of Q; the padding holds zeros and stays zero
    std::vector<double> store((size_t)4 * ld * std::max(n, 1) + 8, 0.0);
    double* hr = (double*)(((uintptr_t)store.data() + 63) & ~(uintptr_t)63);  // hr[i * ld + j] = Re H(i, j)
    double* hi = hr + (size_t)ld * n;
    double* qr = hi + (size_t)ld * n;     // qr[j * ld + i] = Re Q(i, j)
    double* qi = qr + (size_t)ld * n;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            hr[(size_t)i * ld + j] = H(i, j).re;
            hi[(size_t)i * ld + j] = H(i, j).im;
            qr[(size_t)j * ld + i] = Q(i, j).re;
            qi[(size_t)j * ld + i] = Q(i, j).im;
        }
    auto h = [&](int i, int j) { return Z{hr[(size_t)i * ld + j], hi[(size_t)i * ld + j]}; };
    auto seth = [&](int i, int j, Z v) {
        hr[(size_t)i * ld + j] = v.re;
        hi[(size_t)i * ld + j] = v.im;
    };
    std::vector<double> rc((size_t)n), rsr((size_t)n), rsi((size_t)n);  // the rotations of the current sweep
    int ihi = n - 1;
    int its = 0, total = 0;
    bool ok = true;
    while (ihi >= 0) {
        int l = ihi;
        for (; l > 0; --l) {
            const double sub = abs1(h(l, l - 1));
            if (sub <= smlnum) break;
            double tst = abs1(h(l - 1, l - 1)) + abs1(h(l, l));
            if (tst == 0.0) {
                if (l - 2 >= 0) tst += abs1(h(l - 1, l - 2));
                if (l + 1 <= ihi) tst += abs1(h(l + 1, l));
            }
            if (sub <= ulp * tst) {
                // (Ahues & Tisseur) a small subdiagonal next to diagonal entries of very different size
                const double ab = std::max(abs1(h(l, l - 1)), abs1(h(l - 1, l))), ba = std::min(abs1(h(l, l - 1)), abs1(h(l - 1, l)));
                const Z dd = h(l - 1, l - 1) - h(l, l);
                const double aa = std::max(abs1(h(l, l)), abs1(dd)), bb = std::min(abs1(h(l, l)), abs1(dd));
                const double s = aa + ab;
                if (ba * (ab / s) <= std::max(smlnum, ulp * (bb * (aa / s)))) break;
            }
        }
        if (l > 0) seth(l, l - 1, {0.0, 0.0});
        if (l == ihi) {  // one eigenvalue has split off
            --ihi;
            its = 0;
            continue;
        }
        // (LAPACK's zlahqr allows 30 * max(10, n) sweeps per eigenvalue with an exceptional shift every tenth; giving up after 30
        //  aborted whole eigen-solves on clustered projected matrices.  The subdiagonals are not kept real here: abs1, not .re)
        if (++its > 30 * std::max(10, n) || ++total > 30 * std::max(10, n) * n) {
            ok = false;
            break;
        }
        Z shift;
        if (its % 20 == 10) {
            shift = h(l, l) + Z{0.75 * abs1(h(l + 1, l)), 0.0};
        } else if (its % 20 == 0) {
            shift = h(ihi, ihi) + Z{0.75 * abs1(h(ihi, ihi - 1)), 0.0};
        } else {
            // the eigenvalue of the trailing 2 x 2 block nearer to its last diagonal entry
            shift = h(ihi, ihi);
            const Z u2 = h(ihi - 1, ihi) * h(ihi, ihi - 1);
            if (abs1(u2) != 0.0) {
                const Z x = 0.5 * (h(ihi - 1, ihi - 1) - shift);
                const Z y = zsqrt(x * x + u2);
                const Z d = x.re * y.re + x.im * y.im < 0.0 ? x - y : x + y;  // the larger of x +- y
                shift = shift - zdiv(u2, d);
            }
        }
        // one implicit single-shift sweep over rows l .. ihi
        Z x = h(l, l) - shift, y = h(l + 1, l);
        for (int k = l; k < ihi; ++k) {
            double c;
            Z s, r;
            givens(x, y, c, s, r);
            rc[(size_t)k] = c;
            rsr[(size_t)k] = s.re;
            rsi[(size_t)k] = s.im;
            // rows k, k + 1 <- R rows, from the 64-byte line that holds column k to the end of the padded row (contiguous whole
            // vectors; left of column k - 1 both rows hold zeros, and column k - 1 -- the bulge -- gets its exact values below)
            const int k0 = k & ~7;
            rot_pair(ld - k0, hr + (size_t)k * ld + k0, hi + (size_t)k * ld + k0, hr + (size_t)(k + 1) * ld + k0, hi + (size_t)(k + 1) * ld + k0, c, s.re, s.im);
            if (k > l) {
                seth(k, k - 1, r);
                seth(k + 1, k - 1, {0.0, 0.0});
            }
            // columns k, k + 1 <- columns R^H, rows l .. min(k + 2, ihi) (the window; strided).  With R^H = [c -s; conj(s) c]:
            // new_k = c a + conj(s) b,  new_k1 = c b - s a
            const int i1 = std::min(k + 2, ihi);
            const v2d cc = {c, c}, srn = {s.re, -s.re}, sii = {s.im, s.im};
            for (int i = l; i <= i1; ++i) rot_cols2(hr + (size_t)i * ld + k, hi + (size_t)i * ld + k, cc, srn, sii);
            // columns k, k + 1 of Q <- columns R^H (contiguous): in the form of rot_pair, [x; y] <- [c conj(s); -s c] [x; y],
            // i.e. the rotation with s replaced by conj(s).  (Tried: Q's rotations of a whole sweep after the sweep, eight rows at
            // a time with column k carried in registers -- half the loads and stores, and slower: 2.16-2.28 against 1.99-2.06 ms.)
            rot_pair(ld, qr + (size_t)k * ld, qi + (size_t)k * ld, qr + (size_t)(k + 1) * ld, qi + (size_t)(k + 1) * ld, c, s.re, -s.im);
            if (k + 1 < ihi) {
                x = h(k + 1, k);
                y = h(k + 2, k);
            }
        }
        // the rows above the window collect the sweep's column rotations, one row after the other (four rows interleaved)
        for (int i0 = 0; i0 < l; i0 += 4) {
            const int ni = std::min(4, l - i0);
            for (int k = l; k < ihi; ++k) {
                const v2d cc = {rc[(size_t)k], rc[(size_t)k]}, srn = {rsr[(size_t)k], -rsr[(size_t)k]}, sii = {rsi[(size_t)k], rsi[(size_t)k]};
                for (int u = 0; u < ni; ++u) rot_cols2(hr + (size_t)(i0 + u) * ld + k, hi + (size_t)(i0 + u) * ld + k, cc, srn, sii);
            }
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            H(i, j) = (ok && i > j) ? Z{0.0, 0.0} : Z{hr[(size_t)i * ld + j], hi[(size_t)i * ld + j]};
            Q(i, j) = {qr[(size_t)j * ld + i], qi[(size_t)j * ld + i]};
        }
    return ok;
}

// T, Q: Schur form.  Swaps the diagonal entries k and k + 1 by one rotation (LAPACK's ztrexc step).
void schur_swap(int n, const Mat& T, const Mat& Q, int k) {
    const Z t11 = T(k, k), t22 = T(k + 1, k + 1);
    double c;
    Z s, r;
    givens(T(k, k + 1), t22 - t11, c, s, r);
    if (k + 2 < n) rot_rows(T, k, k + 1, k + 2, n, c, s);
    rot_cols(T, k, k + 1, 0, k, c, s);
    T(k, k) = t22;
    T(k + 1, k + 1) = t11;
    rot_cols(Q, k, k + 1, 0, n, c, s);
}

// moves the selected diagonal entries to the leading block (their relative order and that of the others kept); returns their count
int schur_reorder(int n, const Mat& T, const Mat& Q, const std::vector<char>& select) {
    int ks = 0;
    for (int k = 0; k < n; ++k) {
        if (!select[(size_t)k]) continue;
        for (int i = k; i > ks; --i) schur_swap(n, T, Q, i - 1);
        ++ks;
    }
    return ks;
}

// right eigenvectors of the upper triangular T: S[:, k], unit 2-norm
void tri_eigenvectors(int n, const Mat& T, const Mat& S) {
    const double ulp = 2.220446049250313e-16, smlnum = 2.2250738585072014e-308 * (n / ulp);
    double tnorm = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) tnorm = std::max(tnorm, abs1(T(i, j)));
    for (int k = 0; k < n; ++k) {
        const Z lam = T(k, k);
        const double smin = std::max(ulp * std::max(abs1(lam), tnorm * 1e-3), smlnum);
        for (int i = k + 1; i < n; ++i) S(i, k) = {0.0, 0.0};
        S(k, k) = {1.0, 0.0};
        for (int i = 0; i < k; ++i) S(i, k) = -T(i, k);
        for (int i = k - 1; i >= 0; --i) {
            Z d = T(i, i) - lam;
            if (abs1(d) < smin) d = {smin, 0.0};
            const Z xi = zdiv(S(i, k), d);
            S(i, k) = xi;
            if (abs1(xi) > 1e150) {  // (never met on Rayleigh quotients; keeps the substitution finite)
                for (int q = 0; q <= k; ++q) S(q, k) = 1e-150 * S(q, k);
            }
            const Z xs = S(i, k);
            for (int q = 0; q < i; ++q) S(q, k) = S(q, k) - T(q, i) * xs;
        }
        double nrm = 0.0;
        for (int i = 0; i <= k; ++i) nrm += abs2(S(i, k));
        nrm = 1.0 / std::sqrt(nrm);
        for (int i = 0; i <= k; ++i) S(i, k) = nrm * S(i, k);
    }
}

bool is_hessenberg(int n, const Mat& A) {
    for (int j = 0; j + 2 < n; ++j)
        for (int i = j + 2; i < n; ++i)
            if (A(i, j).re != 0.0 || A(i, j).im != 0.0) return false;
    return true;
}

// complex Schur form of a general square matrix: A = Q T Q^H, T over A, Q written
bool schur(int n, const Mat& A, const Mat& Q) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Q(i, j) = {i == j ? 1.0 : 0.0, 0.0};
    if (!is_hessenberg(n, A)) hessenberg_reduce(n, A, Q);
    return hessenberg_qr(n, A, Q);
}

// ---- eigenvalue selection (SLEPc's EPSWhich on lambda, Solver/utils.py:152-187 of the reference) -------------------------------
struct Selector {
    int which, transform;
    Z sigma, nu, target;
    Z back(Z th) const {  // Ritz value of the transformed operator -> eigenvalue of the pencil
        const double tiny = 2.2250738585072014e-308;
        if (transform == 0) {  // shift-invert: theta = 1 / (lambda - sigma)
            if (th.re == 0.0 && th.im == 0.0) th = {tiny, 0.0};
            return sigma + zdiv({1.0, 0.0}, th);
        }
        if (transform == 2) {  // Cayley: theta = (lambda + nu) / (lambda - sigma)
            Z d = th - Z{1.0, 0.0};
            if (th.re == 1.0 && th.im == 0.0) d = {1e-300, 0.0};
            return zdiv(sigma * th + nu, d);
        }
        return th + sigma;  // shift
    }
    double key(Z th) const {  // small = wanted
        const Z lam = back(th);
        switch (which) {
            case LSA_WHICH_LARGEST_MAGNITUDE: return -zabs(lam);
            case LSA_WHICH_LARGEST_REAL: return -lam.re;
            case LSA_WHICH_SMALLEST_REAL: return lam.re;
            case LSA_WHICH_LARGEST_IMAGINARY: return -lam.im;
            case LSA_WHICH_SMALLEST_IMAGINARY: return lam.im;
            case LSA_WHICH_TARGET_REAL: return std::fabs(lam.re - target.re);
            case LSA_WHICH_TARGET_IMAGINARY: return std::fabs(lam.im - target.im);
            default: return zabs(lam - target);  // LSA_WHICH_TARGET_MAGNITUDE
        }
    }
};

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rng {  // splitmix64 + Box-Muller: start vectors and the fresh directions after a breakdown
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    void normal_pair(double& a, double& b) {
        const double r = std::sqrt(-2.0 * std::log(uniform())), t = 6.283185307179586 * uniform();
        a = r * std::cos(t);
        b = r * std::sin(t);
    }
};

}  // namespace

extern "C" {

int lsa_dense_schur(int32_t n, void* A, int32_t lda, void* Q, int32_t ldq) {
    if (n < 0 || !A || !Q || lda < std::max(1, n) || ldq < std::max(1, n)) return LSA_ERR_ARG;
    return schur(n, Mat{(Z*)A, lda}, Mat{(Z*)Q, ldq}) ? LSA_OK : LSA_ERR_DIVERGED;
}

int lsa_dense_schur_reorder(int32_t n, void* T, int32_t ldt, void* Q, int32_t ldq, const int32_t* select, int32_t* nselected) {
    if (n < 0 || !T || !Q || !select || ldt < std::max(1, n) || ldq < std::max(1, n)) return LSA_ERR_ARG;
    std::vector<char> sel((size_t)n);
    for (int32_t k = 0; k < n; ++k) sel[(size_t)k] = select[k] != 0;
    const int ks = schur_reorder(n, Mat{(Z*)T, ldt}, Mat{(Z*)Q, ldq}, sel);
    if (nselected) *nselected = ks;
    return LSA_OK;
}

int lsa_dense_tri_eigenvectors(int32_t n, const void* T, int32_t ldt, void* S, int32_t lds) {
    if (n < 0 || !T || !S || ldt < std::max(1, n) || lds < std::max(1, n)) return LSA_ERR_ARG;
    tri_eigenvectors(n, Mat{(Z*)T, ldt}, Mat{(Z*)S, lds});
    return LSA_OK;
}

// ---- inertia of a dense real symmetric matrix (host): Bunch-Kaufman diagonal pivoting -------------------------------------
// What SLEPc's spectrum slicing asks of the factorisation behind set_interval / iEpsWhich.ALL (Solver/utils.py:248-254): the
// number of negative eigenvalues of A - sigma M = the number of eigenvalues of the definite pencil below sigma (Sylvester).  The
// multifrontal LU is a block congruence C = L D L^T with D = diag(pivot blocks) when C is symmetric, so the inertia of C is the
// sum over the tree nodes of the inertia of their (dense, symmetric) pivot blocks -- lsa_ndlu_inertia hands them to this routine.
// A = P L D L^T P^T with 1 x 1 and 2 x 2 diagonal blocks in D (a 2 x 2 block of the algorithm has one eigenvalue of each sign);
// the whole symmetric matrix is kept and updated (the pivot blocks of a forest are a few hundred rows; O(2 n^3 / 3)).
// Pivots below tol_rel * max|A| count as zero.  Returns negative / zero / positive counts.
int lsa_dense_sym_inertia(int32_t n, const double* A, int32_t lda, double tol_rel, int64_t* negative, int64_t* zero, int64_t* positive) {
    if (n < 0 || (n > 0 && !A) || lda < std::max(1, n) || !negative || !zero || !positive) return LSA_ERR_ARG;
    *negative = *zero = *positive = 0;
    if (n == 0) return LSA_OK;
    std::vector<double> w((size_t)n * n);
    double amax = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double v = 0.5 * (A[(size_t)j * lda + i] + A[(size_t)i * lda + j]);  // (rounding leaves a computed inverse unsymmetric in the last bits)
            if (!std::isfinite(v)) return LSA_ERR_NONFINITE;
            w[(size_t)j * n + i] = v;
            amax = std::max(amax, std::fabs(v));
        }
    auto W = [&](int i, int j) -> double& { return w[(size_t)j * n + i]; };
    auto swap_sym = [&](int p, int q) {  // rows and columns p <-> q of the symmetric matrix
        if (p == q) return;
        for (int j = 0; j < n; ++j) std::swap(W(p, j), W(q, j));
        for (int i = 0; i < n; ++i) std::swap(W(i, p), W(i, q));
    };
    const double alpha = (1.0 + std::sqrt(17.0)) / 8.0, tiny = std::max(tol_rel, 0.0) * amax;
    int k = 0;
    while (k < n) {
        const double absakk = std::fabs(W(k, k));
        int imax = k;
        double colmax = 0.0;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(W(i, k)) > colmax) colmax = std::fabs(W(i, k)), imax = i;
        int step = 1;
        if (std::max(absakk, colmax) <= tiny) {  // the whole column vanishes: a zero eigenvalue (to the tolerance)
            ++*zero;
            ++k;
            continue;
        }
        if (absakk < alpha * colmax) {
            double rowmax = 0.0;
            for (int j = k; j < n; ++j)
                if (j != imax) rowmax = std::max(rowmax, std::fabs(W(imax, j)));
            if (absakk >= alpha * colmax * (colmax / rowmax)) {
                // 1 x 1 pivot in place
            } else if (std::fabs(W(imax, imax)) >= alpha * rowmax) {
                swap_sym(k, imax);  // 1 x 1 pivot, the diagonal entry of row imax
            } else {
                swap_sym(k + 1, imax);  // 2 x 2 pivot: rows k and imax
                step = 2;
            }
        }
        if (step == 1) {
            const double d = W(k, k);
            if (std::fabs(d) <= tiny) ++*zero;
            else if (d < 0.0) ++*negative;
            else ++*positive;
            if (std::fabs(d) > tiny) {
                const double dinv = 1.0 / d;
                for (int j = k + 1; j < n; ++j) {
                    const double f = W(k, j) * dinv;
                    if (f == 0.0) continue;
                    for (int i = k + 1; i < n; ++i) W(i, j) -= W(i, k) * f;
                }
            }
        } else {
            // D = [a b; b c] with |b| the largest entry of its rows: det < 0, one eigenvalue of each sign
            const double a = W(k, k), b = W(k + 1, k), c = W(k + 1, k + 1), det = a * c - b * b;
            if (std::fabs(det) <= tiny * tiny) {
                *zero += 2;
            } else {
                if (det < 0.0) {
                    ++*negative;
                    ++*positive;
                } else if (a + c < 0.0) *negative += 2;
                else *positive += 2;
                for (int j = k + 2; j < n; ++j) {
                    // column j of the update: W22 -= W21 D^-1 W12
                    const double u = W(k, j), v = W(k + 1, j);
                    const double f0 = (c * u - b * v) / det, f1 = (a * v - b * u) / det;
                    if (f0 == 0.0 && f1 == 0.0) continue;
                    for (int i = k + 2; i < n; ++i) W(i, j) -= W(i, k) * f0 + W(i, k + 1) * f1;
                }
            }
        }
        k += step;
    }
    return LSA_OK;
}

static_assert(sizeof(lsa_ks_result) == 56 && sizeof(lsa_ks_options) == 88, "lsa_ks_options layout is part of the C-ABI (tests/test_abi.py)");

int lsa_krylov_solve(lsa_ctx* ctx, lsa_krylov* k, const lsa_ks_options* o, const void* v0, const double* mask, int32_t max_out, void* theta_out,
                     void* lambda_out, void* X_out, double* est_out, lsa_ks_result* result) {
    if (!ctx || !k || !o || !result) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_solve: null argument");
    int32_t m = 0;
    int64_t n = 0;
    LSA_CHECK(lsa_krylov_shape(k, &n, &m));
    if (m > n) return lsa_set_error(ctx, LSA_ERR_ARG, "ncv = %d exceeds the problem size %lld", m, (long long)n);
    if (o->nev < 1 || o->max_restarts < 0 || !(o->tol > 0.0) || max_out < 0 || (max_out > 0 && (!theta_out || !lambda_out)))
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_solve: nev, tol must be positive, output buffers are required");
    const int nev = std::min<int>(o->nev, m);
    const double keep_fraction = o->keep_fraction > 0.0 && o->keep_fraction < 1.0 ? o->keep_fraction : 0.5;
    const Selector sel{o->which, o->transform, {o->sigma[0], o->sigma[1]}, {o->antishift[0], o->antishift[1]}, {o->target[0], o->target[1]}};
    Rng rng{o->seed * 0x2545F4914F6CDD1Dull + 0x1234567ull};
    std::vector<Z> vec((size_t)n);
    auto random_vector = [&]() {
        for (int64_t i = 0; i < n; ++i) {
            rng.normal_pair(vec[(size_t)i].re, vec[(size_t)i].im);
            if (mask && mask[i] == 0.0) vec[(size_t)i] = {0.0, 0.0};
        }
    };
    if (v0) {
        memcpy(vec.data(), v0, (size_t)n * sizeof(Z));
        if (mask)
            for (int64_t i = 0; i < n; ++i)
                if (mask[i] == 0.0) vec[(size_t)i] = {0.0, 0.0};
    } else {
        random_vector();
    }
    LSA_CHECK(lsa_krylov_inject(ctx, k, 0, vec.data()));
    const int ldh = m + 1;
    std::vector<Z> H((size_t)ldh * m, Z{0.0, 0.0}), T((size_t)m * m), Q((size_t)m * m), S((size_t)m * m), b((size_t)m), bq((size_t)m), w((size_t)m);
    std::vector<double> est((size_t)m), rel((size_t)m), keys((size_t)m);
    std::vector<int> rank((size_t)m);
    std::vector<char> select((size_t)m);
    const Mat Hm{H.data(), ldh}, Tm{T.data(), m}, Qm{Q.data(), m}, Sm{S.data(), m};
    int kept = 0, restarts = 0;
    int64_t applies = 0;
    memset(result, 0, sizeof *result);
    double t_expand = 0.0, t_dense = 0.0, t_restart = 0.0;
    while (true) {
        // ---- expand to m vectors; continue past exact breakdowns (invariant subspace) with a fresh direction ----
        double t0 = now_s();
        int j = kept;
        while (j < m) {
            int32_t bd = -1;
            LSA_CHECK(lsa_krylov_extend(ctx, k, j, m, H.data(), ldh, &bd));
            if (bd < 0) {
                applies += m - j;
                break;
            }
            applies += bd - j + 1;
            Hm(bd + 1, bd) = {0.0, 0.0};
            if (bd + 1 >= m) break;  // broke down on the last step: the m vectors span an invariant subspace
            random_vector();
            LSA_CHECK(lsa_krylov_inject(ctx, k, bd + 1, vec.data()));
            j = bd + 1;
        }
        t_expand += now_s() - t0;
        t0 = now_s();
        for (int c = 0; c < m; ++c) {
            b[(size_t)c] = Hm(m, c);  // b^H: the row under the square part
            for (int r = 0; r < m; ++r) Tm(r, c) = Hm(r, c);
        }
        // ---- Ritz pairs and residual estimates ----
        if (!schur(m, Tm, Qm)) return lsa_set_error(ctx, LSA_ERR_DIVERGED, "Krylov-Schur: the QR algorithm on the projected matrix did not converge");
        tri_eigenvectors(m, Tm, Sm);
        for (int c = 0; c < m; ++c) {
            Z acc = {0.0, 0.0};
            for (int r = 0; r < m; ++r) acc = acc + b[(size_t)r] * Qm(r, c);
            bq[(size_t)c] = acc;
        }
        for (int c = 0; c < m; ++c) {
            Z acc = {0.0, 0.0};
            for (int r = 0; r <= c; ++r) acc = acc + bq[(size_t)r] * Sm(r, c);
            est[(size_t)c] = zabs(acc);
            w[(size_t)c] = Tm(c, c);
            keys[(size_t)c] = sel.key(w[(size_t)c]);
            rel[(size_t)c] = est[(size_t)c] / std::max(zabs(w[(size_t)c]), 2.2250738585072014e-308);
            rank[(size_t)c] = c;
        }
        std::stable_sort(rank.begin(), rank.end(), [&](int x, int y) { return keys[(size_t)x] < keys[(size_t)y]; });
        int nconv = 0;
        while (nconv < m && rel[(size_t)rank[(size_t)nconv]] <= o->tol) ++nconv;
        if (nconv >= nev || nconv >= m || restarts >= o->max_restarts) {
            const int nout = std::min<int>(nconv, max_out);
            if (nout > 0) {
                std::vector<Z> Y((size_t)m * nout);
                for (int c = 0; c < nout; ++c) {
                    const int src = rank[(size_t)c];
                    for (int r = 0; r < m; ++r) {
                        Z acc = {0.0, 0.0};
                        for (int q = 0; q <= src; ++q) acc = acc + Qm(r, q) * Sm(q, src);
                        Y[(size_t)c * m + r] = acc;
                    }
                    ((Z*)theta_out)[c] = w[(size_t)src];
                    ((Z*)lambda_out)[c] = sel.back(w[(size_t)src]);
                    if (est_out) est_out[c] = rel[(size_t)src];
                }
                t_dense += now_s() - t0;
                t0 = now_s();
                if (X_out) LSA_CHECK(lsa_krylov_ritz_vectors(ctx, k, m, nout, Y.data(), m, 3, X_out));
                t_restart += now_s() - t0;
                t0 = now_s();
            }
            t_dense += now_s() - t0;
            result->seconds_expand = t_expand;
            result->seconds_dense = t_dense;
            result->seconds_restart = t_restart;
            result->nconv = nconv;
            result->nout = nout;
            result->restarts = restarts;
            result->op_applies = applies;
            result->next_unconverged = nconv < m ? rel[(size_t)rank[(size_t)nconv]] : 0.0;
            return k_agree_in_step(ctx, "lsa_krylov_solve");  // (a sharded solve ends with the ranks comparing their exchange counts)
        }
        // ---- truncate to the wanted part of the Schur form and restart ----
        int knew = nconv + (int)((m - nconv) * keep_fraction);
        knew = std::max(std::min(knew, m - 1), 1);
        {
            // ties at the selection threshold are all selected (what a sort callback of the Schur routine would do)
            std::vector<double> sorted(keys);
            std::sort(sorted.begin(), sorted.end());
            const double thr = sorted[(size_t)knew] > sorted[(size_t)knew - 1] ? 0.5 * (sorted[(size_t)knew - 1] + sorted[(size_t)knew]) : sorted[(size_t)knew - 1];
            for (int c = 0; c < m; ++c) select[(size_t)c] = keys[(size_t)c] <= thr;
            const int sdim = schur_reorder(m, Tm, Qm, select);
            knew = std::max(std::min(sdim, m - 1), 1);
        }
        t_dense += now_s() - t0;
        t0 = now_s();
        LSA_CHECK(lsa_krylov_restart(ctx, k, m, knew, Q.data(), m));
        t_restart += now_s() - t0;
        std::fill(H.begin(), H.end(), Z{0.0, 0.0});
        for (int c = 0; c < knew; ++c) {
            for (int r = 0; r <= c; ++r) Hm(r, c) = Tm(r, c);
            Z acc = {0.0, 0.0};
            for (int r = 0; r < m; ++r) acc = acc + b[(size_t)r] * Qm(r, c);
            Hm(knew, c) = acc;
        }
        kept = knew;
        ++restarts;
    }
}

int lsa_eigs_sinvert(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* M, const double sigma[2], int32_t nev, int32_t ncv, double tol, int32_t max_restarts,
                     const lsa_op_options* opts, const void* v0, const int32_t* row_perm, int32_t max_out, void* lambda_out, void* X_out, double* est_out,
                     lsa_ks_result* result, lsa_stats* stats) {
    if (!ctx || !A || !sigma || !opts || !result) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_eigs_sinvert: null argument");
    if (ncv <= 0) ncv = std::max(2 * nev, nev + 15);
    ncv = (int32_t)std::min<int64_t>(ncv, lsa_mat_rows(A));
    lsa_op* op = nullptr;
    lsa_krylov* kr = nullptr;
    int rc = lsa_op_create(ctx, A, M, sigma, 0, opts, &op);
    if (rc == LSA_OK) rc = lsa_krylov_create(ctx, op, ncv, &kr);
    if (rc == LSA_OK && row_perm) rc = lsa_krylov_set_row_permutation(ctx, kr, row_perm);
    if (rc == LSA_OK) {
        lsa_ks_options o;
        memset(&o, 0, sizeof o);
        o.nev = nev;
        o.max_restarts = max_restarts;
        o.tol = tol;
        o.which = LSA_WHICH_TARGET_MAGNITUDE;
        o.transform = 0;
        o.sigma[0] = o.target[0] = sigma[0];
        o.sigma[1] = o.target[1] = sigma[1];
        o.keep_fraction = 0.5;
        std::vector<Z> theta((size_t)std::max(max_out, 1));
        rc = lsa_krylov_solve(ctx, kr, &o, v0, nullptr, max_out, theta.data(), lambda_out, X_out, est_out, result);
    }
    if (op && stats) (void)lsa_op_stats(op, stats);
    if (kr) lsa_krylov_destroy(kr);
    if (op) lsa_op_destroy(op);
    return rc;
}

}  // extern "C"
