// Exact block-tridiagonal LU of the shifted operator: the device counterpart of PreconditionerType.LU, which is what
// the reference's cylinder runs use for the ST's inner solve (.examples/eigenvalues.py:100, Sensitivity/__init__.py:182).
//
// In RCM order C = A - sigma M has bandwidth w (723 at S30k).  With a block size B > w the matrix is block tridiagonal,
//
//     C = [ C_00 C_01           ]         S_0 = C_00,   S_b = C_bb - C_{b,b-1} S_{b-1}^-1 C_{b-1,b}
//         [ C_10 C_11 C_12      ]
//         [      C_21 C_22 ...  ]         forward :  y_b = v_b - C_{b,b-1} (S_{b-1}^-1 y_{b-1})
//                                         backward:  x_b = S_b^-1 (y_b - C_{b,b+1} x_{b+1})
//
// and only a w x w corner of every Schur complement differs from C_bb.  The Schur blocks are inverted ONCE per shift
// into dense B x B matrices that stay in HBM (n*B scalars: 0.5 GB at S30k, 25 GB at S500k -- the 288 GB at work), by an
// in-place Gauss-Jordan elimination with partial pivoting; the off-diagonal blocks stay the sparse rows of C.  A solve is
// then 4 dependent launches per block (dense mat-vec + sparse update, forward and backward), replayed from a hipGraph,
// and it is a *direct* solve (residual ~1e-15): the GMRES around it converges in one iteration and only guards accuracy.
// Not usable when the band does not fit (3D meshes): lsa_blu_create then fails and the ILU(k) path is used.
#include <algorithm>
#include <chrono>

#include "lsa_internal.h"

struct lsa_blu {
    lsa_ctx* ctx;
    const lsa_mat* C;  // borrowed: the sparse off-diagonal blocks are read from C at every solve
    int32_t n, B, nb, bandwidth;
    int32_t ld = 0;  // row stride of the dense Schur inverses: B + 16, NOT a power of two (column sweeps of the
                     // Gauss-Jordan panels would otherwise hit one memory channel: 1024 rows x 16 KiB apart)
    int dtype;
    int32_t *lsplit = nullptr, *usplit = nullptr;                 // device: per row, first entry with col >= block start / end
    int32_t *cptr = nullptr, *crow = nullptr, *cpos = nullptr;    // device CSC view of C (setup only)
    void* sinv = nullptr;                                         // device: n x B row-major, block b = rows [b*B, ...)
    int32_t* ipiv[2] = {nullptr, nullptr};   // Gauss-Jordan workspaces, one per chain
    void* colbuf[2] = {nullptr, nullptr};
    int32_t* flag = nullptr;
    hipStream_t stream2 = nullptr;          // second chain of the twisted factorisation / solve
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int32_t mid = 0;                        // middle block: chains run 0 .. mid-1 and nb-1 .. mid+1
    void *t[2] = {nullptr, nullptr}, *y[2] = {nullptr, nullptr}, *z[2] = {nullptr, nullptr};
    void *in[2] = {nullptr, nullptr}, *out[2] = {nullptr, nullptr};
    void* graph[2] = {nullptr, nullptr};
    double seconds = 0.0;
};

namespace {

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// out[0] = max |v|^2 (bit pattern of a non-negative double orders like an unsigned integer)
template <typename T>
__global__ void maxabs2_kernel(int64_t nnz, const T* __restrict__ v, unsigned long long* __restrict__ out) {
    double best = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) best = fmax(best, s_abs2(v[p]));
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) best = fmax(best, __shfl_xor(best, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(best));
}

// S[r - bs, c - bs] = C[r, c] for the entries of block row b that fall into the diagonal block
template <typename T>
__global__ void blu_scatter_kernel(int32_t bs, int32_t be, int32_t ld, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                   const int32_t* __restrict__ lsplit, const int32_t* __restrict__ usplit,
                                   const T* __restrict__ val, T* __restrict__ S) {
    const int32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 15;
    const int32_t r = bs + (gid >> 4);
    if (r >= be) return;
    for (int32_t p = lsplit[r] + lane; p < usplit[r]; p += 16) S[(size_t)(r - bs) * ld + (ci[p] - bs)] = val[p];
}

// S[r, :] -= (C[r, nbr block] * Sinv_nbr) * C[nbr block, this block] for the neighbour block [ns, ne) on the left
// (RIGHT = false) or on the right (RIGHT = true) of block [bs, be); only rows with entries in that neighbour do work
template <typename T, bool RIGHT>
__global__ __launch_bounds__(256) void blu_corner_kernel(int32_t ns, int32_t ne, int32_t bs, int32_t be, int32_t ld,
                                                         const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                         const int32_t* __restrict__ lsplit, const int32_t* __restrict__ usplit,
                                                         const T* __restrict__ val, const int32_t* __restrict__ cptr,
                                                         const int32_t* __restrict__ crow, const int32_t* __restrict__ cpos,
                                                         const T* __restrict__ sinv_nbr, T* __restrict__ S) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* X = (T*)dyn;  // row r of C_{b,nbr} * Sinv_nbr
    const int32_t r = bs + blockIdx.x;
    const int32_t g0 = RIGHT ? usplit[r] : rp[r];
    const int32_t g1 = RIGHT ? rp[r + 1] : lsplit[r];
    if (g0 == g1) return;  // no coupling to that neighbour
    const int32_t m1 = ne - ns;
    for (int32_t k = threadIdx.x; k < m1; k += 256) {
        T acc = scalar_traits<T>::zero();
        for (int32_t p = g0; p < g1; ++p) fma_acc(acc, val[p], sinv_nbr[(size_t)(ci[p] - ns) * ld + k]);
        X[k] = acc;
    }
    __syncthreads();
    const int32_t m = be - bs;
    for (int32_t j = threadIdx.x; j < m; j += 256) {
        const int32_t q0 = cptr[bs + j], q1 = cptr[bs + j + 1];
        T acc = scalar_traits<T>::zero();
        bool any = false;
        for (int32_t q = q0; q < q1; ++q) {
            const int32_t k = crow[q];
            if (k >= ns && k < ne) {
                fma_acc(acc, X[k - ns], val[cpos[q]]);
                any = true;
            }
        }
        if (any) {
            T* out = S + (size_t)(r - bs) * ld + j;
            *out = s_sub(*out, acc);
        }
    }
}

// ---- in-place Gauss-Jordan inversion with partial pivoting: two launches per pivot column -----------------------------
template <typename T>
__global__ __launch_bounds__(1024) void gj_pivot_kernel(T* __restrict__ a, int32_t ld, int32_t m, int32_t k, int32_t* __restrict__ ipiv,
                                                        T* __restrict__ colbuf, int32_t* __restrict__ flag, double tiny2) {
    __shared__ double smag[16];
    __shared__ int32_t sidx[16];
    __shared__ int32_t spiv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double best = -1.0;
    int32_t bi = k;
    for (int32_t i = k + tid; i < m; i += 1024) {
        const double mag = s_abs2(a[(size_t)i * ld + k]);
        if (mag > best) {
            best = mag;
            bi = i;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const double ob = __shfl_xor(best, s, 64);
        const int32_t oi = __shfl_xor(bi, s, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) {
        smag[wave] = best;
        sidx[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        double b = smag[0];
        int32_t i0 = sidx[0];
        for (int w = 1; w < 16; ++w)
            if (smag[w] > b || (smag[w] == b && sidx[w] < i0)) {
                b = smag[w];
                i0 = sidx[w];
            }
        spiv = i0;
        ipiv[k] = i0;
        if (!(b > tiny2)) atomicCAS(&flag[1], 0, k + 1);  // (numerically) singular block
    }
    __syncthreads();
    const int32_t p = spiv;
    T* rk = a + (size_t)k * ld;
    T* rpv = a + (size_t)p * ld;
    if (p != k) {
        for (int32_t j = tid; j < m; j += 1024) {
            const T tmp = rk[j];
            rk[j] = rpv[j];
            rpv[j] = tmp;
        }
    }
    __syncthreads();
    T piv = rk[k];
    if (s_abs2(piv) == 0.0) s_from(piv, 1.0, 0.0);
    const T pinv = s_inv(piv);
    __syncthreads();
    for (int32_t j = tid; j < m; j += 1024) rk[j] = (j == k) ? pinv : s_mul(pinv, rk[j]);
    // multipliers: column k of the other rows, which is then cleared
    for (int32_t i = tid; i < m; i += 1024) {
        if (i == k) {
            colbuf[i] = scalar_traits<T>::zero();
        } else {
            colbuf[i] = a[(size_t)i * ld + k];
            a[(size_t)i * ld + k] = scalar_traits<T>::zero();
        }
    }
}

// a[i, :] -= colbuf[i] * a[k, :] for every row i != k; one wavefront per row, rows with a zero multiplier are skipped
template <typename T>
__global__ __launch_bounds__(256) void gj_update_kernel(T* __restrict__ a, int32_t ld, int32_t m, int32_t k, const T* __restrict__ colbuf) {
    const int lane = threadIdx.x & 63;
    const int32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m || i == k) return;
    const T f = colbuf[i];
    if (s_abs2(f) == 0.0) return;
    const T* rk = a + (size_t)k * ld;
    T* ri = a + (size_t)i * ld;
    for (int32_t j = lane; j < m; j += 64) {
        T v = ri[j];
        const T prod = s_mul(f, rk[j]);
        ri[j] = s_sub(v, prod);
    }
}

// ---- blocked form: a panel of w <= kPanelW pivot columns is eliminated by ONE workgroup, then the other columns get the
// accumulated rank-w update in one grid-wide launch:
//     A[:, J] <- (P A)[:, J] with the pivot rows zeroed  +  W * (P A)[K, J]
// where W (m x w) is what the in-place elimination leaves in the panel columns and K are the panel's pivot rows.
// 3 launches per w pivots instead of 2 per pivot; the arithmetic is the same elimination in the same order.
//
// The panel lives in REGISTERS: thread t owns rows t, t + 1024, ... (RPT of them), w complex values each.  Per pivot the
// workgroup does a shuffle/LDS arg-max, the two owner threads publish rows k and p through LDS (2w values), and every
// thread updates its own rows from the broadcast pivot row: three barriers and a few hundred bytes of LDS traffic per
// pivot (an LDS-resident panel moved 24 KB per wavefront per pivot and took 5 us per pivot).
constexpr int kPanelW = 8;
template <typename T, int NT, int RPT, int W>
__global__ __launch_bounds__(NT) void gj_panel_kernel(T* __restrict__ a, int32_t ld, int32_t m, int32_t k0, int32_t w,
                                                        int32_t* __restrict__ ipiv, int32_t* __restrict__ perm, int32_t* __restrict__ flag,
                                                        double tiny2) {
    __shared__ T rowk[kPanelW], rowp[kPanelW];
    __shared__ double smag[16];
    __shared__ int32_t sidx[16];
    __shared__ int32_t spiv[kPanelW];
    __shared__ int32_t aff[2 * kPanelW], src[2 * kPanelW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T r[RPT][W];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int32_t i = tid + NT * q;
#pragma unroll
        for (int j = 0; j < W; ++j) r[q][j] = (i < m && j < w) ? a[(size_t)i * ld + k0 + j] : scalar_traits<T>::zero();
    }
    for (int32_t jj = 0; jj < w; ++jj) {
        const int32_t k = k0 + jj;
        double best = -1.0;
        int32_t bi = k;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            T cv = scalar_traits<T>::zero();
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (j == jj) cv = r[q][j];
            const double mag = s_abs2(cv);
            if (i >= k && i < m && mag > best) {
                best = mag;
                bi = i;
            }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const double ob = __shfl_xor(best, s, 64);
            const int32_t oi = __shfl_xor(bi, s, 64);
            if (ob > best || (ob == best && oi < bi)) {
                best = ob;
                bi = oi;
            }
        }
        if (lane == 0) {
            smag[wave] = best;
            sidx[wave] = bi;
        }
        __syncthreads();
        if (wave == 0) {
            double b = (lane < NT / 64) ? smag[lane] : -2.0;
            int32_t i0 = (lane < NT / 64) ? sidx[lane] : 0;
#pragma unroll
            for (int s = 8; s > 0; s >>= 1) {
                const double ob = __shfl_xor(b, s, 64);
                const int32_t oi = __shfl_xor(i0, s, 64);
                if (ob > b || (ob == b && oi < i0)) {
                    b = ob;
                    i0 = oi;
                }
            }
            if (lane == 0) {
                spiv[jj] = i0;
                ipiv[k] = i0;
                if (!(b > tiny2)) atomicCAS(&flag[1], 0, k + 1);
            }
        }
        __syncthreads();
        const int32_t p = spiv[jj];
        // the owners of rows k and p publish them
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            if (i == k) {
#pragma unroll
                for (int j = 0; j < W; ++j) rowk[j] = r[q][j];
            }
            if (i == p) {
#pragma unroll
                for (int j = 0; j < W; ++j) rowp[j] = r[q][j];
            }
        }
        __syncthreads();
        // scaled pivot row (row p moves to position k); every thread forms it from the broadcast copy
        T piv = scalar_traits<T>::zero();
#pragma unroll
        for (int j = 0; j < W; ++j)
            if (j == jj) piv = rowp[j];
        if (s_abs2(piv) == 0.0) s_from(piv, 1.0, 0.0);
        const T pinv = s_inv(piv);
        T prow[W];
#pragma unroll
        for (int j = 0; j < W; ++j) prow[j] = (j == jj) ? pinv : s_mul(pinv, rowp[j]);
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            if (i >= m) continue;
            if (i == k) {
#pragma unroll
                for (int j = 0; j < W; ++j) r[q][j] = prow[j];
                continue;
            }
            if (i == p) {  // p != k here: this row receives the old row k
#pragma unroll
                for (int j = 0; j < W; ++j) r[q][j] = rowk[j];
            }
            T fm = scalar_traits<T>::zero();
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (j == jj) fm = r[q][j];
            if (s_abs2(fm) == 0.0) continue;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const T base = (j == jj) ? scalar_traits<T>::zero() : r[q][j];
                r[q][j] = s_sub(base, s_mul(fm, prow[j]));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int32_t i = tid + NT * q;
        if (i < m) {
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (j < w) a[(size_t)i * ld + k0 + j] = r[q][j];
        }
    }
    // Row interchanges on the columns outside the panel are left to the whole grid (one CU moves ~50 GB/s, and the w
    // swaps touch up to 2w rows of m entries): thread 0 replays them on an index list and publishes, for every touched
    // row, the row its final content comes from.  perm[0] = count, perm[1 + q] = row, perm[33 + q] = slot of its source.
    if (tid == 0) {
        int cnt = 0;
        auto slot = [&](int32_t rr) {
            for (int q = 0; q < cnt; ++q)
                if (aff[q] == rr) return q;
            aff[cnt] = rr;
            src[cnt] = rr;
            return cnt++;
        };
        for (int32_t jj = 0; jj < w; ++jj) {
            const int qa = slot(k0 + jj), qb = slot(spiv[jj]);
            const int32_t t = src[qa];
            src[qa] = src[qb];
            src[qb] = t;
        }
        perm[0] = cnt;
        for (int q = 0; q < cnt; ++q) {
            perm[1 + q] = aff[q];
            int qs = 0;
            for (int t = 0; t < cnt; ++t)
                if (aff[t] == src[q]) qs = t;
            perm[33 + q] = qs;
        }
    }
}

// Z[q, :] = a[perm row q, :]: the rows touched by the panel's interchanges, staged before anything overwrites them
template <typename T>
__global__ __launch_bounds__(256) void gj_stage_kernel(const T* __restrict__ a, int32_t ld, int32_t m, const int32_t* __restrict__ perm,
                                                       T* __restrict__ Z) {
    const int32_t q = blockIdx.y;
    if (q >= perm[0]) return;
    const int32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < m) Z[(size_t)q * m + c] = a[(size_t)perm[1 + q] * ld + c];
}

// For the columns c outside the panel:  a[i, c] = base(i, c) + sum_j W[i, j] Y[j, c],  where Y[j, :] is pivot row k0 + j
// after the interchanges (read from the staged rows Z), and base is 0 for the pivot rows, the interchanged content
// (from Z) for the other touched rows, and a[i, c] itself elsewhere.  A 256-thread workgroup owns kUpdRows rows; a
// thread walks columns (coalesced), loads the w pivot-row values ONCE and applies them to all kUpdRows rows.
constexpr int kUpdRows = 4;
template <typename T>
__global__ __launch_bounds__(256) void gj_panel_update_kernel(T* __restrict__ a, int32_t ld, int32_t m, int32_t k0, int32_t w,
                                                              const int32_t* __restrict__ perm, const T* __restrict__ Z) {
    __shared__ T Ws[kUpdRows][kPanelW];
    __shared__ int32_t yslot[kPanelW];     // staged row holding pivot row k0 + j
    __shared__ int32_t rslot[kUpdRows];    // staged row holding the new content of my row r, or -1
    const int32_t i0 = blockIdx.x * kUpdRows;
    if (threadIdx.x < kUpdRows * kPanelW) {
        const int r = threadIdx.x / kPanelW, j = threadIdx.x % kPanelW;
        Ws[r][j] = (i0 + r < m && j < w) ? a[(size_t)(i0 + r) * ld + k0 + j] : scalar_traits<T>::zero();
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kPanelW + kUpdRows) {
        const int32_t t = (int32_t)threadIdx.x - 64;
        const int32_t na = perm[0];
        const int32_t row = (t < kPanelW) ? k0 + t : i0 + (t - kPanelW);
        int32_t sl = -1;
        for (int32_t q = 0; q < na; ++q)
            if (perm[1 + q] == row) sl = perm[33 + q];
        if (t < kPanelW) yslot[t] = (t < w) ? sl : 0;
        else rslot[t - kPanelW] = sl;
    }
    __syncthreads();
    // (columns are not unrolled: the W tile and the y values already fill the register budget of two waves per SIMD)
#pragma unroll 1
    for (int32_t c = threadIdx.x; c < m; c += 256) {
        if (c >= k0 && c < k0 + w) continue;
        T y[kPanelW];
#pragma unroll
        for (int j = 0; j < kPanelW; ++j) y[j] = Z[(size_t)yslot[j] * m + c];  // W is zero beyond w, slot 0 is always valid
#pragma unroll
        for (int r = 0; r < kUpdRows; ++r) {
            const int32_t i = i0 + r;
            if (i >= m) break;
            T* e = a + (size_t)i * ld + c;
            T acc;
            if (i >= k0 && i < k0 + w) acc = scalar_traits<T>::zero();
            else if (rslot[r] >= 0) acc = Z[(size_t)rslot[r] * m + c];
            else acc = *e;
#pragma unroll
            for (int j = 0; j < kPanelW; ++j) fma_acc(acc, Ws[r][j], y[j]);
            *e = acc;
        }
    }
}

// undo the row interchanges on the columns of the inverse: thread per row, swaps in reverse order
template <typename T>
__global__ void gj_unpivot_kernel(T* __restrict__ a, int32_t ld, int32_t m, const int32_t* __restrict__ ipiv) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    T* ri = a + (size_t)i * ld;
    for (int32_t k = m - 1; k >= 0; --k) {
        const int32_t p = ipiv[k];
        if (p != k) {
            const T tmp = ri[k];
            ri[k] = ri[p];
            ri[p] = tmp;
        }
    }
}

// ---- solve kernels ------------------------------------------------------------------------------------------------------
// out[r] = rhs[r] - sum over the entries of row r left of (LEFT) / right of its diagonal block of val * x[col]
template <typename MT, typename VT, bool LEFT>
__global__ __launch_bounds__(256) void blu_sparse_kernel(int32_t bs, int32_t be, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                         const int32_t* __restrict__ lsplit, const int32_t* __restrict__ usplit,
                                                         const MT* __restrict__ val, const VT* __restrict__ rhs, const VT* __restrict__ x,
                                                         VT* __restrict__ out) {
    const int32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 15;
    const int32_t r = bs + (gid >> 4);
    if (r >= be) return;
    const int32_t p0 = LEFT ? rp[r] : usplit[r];
    const int32_t p1 = LEFT ? lsplit[r] : rp[r + 1];
    VT acc = scalar_traits<VT>::zero();
    for (int32_t p = p0 + lane; p < p1; p += 16) fma_acc(acc, val[p], x[ci[p]]);
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) {
        if constexpr (sizeof(VT) == 16) {
            acc.re += __shfl_xor(acc.re, s, 64);
            acc.im += __shfl_xor(acc.im, s, 64);
        } else {
            acc += __shfl_xor(acc, s, 64);
        }
    }
    if (lane == 0) out[r] = s_sub(rhs[r], acc);
}

// middle block of the twisted sweep: out[r] = rhs[r] - (entries left AND right of the diagonal block) * x
template <typename MT, typename VT>
__global__ __launch_bounds__(256) void blu_sparse_both_kernel(int32_t bs, int32_t be, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                              const int32_t* __restrict__ lsplit, const int32_t* __restrict__ usplit,
                                                              const MT* __restrict__ val, const VT* __restrict__ rhs, const VT* __restrict__ x,
                                                              VT* __restrict__ out) {
    const int32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 15;
    const int32_t r = bs + (gid >> 4);
    if (r >= be) return;
    VT acc = scalar_traits<VT>::zero();
    for (int32_t p = rp[r] + lane; p < lsplit[r]; p += 16) fma_acc(acc, val[p], x[ci[p]]);
    for (int32_t p = usplit[r] + lane; p < rp[r + 1]; p += 16) fma_acc(acc, val[p], x[ci[p]]);
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) {
        if constexpr (sizeof(VT) == 16) {
            acc.re += __shfl_xor(acc.re, s, 64);
            acc.im += __shfl_xor(acc.im, s, 64);
        } else {
            acc += __shfl_xor(acc, s, 64);
        }
    }
    if (lane == 0) out[r] = s_sub(rhs[r], acc);
}

// out[r] = sum_s Sinv[r, s] in[s] over the whole diagonal block; one wavefront per row, 8 loads in flight per lane
template <typename MT, typename VT>
__global__ __launch_bounds__(256) void blu_dense_kernel(int32_t bs, int32_t be, int32_t ld, const MT* __restrict__ sinv,
                                                        const VT* __restrict__ in, VT* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int32_t r = bs + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= be) return;
    const MT* row = sinv + (size_t)r * ld - bs;
    VT acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = scalar_traits<VT>::zero();
    int32_t s = bs + lane;
    for (; s + 7 * 64 < be; s += 8 * 64) {
        MT a[8];
        VT tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = row[s + k * 64];
            tv[k] = in[s + k * 64];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) fma_acc(acc[k & 3], a[k], tv[k]);
    }
    for (; s < be; s += 64) fma_acc(acc[0], row[s], in[s]);
    VT v = s_add(s_add(acc[0], acc[1]), s_add(acc[2], acc[3]));
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        if constexpr (sizeof(VT) == 16) {
            v.re += __shfl_xor(v.re, m, 64);
            v.im += __shfl_xor(v.im, m, 64);
        } else {
            v += __shfl_xor(v, m, 64);
        }
    }
    if (lane == 0) out[r] = v;
}

template <typename T>
int factorize(lsa_ctx* ctx, lsa_blu* f) {
    const int32_t B = f->B, n = f->n, ld = f->ld;
    const lsa_mat* C = f->C;
    const size_t inv_bytes = (size_t)n * ld * sizeof(T);
    LSA_HIP_CHECK(ctx, hipMemsetAsync(f->sinv, 0, inv_bytes, ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemsetAsync(f->flag, 0, 4 * sizeof(int32_t), ctx->stream));
    // pivots below 1e-12 of the largest entry of C mean a singular Schur block (a singular leading block of C): block
    // elimination without pivoting across blocks cannot continue; the caller then falls back to ILU(k) + GMRES
    double tiny2 = 0.0;
    {
        unsigned long long* dmax = (unsigned long long*)ctx->dscratch;
        LSA_HIP_CHECK(ctx, hipMemsetAsync(dmax, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL((maxabs2_kernel<T>), dim3(std::max(1, std::min(ctx->num_cu * 4, (int)((C->nnz + 255) / 256)))), dim3(256), 0, ctx->stream,
                           C->nnz, (const T*)C->val, dmax);
        unsigned long long hmax = 0;
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(&hmax, dmax, sizeof hmax, hipMemcpyDeviceToHost, ctx->stream));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        double m2;
        memcpy(&m2, &hmax, sizeof m2);
        tiny2 = 1e-24 * m2;
    }
    const size_t lds = (size_t)B * sizeof(T);
    if (lds > 64 * 1024) {
        LSA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blu_corner_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LSA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blu_corner_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // panel width of the blocked Gauss-Jordan (LSA_GJ_PANEL=1 selects the unblocked form: two launches per pivot): 8
    // columns while a thread of the panel kernel holds at most 4 rows, 4 columns up to 8 rows (blocks of 4096 rows)
    int32_t panel_w = B <= 2048 ? 8 : B <= 4096 ? 4 : 1;
    if (const char* e = getenv("LSA_GJ_PANEL")) panel_w = std::max(1, std::min(panel_w, atoi(e)));
    // LSA_BLU_TIMING=2: HIP events around every launch of block 1 (development aid; per-kernel sums go to stderr)
    const bool probe = getenv("LSA_BLU_TIMING") && atoi(getenv("LSA_BLU_TIMING")) >= 2;
    std::vector<std::pair<int, hipEvent_t>> marks;
    auto factor_block = [&](hipStream_t st, int chain, int32_t b, bool corr_left, bool corr_right) {
        const int32_t bs = b * B, be = std::min(n, bs + B), m = be - bs;
        auto mark = [&](int kind) {
            if (!probe || b != 1) return;
            hipEvent_t ev;
            (void)hipEventCreate(&ev);
            (void)hipEventRecord(ev, st);
            marks.push_back({kind, ev});
        };
        mark(-1);
        T* S = (T*)f->sinv + (size_t)bs * ld;
        hipLaunchKernelGGL((blu_scatter_kernel<T>), dim3((m * 16 + 255) / 256), dim3(256), 0, st, bs, be, ld, C->rp, C->ci, f->lsplit, f->usplit,
                           (const T*)C->val, S);
        if (corr_left) {
            const int32_t ns = bs - B;
            hipLaunchKernelGGL((blu_corner_kernel<T, false>), dim3(m), dim3(256), lds, st, ns, bs, bs, be, ld, C->rp, C->ci, f->lsplit, f->usplit,
                               (const T*)C->val, f->cptr, f->crow, f->cpos, (const T*)f->sinv + (size_t)ns * ld, S);
        }
        if (corr_right) {
            const int32_t ns = be, ne = std::min(n, be + B);
            hipLaunchKernelGGL((blu_corner_kernel<T, true>), dim3(m), dim3(256), lds, st, ns, ne, bs, be, ld, C->rp, C->ci, f->lsplit, f->usplit,
                               (const T*)C->val, f->cptr, f->crow, f->cpos, (const T*)f->sinv + (size_t)ns * ld, S);
        }
        mark(0);
        int32_t* ipiv = f->ipiv[chain];
        int32_t* perm = ipiv + B;  // interchange lists of the current panel (80 ints behind the pivot indices)
        T* ws = (T*)f->colbuf[chain];
        if (panel_w >= 2) {
            for (int32_t k0 = 0; k0 < m; k0 += panel_w) {
                const int32_t w = std::min(panel_w, m - k0);
                if (B <= 1024) hipLaunchKernelGGL((gj_panel_kernel<T, 1024, 1, 8>), dim3(1), dim3(1024), 0, st, S, ld, m, k0, w, ipiv, perm, f->flag, tiny2);
                else if (B <= 2048) hipLaunchKernelGGL((gj_panel_kernel<T, 512, 4, 8>), dim3(1), dim3(512), 0, st, S, ld, m, k0, w, ipiv, perm, f->flag, tiny2);
                else hipLaunchKernelGGL((gj_panel_kernel<T, 512, 8, 4>), dim3(1), dim3(512), 0, st, S, ld, m, k0, w, ipiv, perm, f->flag, tiny2);
                mark(1);
                hipLaunchKernelGGL((gj_stage_kernel<T>), dim3((m + 255) / 256, 2 * w), dim3(256), 0, st, (const T*)S, ld, m, (const int32_t*)perm, ws);
                mark(2);
                hipLaunchKernelGGL((gj_panel_update_kernel<T>), dim3((m + kUpdRows - 1) / kUpdRows), dim3(256), 0, st, S, ld, m, k0, w,
                                   (const int32_t*)perm, (const T*)ws);
                mark(3);
            }
        } else {
            for (int32_t k = 0; k < m; ++k) {
                hipLaunchKernelGGL((gj_pivot_kernel<T>), dim3(1), dim3(1024), 0, st, S, ld, m, k, ipiv, ws, f->flag, tiny2);
                hipLaunchKernelGGL((gj_update_kernel<T>), dim3((m + 3) / 4), dim3(256), 0, st, S, ld, m, k, (const T*)ws);
            }
        }
        hipLaunchKernelGGL((gj_unpivot_kernel<T>), dim3((m + 255) / 256), dim3(256), 0, st, S, ld, m, ipiv);
    };
    // twisted order: chain 0 eliminates downwards from block 0, chain 1 upwards from the last block, on two streams;
    // they meet at the middle block, which receives both corrections
    const int32_t nb = f->nb, mid = f->mid;
    LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_fork, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamWaitEvent(f->stream2, f->ev_fork, 0));
    // (the launches of the two chains are enqueued alternately so that neither queue waits for the host)
    const double t_enq = now_s();
    for (int32_t lo = 0, hi = nb - 1; lo < mid || hi > mid; ++lo, --hi) {
        if (lo < mid) factor_block(ctx->stream, 0, lo, lo > 0, false);
        if (hi > mid) factor_block(f->stream2, 1, hi, false, hi < nb - 1);
    }
    LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_join, f->stream2));
    LSA_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, f->ev_join, 0));
    if (nb > 0) factor_block(ctx->stream, 0, mid, mid > 0, mid < nb - 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "block LU launch failed: %s", hipGetErrorString(e));
    const double t_sub = now_s();
    int32_t hflag[4];
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(hflag, f->flag, sizeof hflag, hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (!marks.empty()) {
        double sum[4] = {0, 0, 0, 0};
        int cnt[4] = {0, 0, 0, 0};
        for (size_t i = 1; i < marks.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second);
            sum[marks[i].first] += ms;
            ++cnt[marks[i].first];
        }
        const char* names[4] = {"scatter+corner", "panel", "stage", "update"};
        for (int q = 0; q < 4; ++q)
            fprintf(stderr, "[lsa_blu] block 1 %-14s %4d intervals, %8.1f us each, %7.2f ms total\n", names[q], cnt[q], cnt[q] ? sum[q] * 1e3 / cnt[q] : 0.0, sum[q]);
        for (auto& mk : marks) (void)hipEventDestroy(mk.second);
    }
    if (getenv("LSA_BLU_TIMING"))
        fprintf(stderr, "[lsa_blu] numeric factorisation: host enqueue %.1f ms, device drained after %.1f ms\n", (t_sub - t_enq) * 1e3,
                (now_s() - t_enq) * 1e3);
    if (hflag[1] != 0) return lsa_set_error(ctx, LSA_ERR_ZERO_PIVOT, "block LU: a Schur block is singular (pivot column %d)", hflag[1] - 1);
    return LSA_OK;
}

template <typename MT, typename VT>
int launch_apply(lsa_ctx* ctx, lsa_blu* f, const VT* v, VT* x, VT* y, VT* z, VT* t) {
    const int32_t B = f->B, nb = f->nb, n = f->n, mid = f->mid;
    const lsa_mat* C = f->C;
    // The sweeps' kernels are a few microseconds each: measured at S30k, running the two chains on two queues costs
    // 796 us per apply against 538 us for the same twisted order on one queue (graph replay), so the solve stays on
    // one stream; the factorisation, whose panel kernels occupy one CU for ~50 us, does gain from two (0.35 -> 0.21 s).
    static const bool two_streams = getenv("LSA_BLU_SOLVE_STREAMS") && atoi(getenv("LSA_BLU_SOLVE_STREAMS")) == 2;
    hipStream_t s0 = ctx->stream, s1 = two_streams ? f->stream2 : ctx->stream;
    auto sparse = [&](hipStream_t st, bool left, int32_t b, const VT* rhs, const VT* xin, VT* out) {
        const int32_t bs = b * B, be = std::min(n, bs + B), m = be - bs;
        if (left)
            hipLaunchKernelGGL((blu_sparse_kernel<MT, VT, true>), dim3((m * 16 + 255) / 256), dim3(256), 0, st, bs, be, C->rp, C->ci, f->lsplit,
                               f->usplit, (const MT*)C->val, rhs, xin, out);
        else
            hipLaunchKernelGGL((blu_sparse_kernel<MT, VT, false>), dim3((m * 16 + 255) / 256), dim3(256), 0, st, bs, be, C->rp, C->ci, f->lsplit,
                               f->usplit, (const MT*)C->val, rhs, xin, out);
    };
    auto dense = [&](hipStream_t st, int32_t b, const VT* in, VT* out) {
        const int32_t bs = b * B, be = std::min(n, bs + B), m = be - bs;
        hipLaunchKernelGGL((blu_dense_kernel<MT, VT>), dim3((m + 3) / 4), dim3(256), 0, st, bs, be, f->ld, (const MT*)f->sinv, in, out);
    };
    // elimination towards the middle, two chains in parallel:  y_b = v_b - C_{b,b-+1} z_{b-+1},  z_b = Sinv_b y_b
    LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_fork, s0));
    LSA_HIP_CHECK(ctx, hipStreamWaitEvent(s1, f->ev_fork, 0));
    // (the two chains are enqueued alternately so that neither queue waits for the host)
    for (int32_t k = 0; k < std::max(mid, nb - 1 - mid); ++k) {
        const int32_t bt = k, bb = nb - 1 - k;
        if (bt < mid) {
            sparse(s0, true, bt, v, (const VT*)z, y);
            dense(s0, bt, (const VT*)y, z);
        }
        if (bb > mid) {
            sparse(s1, false, bb, v, (const VT*)z, y);
            dense(s1, bb, (const VT*)y, z);
        }
    }
    LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_join, s1));
    LSA_HIP_CHECK(ctx, hipStreamWaitEvent(s0, f->ev_join, 0));
    // middle block: x_m = Sinv_m (v_m - C_{m,m-1} z_{m-1} - C_{m,m+1} z_{m+1})
    {
        const int32_t bs = mid * B, be = std::min(n, bs + B), m = be - bs;
        hipLaunchKernelGGL((blu_sparse_both_kernel<MT, VT>), dim3((m * 16 + 255) / 256), dim3(256), 0, s0, bs, be, C->rp, C->ci, f->lsplit,
                           f->usplit, (const MT*)C->val, v, (const VT*)z, t);
        dense(s0, mid, (const VT*)t, x);
    }
    // substitution outwards, two chains in parallel:  x_b = Sinv_b (y_b - C_{b,b+-1} x_{b+-1})
    LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_fork, s0));
    LSA_HIP_CHECK(ctx, hipStreamWaitEvent(s1, f->ev_fork, 0));
    for (int32_t k = 1; k <= std::max(mid, nb - 1 - mid); ++k) {
        const int32_t bt = mid - k, bb = mid + k;
        if (bt >= 0) {
            sparse(s0, false, bt, (const VT*)y, (const VT*)x, t);
            dense(s0, bt, (const VT*)t, x);
        }
        if (bb < nb) {
            sparse(s1, true, bb, (const VT*)y, (const VT*)x, t);
            dense(s1, bb, (const VT*)t, x);
        }
    }
    LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_join, s1));
    LSA_HIP_CHECK(ctx, hipStreamWaitEvent(s0, f->ev_join, 0));
    return LSA_OK;
}

template <typename MT, typename VT>
int apply_typed(lsa_ctx* ctx, lsa_blu* f, const void* b, void* x) {
    const int vd = scalar_traits<VT>::dtype;
    const size_t vb = (size_t)std::max<int32_t>(f->n, 1) * sizeof(VT);
    if (!f->t[vd]) {
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (void** p : {&f->t[vd], &f->y[vd], &f->z[vd], &f->in[vd], &f->out[vd]}) {
            LSA_HIP_CHECK(ctx, hipMalloc(p, vb));
            LSA_HIP_CHECK(ctx, hipMemsetAsync(*p, 0, vb, ctx->stream));
        }
    }
    VT *t = (VT*)f->t[vd], *y = (VT*)f->y[vd], *z = (VT*)f->z[vd], *in = (VT*)f->in[vd], *out = (VT*)f->out[vd];
    static const bool use_graph = !(getenv("LSA_SPTRSV_GRAPH") && atoi(getenv("LSA_SPTRSV_GRAPH")) == 0);
    if (!use_graph) {
        LSA_CHECK((launch_apply<MT, VT>(ctx, f, (const VT*)b, (VT*)x, y, z, t)));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "block LU solve launch failed: %s", hipGetErrorString(le));
        return LSA_OK;
    }
    if (!f->graph[vd]) {
        hipGraph_t graph = nullptr;
        LSA_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        const int crc = launch_apply<MT, VT>(ctx, f, in, out, y, z, t);
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (crc != LSA_OK) return crc;
        if (e != hipSuccess || !graph) return lsa_set_error(ctx, LSA_ERR_HIP, "block LU: graph capture failed: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "block LU: graph instantiate failed: %s", hipGetErrorString(e));
        f->graph[vd] = (void*)exec;
    }
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(in, b, (size_t)f->n * sizeof(VT), hipMemcpyDeviceToDevice, ctx->stream));
    LSA_HIP_CHECK(ctx, hipGraphLaunch((hipGraphExec_t)f->graph[vd], ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(x, out, (size_t)f->n * sizeof(VT), hipMemcpyDeviceToDevice, ctx->stream));
    return LSA_OK;
}

}  // namespace

int blu_solve_dev(lsa_ctx* ctx, lsa_blu* f, int vdtype, const void* b, void* x) {
    if (b == x) return lsa_set_error(ctx, LSA_ERR_ARG, "blu_solve: b and x must not alias");
    if (f->dtype == LSA_F64 && vdtype == LSA_F64) return apply_typed<double, double>(ctx, f, b, x);
    if (f->dtype == LSA_F64 && vdtype == LSA_C128) return apply_typed<double, cplx>(ctx, f, b, x);
    if (f->dtype == LSA_C128 && vdtype == LSA_C128) return apply_typed<cplx, cplx>(ctx, f, b, x);
    return lsa_set_error(ctx, LSA_ERR_ARG, "blu_solve: complex factors need complex vectors");
}

extern "C" {

void lsa_blu_destroy(lsa_blu* f) {
    if (!f) return;
    if (f->ctx && f->ctx->stream) (void)hipStreamSynchronize(f->ctx->stream);
    for (int vd = 0; vd < 2; ++vd) {
        if (f->graph[vd]) (void)hipGraphExecDestroy((hipGraphExec_t)f->graph[vd]);
        for (void* p : {f->t[vd], f->y[vd], f->z[vd], f->in[vd], f->out[vd]})
            if (p) (void)hipFree(p);
    }
    if (f->stream2) (void)hipStreamSynchronize(f->stream2);
    for (void* p : {(void*)f->lsplit, (void*)f->usplit, (void*)f->cptr, (void*)f->crow, (void*)f->cpos, f->sinv, (void*)f->ipiv[0], (void*)f->ipiv[1],
                    f->colbuf[0], f->colbuf[1], (void*)f->flag})
        if (p) (void)hipFree(p);
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    if (f->ev_join) (void)hipEventDestroy(f->ev_join);
    if (f->stream2) (void)hipStreamDestroy(f->stream2);
    delete f;
}

int lsa_blu_create(lsa_ctx* ctx, const lsa_mat* C, int32_t block_size, lsa_blu** out) {
    if (!ctx || !C || !out) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_blu_create: null argument");
    if (C->n != C->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_blu_create: the matrix must be square");
    const double t0 = now_s();
    const int32_t n = C->n;
    int32_t bw = 0;
    for (int32_t i = 0; i < n; ++i)
        for (int32_t p = C->h_rp[i]; p < C->h_rp[i + 1]; ++p) bw = std::max(bw, std::abs(C->h_ci[p] - i));
    int32_t B = std::max(block_size > 0 ? block_size : 1024, bw + 1);
    B = ((B + 255) / 256) * 256;
    const size_t esz = C->dtype == LSA_C128 ? 16 : 8;
    const int32_t ld = B + 16;
    const size_t inv_bytes = (size_t)std::max(n, 1) * ld * esz;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (inv_bytes > free_b / 2)
        return lsa_set_error(ctx, LSA_ERR_HIP, "block LU: bandwidth %d needs %.1f GB of inverted Schur blocks (%.1f GB free); use ILU(k)", bw,
                             inv_bytes / 1e9, free_b / 1e9);
    lsa_blu* f = new lsa_blu();
    f->ctx = ctx;
    f->C = C;
    f->n = n;
    f->B = B;
    f->ld = ld;
    f->nb = n > 0 ? (n + B - 1) / B : 0;
    f->mid = f->nb / 2;
    if (const char* e = getenv("LSA_BLU_TWIST"))  // 0: one chain from the first block (the middle block is the last one)
        if (atoi(e) == 0) f->mid = f->nb > 0 ? f->nb - 1 : 0;
    f->bandwidth = bw;
    f->dtype = C->dtype;
    // splits on C's pattern and a CSC view (positions into C's value array) for the corner update
    std::vector<int32_t> ls((size_t)n), us((size_t)n), cptr((size_t)n + 1, 0), crow((size_t)C->nnz), cpos((size_t)C->nnz);
    const int32_t* c = C->h_ci.data();
    for (int32_t r = 0; r < n; ++r) {
        const int32_t bs = (r / B) * B, be = std::min(n, bs + B);
        ls[r] = (int32_t)(std::lower_bound(c + C->h_rp[r], c + C->h_rp[r + 1], bs) - c);
        us[r] = (int32_t)(std::lower_bound(c + C->h_rp[r], c + C->h_rp[r + 1], be) - c);
        for (int32_t p = C->h_rp[r]; p < C->h_rp[r + 1]; ++p) ++cptr[(size_t)c[p] + 1];
    }
    for (int32_t j = 0; j < n; ++j) cptr[(size_t)j + 1] += cptr[j];
    {
        std::vector<int32_t> cur(cptr.begin(), cptr.end() - 1);
        for (int32_t r = 0; r < n; ++r)
            for (int32_t p = C->h_rp[r]; p < C->h_rp[r + 1]; ++p) {
                const int32_t q = cur[c[p]]++;
                crow[q] = r;
                cpos[q] = p;
            }
    }
    const size_t n1 = (size_t)std::max(n, 1), z1 = (size_t)std::max<int64_t>(C->nnz, 1);
    bool ok = hipMalloc((void**)&f->lsplit, 4 * n1) == hipSuccess && hipMalloc((void**)&f->usplit, 4 * n1) == hipSuccess &&
              hipMalloc((void**)&f->cptr, 4 * (n1 + 1)) == hipSuccess && hipMalloc((void**)&f->crow, 4 * z1) == hipSuccess &&
              hipMalloc((void**)&f->cpos, 4 * z1) == hipSuccess && hipMalloc(&f->sinv, inv_bytes) == hipSuccess &&
              hipMalloc((void**)&f->ipiv[0], 4 * ((size_t)B + 128)) == hipSuccess && hipMalloc((void**)&f->ipiv[1], 4 * ((size_t)B + 128)) == hipSuccess &&
              hipMalloc(&f->colbuf[0], esz * (size_t)B * 48) == hipSuccess && hipMalloc(&f->colbuf[1], esz * (size_t)B * 48) == hipSuccess &&
              hipMalloc((void**)&f->flag, 16) == hipSuccess && hipStreamCreateWithFlags(&f->stream2, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming) == hipSuccess;
    hipStream_t s = ctx->stream;
    ok = ok && hipMemcpyAsync(f->lsplit, ls.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->usplit, us.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->cptr, cptr.data(), 4 * ((size_t)n + 1), hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->crow, crow.data(), 4 * (size_t)C->nnz, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->cpos, cpos.data(), 4 * (size_t)C->nnz, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) {
        lsa_blu_destroy(f);
        return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_blu_create: out of device memory");
    }
    if (getenv("LSA_BLU_TIMING")) fprintf(stderr, "[lsa_blu] symbolic setup + allocation + upload %.1f ms\n", (now_s() - t0) * 1e3);
    int rc = f->dtype == LSA_C128 ? factorize<cplx>(ctx, f) : factorize<double>(ctx, f);
    if (rc != LSA_OK) {
        lsa_blu_destroy(f);
        return rc;
    }
    // the CSC view is only needed by the factorisation
    for (int32_t** p : {&f->cptr, &f->crow, &f->cpos}) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    f->seconds = now_s() - t0;
    *out = f;
    return LSA_OK;
}

int lsa_blu_solve(lsa_ctx* ctx, lsa_blu* f, const lsa_vec* b, lsa_vec* x) {
    if (!ctx || !f || !b || !x) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_blu_solve: null argument");
    if (b->n != f->n || x->n != f->n || b->dtype != x->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_blu_solve: vector shape/dtype mismatch");
    LSA_CHECK(blu_solve_dev(ctx, f, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

int lsa_blu_solve_time(lsa_ctx* ctx, lsa_blu* f, const lsa_vec* b, lsa_vec* x, int iters, double* avg_ms) {
    if (!ctx || !f || !b || !x || !avg_ms || iters <= 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_blu_solve_time: bad argument");
    if (b->n != f->n || x->n != f->n || b->dtype != x->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_blu_solve_time: vector shape/dtype mismatch");
    LSA_CHECK(blu_solve_dev(ctx, f, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; ++i) LSA_CHECK(blu_solve_dev(ctx, f, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    LSA_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    LSA_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / iters;
    return LSA_OK;
}

int lsa_blu_info(const lsa_blu* f, int32_t* block_size, int32_t* nblocks, int32_t* bandwidth, double* seconds) {
    if (!f) return LSA_ERR_ARG;
    if (block_size) *block_size = f->B;
    if (nblocks) *nblocks = f->nb;
    if (bandwidth) *bandwidth = f->bandwidth;
    if (seconds) *seconds = f->seconds;
    return LSA_OK;
}

}  // extern "C"
