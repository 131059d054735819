// Device-resident Krylov loops: restarted GMRES (the ST's inner KSP), the shift-invert operator, and the
// Arnoldi recurrences + basis updates that Krylov-Schur needs (EPS.solve of the reference, Solver/utils.py:270).
// Only O(m) scalars per step cross PCIe (one Hessenberg column); vectors never leave HBM.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>

#include "lsa_internal.h"

int ilu_check_abort(lsa_ctx* ctx, lsa_ilu* pc);
int k_allgather_inplace(lsa_ctx* ctx, void* vec, size_t bytes_per_rank);  // comm.hip

namespace {

using zc = std::complex<double>;
inline size_t esize(int dtype) { return dtype == LSA_C128 ? 16 : 8; }
inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// small device arrays used by the orthogonalisation (sized for `cap` basis vectors)
struct OrthWork {
    void *h1 = nullptr, *h2 = nullptr, *hcol = nullptr;
    double* nrm2 = nullptr;
    void* fused = nullptr;  // workspace of the five-launch CGS2 (k_cgs2_fused), allocated on first use
    int cap = 0;
    int ensure_fused(lsa_ctx* ctx, int64_t n) {
        if (fused) return LSA_OK;
        LSA_HIP_ALLOC(ctx, hipMalloc(&fused, k_cgs2_fused_work_bytes(ctx, n, cap)));
        return LSA_OK;
    }
    int alloc(lsa_ctx* ctx, int cap_, int dtype) {
        cap = cap_;
        const size_t b = (size_t)(cap + 2) * esize(dtype);
        LSA_HIP_CHECK(ctx, hipMalloc(&h1, b));
        LSA_HIP_CHECK(ctx, hipMalloc(&h2, b));
        LSA_HIP_CHECK(ctx, hipMalloc(&hcol, b));
        LSA_HIP_CHECK(ctx, hipMalloc((void**)&nrm2, 4 * sizeof(double)));
        return LSA_OK;
    }
    void release() {
        for (void* p : {h1, h2, hcol, (void*)nrm2, fused})
            if (p) (void)hipFree(p);
        h1 = h2 = hcol = fused = nullptr;
        nrm2 = nullptr;
    }
};

// the device part of CGS2: orthogonalise w against V[:, 0:j] and normalise into vnext; the j+1 Hessenberg entries go to
// `hcol_dev`, nothing is read back
int orthonormalize_enqueue(lsa_ctx* ctx, int dtype, int64_t n, const void* V, int64_t ldv, int j, void* w, void* vnext, OrthWork& ow,
                           void* hcol_dev) {
    static const bool fuse = !(getenv("LSA_KRYLOV_FUSED") && atoi(getenv("LSA_KRYLOV_FUSED")) == 0);
    if (fuse) {  // five launches instead of nine where the shape allows it (same arithmetic as the batched Arnoldi steps)
        LSA_CHECK(ow.ensure_fused(ctx, n));
        const int frc = k_cgs2_fused(ctx, dtype, n, j, V, ldv, w, vnext, hcol_dev, ow.fused, nullptr, nullptr, nullptr);
        if (frc <= 0) return frc;
    }
    LSA_CHECK(k_multi_dot(ctx, dtype, n, j, V, ldv, w, ow.h1));
    // long vectors: the first projection and the second dot product share one pass over the basis (three passes instead of four)
    static const bool three = !(getenv("LSA_KRYLOV_PASSES") && atoi(getenv("LSA_KRYLOV_PASSES")) == 4);
    const int arc = three ? k_multi_axpy_dot(ctx, dtype, n, j, V, ldv, ow.h1, w, ow.h2) : 1;
    if (arc < 0) return arc;
    if (arc > 0) {
        LSA_CHECK(k_multi_axpy(ctx, dtype, n, j, V, ldv, ow.h1, w, nullptr));
        LSA_CHECK(k_multi_dot(ctx, dtype, n, j, V, ldv, w, ow.h2));
    }
    LSA_CHECK(k_multi_axpy(ctx, dtype, n, j, V, ldv, ow.h2, w, ow.nrm2));
    LSA_CHECK(k_hess_column(ctx, dtype, j, ow.h1, ow.h2, ow.nrm2, hcol_dev));
    LSA_CHECK(k_scale_by_inv_norm(ctx, dtype, n, w, ow.nrm2, vnext));
    return LSA_OK;
}

// CGS2 with the Hessenberg entries brought to the host as complex numbers (real dtype is widened).  One stream
// synchronisation.
int orthonormalize(lsa_ctx* ctx, int dtype, int64_t n, const void* V, int64_t ldv, int j, void* w, void* vnext,
                   OrthWork& ow, zc* h_host) {
    LSA_CHECK(orthonormalize_enqueue(ctx, dtype, n, V, ldv, j, w, vnext, ow, ow.hcol));
    const size_t bytes = (size_t)(j + 1) * esize(dtype);
    LSA_CHECK(lsa_ensure_scratch(ctx, 0, bytes));
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->pinned, ow.hcol, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (dtype == LSA_C128) {
        const cplx* p = (const cplx*)ctx->pinned;
        for (int c = 0; c <= j; ++c) h_host[c] = zc(p[c].re, p[c].im);
    } else {
        const double* p = (const double*)ctx->pinned;
        for (int c = 0; c <= j; ++c) h_host[c] = zc(p[c], 0.0);
    }
    return LSA_OK;
}

int device_norm(lsa_ctx* ctx, int dtype, int64_t n, const void* x, double* nrm2_dev, double* out) {
    LSA_CHECK(k_nrm2(ctx, dtype, n, x, nrm2_dev));
    LSA_CHECK(lsa_ensure_scratch(ctx, 0, sizeof(double)));
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->pinned, nrm2_dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = std::sqrt(*(const double*)ctx->pinned);
    return LSA_OK;
}

// upload a small host array of complex numbers as `dtype`
int upload_small(lsa_ctx* ctx, int dtype, const zc* src, size_t count, void* dst) {
    LSA_CHECK(lsa_ensure_scratch(ctx, 0, count * 16));
    if (dtype == LSA_C128) {
        cplx* p = (cplx*)ctx->pinned;
        for (size_t i = 0; i < count; ++i) p[i] = cplx{src[i].real(), src[i].imag()};
    } else {
        double* p = (double*)ctx->pinned;
        for (size_t i = 0; i < count; ++i) p[i] = src[i].real();
    }
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(dst, ctx->pinned, count * esize(dtype), hipMemcpyHostToDevice, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // pinned buffer is reused
    return LSA_OK;
}

// y = A x for a (possibly row-sharded) matrix; x and y are global-length vectors replicated on every rank: the
// shard writes its own rows, then the equal-sized padded blocks are exchanged with one in-place all-gather
int spmv_global(lsa_ctx* ctx, const lsa_mat* A, int dtype, const void* x, void* y, bool adjoint = false) {
    // y = A^H x with the whole matrix (lsa_op_set_adjoint sees to that on every rank); the pull-form product is reproducible bit for
    // bit, so the ranks' replicated vectors stay alike without an exchange
    if (adjoint) return k_spmv_transpose(ctx, A, 1, dtype, x, y);
    const size_t es = esize(dtype);
    LSA_CHECK(k_spmv(ctx, A, dtype, x, (char*)y + (size_t)A->row0 * es));
    // (a whole square matrix was multiplied on every rank: nothing to exchange)
    if (ctx->nranks > 1 && A->n != A->ncols) LSA_CHECK(k_allgather_inplace(ctx, y, (size_t)(A->ncols / ctx->nranks) * es));
    return LSA_OK;
}

// x = P^-1 b: block-Jacobi over ranks (each rank holds the ILU(k) of its diagonal block), then all-gather
struct PcRef {
    lsa_ilu* ilu = nullptr;  // ILU(k) + triangular solves
    lsa_ndlu* nd = nullptr;  // exact nested-dissection multifrontal LU
    bool nd_dist = false;    // the subtree-parallel form: reads and writes whole replicated vectors
    bool adjoint = false;    // solve with C^H on the same factors (nd only)
    double normF = 0.0;      // ||C||_F when known: lets a direct solve be judged by its backward error
    explicit operator bool() const { return ilu || nd; }
    bool exact() const { return nd != nullptr; }
};

int pc_global(lsa_ctx* ctx, PcRef pc, int32_t row0, int64_t nglobal, int dtype, const void* b, void* x) {
    const size_t es = esize(dtype);
    const char* bl = (const char*)b + (size_t)row0 * es;
    char* xl = (char*)x + (size_t)row0 * es;
    if (pc.nd && pc.adjoint) LSA_CHECK(ndlu_solve_adjoint_dev(ctx, pc.nd, 1, dtype, pc.nd_dist ? b : bl, pc.nd_dist ? x : xl));
    else if (pc.nd && pc.nd_dist) LSA_CHECK(ndlu_solve_dev(ctx, pc.nd, dtype, b, x));  // own subtrees + replicated top; x completed below
    else if (pc.nd) LSA_CHECK(ndlu_solve_dev(ctx, pc.nd, dtype, bl, xl));
    else LSA_CHECK(ilu_solve_dev(ctx, pc.ilu, 2, dtype, bl, xl));
    if (ctx->nranks > 1) LSA_CHECK(k_allgather_inplace(ctx, x, (size_t)(nglobal / ctx->nranks) * es));
    return LSA_OK;
}

struct GmresWork {
    int dtype = LSA_C128;
    int64_t n = 0;
    int restart = 0;
    void *V = nullptr, *w = nullptr, *z = nullptr, *ydev = nullptr;
    OrthWork ow;
    std::vector<zc> H, cs, sn, g, y, hcol;
    int alloc(lsa_ctx* ctx, int64_t n_, int restart_, int dtype_) {
        n = n_;
        restart = restart_;
        dtype = dtype_;
        const size_t vb = (size_t)std::max<int64_t>(n, 1) * esize(dtype);
        // (the basis itself is allocated when an iteration actually starts: behind an exact LU solve that is the exception,
        //  and 41 vectors are 330 MB at 500 k unknowns, per operator, i.e. per solve)
        LSA_HIP_CHECK(ctx, hipMalloc(&w, vb));
        LSA_HIP_CHECK(ctx, hipMalloc(&z, vb));
        // shards write only their own rows: the padding rows of the block layout must read as zero
        LSA_HIP_CHECK(ctx, hipMemsetAsync(w, 0, vb, ctx->stream));
        LSA_HIP_CHECK(ctx, hipMemsetAsync(z, 0, vb, ctx->stream));
        LSA_HIP_CHECK(ctx, hipMalloc(&ydev, (size_t)(restart + 2) * esize(dtype)));
        LSA_CHECK(ow.alloc(ctx, restart + 1, dtype));
        H.assign((size_t)(restart + 1) * restart, zc(0));
        cs.assign(restart + 1, zc(0));
        sn.assign(restart + 1, zc(0));
        g.assign(restart + 2, zc(0));
        y.assign(restart + 1, zc(0));
        hcol.assign(restart + 2, zc(0));
        return LSA_OK;
    }
    int ensure_basis(lsa_ctx* ctx) {
        if (V) return LSA_OK;
        const size_t vb = (size_t)std::max<int64_t>(n, 1) * esize(dtype);
        LSA_HIP_ALLOC(ctx, hipMalloc(&V, vb * (size_t)(restart + 1)));
        return LSA_OK;
    }
    void release() {
        for (void* p : {V, w, z, ydev})
            if (p) (void)hipFree(p);
        V = w = z = ydev = nullptr;
        ow.release();
    }
};

// right-preconditioned restarted GMRES on the device; x0 = x when use_x0, else 0
int gmres_run(lsa_ctx* ctx, const lsa_mat* C, PcRef pc, int dtype, const void* b, void* x, bool use_x0, double rtol,
              int maxit, GmresWork& W, int32_t* iters_out, double* relres_out, lsa_stats* st) {
    const int64_t n = W.n;
    const int m = W.restart;
    const size_t vb = (size_t)n * esize(dtype);
    auto col = [&](int j) { return (void*)((char*)W.V + (size_t)j * vb); };
    double bnorm = 0.0;
    double beta0 = -1.0;  // ||b - C x|| of the first cycle when it was computed together with ||b||
    int total = 0;
    double relres = 0.0;
    if (pc.exact() && !use_x0) {
        // exact (LU) preconditioner: x = P^-1 b is already the solution on one GPU; the loop below then only
        // checks b - C x and iterates on the residual when the factors are block-Jacobi over ranks.  The check and
        // ||b|| come from one fused pass and one stream synchronisation.
        LSA_CHECK(pc_global(ctx, pc, C->row0, n, dtype, b, x));
        if (st) st->sptrsv_calls += 2;
        LSA_CHECK(spmv_global(ctx, C, dtype, x, W.z, pc.adjoint));
        if (st) ++st->spmv_calls;
        LSA_CHECK(k_residual_norms(ctx, dtype, n, b, W.z, W.w, W.ow.nrm2));
        LSA_CHECK(lsa_ensure_scratch(ctx, 0, 2 * sizeof(double)));
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->pinned, W.ow.nrm2, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        beta0 = std::sqrt(((const double*)ctx->pinned)[0]);
        bnorm = std::sqrt(((const double*)ctx->pinned)[1]);
        use_x0 = true;
        // Iterative refinement before any looser judgement: the residual b - C x is on the device already (W.w), one more
        // pair of sweeps and one product give x += C^-1 (b - C x).  Large 3D factorisations leave ||b - C x|| / ||b|| at 1e-11
        // (growth over tens of thousands of pivots per front); one step brings that to rounding level.  Repeated while it
        // helps (at most twice); what is left after that is the conditioning of C itself (a shift next to an eigenvalue).
        for (int pass = 0; pass < 2 && beta0 > rtol * bnorm && std::isfinite(beta0) && bnorm > 0.0; ++pass) {
            LSA_CHECK(pc_global(ctx, pc, C->row0, n, dtype, W.w, W.z));
            const double one[2] = {1.0, 0.0};
            LSA_CHECK(k_axpy(ctx, dtype, n, one, W.z, x));
            LSA_CHECK(spmv_global(ctx, C, dtype, x, W.z, pc.adjoint));
            LSA_CHECK(k_residual_norms(ctx, dtype, n, b, W.z, W.w, W.ow.nrm2));
            LSA_HIP_CHECK(ctx, hipMemcpyAsync(ctx->pinned, W.ow.nrm2, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const double before = beta0;
            beta0 = std::sqrt(((const double*)ctx->pinned)[0]);
            if (st) {
                st->sptrsv_calls += 2;
                ++st->spmv_calls;
                if (pass == 0) ++st->refined_solves;
            }
            if (!(beta0 < 0.5 * before)) break;  // no longer helping: x is as good as these factors make it
        }
        if (beta0 > rtol * bnorm && pc.normF > 0.0 && std::isfinite(beta0)) {
            // A direct solve is judged by its backward error.  For a shift next to an eigenvalue ||x|| >> ||b|| / ||C|| and
            // ||b - C x|| / ||b|| cannot go below eps ||C|| ||x|| / ||b||, whatever the solver (the reference shifts the
            // adjoint problem exactly at a converged eigenvalue, Sensitivity/__init__.py:260-262).  x is accepted when it
            // solves a system within 1e-12 ||C||_F of C exactly; GMRES could not improve on that.
            double xnorm = 0.0;
            LSA_CHECK(device_norm(ctx, dtype, n, x, W.ow.nrm2, &xnorm));
            if (beta0 <= 1e-12 * pc.normF * xnorm) {
                if (st) {
                    ++st->backward_accepted;
                    st->last_rel_res = beta0 / bnorm;
                    st->max_rel_res = std::max(st->max_rel_res, beta0 / bnorm);  // (reported as it is: the Python layer warns above 10 rtol)
                }
                if (iters_out) *iters_out = 0;
                if (relres_out) *relres_out = beta0 / bnorm;
                return LSA_OK;
            }
        }
    } else {
        LSA_CHECK(device_norm(ctx, dtype, n, b, W.ow.nrm2, &bnorm));
    }
    if (!std::isfinite(bnorm)) return lsa_set_error(ctx, LSA_ERR_NONFINITE, "GMRES: right-hand side is not finite");
    if (!use_x0) LSA_CHECK(k_set_zero(ctx, dtype, n, x));
    if (bnorm == 0.0) {
        LSA_CHECK(k_set_zero(ctx, dtype, n, x));
        if (iters_out) *iters_out = 0;
        if (relres_out) *relres_out = 0.0;
        return LSA_OK;
    }
    bool converged = false;
    bool first_cycle = true;
    int verified_cycles = 0;  // cycles that ended on the recurrence estimate and were then checked against b - C x
    bool pending = false;     // a cycle claimed convergence; the claim still has to be checked
    double last_verified = 1e300;
    while (!converged && (total < maxit || pending)) {
        // r = b - C x  -> w
        if (beta0 >= 0.0) {
            // already in W.w
        } else if (first_cycle && !use_x0) {
            LSA_CHECK(k_copy(ctx, dtype, n, b, W.w));
        } else {
            LSA_CHECK(spmv_global(ctx, C, dtype, x, W.z, pc.adjoint));
            if (st) ++st->spmv_calls;
            LSA_CHECK(k_copy(ctx, dtype, n, b, W.w));
            const double minus1[2] = {-1.0, 0.0};
            LSA_CHECK(k_axpy(ctx, dtype, n, minus1, W.z, W.w));
        }
        first_cycle = false;
        double beta = beta0;
        if (beta0 < 0.0) LSA_CHECK(device_norm(ctx, dtype, n, W.w, W.ow.nrm2, &beta));
        beta0 = -1.0;
        if (!std::isfinite(beta)) return lsa_set_error(ctx, LSA_ERR_NONFINITE, "GMRES: residual is not finite");
        relres = beta / bnorm;
        pending = false;
        // The true residual decides.  After a cycle that claimed convergence, accept the rounding floor of b - C x:
        // for a shift close to an eigenvalue ||x|| >> ||b|| / ||C|| and the attainable ||r|| / ||b|| is
        // eps * ||C|| ||x|| / ||b||, above a tight rtol.  A residual that no longer halves from one verified cycle to
        // the next has reached that floor (a direct solver does no better); it is reported through the statistics.
        if (relres <= rtol || (verified_cycles > 0 && relres <= 10.0 * rtol)) {
            converged = true;
            break;
        }
        if (verified_cycles >= 2 && relres > 0.5 * last_verified) {
            // stagnation at the rounding floor of b - C x.  Accepted within 1000 rtol (reported through max_rel_res and
            // the stagnation counter, which the Python layer turns into a warning); anything looser is a failed solve.
            converged = relres <= 1e3 * rtol;
            if (converged && st) ++st->stagnated_solves;
            break;
        }
        if (verified_cycles > 0) last_verified = relres;
        LSA_CHECK(W.ensure_basis(ctx));
        LSA_CHECK(k_scale_by_inv_norm(ctx, dtype, n, W.w, W.ow.nrm2, col(0)));
        std::fill(W.g.begin(), W.g.end(), zc(0));
        W.g[0] = zc(beta, 0);
        int jj = 0;
        for (int j = 0; j < m && total < maxit; ++j) {
            const void* vj = col(j);
            const void* src = vj;
            if (pc) {
                LSA_CHECK(pc_global(ctx, pc, C->row0, n, dtype, vj, W.z));
                if (st) st->sptrsv_calls += 2;
                src = W.z;
            }
            LSA_CHECK(spmv_global(ctx, C, dtype, src, W.w, pc.adjoint));
            if (st) ++st->spmv_calls;
            LSA_CHECK(orthonormalize(ctx, dtype, n, W.V, n, j + 1, W.w, col(j + 1), W.ow, W.hcol.data()));
            ++total;
            zc* Hj = &W.H[(size_t)j * (m + 1)];
            for (int i = 0; i <= j + 1; ++i) Hj[i] = W.hcol[i];
            if (!std::isfinite(Hj[j + 1].real())) {
                if (pc.ilu) LSA_CHECK(ilu_check_abort(ctx, pc.ilu));
                return lsa_set_error(ctx, LSA_ERR_NONFINITE, "GMRES: non-finite Hessenberg entry at iteration %d", total);
            }
            for (int i = 0; i < j; ++i) {
                const zc t = std::conj(W.cs[i]) * Hj[i] + std::conj(W.sn[i]) * Hj[i + 1];
                Hj[i + 1] = -W.sn[i] * Hj[i] + W.cs[i] * Hj[i + 1];
                Hj[i] = t;
            }
            // rotation that zeroes Hj[j+1]
            const double a = std::abs(Hj[j]), bb = std::abs(Hj[j + 1]);
            const double rr = std::hypot(a, bb);
            if (rr == 0.0) {
                W.cs[j] = zc(1);
                W.sn[j] = zc(0);
            } else {
                W.cs[j] = Hj[j] / rr;
                W.sn[j] = Hj[j + 1] / rr;
            }
            Hj[j] = std::conj(W.cs[j]) * Hj[j] + std::conj(W.sn[j]) * Hj[j + 1];
            Hj[j + 1] = zc(0);
            W.g[j + 1] = -W.sn[j] * W.g[j];
            W.g[j] = std::conj(W.cs[j]) * W.g[j];
            jj = j + 1;
            relres = std::abs(W.g[j + 1]) / bnorm;
            if (relres <= rtol) {  // recurrence estimate: verified against b - C x at the top of the next cycle
                ++verified_cycles;
                pending = true;
                break;
            }
        }
        // y = R^-1 g ;  x += P^-1 (V y)
        for (int i = jj - 1; i >= 0; --i) {
            zc s = W.g[i];
            for (int k = i + 1; k < jj; ++k) s -= W.H[(size_t)k * (m + 1) + i] * W.y[k];
            W.y[i] = s / W.H[(size_t)i * (m + 1) + i];
        }
        if (jj > 0) {
            LSA_CHECK(upload_small(ctx, dtype, W.y.data(), (size_t)jj, W.ydev));
            LSA_CHECK(k_basis_gemm(ctx, dtype, n, jj, 1, W.V, n, W.ydev, jj, W.w, n));
            const double one[2] = {1.0, 0.0};
            if (pc) {
                LSA_CHECK(pc_global(ctx, pc, C->row0, n, dtype, W.w, W.z));
                if (st) st->sptrsv_calls += 2;
                LSA_CHECK(k_axpy(ctx, dtype, n, one, W.z, x));
            } else {
                LSA_CHECK(k_axpy(ctx, dtype, n, one, W.w, x));
            }
        }
    }
    if (pc.ilu) LSA_CHECK(ilu_check_abort(ctx, pc.ilu));
    if (iters_out) *iters_out = total;
    if (relres_out) *relres_out = relres;
    if (st) {
        st->gmres_iters += total;
        st->last_rel_res = relres;
        st->max_rel_res = std::max(st->max_rel_res, relres);
    }
    if (!converged)
        return lsa_set_error(ctx, LSA_ERR_DIVERGED, "GMRES did not reach rtol %.1e in %d iterations (relative residual %.3e)", rtol,
                             total, relres);
    return LSA_OK;
}

// out[perm[i], c] = in[i, c]  (column-major n x ncols; grid.y = column)
__global__ void scatter_rows_kernel(int64_t n, const int32_t* __restrict__ perm, const cplx* __restrict__ in, cplx* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const size_t off = (size_t)blockIdx.y * (size_t)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[off + (size_t)perm[i]] = in[off + (size_t)i];
}

template <typename T>
__global__ void shift_diag_kernel(int32_t n, int32_t row0, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                  T* __restrict__ val, cplx shift, int32_t* __restrict__ missing) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t target = (int32_t)i + row0;
        int32_t lo = rp[i], hi = rp[i + 1];
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (ci[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        if (lo < rp[i + 1] && ci[lo] == target) {
            cplx v = to_cplx(val[lo]);
            s_from(val[lo], v.re - shift.re, v.im - shift.im);
        } else {
            atomicAdd(missing, 1);
        }
    }
}

}  // namespace

// ---- objects ------------------------------------------------------------------------------------------------------
static_assert(sizeof(lsa_op_options) == 56, "lsa_op_options layout is part of the C-ABI (tests/test_abi.py)");

// the caller's elimination forest for the subtree-parallel factorisation (lsa_op_create_dist)
struct TreeArg {
    int32_t nt;
    const int32_t *first, *size, *parent, *owner;
    int32_t row0, row1;  // this rank's rows of the padded layout
};

struct lsa_op {
    lsa_ctx* ctx;
    int64_t n;
    const lsa_mat* Kmul;  // y = Kfac^-1 (Kmul x); either may be null (identity)
    const lsa_mat* Kfac;
    lsa_mat* owned;       // the matrix built here (C = A - sigma M), destroyed with the operator
    lsa_mat* owned_mul = nullptr;  // Cayley: A + nu M
    lsa_mat* owned_diag;  // sharded layout: this rank's diagonal block of C (input of the block-Jacobi ILU)
    lsa_ilu* pc;
    lsa_ndlu* nd = nullptr;  // exact nested-dissection LU (opts.pc_type == 2)
    bool nd_dist = false;    // ... subtree-parallel over the ranks: works on whole replicated vectors
    bool adjoint = false;    // y = Kfac^-H Kmul^H x on the same factors (lsa_op_set_adjoint)
    double normF = 0.0;      // ||Kfac||_F
    lsa_mat *view_fac = nullptr, *view_mul = nullptr;  // this rank's rows of the whole matrices (subtree-parallel layout)
    const lsa_mat* M_whole = nullptr;                  // the caller's M (modes 0 and 2)
    lsa_op_options opts;
    GmresWork gw;
    bool gw_ready;
    void* t;  // device temp vector (complex)
    double* keep = nullptr;  // device 0/1 mask of the projected operator (lsa_op_set_projection), or null
    bool refine = false;     // queued Arnoldi steps carry one step of iterative refinement (set the first time a direct solve misses rtol)
    lsa_stats st;
};

struct lsa_krylov {
    lsa_ctx* ctx;
    lsa_op* op;
    int64_t n;
    int32_t ncv;
    int32_t* row_perm = nullptr;  // device: row i of the basis is unknown row_perm[i] of the caller (lsa_krylov_set_row_permutation)
    void *V, *V2, *w, *qdev;
    OrthWork ow;
    std::vector<zc> hcol;
    // pipelined Arnoldi steps (exact inner solves): Hessenberg columns and the b - C x checks of a batch of steps stay on
    // the device until the batch is read back with one synchronisation
    double* imag2 = nullptr;         // device: sum Im^2 of the columns of the last canonical Ritz vectors
    void* xtmp = nullptr;            // device: the Ritz vectors in the caller's row order (ncv + 1 columns), allocated on first use
    std::vector<double> imag_norms;  // ... their square roots on the host (lsa_krylov_imag_norms)
    void* Hdev = nullptr;      // batch x (ncv + 2) complex
    double* checks = nullptr;  // batch x 2: ||b - C x||^2, ||b||^2
    int32_t batch = 0;
    bool pipeline = false;
    // tail form of a pipelined step (k_cgs2_fused_tail): the step's last launch also multiplies t = M v_{j+1} for the next step
    // and leaves the pairs of this step's check of the inner solve; w2 takes the orthogonalised vector (w keeps the solve's result)
    void* w2 = nullptr;
    double* tail_parts = nullptr;  // batch x tail_nparts pairs
    int32_t tail_nparts = 0;
    int32_t t_for = -1;            // op->t holds M v_j for this j (valid inside one lsa_krylov_extend call), -1: nothing
};

static void krylov_free(lsa_krylov* k) {
    for (void* p : {k->V, k->V2, k->w, k->qdev, k->Hdev, (void*)k->checks, (void*)k->imag2, k->xtmp, (void*)k->row_perm, k->w2, (void*)k->tail_parts})
        if (p) (void)hipFree(p);
    k->ow.release();
    delete k;
}

extern "C" {

int lsa_gmres(lsa_ctx* ctx, const lsa_mat* C, lsa_ilu* pc, const lsa_vec* b, lsa_vec* x, int use_x0, double rtol, int restart,
              int maxit, int32_t* iters, double* rel_res) {
    if (!ctx || !C || !b || !x) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_gmres: null argument");
    if (C->n != C->ncols || b->n != C->n || x->n != C->n || b->dtype != x->dtype)
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_gmres: shape/dtype mismatch");
    if (C->dtype == LSA_C128 && b->dtype != LSA_C128) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_gmres: complex matrix needs complex vectors");
    if (restart < 1 || maxit < 1 || !(rtol > 0.0)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_gmres: restart, maxit, rtol must be positive");
    restart = std::min(restart, maxit);
    GmresWork W;
    int rc = W.alloc(ctx, C->n, restart, b->dtype);
    PcRef pcr;
    pcr.ilu = pc;
    if (rc == LSA_OK) rc = gmres_run(ctx, C, pcr, b->dtype, b->d, x->d, use_x0 != 0, rtol, maxit, W, iters, rel_res, nullptr);
    (void)hipStreamSynchronize(ctx->stream);
    W.release();
    return rc;
}

// C = A - sigma M on the shared pattern (M null: A - sigma I through a diagonal shift); rows may be a shard
static int build_shifted(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* M, const double sigma[2], bool cdt, lsa_mat** out) {
    const bool zshift = sigma[0] == 0.0 && sigma[1] == 0.0;
    const double one[2] = {1.0, 0.0}, ms[2] = {-sigma[0], -sigma[1]}, zero[2] = {0.0, 0.0};
    lsa_mat* C = nullptr;
    int rc;
    if (M) rc = lsa_csr_axpby(ctx, A, M, one, ms, cdt ? LSA_C128 : LSA_F64, &C);
    else {
        rc = lsa_csr_axpby(ctx, A, A, one, zero, cdt ? LSA_C128 : LSA_F64, &C);
        if (rc == LSA_OK && !zshift) {
            int32_t* miss = (int32_t*)ctx->dscratch;
            (void)hipMemsetAsync(miss, 0, sizeof(int32_t), ctx->stream);
            const int blocks = std::max(1, std::min((int)((C->n + 255) / 256), ctx->num_cu * 8));
            if (cdt) hipLaunchKernelGGL((shift_diag_kernel<cplx>), dim3(blocks), dim3(256), 0, ctx->stream, C->n, C->row0, C->rp, C->ci, (cplx*)C->val, cplx{sigma[0], sigma[1]}, miss);
            else hipLaunchKernelGGL((shift_diag_kernel<double>), dim3(blocks), dim3(256), 0, ctx->stream, C->n, C->row0, C->rp, C->ci, (double*)C->val, cplx{sigma[0], sigma[1]}, miss);
            int32_t hm = 0;
            (void)hipMemcpyAsync(&hm, miss, sizeof hm, hipMemcpyDeviceToHost, ctx->stream);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = lsa_set_error(ctx, LSA_ERR_HIP, "diagonal shift failed");
            else if (hm != 0) rc = lsa_set_error(ctx, LSA_ERR_ARG, "A - sigma*I needs a structurally present diagonal (%d rows have none)", hm);
        }
    }
    if (rc != LSA_OK) {
        if (C) lsa_mat_destroy(C);
        return rc;
    }
    *out = C;
    return LSA_OK;
}

// shared builder: (A, M) are the full matrices or this rank's row shards; (Ad, Md) are null or the rank's diagonal blocks
static int op_build(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* M, const lsa_mat* Ad, const lsa_mat* Md, const double sigma[2],
                    int mode, const lsa_op_options* opts, lsa_op** out, const TreeArg* tree = nullptr) {
    if (!ctx || !A || !sigma || !opts || !out) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create: null argument");
    const bool sharded = Ad != nullptr;
    if (tree) {
        const int nr = std::max(1, ctx->nranks);
        if (sharded || A->n != A->ncols || A->row0 != 0 || (mode != 0 && mode != 2) || opts->pc_type != 2)
            return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_dist: needs the whole square matrices, a factorising mode (0 or 2) and pc_type 2");
        if (A->ncols % nr != 0 || tree->row0 != (int64_t)ctx->rank * (A->ncols / nr) || tree->row1 < tree->row0 || tree->row1 > tree->row0 + A->ncols / nr)
            return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_dist: this rank's rows do not sit on its block of the padded layout");
    }
    if (!sharded && A->n != A->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create: A must be square (got %d x %d)", A->n, A->ncols);
    if (M && (M->n != A->n || M->ncols != A->ncols || M->row0 != A->row0)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create: M does not match A's shape");
    if (mode < 0 || mode > 2) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create: mode must be 0 (sinvert), 1 (shift) or 2 (Cayley)");
    if (sharded) {
        if (Ad->n != Ad->ncols || Ad->n != A->n || (Md && (Md->n != Ad->n || Md->ncols != Ad->ncols)) || (M != nullptr) != (Md != nullptr))
            return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_sharded: diagonal blocks must be square with the shard's row count");
        if (A->ncols % std::max(1, ctx->nranks) != 0 || A->row0 != (int64_t)ctx->rank * (A->ncols / std::max(1, ctx->nranks)))
            return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_sharded: shard does not sit on this rank's block of the padded layout");
        if (A->n > A->ncols / std::max(1, ctx->nranks)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_sharded: shard larger than its padded block");
    }
    const double t0 = now_s();
    lsa_op* op = new lsa_op();
    op->ctx = ctx;
    op->n = A->ncols;  // vectors are global-length (padded block layout when sharded)
    op->Kmul = op->Kfac = nullptr;
    op->owned = op->owned_diag = nullptr;
    op->pc = nullptr;
    op->opts = *opts;
    op->gw_ready = false;
    op->t = nullptr;
    memset(&op->st, 0, sizeof op->st);
    const bool zshift = sigma[0] == 0.0 && sigma[1] == 0.0;
    const bool cdt = sigma[1] != 0.0 || A->dtype == LSA_C128 || (M && M->dtype == LSA_C128);
    int rc = LSA_OK;
    lsa_mat* C = nullptr;
    if (!(mode == 1 && zshift)) {
        rc = build_shifted(ctx, A, M, sigma, cdt, &C);
        if (rc != LSA_OK) {
            delete op;
            return rc;
        }
        op->owned = C;
    }
    const lsa_mat* fac_src = nullptr;  // the square matrix the preconditioner is built from
    if (mode == 0 || mode == 2) {
        op->Kfac = C;
        op->Kmul = M;
        op->M_whole = M;
        // Subtree-parallel layout: every rank holds the whole matrices.  A product over this rank's rows only must be
        // completed by an all-gather of the result (16 B per unknown over xGMI); the whole product costs 20 B per stored
        // entry from HBM.  With the ~30 entries per row of the 2D pattern the replicated product is the cheaper one at any
        // size (and bit-identical on every rank); with the ~100 of the 3D pattern the sharded one is.
        bool shard_rows = false;
        if (tree) {
            const char* e = getenv("LSA_DIST_SPMV");  // "shard" / "replicate": override (measurement)
            shard_rows = (e && *e) ? (e[0] == 's') : (C->nnz > 60 * (int64_t)C->n);
        }
        if (tree && shard_rows) {  // SpMV on this rank's rows of the whole matrices
            op->view_fac = mat_row_view(C, tree->row0, tree->row1);
            op->Kfac = op->view_fac;
            if (M) {
                op->view_mul = mat_row_view(M, tree->row0, tree->row1);
                op->Kmul = op->view_mul;
            }
        }
        if (mode == 2) {  // Cayley: multiply by A + nu M = A - (-nu) M
            const double mnu[2] = {-opts->antishift[0], -opts->antishift[1]};
            const bool cmul = cdt || opts->antishift[1] != 0.0;
            rc = build_shifted(ctx, A, M, mnu, cmul, &op->owned_mul);
            if (rc != LSA_OK) {
                lsa_op_destroy(op);
                return rc;
            }
            if (tree && shard_rows) {
                if (op->view_mul) lsa_mat_destroy(op->view_mul);
                op->view_mul = mat_row_view(op->owned_mul, tree->row0, tree->row1);
                op->Kmul = op->view_mul;
            } else op->Kmul = op->owned_mul;
        }
        if (sharded) {
            rc = build_shifted(ctx, Ad, Md, sigma, cdt, &op->owned_diag);
            if (rc != LSA_OK) {
                lsa_op_destroy(op);
                return rc;
            }
            fac_src = op->owned_diag;
        } else fac_src = C;
    } else {
        op->Kmul = C ? C : A;
        op->Kfac = M;
        fac_src = sharded ? Md : M;
    }
    bool lu_fell_back = false;
    if (opts->pc_type == 3) {
        lsa_op_destroy(op);
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create: pc_type 3 (the banded block LU of round 1) left the product library; it lives on as a "
                                               "cross-check library for the tests (tests/xcheck).  Use pc_type 2, the nested-dissection LU");
    }
    if (op->Kfac && fac_src && opts->pc_type == 2) {
        // exact LU.  Only running out of device memory is answered by the leaner ILU(k) + GMRES (and said so on stderr
        // and in the statistics); a singular pivot block or any other failure is an error, as with PETSc's PC LU.
        if (tree) {
            rc = lsa_ndlu_create_tree(ctx, fac_src, tree->nt, tree->first, tree->size, tree->parent, tree->owner, &op->nd);
            op->nd_dist = rc == LSA_OK;
            if (rc == LSA_ERR_OOM) {  // no leaner collective method to agree on: an error on every rank that hits it
                lsa_op_destroy(op);
                return rc;
            }
        } else
            rc = lsa_ndlu_create(ctx, fac_src, 0, &op->nd);
        if (rc == LSA_OK && op->nd) {
            double sa = 1.0;
            (void)lsa_ndlu_info(op->nd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &sa, nullptr);
            op->st.analysis_reused = sa == 0.0 ? 1 : 0;
        }
        if (rc == LSA_ERR_OOM) {
            fprintf(stderr, "[lsa_hip] exact LU does not fit the device memory (%s): falling back to ILU(%d) + GMRES\n", ctx->err.c_str(),
                    opts->ilu_levels);
            lu_fell_back = true;
            op->st.pc_fallback = 1;
        } else if (rc != LSA_OK) {
            lsa_op_destroy(op);
            return rc;
        }
    }
    if (op->Kfac && fac_src && (opts->pc_type == 1 || lu_fell_back)) {
        rc = lsa_ilu_create(ctx, fac_src, opts->ilu_levels, opts->ilu_shift, &op->pc);
        if (rc != LSA_OK) {
            lsa_op_destroy(op);
            return rc;
        }
    }
    const size_t tb = (size_t)std::max<int64_t>(op->n, 1) * 16;
    if (hipMalloc(&op->t, tb) != hipSuccess || hipMemsetAsync(op->t, 0, tb, ctx->stream) != hipSuccess) {
        lsa_op_destroy(op);
        return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_op_create: out of device memory");
    }
    if (op->Kfac && op->nd && fac_src && fac_src->nnz > 0) {
        // ||C||_F (one reduction over the values): the scale of the backward-error test of the direct solves
        // (the result goes to the head of the operator's own temp vector, not to the context's scratch: k_nrm2 keeps its partial
        //  sums there and may reallocate it; the head is zeroed again afterwards -- padding rows must read zero)
        double* nr = (double*)op->t;
        double v2 = 0.0;
        if (k_nrm2(ctx, fac_src->dtype, fac_src->nnz, fac_src->val, nr) == LSA_OK &&
            hipMemcpyAsync(&v2, nr, sizeof v2, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess &&
            std::isfinite(v2))
            op->normF = std::sqrt(v2);
        (void)hipMemsetAsync(op->t, 0, sizeof(double), ctx->stream);
    }
    op->st.seconds_factor = now_s() - t0;
    *out = op;
    return LSA_OK;
}

int lsa_op_create(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* M, const double sigma[2], int mode, const lsa_op_options* opts,
                  lsa_op** out) {
    return op_build(ctx, A, M, nullptr, nullptr, sigma, mode, opts, out);
}

int lsa_op_create_sharded(lsa_ctx* ctx, const lsa_mat* A_rows, const lsa_mat* M_rows, const lsa_mat* A_diag, const lsa_mat* M_diag,
                          const double sigma[2], int mode, const lsa_op_options* opts, lsa_op** out) {
    if (!A_diag) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_sharded: the diagonal block of A is required");
    return op_build(ctx, A_rows, M_rows, A_diag, M_diag, sigma, mode, opts, out);
}

int lsa_op_create_dist(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* M, int32_t row0, int32_t row1, int32_t ntree, const int32_t* first,
                       const int32_t* size, const int32_t* parent, const int32_t* owner, const double sigma[2], int mode, const lsa_op_options* opts,
                       lsa_op** out) {
    if (!first || !size || !parent || !owner || ntree < 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_create_dist: the forest is required");
    const TreeArg tree{ntree, first, size, parent, owner, row0, row1};
    return op_build(ctx, A, M, nullptr, nullptr, sigma, mode, opts, out, &tree);
}

void lsa_op_destroy(lsa_op* op) {
    if (!op) return;
    if (op->ctx && op->ctx->stream) (void)hipStreamSynchronize(op->ctx->stream);
    if (op->pc) lsa_ilu_destroy(op->pc);
    if (op->nd) lsa_ndlu_destroy(op->nd);
    if (op->view_fac) lsa_mat_destroy(op->view_fac);
    if (op->view_mul) lsa_mat_destroy(op->view_mul);
    if (op->owned) lsa_mat_destroy(op->owned);
    if (op->owned_mul) lsa_mat_destroy(op->owned_mul);
    if (op->owned_diag) lsa_mat_destroy(op->owned_diag);
    if (op->gw_ready) op->gw.release();
    if (op->t) (void)hipFree(op->t);
    if (op->keep) (void)hipFree(op->keep);
    delete op;
}

// y = Kfac^-1 Kmul x on device pointers (complex vectors)
static int op_apply_dev(lsa_ctx* ctx, lsa_op* op, const void* x, void* y) {
    const int dtype = LSA_C128;
    ++op->st.op_applies;
    const void* rhs = x;
    if (op->Kmul) {
        void* dst = op->Kfac ? op->t : y;
        LSA_CHECK(spmv_global(ctx, op->Kmul, dtype, x, dst, op->adjoint));
        ++op->st.spmv_calls;
        rhs = dst;
    }
    if (!op->Kfac) {
        if (!op->Kmul) LSA_CHECK(k_copy(ctx, dtype, op->n, x, y));
        if (op->keep) LSA_CHECK(k_mask(ctx, dtype, op->n, op->keep, y));
        return LSA_OK;
    }
    if (!op->gw_ready) {
        int restart = std::max(1, std::min(op->opts.ksp_restart, op->opts.ksp_maxit));
        // with an exact factorisation on one GPU GMRES only polishes (0-2 iterations): keep its basis small
        if (op->nd && (ctx->nranks == 1 || op->nd_dist)) restart = std::min(restart, 40);
        LSA_CHECK(op->gw.alloc(ctx, op->n, restart, dtype));
        op->gw_ready = true;
    }
    PcRef pcr;
    pcr.ilu = op->pc;
    pcr.nd = op->nd;
    pcr.nd_dist = op->nd_dist;
    pcr.adjoint = op->adjoint;
    pcr.normF = op->normF;
    LSA_CHECK(gmres_run(ctx, op->Kfac, pcr, dtype, rhs, y, false, op->opts.ksp_rtol, op->opts.ksp_maxit, op->gw, nullptr, nullptr, &op->st));
    if (op->keep) LSA_CHECK(k_mask(ctx, dtype, op->n, op->keep, y));
    return LSA_OK;
}

int lsa_op_set_adjoint(lsa_ctx* ctx, lsa_op* op, int on) {
    if (!ctx || !op) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_set_adjoint: null argument");
    if (on && (!op->nd || !op->Kfac || (ctx->nranks != 1 && !op->nd_dist)))
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_set_adjoint: needs the nested-dissection LU (pc_type 2) of a factorising mode, on one rank or in the "
                                                "subtree-parallel layout");
    if (op->nd_dist) {
        // Subtree-parallel layout: the transposed sweeps walk the same forest with the same exchanges (own subtrees, all-gather of
        // the subtree roots' update vectors, replicated top, all-gather of the solution blocks).  The transposed sparse products
        // are taken with the WHOLE matrices every rank holds (a transposed product over a row block would need a reduction
        // over the ranks instead of an all-gather), whatever the forward products use.
        op->Kfac = (on || !op->view_fac) ? op->owned : op->view_fac;
        op->Kmul = (on || !op->view_mul) ? (op->owned_mul ? op->owned_mul : op->M_whole) : op->view_mul;
    }
    op->adjoint = on != 0;
    return LSA_OK;
}

int lsa_op_set_projection(lsa_ctx* ctx, lsa_op* op, const double* keep) {
    if (!ctx || !op) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_set_projection: null argument");
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (!keep) {
        if (op->keep) (void)hipFree(op->keep);
        op->keep = nullptr;
        return LSA_OK;
    }
    for (int64_t i = 0; i < op->n; ++i)
        if (keep[i] != 0.0 && keep[i] != 1.0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_set_projection: keep[%lld] is neither 0 nor 1", (long long)i);
    const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(op->n, 1);
    if (!op->keep) LSA_HIP_CHECK(ctx, hipMalloc((void**)&op->keep, bytes));
    LSA_HIP_CHECK(ctx, hipMemcpy(op->keep, keep, sizeof(double) * (size_t)op->n, hipMemcpyHostToDevice));
    return LSA_OK;
}

int lsa_op_apply(lsa_ctx* ctx, lsa_op* op, const lsa_vec* x, lsa_vec* y) {
    if (!ctx || !op || !x || !y) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_apply: null argument");
    if (x->n != op->n || y->n != op->n || x->dtype != LSA_C128 || y->dtype != LSA_C128 || x->d == y->d)
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_op_apply: x and y must be distinct complex vectors of length n");
    const double t0 = now_s();
    int rc = op_apply_dev(ctx, op, x->d, y->d);
    if (rc == LSA_OK) LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    op->st.seconds_solve += now_s() - t0;
    return rc;
}

int lsa_op_stats(const lsa_op* op, lsa_stats* out) {
    if (!op || !out) return LSA_ERR_ARG;
    *out = op->st;
    return LSA_OK;
}

// ---- Krylov basis ------------------------------------------------------------------------------------------------
int lsa_krylov_create(lsa_ctx* ctx, lsa_op* op, int32_t ncv, lsa_krylov** out) {
    if (!ctx || !op || !out || ncv < 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_create: bad argument");
    const size_t vb = (size_t)std::max<int64_t>(op->n, 1) * 16;
    const char* be = getenv("LSA_KRYLOV_BATCH");
    const int32_t batch = be && *be ? std::max(0, std::min(atoi(be), 256)) : 16;
    if (lsa_krylov* c = ctx->krylov_cache) {
        // A shift sweep (and every repeated solve) builds one Krylov workspace per solve: two bases of ncv + 1 vectors
        // (80 MB at 30 k unknowns, 1.3 GB at 500 k).  The last one destroyed is kept and handed out again when the shape
        // matches, instead of being freed and allocated anew (LSA_KRYLOV_NO_CACHE switches this off).
        ctx->krylov_cache = nullptr;
        if (c->n == op->n && c->ncv == ncv && c->batch == batch) {
            c->op = op;
            c->pipeline = batch > 1;
            c->imag_norms.clear();
            (void)hipMemsetAsync(c->w, 0, vb, ctx->stream);
            *out = c;
            return LSA_OK;
        }
        krylov_free(c);
    }
    lsa_krylov* k = new lsa_krylov();
    k->ctx = ctx;
    k->op = op;
    k->n = op->n;
    k->ncv = ncv;
    k->V = k->V2 = k->w = k->qdev = nullptr;
    bool ok = hipMalloc(&k->V, vb * (size_t)(ncv + 1)) == hipSuccess && hipMalloc(&k->V2, vb * (size_t)(ncv + 1)) == hipSuccess &&
              hipMalloc(&k->w, vb) == hipSuccess && hipMalloc(&k->qdev, (size_t)(ncv + 1) * (size_t)(ncv + 1) * 16) == hipSuccess;
    if (!ok || k->ow.alloc(ctx, ncv + 1, LSA_C128) != LSA_OK) {
        lsa_krylov_destroy(k);
        return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_krylov_create: out of device memory (n=%lld, ncv=%d)", (long long)op->n, ncv);
    }
    (void)hipMemsetAsync(k->w, 0, vb, ctx->stream);
    k->hcol.assign((size_t)ncv + 2, zc(0));
    // Arnoldi steps are queued in batches when the inner solve is one exact LU solve and the exchange (if any) is
    // stream-ordered (RCCL): LSA_KRYLOV_BATCH steps per read-back (default 16; 0 or 1 = one step at a time)
    k->batch = batch;
    if (k->batch > 1) {
        if (hipMalloc(&k->Hdev, (size_t)k->batch * (size_t)(ncv + 2) * 16) != hipSuccess ||
            hipMalloc((void**)&k->checks, (size_t)k->batch * 2 * sizeof(double)) != hipSuccess) {
            lsa_krylov_destroy(k);
            return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_krylov_create: out of device memory (pipeline buffers)");
        }
        k->pipeline = true;
    }
    *out = k;
    return LSA_OK;
}

void lsa_krylov_destroy(lsa_krylov* k) {
    if (!k) return;
    lsa_ctx* ctx = k->ctx;
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (k->row_perm) (void)hipFree(k->row_perm);
    k->row_perm = nullptr;
    k->op = nullptr;
    if (ctx && k->V && k->V2 && k->w && k->qdev && k->ow.h1 && !getenv("LSA_KRYLOV_NO_CACHE")) {  // (a half-built workspace is not worth keeping)
        if (ctx->krylov_cache) krylov_free(ctx->krylov_cache);
        ctx->krylov_cache = k;
        return;
    }
    krylov_free(k);
}

void lsa_krylov_drop_cache(lsa_ctx* ctx) {
    if (ctx && ctx->krylov_cache) {
        krylov_free(ctx->krylov_cache);
        ctx->krylov_cache = nullptr;
    }
}

int lsa_krylov_shape(const lsa_krylov* k, int64_t* n, int32_t* ncv) {
    if (!k) return LSA_ERR_ARG;
    if (n) *n = k->n;
    if (ncv) *ncv = k->ncv;
    return LSA_OK;
}

int lsa_krylov_set_row_permutation(lsa_ctx* ctx, lsa_krylov* k, const int32_t* perm) {
    if (!ctx || !k) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_set_row_permutation: null argument");
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (k->row_perm) (void)hipFree(k->row_perm);
    k->row_perm = nullptr;
    if (!perm) return LSA_OK;
    std::vector<char> seen((size_t)k->n, 0);
    for (int64_t i = 0; i < k->n; ++i) {
        if (perm[i] < 0 || perm[i] >= k->n || seen[(size_t)perm[i]]) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_set_row_permutation: not a permutation of 0..n-1");
        seen[(size_t)perm[i]] = 1;
    }
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&k->row_perm, sizeof(int32_t) * (size_t)std::max<int64_t>(k->n, 1)));
    LSA_HIP_CHECK(ctx, hipMemcpy(k->row_perm, perm, sizeof(int32_t) * (size_t)k->n, hipMemcpyHostToDevice));
    return LSA_OK;
}

int lsa_krylov_set_start(lsa_ctx* ctx, lsa_krylov* k, const void* host_v) {
    if (!ctx || !k || !host_v) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_set_start: null argument");
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(k->w, host_v, (size_t)k->n * 16, hipMemcpyHostToDevice, ctx->stream));
    double nrm = 0.0;
    LSA_CHECK(device_norm(ctx, LSA_C128, k->n, k->w, k->ow.nrm2, &nrm));
    if (!(nrm > 0.0) || !std::isfinite(nrm)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_set_start: start vector is zero or not finite");
    LSA_CHECK(k_scale_by_inv_norm(ctx, LSA_C128, k->n, k->w, k->ow.nrm2, k->V));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

int lsa_krylov_inject(lsa_ctx* ctx, lsa_krylov* k, int32_t j, const void* host_v) {
    if (!ctx || !k || !host_v) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_inject: null argument");
    if (j < 0 || j > k->ncv) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_inject: index %d out of range", j);
    if (j == 0) return lsa_krylov_set_start(ctx, k, host_v);
    const size_t vb = (size_t)k->n * 16;
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(k->w, host_v, vb, hipMemcpyHostToDevice, ctx->stream));
    LSA_CHECK(orthonormalize(ctx, LSA_C128, k->n, k->V, k->n, j, k->w, (char*)k->V + (size_t)j * vb, k->ow, k->hcol.data()));
    const double beta = k->hcol[j].real();
    if (!(beta > 0.0) || !std::isfinite(beta)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_inject: vector lies in the span of the basis");
    return LSA_OK;
}

// One batch of Arnoldi steps [j, j + nb) queued without a host round trip: operator apply by one exact LU solve, the
// check b - C x of that solve, CGS2.  Everything the host decides on (the check, the Hessenberg column, breakdown) is read
// back once per batch.
// the steps of a batch can take the tail form: a generalised problem whose M and C share their index arrays, whole on this rank,
// plain forward products, nothing between the solve and the orthogonalisation (no projection mask, no refinement step)
static bool krylov_tail_ok(lsa_ctx* ctx, const lsa_krylov* k) {
    static const bool enabled = !(getenv("LSA_KRYLOV_TAIL") && atoi(getenv("LSA_KRYLOV_TAIL")) == 0);
    static const bool fuse = !(getenv("LSA_KRYLOV_FUSED") && atoi(getenv("LSA_KRYLOV_FUSED")) == 0);
    const lsa_op* op = k->op;
    if (!enabled || !fuse || !op->Kmul || !op->Kfac || op->adjoint || op->keep || op->refine || op->n != k->n) return false;
    return k_cgs2_tail_fits(ctx, k->n, k->ncv, op->Kmul, op->Kfac);
}

static int krylov_enqueue_step(lsa_ctx* ctx, lsa_krylov* k, int32_t j, int32_t slot, bool tail) {
    lsa_op* op = k->op;
    const int dtype = LSA_C128;
    const size_t vb = (size_t)k->n * 16;
    const void* vj = (char*)k->V + (size_t)j * vb;
    void* vn = (char*)k->V + (size_t)(j + 1) * vb;
    const void* rhs = vj;
    PcRef pcr;
    pcr.nd = op->nd;
    pcr.nd_dist = op->nd_dist;
    pcr.adjoint = op->adjoint;
    void* hcol_dev = (char*)k->Hdev + (size_t)slot * (size_t)(k->ncv + 2) * 16;
    if (tail) {
        // 18 launches: [M v_j unless the previous step's tail left it] + the sweeps + dot, update, dot, update + tail
        if (k->t_for != j) LSA_CHECK(spmv_global(ctx, op->Kmul, dtype, vj, op->t, false));
        k->t_for = -1;
        LSA_CHECK(pc_global(ctx, pcr, op->Kfac->row0, op->n, dtype, op->t, k->w));
        LSA_CHECK(k->ow.ensure_fused(ctx, k->n));
        LSA_CHECK(k_cgs2_fused_tail(ctx, k->n, j + 1, k->V, k->n, k->w, k->w2, vn, hcol_dev, k->ow.fused, op->Kmul, op->Kfac, op->t,
                                    k->tail_parts + (size_t)slot * 2 * (size_t)k->tail_nparts));
        k->t_for = j + 1;
        return LSA_OK;
    }
    k->t_for = -1;
    if (op->Kmul) {
        LSA_CHECK(spmv_global(ctx, op->Kmul, dtype, vj, op->t, op->adjoint));
        rhs = op->t;
    }
    LSA_CHECK(pc_global(ctx, pcr, op->Kfac->row0, op->n, dtype, rhs, k->w));
    LSA_CHECK(spmv_global(ctx, op->Kfac, dtype, k->w, op->gw.z, op->adjoint));
    if (op->refine) {
        // y += C^-1 (t - C y): the factors of a large 3D problem leave 1e-11 of the right-hand side behind, this step takes it to
        // rounding level (the norms this residual pass leaves in the check slot are overwritten by the final check below)
        LSA_CHECK(k_residual_norms(ctx, dtype, op->n, rhs, op->gw.z, op->gw.w, k->checks + 2 * (size_t)slot));
        LSA_CHECK(pc_global(ctx, pcr, op->Kfac->row0, op->n, dtype, op->gw.w, op->gw.z));
        const double one[2] = {1.0, 0.0};
        LSA_CHECK(k_axpy(ctx, dtype, op->n, one, op->gw.z, k->w));
        LSA_CHECK(spmv_global(ctx, op->Kfac, dtype, k->w, op->gw.z, op->adjoint));
    }
    static const bool fuse = !(getenv("LSA_KRYLOV_FUSED") && atoi(getenv("LSA_KRYLOV_FUSED")) == 0);
    if (fuse) {
        // the check ||b - C y||, ||b|| rides in the first reduction of the orthogonalisation (its vector b - C y is needed only
        // by the one-step path, which recomputes it)
        LSA_CHECK(k->ow.ensure_fused(ctx, k->n));
        if (op->keep) LSA_CHECK(k_mask(ctx, dtype, op->n, op->keep, k->w));
        const int frc = k_cgs2_fused(ctx, dtype, k->n, j + 1, k->V, k->n, k->w, vn, hcol_dev, k->ow.fused, rhs, op->gw.z, k->checks + 2 * (size_t)slot);
        if (frc <= 0) return frc;
    }
    LSA_CHECK(k_residual_norms(ctx, dtype, op->n, rhs, op->gw.z, op->gw.w, k->checks + 2 * (size_t)slot));
    if (op->keep && !fuse) LSA_CHECK(k_mask(ctx, dtype, op->n, op->keep, k->w));
    return orthonormalize_enqueue(ctx, dtype, k->n, k->V, k->n, j + 1, k->w, vn, k->ow, hcol_dev);
}

// true when an operator apply is one exact LU solve whose result only needs checking, with nothing on the way that
// synchronises with the host by itself (the host-staged exchange does)
static bool krylov_can_pipeline(const lsa_ctx* ctx, const lsa_krylov* k) {
    const lsa_op* op = k->op;
    if (!k->pipeline || !op->Kfac || !op->nd || op->pc) return false;
    if (ctx->nranks > 1 && (!op->nd_dist || ctx->host_gather)) return false;
    return true;
}

int lsa_krylov_extend(lsa_ctx* ctx, lsa_krylov* k, int32_t j0, int32_t j1, void* H, int32_t ldh, int32_t* breakdown) {
    if (!ctx || !k || !H) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_extend: null argument");
    if (j0 < 0 || j1 < j0 || j1 > k->ncv || ldh < j1 + 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_extend: bad step range [%d, %d) for ncv %d", j0, j1, k->ncv);
    const double t0 = now_s();
    const size_t vb = (size_t)k->n * 16;
    if (breakdown) *breakdown = -1;
    cplx* Hh = (cplx*)H;
    lsa_op* op = k->op;
    // fills column j of H from hcol (j + 2 entries) and tells whether the step broke down; < 0: non-finite
    auto take_column = [&](int32_t j, const zc* hcol) -> int {
        for (int32_t i = 0; i < ldh; ++i) Hh[(size_t)j * ldh + i] = cplx{0.0, 0.0};
        for (int32_t i = 0; i <= j + 1; ++i) Hh[(size_t)j * ldh + i] = cplx{hcol[i].real(), hcol[i].imag()};
        const double beta = hcol[j + 1].real();
        if (!std::isfinite(beta)) return -1;
        double colmax = 0.0;
        for (int32_t i = 0; i <= j; ++i) colmax = std::max(colmax, std::abs(hcol[i]));
        return beta <= 1e-14 * std::max(colmax, 1e-300) ? 1 : 0;
    };
    int32_t j = j0;
    if (krylov_can_pipeline(ctx, k)) {
        if (!op->gw_ready) {
            LSA_CHECK(op->gw.alloc(ctx, op->n, std::max(1, std::min({op->opts.ksp_restart, op->opts.ksp_maxit, 40})), LSA_C128));
            op->gw_ready = true;
        }
        const size_t colb = (size_t)(k->ncv + 2) * 16;
        LSA_CHECK(lsa_ensure_scratch(ctx, 0, (size_t)k->batch * (colb + 2 * sizeof(double))));
        const double rtol = op->opts.ksp_rtol;
        k->t_for = -1;  // (op->t is anybody's between calls)
        while (j < j1 && k->pipeline) {
            const int32_t nb = std::min<int32_t>(k->batch, j1 - j);
            const double tq = now_s();
            const bool tail = krylov_tail_ok(ctx, k);
            if (tail && !k->tail_parts) {
                k->tail_nparts = k_cgs2_tail_parts(k->n);
                LSA_HIP_ALLOC(ctx, hipMalloc(&k->w2, (size_t)k->n * 16));
                LSA_HIP_ALLOC(ctx, hipMalloc((void**)&k->tail_parts, (size_t)k->batch * 2 * (size_t)k->tail_nparts * sizeof(double)));
            }
            for (int32_t s = 0; s < nb; ++s) {
                int rc = krylov_enqueue_step(ctx, k, j + s, s, tail);
                if (rc != LSA_OK) {
                    (void)hipStreamSynchronize(ctx->stream);
                    op->st.seconds_solve += now_s() - t0;
                    return rc;
                }
            }
            if (tail) LSA_CHECK(k_cgs2_tail_checks(ctx, nb, k->tail_nparts, k->tail_parts, k->checks));
            char* host = (char*)ctx->pinned;
            LSA_HIP_CHECK(ctx, hipMemcpyAsync(host, k->Hdev, (size_t)nb * colb, hipMemcpyDeviceToHost, ctx->stream));
            LSA_HIP_CHECK(ctx, hipMemcpyAsync(host + (size_t)k->batch * colb, k->checks, (size_t)nb * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            const double tw = now_s();
            LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            static const bool timing = getenv("LSA_KRYLOV_TIMING") != nullptr;
            if (timing) fprintf(stderr, "[lsa_krylov] %d steps queued in %.3f ms, waited %.3f ms\n", nb, 1e3 * (tw - tq), 1e3 * (now_s() - tw));
            const double* chk = (const double*)(host + (size_t)k->batch * colb);
            int32_t accepted = 0;
            for (int32_t s = 0; s < nb; ++s, ++accepted) {
                const double beta0 = std::sqrt(chk[2 * s]), bnorm = std::sqrt(chk[2 * s + 1]);
                if (!(beta0 <= rtol * bnorm)) {
                    if (!op->refine && std::isfinite(beta0)) {
                        // a direct solve missed rtol: from here on every queued step carries one refinement step (large 3D
                        // factors; the steps of this batch from s on are queued again)
                        op->refine = true;
                        break;
                    }
                    // refined and still short of rtol: this solve needs the judgement of the one-step path (backward error
                    // next to an eigenvalue, GMRES), and so will its neighbours: the rest of this basis runs one step at a time
                    k->pipeline = false;
                    break;
                }
                const cplx* hc = (const cplx*)(host + (size_t)s * colb);
                for (int32_t i = 0; i <= j + s + 1; ++i) k->hcol[i] = zc(hc[i].re, hc[i].im);
                ++op->st.op_applies;
                op->st.spmv_calls += (op->Kmul ? 2 : 1) + (op->refine ? 1 : 0);
                op->st.sptrsv_calls += op->refine ? 4 : 2;
                if (op->refine) ++op->st.refined_solves;
                op->st.last_rel_res = bnorm > 0.0 ? beta0 / bnorm : 0.0;
                op->st.max_rel_res = std::max(op->st.max_rel_res, op->st.last_rel_res);
                const int what = take_column(j + s, k->hcol.data());
                if (what < 0) {
                    op->st.seconds_solve += now_s() - t0;
                    return lsa_set_error(ctx, LSA_ERR_NONFINITE, "Arnoldi: non-finite norm at step %d", j + s);
                }
                if (what > 0) {  // the steps queued behind a breakdown worked on noise: the caller restarts from here
                    if (breakdown) *breakdown = j + s;
                    op->st.seconds_solve += now_s() - t0;
                    return LSA_OK;
                }
            }
            j += accepted;
            if (accepted < nb) k->t_for = -1;  // the steps behind the one that stopped the batch ran on; t is theirs
        }
    }
    k->t_for = -1;
    for (; j < j1; ++j) {
        const void* vj = (char*)k->V + (size_t)j * vb;
        void* vn = (char*)k->V + (size_t)(j + 1) * vb;
        int rc = op_apply_dev(ctx, op, vj, k->w);
        if (rc != LSA_OK) {
            op->st.seconds_solve += now_s() - t0;
            return rc;
        }
        LSA_CHECK(orthonormalize(ctx, LSA_C128, k->n, k->V, k->n, j + 1, k->w, vn, k->ow, k->hcol.data()));
        const int what = take_column(j, k->hcol.data());
        if (what < 0) return lsa_set_error(ctx, LSA_ERR_NONFINITE, "Arnoldi: non-finite norm at step %d", j);
        if (what > 0) {
            if (breakdown) *breakdown = j;
            break;
        }
    }
    op->st.seconds_solve += now_s() - t0;
    return LSA_OK;
}

int lsa_krylov_restart(lsa_ctx* ctx, lsa_krylov* k, int32_t m, int32_t knew, const void* Q, int32_t ldq) {
    if (!ctx || !k || !Q) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_restart: null argument");
    if (m < 1 || m > k->ncv || knew < 0 || knew > m || ldq < m) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_restart: bad sizes m=%d knew=%d", m, knew);
    const size_t vb = (size_t)k->n * 16;
    if (knew > 0) {
        // pack Q densely (m x knew) and upload
        LSA_CHECK(lsa_ensure_scratch(ctx, 0, (size_t)m * knew * 16));
        const cplx* Qh = (const cplx*)Q;
        cplx* p = (cplx*)ctx->pinned;
        for (int32_t c = 0; c < knew; ++c)
            for (int32_t r = 0; r < m; ++r) p[(size_t)c * m + r] = Qh[(size_t)c * ldq + r];
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(k->qdev, p, (size_t)m * knew * 16, hipMemcpyHostToDevice, ctx->stream));
        LSA_CHECK(k_basis_gemm(ctx, LSA_C128, k->n, m, knew, k->V, k->n, k->qdev, m, k->V2, k->n));
    }
    LSA_CHECK(k_copy(ctx, LSA_C128, k->n, (char*)k->V + (size_t)m * vb, (char*)k->V2 + (size_t)knew * vb));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::swap(k->V, k->V2);
    return LSA_OK;
}

int lsa_krylov_ritz_vectors(lsa_ctx* ctx, lsa_krylov* k, int32_t m, int32_t nvec, const void* Y, int32_t ldy, int normalise,
                            void* X) {
    if (!ctx || !k || !Y || !X) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_ritz_vectors: null argument");
    if (m < 1 || m > k->ncv + 1 || nvec < 0 || nvec > k->ncv + 1 || ldy < m) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_krylov_ritz_vectors: bad sizes");
    if (nvec == 0) return LSA_OK;
    const size_t vb = (size_t)k->n * 16;
    LSA_CHECK(lsa_ensure_scratch(ctx, 0, (size_t)m * nvec * 16 + (size_t)nvec * sizeof(double)));
    const cplx* Yh = (const cplx*)Y;
    cplx* p = (cplx*)ctx->pinned;
    for (int32_t c = 0; c < nvec; ++c)
        for (int32_t r = 0; r < m; ++r) p[(size_t)c * m + r] = Yh[(size_t)c * ldy + r];
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(k->qdev, p, (size_t)m * nvec * 16, hipMemcpyHostToDevice, ctx->stream));
    LSA_CHECK(k_basis_gemm(ctx, LSA_C128, k->n, m, nvec, k->V, k->n, k->qdev, m, k->V2, k->n));
    k->imag_norms.clear();
    if (normalise & 2) {
        // unit norm and canonical phase of all columns in three launches; the norms of the imaginary parts come back with them
        if (!k->imag2) LSA_HIP_ALLOC(ctx, hipMalloc((void**)&k->imag2, (size_t)(k->ncv + 2) * sizeof(double)));
        LSA_CHECK(k_columns_canonical(ctx, k->n, nvec, k->V2, k->n, normalise & 1, k->imag2));
        k->imag_norms.assign((size_t)nvec, 0.0);
        // (into the pinned scratch, behind the packed Y: an asynchronous copy to pageable memory is staged by the runtime)
        LSA_HIP_CHECK(ctx, hipMemcpyAsync((char*)ctx->pinned + (size_t)m * nvec * 16, k->imag2, (size_t)nvec * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    } else if (normalise) {
        for (int32_t c = 0; c < nvec; ++c) {
            void* col = (char*)k->V2 + (size_t)c * vb;
            LSA_CHECK(k_nrm2(ctx, LSA_C128, k->n, col, k->ow.nrm2));
            LSA_CHECK(k_copy(ctx, LSA_C128, k->n, col, k->w));
            LSA_CHECK(k_scale_by_inv_norm(ctx, LSA_C128, k->n, k->w, k->ow.nrm2, col));
        }
    }
    const void* src = k->V2;
    void* tmp = nullptr;
    if (k->row_perm) {  // rows back into the caller's numbering on the device (a fancy-indexed scatter of n x nvec on the host costs more than the solve's Schur forms)
        if (!k->xtmp) LSA_HIP_ALLOC(ctx, hipMalloc(&k->xtmp, vb * (size_t)(k->ncv + 1)));
        tmp = k->xtmp;
        const int blocks = (int)std::min<int64_t>((k->n + 255) / 256, (int64_t)ctx->num_cu * 16);
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks, nvec), dim3(256), 0, ctx->stream, k->n, k->row_perm, (const cplx*)k->V2, (cplx*)tmp);
        src = tmp;
    }
    hipError_t e = hipMemcpyAsync(X, src, vb * (size_t)nvec, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_krylov_ritz_vectors: download failed: %s", hipGetErrorString(e));
    for (size_t c = 0; c < k->imag_norms.size(); ++c) k->imag_norms[c] = std::sqrt(((const double*)((const char*)ctx->pinned + (size_t)m * nvec * 16))[c]);
    return LSA_OK;
}

int lsa_krylov_imag_norms(const lsa_krylov* k, int32_t nvec, double* out) {
    if (!k || !out || nvec < 0 || (size_t)nvec > k->imag_norms.size()) return LSA_ERR_ARG;
    for (int32_t c = 0; c < nvec; ++c) out[c] = k->imag_norms[(size_t)c];
    return LSA_OK;
}

int lsa_eig_residuals(lsa_ctx* ctx, const lsa_mat* A, const lsa_mat* M, int32_t nvec, const void* lam, const void* X, double* res) {
    if (!ctx || !A || !lam || !X || !res || nvec < 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_eig_residuals: bad argument");
    const int64_t n = A->n;
    const size_t vb = (size_t)std::max<int64_t>(n, 1) * 16;
    void *x = nullptr, *ax = nullptr, *mx = nullptr;
    double* nr = nullptr;
    int rc = LSA_OK;
    if (hipMalloc(&x, vb) != hipSuccess || hipMalloc(&ax, vb) != hipSuccess || hipMalloc(&mx, vb) != hipSuccess ||
        hipMalloc((void**)&nr, 4 * sizeof(double)) != hipSuccess)
        rc = lsa_set_error(ctx, LSA_ERR_HIP, "lsa_eig_residuals: out of device memory");
    const cplx* lamh = (const cplx*)lam;
    for (int32_t c = 0; c < nvec && rc == LSA_OK; ++c) {
        double nax = 0, nmx = 0, nrr = 0;
        if (hipMemcpyAsync(x, (const char*)X + (size_t)c * (size_t)n * 16, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = lsa_set_error(ctx, LSA_ERR_HIP, "lsa_eig_residuals: upload failed");
            break;
        }
        rc = k_spmv(ctx, A, LSA_C128, x, ax);
        if (rc == LSA_OK) rc = M ? k_spmv(ctx, M, LSA_C128, x, mx) : k_copy(ctx, LSA_C128, n, x, mx);
        if (rc == LSA_OK) rc = device_norm(ctx, LSA_C128, n, ax, nr, &nax);
        if (rc == LSA_OK) rc = device_norm(ctx, LSA_C128, n, mx, nr, &nmx);
        const double ml[2] = {-lamh[c].re, -lamh[c].im};
        if (rc == LSA_OK) rc = k_axpy(ctx, LSA_C128, n, ml, mx, ax);
        if (rc == LSA_OK) rc = device_norm(ctx, LSA_C128, n, ax, nr, &nrr);
        const double lmag = std::hypot(lamh[c].re, lamh[c].im);
        res[c] = nrr / (nax + lmag * nmx + 1e-16);
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : {x, ax, mx, (void*)nr})
        if (p) (void)hipFree(p);
    return rc;
}

}  // extern "C"
