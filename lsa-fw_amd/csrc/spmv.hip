// CSR SpMV for gfx950 (MatMult of the eigen path: y = M x once per Arnoldi step, y = C x once per inner
// GMRES iteration).  HBM-bound: 12|20 B per stored entry plus the vectors; no MFMA (sparse contraction).
//
// Kernel shape: a 64-lane wavefront is split into sub-waves of LPR lanes, one sub-wave per row, so the
// column/value loads of a row are contiguous across lanes (coalesced HBM row reads) and the gathers of x
// hit L2 / Infinity Cache (FEM rows after RCM reference a narrow band of x).  Partial sums are combined
// with DPP/shuffle butterflies inside the sub-wave.  LPR is picked from the mean row length.
#include "lsa_internal.h"

template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int mask);
template <>
__device__ __forceinline__ double shfl_xor_t<double>(double v, int mask) {
    return __shfl_xor(v, mask, 64);
}
template <>
__device__ __forceinline__ cplx shfl_xor_t<cplx>(cplx v, int mask) {
    return cplx{__shfl_xor(v.re, mask, 64), __shfl_xor(v.im, mask, 64)};
}

template <typename MT, typename VT, int LPR>
__global__ __launch_bounds__(256) void spmv_subwave_kernel(int32_t n, const int32_t* __restrict__ rp,
                                                           const int32_t* __restrict__ ci, const MT* __restrict__ val,
                                                           const VT* __restrict__ x, VT* __restrict__ y) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    int64_t row = gid / LPR;
    const int64_t row_stride = ((int64_t)gridDim.x * blockDim.x) / LPR;
    for (; row < n; row += row_stride) {
        const int32_t p0 = rp[row], p1 = rp[row + 1];
        VT acc = scalar_traits<VT>::zero();
        for (int32_t p = p0 + lane; p < p1; p += LPR) fma_acc(acc, val[p], x[ci[p]]);
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) acc = s_add(acc, shfl_xor_t<VT>(acc, m));
        if (lane == 0) y[row] = acc;
    }
}

template <typename MT, typename VT, int LPR>
static void launch_spmv(lsa_ctx* ctx, const lsa_mat* A, const void* x, void* y) {
    const int threads = 256;
    int64_t want = ((int64_t)A->n * LPR + threads - 1) / threads;
    int64_t cap = (int64_t)ctx->num_cu * 64;  // grid-stride beyond this
    int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    hipLaunchKernelGGL((spmv_subwave_kernel<MT, VT, LPR>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, A->rp, A->ci,
                       (const MT*)A->val, (const VT*)x, (VT*)y);
}

template <typename MT, typename VT>
static void dispatch_spmv(lsa_ctx* ctx, const lsa_mat* A, const void* x, void* y) {
    const double mean = A->n > 0 ? (double)A->nnz / (double)A->n : 0.0;
    if (mean <= 6.0) launch_spmv<MT, VT, 4>(ctx, A, x, y);
    else if (mean <= 12.0) launch_spmv<MT, VT, 8>(ctx, A, x, y);
    else if (mean <= 48.0) launch_spmv<MT, VT, 16>(ctx, A, x, y);
    else if (mean <= 96.0) launch_spmv<MT, VT, 32>(ctx, A, x, y);
    else launch_spmv<MT, VT, 64>(ctx, A, x, y);
}

int k_spmv(lsa_ctx* ctx, const lsa_mat* A, int xdtype, const void* x, void* y) {
    if (A->dtype == LSA_F64 && xdtype == LSA_F64) dispatch_spmv<double, double>(ctx, A, x, y);
    else if (A->dtype == LSA_F64 && xdtype == LSA_C128) dispatch_spmv<double, cplx>(ctx, A, x, y);
    else if (A->dtype == LSA_C128 && xdtype == LSA_C128) dispatch_spmv<cplx, cplx>(ctx, A, x, y);
    else return lsa_set_error(ctx, LSA_ERR_ARG, "spmv: a complex matrix needs complex vectors");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "spmv launch failed: %s", hipGetErrorString(e));
    return LSA_OK;
}

// ---- y = A^T x / A^H x by scatter (adjoint eigenproblem; not on the inner-loop path) ----------------------------
__device__ __forceinline__ void atomic_add_t(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_t(cplx* p, cplx v) {
    unsafeAtomicAdd(&p->re, v.re);
    unsafeAtomicAdd(&p->im, v.im);
}

template <typename MT, typename VT>
__global__ __launch_bounds__(256) void spmv_transpose_kernel(int32_t n, int conj, const int32_t* __restrict__ rp,
                                                             const int32_t* __restrict__ ci, const MT* __restrict__ val,
                                                             const VT* __restrict__ x, VT* __restrict__ y) {
    constexpr int LPR = 16;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    const int64_t row_stride = ((int64_t)gridDim.x * blockDim.x) / LPR;
    for (int64_t row = gid / LPR; row < n; row += row_stride) {
        const VT xr = x[row];
        for (int32_t p = rp[row] + lane; p < rp[row + 1]; p += LPR) {
            MT a = val[p];
            if (conj) a = s_conj(a);
            atomic_add_t(&y[ci[p]], s_mul(a, xr));
        }
    }
}

int k_spmv_transpose(lsa_ctx* ctx, const lsa_mat* A, int conj, int xdtype, const void* x, void* y) {
    if (A->row0 != 0 || A->n != A->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "spmv_transpose: sharded matrices are not supported");
    LSA_CHECK(k_set_zero(ctx, xdtype, A->ncols, y));
    const int threads = 256;
    int64_t want = ((int64_t)A->n * 16 + threads - 1) / threads;
    int64_t cap = (int64_t)ctx->num_cu * 32;
    int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    if (A->dtype == LSA_F64 && xdtype == LSA_F64)
        hipLaunchKernelGGL((spmv_transpose_kernel<double, double>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, conj, A->rp,
                           A->ci, (const double*)A->val, (const double*)x, (double*)y);
    else if (A->dtype == LSA_F64 && xdtype == LSA_C128)
        hipLaunchKernelGGL((spmv_transpose_kernel<double, cplx>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, conj, A->rp,
                           A->ci, (const double*)A->val, (const cplx*)x, (cplx*)y);
    else if (A->dtype == LSA_C128 && xdtype == LSA_C128)
        hipLaunchKernelGGL((spmv_transpose_kernel<cplx, cplx>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, conj, A->rp,
                           A->ci, (const cplx*)A->val, (const cplx*)x, (cplx*)y);
    else return lsa_set_error(ctx, LSA_ERR_ARG, "spmv_transpose: a complex matrix needs complex vectors");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "spmv_transpose launch failed: %s", hipGetErrorString(e));
    return LSA_OK;
}

// ---- C-ABI -----------------------------------------------------------------------------------------------------
extern "C" {

static int check_spmv_args(lsa_ctx* ctx, const lsa_mat* A, const lsa_vec* x, const lsa_vec* y, const char* who) {
    if (!ctx || !A || !x || !y) return lsa_set_error(ctx, LSA_ERR_ARG, "%s: null argument", who);
    if (x->n != A->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "%s: x has length %lld, matrix has %d columns", who, (long long)x->n, A->ncols);
    if (x->dtype != y->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "%s: x and y must have one dtype", who);
    if (x->d == y->d) return lsa_set_error(ctx, LSA_ERR_ARG, "%s: x and y must not alias", who);
    return LSA_OK;
}

int lsa_spmv(lsa_ctx* ctx, const lsa_mat* A, const lsa_vec* x, lsa_vec* y) {
    LSA_CHECK(check_spmv_args(ctx, A, x, y, "lsa_spmv"));
    if (y->n != A->n && y->n != A->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_spmv: y has length %lld, matrix has %d rows", (long long)y->n, A->n);
    // a shard writes its own rows of a global-length y
    char* yp = (char*)y->d;
    if (y->n == A->ncols && A->n != A->ncols) yp += (size_t)A->row0 * (y->dtype == LSA_C128 ? 16 : 8);
    return k_spmv(ctx, A, x->dtype, x->d, yp);
}

int lsa_spmv_transpose(lsa_ctx* ctx, const lsa_mat* A, int conj, const lsa_vec* x, lsa_vec* y) {
    LSA_CHECK(check_spmv_args(ctx, A, x, y, "lsa_spmv_transpose"));
    if (y->n != A->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_spmv_transpose: y has the wrong length");
    return k_spmv_transpose(ctx, A, conj, x->dtype, x->d, y->d);
}

int lsa_spmv_time(lsa_ctx* ctx, const lsa_mat* A, const lsa_vec* x, lsa_vec* y, int iters, double* avg_ms) {
    LSA_CHECK(check_spmv_args(ctx, A, x, y, "lsa_spmv_time"));
    if (iters <= 0 || !avg_ms) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_spmv_time: iters must be positive");
    if (y->n != A->n) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_spmv_time: y has the wrong length");
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; ++i) LSA_CHECK(k_spmv(ctx, A, x->dtype, x->d, y->d));
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    LSA_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    LSA_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / iters;
    return LSA_OK;
}

}  // extern "C"
