// Blocked sparse triangular solve for the ILU(k) factors (algo 2).
//
// FEM factors in RCM order have thousands of dependency levels with a handful of rows each, so a row-parallel
// SpTRSV is a chain of ~n/8 latency-bound hand-offs.  Here the rows are cut into contiguous blocks of B rows:
//
//     x_b = inv(L_bb) * ( rhs_b - L_b,<b x_<b )        b = 0 .. nb-1        (U: b = nb-1 .. 0 with blocks to the right)
//
// * the diagonal blocks L_bb / U_bb are inverted ONCE on the device into dense B x B triangles (row-major, one
//   thread per column, forward/backward substitution over the sparse rows) -- n*B scalars per factor, which is what
//   288 GB of HBM is for;
// * every solve is then 2*nb dependent launches per factor instead of thousands of levels: a sparse update with the
//   entries outside the diagonal block (coalesced CSR row reads, x gathered from earlier blocks) and a dense
//   triangular mat-vec (one wavefront per row, 1 KiB contiguous loads).  Both are HBM streams.
// * the launch chain of a full apply (L then U) is captured once into a hipGraph and replayed, so the host pays one
//   graph launch per preconditioner apply.
//
// The result is the same triangular solve (same factors, different summation order), checked against the C oracle.
#include <algorithm>

#include "lsa_internal.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void build_linv_kernel(int32_t n, int32_t B, const int32_t* __restrict__ ci,
                                                         const int32_t* __restrict__ diag, const int32_t* __restrict__ lsplit,
                                                         const T* __restrict__ val, T* inv) {
    // blockDim.x divides B, so all threads of a workgroup sit in one diagonal block and share the row loop
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t bs = ((blockIdx.x * blockDim.x) / B) * B;
    const int32_t be = (bs + B < n) ? bs + B : n;
    const bool live = j < n;
    const int32_t jl = j - bs;
    for (int32_t i = bs; i < be; ++i) {
        if (!live || i < j) continue;
        T* out = inv + (size_t)i * B + jl;
        if (i == j) {
            s_from(*out, 1.0, 0.0);
            continue;
        }
        T s = scalar_traits<T>::zero();
        for (int32_t p = lsplit[i]; p < diag[i]; ++p) {
            const int32_t c = ci[p];
            if (c >= j) fma_acc(s, val[p], inv[(size_t)c * B + jl]);
        }
        s_from(*out, 0.0, 0.0);
        *out = s_sub(*out, s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void build_uinv_kernel(int32_t n, int32_t B, const int32_t* __restrict__ ci,
                                                         const int32_t* __restrict__ diag, const int32_t* __restrict__ usplit,
                                                         const T* __restrict__ val, const T* __restrict__ dinv, T* inv) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t bs = ((blockIdx.x * blockDim.x) / B) * B;
    const int32_t be = (bs + B < n) ? bs + B : n;
    const bool live = j < n;
    const int32_t jl = j - bs;
    for (int32_t i = be - 1; i >= bs; --i) {
        if (!live || i > j) continue;
        T* out = inv + (size_t)i * B + jl;
        if (i == j) {
            *out = dinv[j];
            continue;
        }
        T s = scalar_traits<T>::zero();
        for (int32_t p = diag[i] + 1; p < usplit[i]; ++p) {
            const int32_t c = ci[p];
            if (c <= j) fma_acc(s, val[p], inv[(size_t)c * B + jl]);
        }
        T z = scalar_traits<T>::zero();
        *out = s_sub(z, s_mul(dinv[i], s));
    }
}

// t[r] = rhs[r] - sum over the entries of row r outside its diagonal block of val * x[col]
template <typename MT, typename VT, bool LOWER>
__global__ __launch_bounds__(256) void blk_sparse_kernel(int32_t bs, int32_t be, const int32_t* __restrict__ rp,
                                                         const int32_t* __restrict__ ci, const int32_t* __restrict__ split,
                                                         const MT* __restrict__ val, const VT* __restrict__ rhs,
                                                         const VT* __restrict__ x, VT* __restrict__ t) {
    const int32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 15;
    const int32_t r = bs + (gid >> 4);
    if (r >= be) return;
    const int32_t p0 = LOWER ? rp[r] : split[r];
    const int32_t p1 = LOWER ? split[r] : rp[r + 1];
    VT acc = scalar_traits<VT>::zero();
    for (int32_t p = p0 + lane; p < p1; p += 16) fma_acc(acc, val[p], x[ci[p]]);
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) {
        if constexpr (sizeof(VT) == 16) {
            acc.re += __shfl_xor(acc.re, m, 64);
            acc.im += __shfl_xor(acc.im, m, 64);
        } else {
            acc += __shfl_xor(acc, m, 64);
        }
    }
    if (lane == 0) t[r] = s_sub(rhs[r], acc);
}

// x[r] = sum_s inv[r, s] t[s] over the triangle of the diagonal block; one wavefront per row, 1 KiB contiguous per
// load instruction, eight independent loads in flight per lane (a row of 1024 entries is two sweeps), because a block
// step moves only a few MB and is bound by memory latency, not bandwidth.
constexpr int kDenseRows = 4;  // rows (= wavefronts) per 256-thread workgroup
template <typename MT, typename VT, bool LOWER>
__global__ __launch_bounds__(256) void blk_dense_kernel(int32_t bs, int32_t be, int32_t B, const MT* __restrict__ inv,
                                                        const VT* __restrict__ t, VT* __restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int32_t r = bs + blockIdx.x * kDenseRows + (threadIdx.x >> 6);
    if (r >= be) return;
    const int32_t s0 = LOWER ? bs : r;
    const int32_t s1 = LOWER ? r + 1 : be;
    const MT* row = inv + (size_t)r * B - bs;  // row[s] for global s
    VT acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = scalar_traits<VT>::zero();
    int32_t s = s0 + lane;
    for (; s + 7 * 64 < s1; s += 8 * 64) {
        MT a[8];
        VT tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = row[s + k * 64];
            tv[k] = t[s + k * 64];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) fma_acc(acc[k & 3], a[k], tv[k]);
    }
    for (; s < s1; s += 64) fma_acc(acc[0], row[s], t[s]);
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
    if (lane == 0) x[r] = v;
}

template <typename MT, typename VT, bool LOWER>
void launch_factor(lsa_ctx* ctx, lsa_ilu* pc, const VT* rhs, VT* x, VT* t) {
    const int32_t B = pc->blk_B, nb = pc->blk_nb, n = pc->n;
    for (int32_t k = 0; k < nb; ++k) {
        const int32_t b = LOWER ? k : nb - 1 - k;
        const int32_t bs = b * B, be = std::min(n, bs + B);
        const int rows = be - bs;
        hipLaunchKernelGGL((blk_sparse_kernel<MT, VT, LOWER>), dim3((rows * 16 + 255) / 256), dim3(256), 0, ctx->stream, bs, be, pc->rp,
                           pc->ci, LOWER ? pc->lsplit : pc->usplit, (const MT*)pc->val, rhs, (const VT*)x, t);
        hipLaunchKernelGGL((blk_dense_kernel<MT, VT, LOWER>), dim3((rows + kDenseRows - 1) / kDenseRows), dim3(256), 0, ctx->stream, bs, be, B,
                           (const MT*)(LOWER ? pc->linv : pc->uinv), (const VT*)t, x);
    }
}

template <typename MT, typename VT>
int solve_blocked(lsa_ctx* ctx, lsa_ilu* pc, int which, const void* b, void* x) {
    const int vd = scalar_traits<VT>::dtype;
    const size_t vb = (size_t)std::max<int32_t>(pc->n, 1) * sizeof(VT);
    if (!pc->blk_t[vd]) {
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        LSA_HIP_CHECK(ctx, hipMalloc(&pc->blk_t[vd], vb));
        LSA_HIP_CHECK(ctx, hipMalloc(&pc->blk_y[vd], vb));
        LSA_HIP_CHECK(ctx, hipMalloc(&pc->blk_in[vd], vb));
        LSA_HIP_CHECK(ctx, hipMalloc(&pc->blk_out[vd], vb));
    }
    VT* t = (VT*)pc->blk_t[vd];
    if (which == 0) launch_factor<MT, VT, true>(ctx, pc, (const VT*)b, (VT*)x, t);
    else if (which == 1) launch_factor<MT, VT, false>(ctx, pc, (const VT*)b, (VT*)x, t);
    else {
        VT *in = (VT*)pc->blk_in[vd], *y = (VT*)pc->blk_y[vd], *out = (VT*)pc->blk_out[vd];
        static const bool use_graph = !(getenv("LSA_SPTRSV_GRAPH") && atoi(getenv("LSA_SPTRSV_GRAPH")) == 0);
        if (!use_graph) {  // plain launches (rocprofv3's kernel trace cannot follow graph replays on ROCm 7.2)
            launch_factor<MT, VT, true>(ctx, pc, (const VT*)b, y, t);
            launch_factor<MT, VT, false>(ctx, pc, y, (VT*)x, t);
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "blocked SpTRSV launch failed: %s", hipGetErrorString(le));
            return LSA_OK;
        }
        // full apply through a captured graph on fixed buffers
        if (!pc->blk_graph[vd]) {
            hipGraph_t graph = nullptr;
            LSA_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            launch_factor<MT, VT, true>(ctx, pc, in, y, t);
            launch_factor<MT, VT, false>(ctx, pc, y, out, t);
            hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (e != hipSuccess || !graph) return lsa_set_error(ctx, LSA_ERR_HIP, "blocked SpTRSV: graph capture failed: %s", hipGetErrorString(e));
            hipGraphExec_t exec = nullptr;
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "blocked SpTRSV: graph instantiate failed: %s", hipGetErrorString(e));
            pc->blk_graph[vd] = (void*)exec;
        }
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(in, b, (size_t)pc->n * sizeof(VT), hipMemcpyDeviceToDevice, ctx->stream));
        LSA_HIP_CHECK(ctx, hipGraphLaunch((hipGraphExec_t)pc->blk_graph[vd], ctx->stream));
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(x, out, (size_t)pc->n * sizeof(VT), hipMemcpyDeviceToDevice, ctx->stream));
        return LSA_OK;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "blocked SpTRSV launch failed: %s", hipGetErrorString(e));
    return LSA_OK;
}

template <typename T>
int build_inverses(lsa_ctx* ctx, lsa_ilu* pc) {
    const int32_t B = pc->blk_B, n = pc->n;
    const int threads = 256;  // divides B (B is a multiple of 256)
    const int blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL((build_linv_kernel<T>), dim3(blocks), dim3(threads), 0, ctx->stream, n, B, pc->ci, pc->diag, pc->lsplit,
                       (const T*)pc->val, (T*)pc->linv);
    hipLaunchKernelGGL((build_uinv_kernel<T>), dim3(blocks), dim3(threads), 0, ctx->stream, n, B, pc->ci, pc->diag, pc->usplit,
                       (const T*)pc->val, (const T*)pc->dinv, (T*)pc->uinv);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "block inverse launch failed: %s", hipGetErrorString(e));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

}  // namespace

void blk_release(lsa_ilu* pc) {
    for (int vd = 0; vd < 2; ++vd) {
        if (pc->blk_graph[vd]) (void)hipGraphExecDestroy((hipGraphExec_t)pc->blk_graph[vd]);
        pc->blk_graph[vd] = nullptr;
        for (void** p : {&pc->blk_t[vd], &pc->blk_y[vd], &pc->blk_in[vd], &pc->blk_out[vd]}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
    }
    for (void** p : {(void**)&pc->lsplit, (void**)&pc->usplit, &pc->linv, &pc->uinv}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    pc->blk_B = pc->blk_nb = 0;
}

// (re)build the blocked form with block size B (rounded up to a multiple of 256)
int blk_setup(lsa_ctx* ctx, lsa_ilu* pc, int32_t B) {
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    blk_release(pc);
    if (pc->n == 0) return LSA_OK;
    B = std::max(256, ((B + 255) / 256) * 256);
    const int32_t n = pc->n;
    pc->blk_B = B;
    pc->blk_nb = (n + B - 1) / B;
    std::vector<int32_t> ls((size_t)n), us((size_t)n);
    for (int32_t r = 0; r < n; ++r) {
        const int32_t bs = (r / B) * B, be = std::min(n, bs + B);
        const int32_t* c = pc->h_ci.data();
        ls[r] = (int32_t)(std::lower_bound(c + pc->h_rp[r], c + pc->h_diag[r], bs) - c);
        us[r] = (int32_t)(std::lower_bound(c + pc->h_diag[r] + 1, c + pc->h_rp[r + 1], be) - c);
    }
    const size_t esz = pc->dtype == LSA_C128 ? 16 : 8;
    const size_t inv_bytes = (size_t)pc->blk_nb * B * (size_t)B * esz;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (2 * inv_bytes > free_b / 2) {
        blk_release(pc);
        return lsa_set_error(ctx, LSA_ERR_HIP, "blocked SpTRSV: 2 x %.1f GB of inverted diagonal blocks do not fit (%.1f GB free)",
                             inv_bytes / 1e9, free_b / 1e9);
    }
    bool ok = hipMalloc((void**)&pc->lsplit, sizeof(int32_t) * (size_t)n) == hipSuccess &&
              hipMalloc((void**)&pc->usplit, sizeof(int32_t) * (size_t)n) == hipSuccess && hipMalloc(&pc->linv, inv_bytes) == hipSuccess &&
              hipMalloc(&pc->uinv, inv_bytes) == hipSuccess;
    if (!ok) {
        blk_release(pc);
        return lsa_set_error(ctx, LSA_ERR_HIP, "blocked SpTRSV: out of device memory");
    }
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(pc->lsplit, ls.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(pc->usplit, us.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemsetAsync(pc->linv, 0, inv_bytes, ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemsetAsync(pc->uinv, 0, inv_bytes, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return pc->dtype == LSA_C128 ? build_inverses<cplx>(ctx, pc) : build_inverses<double>(ctx, pc);
}

int blk_solve(lsa_ctx* ctx, lsa_ilu* pc, int which, int vdtype, const void* b, void* x) {
    if (pc->dtype == LSA_F64 && vdtype == LSA_F64) return solve_blocked<double, double>(ctx, pc, which, b, x);
    if (pc->dtype == LSA_F64 && vdtype == LSA_C128) return solve_blocked<double, cplx>(ctx, pc, which, b, x);
    if (pc->dtype == LSA_C128 && vdtype == LSA_C128) return solve_blocked<cplx, cplx>(ctx, pc, which, b, x);
    return lsa_set_error(ctx, LSA_ERR_ARG, "ilu_solve: complex factors need complex vectors");
}
