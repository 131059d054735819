// CSR SpMV for gfx950 (MatMult of the eigen path: y = M x once per Arnoldi step, y = C x once per inner
// GMRES iteration).  HBM-bound: 12|20 B per stored entry plus the vectors; no MFMA (sparse contraction).
//
// Kernel shape: a 64-lane wavefront is split into sub-waves of LPR lanes, one sub-wave per row, so the
// column/value loads of a row are contiguous across lanes (coalesced HBM row reads) and the gathers of x
// hit L2 / Infinity Cache (FEM rows after RCM reference a narrow band of x).  Partial sums are combined
// with DPP/shuffle butterflies inside the sub-wave.  LPR is picked from the mean row length.
#include <algorithm>

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

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
    // matrix entries are read exactly once: a non-temporal load keeps them from evicting x out of L2 / Infinity Cache
    if constexpr (NT) {
        if constexpr (sizeof(T) == 16) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d v = __builtin_nontemporal_load((const v2d*)p);  // one 16-byte load
            T out;
            s_from(out, v.x, v.y);
            return out;
        } else {
            return __builtin_nontemporal_load(p);
        }
    } else {
        return *p;
    }
}

template <typename MT, typename VT, int LPR, bool NT>
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
        for (int32_t p = p0 + lane; p < p1; p += LPR) fma_acc(acc, stream_load<NT>(val + p), x[stream_load<NT>(ci + p)]);
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) acc = s_add(acc, shfl_xor_t<VT>(acc, m));
        if (lane == 0) y[row] = acc;
    }
}

// Two rows per sub-wave in flight (more independent loads per lane): rows r and r + stride/2 are processed together.
template <typename MT, typename VT, int LPR, bool NT>
__global__ __launch_bounds__(256) void spmv_subwave2_kernel(int32_t n, const int32_t* __restrict__ rp,
                                                            const int32_t* __restrict__ ci, const MT* __restrict__ val,
                                                            const VT* __restrict__ x, VT* __restrict__ y) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    const int64_t half = ((int64_t)gridDim.x * blockDim.x) / LPR;
    for (int64_t row = gid / LPR; row < n; row += 2 * half) {
        const int64_t row2 = row + half;
        const bool has2 = row2 < n;
        const int32_t p0 = rp[row], p1 = rp[row + 1];
        const int32_t q0 = has2 ? rp[row2] : 0, q1 = has2 ? rp[row2 + 1] : 0;
        VT acc = scalar_traits<VT>::zero(), acc2 = scalar_traits<VT>::zero();
        int32_t p = p0 + lane, q = q0 + lane;
        for (; p < p1 && q < q1; p += LPR, q += LPR) {
            const int32_t c1 = stream_load<NT>(ci + p), c2 = stream_load<NT>(ci + q);
            const MT a1 = stream_load<NT>(val + p), a2 = stream_load<NT>(val + q);
            fma_acc(acc, a1, x[c1]);
            fma_acc(acc2, a2, x[c2]);
        }
        for (; p < p1; p += LPR) fma_acc(acc, stream_load<NT>(val + p), x[stream_load<NT>(ci + p)]);
        for (; q < q1; q += LPR) fma_acc(acc2, stream_load<NT>(val + q), x[stream_load<NT>(ci + q)]);
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) {
            acc = s_add(acc, shfl_xor_t<VT>(acc, m));
            acc2 = s_add(acc2, shfl_xor_t<VT>(acc2, m));
        }
        if (lane == 0) {
            y[row] = acc;
            if (has2) y[row2] = acc2;
        }
    }
}

// XCD-aware form: workgroup b is dealt to XCD (b % 8) by the dispatcher, and each XCD has its own 4 MB L2.  With rows
// interleaved over workgroups every XCD ends up gathering the WHOLE of x (8x the vector in L2 misses: measured 0.63 GB
// of extra fetch on SROOF = 8 x 80 MB).  Here workgroup b works on the contiguous row chunk
// (b % 8) * (G / 8) + b / 8, so an XCD walks one eighth of the rows and touches one eighth of x (plus the band).
template <typename MT, typename VT, int LPR, bool NT>
__global__ __launch_bounds__(256) void spmv_xcd_kernel(int32_t n, int32_t rows_per_wg, const int32_t* __restrict__ rp,
                                                       const int32_t* __restrict__ ci, const MT* __restrict__ val,
                                                       const VT* __restrict__ x, VT* __restrict__ y) {
    const int32_t G = gridDim.x;  // multiple of 8
    const int32_t chunk = (int32_t)(blockIdx.x & 7) * (G >> 3) + (int32_t)(blockIdx.x >> 3);
    const int32_t lane = threadIdx.x % LPR;
    const int64_t r0 = (int64_t)chunk * rows_per_wg;
    const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
    for (int64_t row = r0 + threadIdx.x / LPR; row < r1; row += 256 / LPR) {
        const int32_t p0 = rp[row], p1 = rp[row + 1];
        VT acc = scalar_traits<VT>::zero();
        for (int32_t p = p0 + lane; p < p1; p += LPR) fma_acc(acc, stream_load<NT>(val + p), x[stream_load<NT>(ci + p)]);
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) acc = s_add(acc, shfl_xor_t<VT>(acc, m));
        if (lane == 0) y[row] = acc;
    }
}

// 16-bit column indices relative to the first column of the row: 18 | 10 bytes per entry instead of 20 | 12
template <typename MT, typename VT, int LPR>
__global__ __launch_bounds__(256) void spmv_subwave16_kernel(int32_t n, const int32_t* __restrict__ rp, const uint16_t* __restrict__ ci16,
                                                             const int32_t* __restrict__ cbase, const MT* __restrict__ val,
                                                             const VT* __restrict__ x, VT* __restrict__ y) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    int64_t row = gid / LPR;
    const int64_t row_stride = ((int64_t)gridDim.x * blockDim.x) / LPR;
    for (; row < n; row += row_stride) {
        const int32_t p0 = rp[row], p1 = rp[row + 1];
        const VT* xr = x + cbase[row];
        VT acc = scalar_traits<VT>::zero();
        for (int32_t p = p0 + lane; p < p1; p += LPR) fma_acc(acc, val[p], xr[ci16[p]]);
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) acc = s_add(acc, shfl_xor_t<VT>(acc, m));
        if (lane == 0) y[row] = acc;
    }
}

// Row groups: the rows of one mesh node (ux, uy[, uz][, p]) have one and the same column pattern.  A sub-wave takes a
// whole group: the column indices are read once (from the group's first row) and every x entry is gathered once and used
// for all rows of the group -- 1 / g of the index bytes and of the gather instructions (g = 2.2 rows on the 2D
// Taylor-Hood pattern), and g independent value loads in flight per lane.  Rows of a group are consecutive and equally
// long, so row k of the group starts at p0 + k * len: no extra row-pointer loads.
// (Tried and dropped: one 16-byte record {first entry, length, first row, rows} per group with the next group's record
// requested ahead, instead of row pointers behind a group pointer: 692 vs 635 us on SROOF.)
// XCD: workgroup b runs on XCD b % 8 (observed dispatch order; speed only), each with its own L2.  With groups interleaved
// over workgroups every XCD gathers the whole of x; here workgroup b takes the contiguous chunk of groups number
// (b % 8) * (G / 8) + b / 8, so an XCD walks one eighth of the rows and touches one eighth of x (plus the band).
// sum over the 16 lanes of a DPP row, returned to every lane of it (quad_perm, row_half_mirror, row_mirror)
template <int CTRL>
__device__ __forceinline__ double dpp_row_mov(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bits & 0xFFFFFFFFull), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bits >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_row_mov<0xB1>(v);
    v += dpp_row_mov<0x4E>(v);
    v += dpp_row_mov<0x141>(v);
    v += dpp_row_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ cplx row16_sum(cplx v) { return cplx{row16_sum(v.re), row16_sum(v.im)}; }

template <typename MT, typename VT, int LPR, bool C16, bool XCD>
__global__ __launch_bounds__(256) void spmv_group_kernel(int32_t ngroups, int32_t chunk, const int4* __restrict__ gstart,
                                                         const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                         const uint16_t* __restrict__ ci16, const int32_t* __restrict__ cbase,
                                                         const MT* __restrict__ val, const VT* __restrict__ x, VT* __restrict__ y) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    int64_t stride = ((int64_t)gridDim.x * blockDim.x) / LPR, grp = gid / LPR, gend = ngroups;
    if constexpr (XCD) {
        const int64_t G = gridDim.x, vb = (int64_t)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        stride = 256 / LPR;
        grp = vb * chunk + threadIdx.x / LPR;
        gend = min((int64_t)ngroups, (vb + 1) * chunk);
    }
    for (; grp < gend; grp += stride) {
        const int4 gr = gstart[grp];  // (first row, rows, first entry, entries per row): one load, no second trip to the row pointers
        const int32_t r0 = gr.x, g = gr.y, p0 = gr.z, len = gr.w;
        const VT* xr = x;
        if constexpr (C16) xr += cbase[r0];
        VT acc[4] = {scalar_traits<VT>::zero(), scalar_traits<VT>::zero(), scalar_traits<VT>::zero(), scalar_traits<VT>::zero()};
        for (int32_t p = lane; p < len; p += LPR) {
            int32_t col;
            if constexpr (C16) col = ci16[p0 + p];
            else col = ci[p0 + p];
            const MT* v = val + p0 + p;
            MT a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < g) a[k] = v[(size_t)k * len];
            const VT xv = xr[col];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < g) fma_acc(acc[k], a[k], xv);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < g) {
                if constexpr (LPR == 16) acc[k] = row16_sum(acc[k]);  // a sub-wave of 16 is a DPP row: no trip through the LDS crossbar
                else {
#pragma unroll
                    for (int m = LPR / 2; m > 0; m >>= 1) acc[k] = s_add(acc[k], shfl_xor_t<VT>(acc[k], m));
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < g) y[r0 + k] = acc[k];
        }
    }
}

// builds A->grp_start from the host pattern; false when grouping does not pay (mean group below 1.5 rows)
static bool ensure_groups(const lsa_mat* A) {
    if (A->grp_state != 0) return A->grp_state > 0;
    A->grp_state = -1;
    const int32_t n = A->n;
    if (n < 1 || (int64_t)A->h_rp.size() != (int64_t)n + 1 || (int64_t)A->h_ci.size() != A->nnz) return false;
    std::vector<int32_t> gs;
    gs.reserve((size_t)n / 2 + 2);
    gs.push_back(0);
    int32_t cur = 1;
    for (int32_t r = 1; r < n; ++r) {
        const int32_t a0 = A->h_rp[(size_t)r - 1], a1 = A->h_rp[r], b1 = A->h_rp[(size_t)r + 1];
        const bool same = cur < 4 && (a1 - a0) == (b1 - a1) && (a1 == a0 || memcmp(&A->h_ci[a0], &A->h_ci[a1], sizeof(int32_t) * (size_t)(a1 - a0)) == 0);
        if (same) ++cur;
        else {
            gs.push_back(r);
            cur = 1;
        }
    }
    gs.push_back(n);
    const int32_t ng = (int32_t)gs.size() - 1;
    if ((double)n / (double)ng < 1.5) return false;
    // (first row, rows) per group; optionally ordered by row length (LSA_SPMV_SORT=1): the four sub-waves of a wavefront then
    // walk rows of one length instead of waiting for the longest of four
    std::vector<int32_t> pairs((size_t)ng * 4);
    std::vector<int32_t> order((size_t)ng);
    for (int32_t q = 0; q < ng; ++q) order[(size_t)q] = q;
    if (const char* e = getenv("LSA_SPMV_SORT"))
        if (atoi(e) != 0)
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                return A->h_rp[(size_t)gs[(size_t)x] + 1] - A->h_rp[gs[(size_t)x]] > A->h_rp[(size_t)gs[(size_t)y] + 1] - A->h_rp[gs[(size_t)y]];
            });
    for (int32_t q = 0; q < ng; ++q) {
        const int32_t r0 = gs[(size_t)order[(size_t)q]];
        pairs[(size_t)4 * q] = r0;
        pairs[(size_t)4 * q + 1] = gs[(size_t)order[(size_t)q] + 1] - r0;
        pairs[(size_t)4 * q + 2] = A->h_rp[r0];
        pairs[(size_t)4 * q + 3] = A->h_rp[(size_t)r0 + 1] - A->h_rp[r0];
    }
    gs.swap(pairs);
    if (hipMalloc((void**)&A->grp_start, gs.size() * sizeof(int32_t)) != hipSuccess) return false;
    if (hipMemcpy(A->grp_start, gs.data(), gs.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(A->grp_start);
        A->grp_start = nullptr;
        return false;
    }
    A->ngroups = ng;
    A->grp_state = 1;
    return true;
}

// builds A->ci16 / A->cbase from the host pattern; false when a row spans 65 536 columns or more
static bool ensure_ci16(const lsa_mat* A) {
    if (A->ci16_state != 0) return A->ci16_state > 0;
    A->ci16_state = -1;
    const int32_t n = A->n;
    if ((int64_t)A->h_rp.size() != (int64_t)n + 1 || (int64_t)A->h_ci.size() != A->nnz) return false;
    std::vector<uint16_t> d((size_t)std::max<int64_t>(A->nnz, 1));
    std::vector<int32_t> base((size_t)std::max(n, 1), 0);
    for (int32_t r = 0; r < n; ++r) {
        const int32_t p0 = A->h_rp[r], p1 = A->h_rp[r + 1];
        if (p0 == p1) continue;
        int32_t lo = A->h_ci[p0];
        for (int32_t p = p0; p < p1; ++p) lo = std::min(lo, A->h_ci[p]);
        base[r] = lo;
        for (int32_t p = p0; p < p1; ++p) {
            const int32_t delta = A->h_ci[p] - lo;
            if (delta > 65535) return false;
            d[p] = (uint16_t)delta;
        }
    }
    if (hipMalloc((void**)&A->ci16, d.size() * sizeof(uint16_t)) != hipSuccess) return false;
    if (hipMalloc((void**)&A->cbase, base.size() * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(A->ci16);
        A->ci16 = nullptr;
        return false;
    }
    if (hipMemcpy(A->ci16, d.data(), d.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(A->cbase, base.data(), base.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(A->ci16);
        (void)hipFree(A->cbase);
        A->ci16 = nullptr;
        A->cbase = nullptr;
        return false;  // state stays -1: the 32-bit kernel runs
    }
    A->ci16_state = 1;
    return true;
}

static int spmv_lanes_per_row(const lsa_mat* A, int variant) {
    int lpr = variant & 0xff;
    if (lpr == 0) {
        const double mean = A->n > 0 ? (double)A->nnz / (double)A->n : 0.0;
        lpr = mean <= 6.0 ? 4 : mean <= 12.0 ? 8 : mean <= 48.0 ? 16 : mean <= 96.0 ? 32 : 64;
    }
    return (lpr == 4 || lpr == 8 || lpr == 32 || lpr == 64) ? lpr : 16;
}
static bool spmv_wants_groups(const lsa_mat* A, int variant) {
    if (variant & 0x4000) return false;
    if (variant & (0x100 | 0x200 | 0x400)) return false;
    // one host pass over the pattern, worth it where the SpMV is a bandwidth question
    return (variant & 0x2000) || A->nnz >= (int64_t)4 << 20;
}
static bool spmv_wants_ci16(const lsa_mat* A, int variant) {
    const size_t msize = A->dtype == LSA_C128 ? 16 : 8;
    return ((variant & 0x800) || (variant == 0 && msize == 16 && A->nnz >= (int64_t)4 << 20)) && !(variant & 0x100) && !(variant & 0x200);
}

// variant word: bits 0-7 lanes per row (0 = from the mean row length), bit 8 non-temporal matrix loads,
// bit 9 XCD-contiguous row chunks, bit 10 two rows per sub-wave, bit 11 16-bit column indices, bit 12 plain 32-bit
// indices even for large matrices, bits 16-31 workgroups per CU (0 = 64)
template <typename MT, typename VT, int LPR, bool NT>
static void launch_spmv(lsa_ctx* ctx, const lsa_mat* A, const void* x, void* y, int variant) {
    const int threads = 256;
    int64_t want = ((int64_t)A->n * LPR + threads - 1) / threads;
    const int per_cu = (variant >> 16) > 0 ? (variant >> 16) : 64;
    int64_t cap = (int64_t)ctx->num_cu * per_cu;  // grid-stride beyond this
    int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    if ((variant & 0x200) && A->n >= 8 * (256 / LPR)) {
        const int rows_per_pass = 256 / LPR;
        int G = (int)(((blocks + 7) / 8) * 8);
        int64_t rows_per_wg = ((int64_t)A->n + G - 1) / G;
        rows_per_wg = ((rows_per_wg + rows_per_pass - 1) / rows_per_pass) * rows_per_pass;
        hipLaunchKernelGGL((spmv_xcd_kernel<MT, VT, LPR, NT>), dim3(G), dim3(threads), 0, ctx->stream, A->n, (int32_t)rows_per_wg, A->rp,
                           A->ci, (const MT*)A->val, (const VT*)x, (VT*)y);
        return;
    }
    // default (no variant word): compressed indices for matrices where the SpMV is a bandwidth question (>= 4 M entries;
    // building them is one host pass over the pattern, not worth it for the 0.9 M-entry matrices rebuilt per shift)
    // (complex matrices only: with 8-byte values the 2-byte index loads cost more than they save -- f64 SROOF 565 -> 592 us)
    if (!NT && spmv_wants_groups(A, variant) && ensure_groups(A)) {
        // with grouped rows the column indices are 1 / g of what they were: 2-byte offsets no longer pay (measured on SROOF:
        // 635 us with 32-bit indices against 663 us with 16-bit ones); they stay available through the variant word
        const bool c16 = (variant & 0x800) && ensure_ci16(A);
        int64_t gwant = ((int64_t)A->ngroups * LPR + threads - 1) / threads;
        const int64_t gcap = (variant >> 16) > 0 ? cap : (int64_t)ctx->num_cu * 256;  // more, shorter workgroups: 2 % faster than 64 per CU
        const int gblocks = (int)(gwant < 1 ? 1 : (gwant > gcap ? gcap : gwant));
        if ((variant & 0x8000) && !c16 && A->ngroups >= 8 * 64) {
            const int G = ((gblocks + 7) / 8) * 8;
            const int32_t chunk = (A->ngroups + G - 1) / G;
            hipLaunchKernelGGL((spmv_group_kernel<MT, VT, LPR, false, true>), dim3(G), dim3(threads), 0, ctx->stream, A->ngroups, chunk, (const int4*)A->grp_start, A->rp,
                               A->ci, (const uint16_t*)nullptr, (const int32_t*)nullptr, (const MT*)A->val, (const VT*)x, (VT*)y);
            return;
        }
        if (c16)
            hipLaunchKernelGGL((spmv_group_kernel<MT, VT, LPR, true, false>), dim3(gblocks), dim3(threads), 0, ctx->stream, A->ngroups, 0, (const int4*)A->grp_start, A->rp,
                               A->ci, (const uint16_t*)A->ci16, (const int32_t*)A->cbase, (const MT*)A->val, (const VT*)x, (VT*)y);
        else
            hipLaunchKernelGGL((spmv_group_kernel<MT, VT, LPR, false, false>), dim3(gblocks), dim3(threads), 0, ctx->stream, A->ngroups, 0, (const int4*)A->grp_start, A->rp,
                               A->ci, (const uint16_t*)nullptr, (const int32_t*)nullptr, (const MT*)A->val, (const VT*)x, (VT*)y);
        return;
    }
    if (spmv_wants_ci16(A, variant) && !NT && ensure_ci16(A)) {
        hipLaunchKernelGGL((spmv_subwave16_kernel<MT, VT, LPR>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, A->rp, (const uint16_t*)A->ci16,
                           (const int32_t*)A->cbase, (const MT*)A->val, (const VT*)x, (VT*)y);
        return;
    }
    if (variant & 0x400) {
        blocks = (blocks + 1) / 2;
        hipLaunchKernelGGL((spmv_subwave2_kernel<MT, VT, LPR, NT>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, A->rp, A->ci,
                           (const MT*)A->val, (const VT*)x, (VT*)y);
        return;
    }
    hipLaunchKernelGGL((spmv_subwave_kernel<MT, VT, LPR, NT>), dim3(blocks), dim3(threads), 0, ctx->stream, A->n, A->rp, A->ci,
                       (const MT*)A->val, (const VT*)x, (VT*)y);
}

template <typename MT, typename VT, bool NT>
static void dispatch_lpr(lsa_ctx* ctx, const lsa_mat* A, const void* x, void* y, int variant) {
    switch (spmv_lanes_per_row(A, variant)) {
        case 4: launch_spmv<MT, VT, 4, NT>(ctx, A, x, y, variant); break;
        case 8: launch_spmv<MT, VT, 8, NT>(ctx, A, x, y, variant); break;
        case 32: launch_spmv<MT, VT, 32, NT>(ctx, A, x, y, variant); break;
        case 64: launch_spmv<MT, VT, 64, NT>(ctx, A, x, y, variant); break;
        default: launch_spmv<MT, VT, 16, NT>(ctx, A, x, y, variant); break;
    }
}

static int spmv_variant() {
    // development knob (A/B runs of tools/spmv_only.py); unset = the tuned default
    const char* e = getenv("LSA_SPMV_VARIANT");
    return e ? (int)strtol(e, nullptr, 0) : 0;
}

// LPR when k_spmv runs the plain row-per-sub-wave kernel on A with complex vectors (no variant word, no row groups, no compressed
// indices), 0 otherwise: the fused tail of an Arnoldi step (blas.hip::cgs_tail_kernel) walks the rows the same way, so that
// its two products are bit for bit those of k_spmv
int k_spmv_plain_subwave_lanes(const lsa_mat* A) {
    const int variant = spmv_variant();
    if (variant != 0 || spmv_wants_groups(A, variant) || spmv_wants_ci16(A, variant)) return 0;
    return spmv_lanes_per_row(A, variant);
}

template <typename MT, typename VT>
static void dispatch_spmv(lsa_ctx* ctx, const lsa_mat* A, const void* x, void* y) {
    const int variant = spmv_variant();
    if (variant & 0x100) dispatch_lpr<MT, VT, true>(ctx, A, x, y, variant);
    else dispatch_lpr<MT, VT, false>(ctx, A, x, y, variant);
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

// ---- y = A^T x / A^H x (adjoint eigenproblem; not on the inner-loop path) --------------------------------------------------------
// Pull form over a transposed index built once per pattern on the host (counting sort of the column indices): output c walks
// its entries q with a sub-wave of 16 lanes, value val[src[q]], vector entry x[ci[q]] -- the additions in a fixed order, so the
// product is reproducible bit for bit.  (Rounds 2-3 scattered with floating-point atomics: the only kernel of the path whose
// rounding depended on the order in which additions arrived -- 11 of 12 adjoint solves in processes sharing a GPU differed in
// the last bits, tools/micro/repro_under_sharing.py, and the ranks of a sharded adjoint solve need replicated vectors alike.)
template <typename MT, typename VT>
__global__ __launch_bounds__(256) void spmv_transpose_pull_kernel(int32_t ncols, int conj, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                                  const int32_t* __restrict__ src, const MT* __restrict__ val,
                                                                  const VT* __restrict__ x, VT* __restrict__ y) {
    constexpr int LPR = 16;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) / LPR;
    for (int64_t c = gid / LPR; c < ncols; c += stride) {
        VT acc = scalar_traits<VT>::zero();
        for (int32_t q = rp[c] + lane; q < rp[c + 1]; q += LPR) {
            MT a = val[src[q]];
            if (conj) a = s_conj(a);
            fma_acc(acc, a, x[ci[q]]);
        }
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) acc = s_add(acc, shfl_xor_t<VT>(acc, m));
        if (lane == 0) y[c] = acc;
    }
}

static int ensure_transpose(lsa_ctx* ctx, const lsa_mat* A) {
    if (!A->h_shared) A->h_shared = std::make_shared<PatternShared>();
    if (A->h_shared->tr && A->h_shared->tr->nnz == A->nnz && A->h_shared->tr->ncols == A->ncols) return LSA_OK;
    const int32_t n = A->n, nc = A->ncols;
    if ((int64_t)A->h_rp.size() != (int64_t)n + 1 || (int64_t)A->h_ci.size() != A->nnz)
        return lsa_set_error(ctx, LSA_ERR_ARG, "spmv_transpose: the matrix has no host copy of its pattern");
    if (A->nnz > 0x7FFFFFFF) return lsa_set_error(ctx, LSA_ERR_ARG, "spmv_transpose: more than 2^31 entries");
    std::vector<int32_t> rp((size_t)nc + 1, 0), ci((size_t)std::max<int64_t>(A->nnz, 1)), src((size_t)std::max<int64_t>(A->nnz, 1));
    for (int64_t p = 0; p < A->nnz; ++p) ++rp[(size_t)A->h_ci[(size_t)p] + 1];
    for (int32_t c = 0; c < nc; ++c) rp[(size_t)c + 1] += rp[(size_t)c];
    std::vector<int32_t> next(rp.begin(), rp.end() - 1);
    for (int32_t r = 0; r < n; ++r)  // rows ascending: every output's entries end up ordered by row
        for (int32_t p = A->h_rp[(size_t)r]; p < A->h_rp[(size_t)r + 1]; ++p) {
            const int32_t q = next[(size_t)A->h_ci[(size_t)p]]++;
            ci[(size_t)q] = r;
            src[(size_t)q] = p;
        }
    auto T = std::make_shared<TransposeIndex>();
    T->ncols = nc;
    T->nnz = A->nnz;
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&T->rp, rp.size() * sizeof(int32_t)));
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&T->ci, ci.size() * sizeof(int32_t)));
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&T->src, src.size() * sizeof(int32_t)));
    LSA_HIP_CHECK(ctx, hipMemcpy(T->rp, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    LSA_HIP_CHECK(ctx, hipMemcpy(T->ci, ci.data(), ci.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    LSA_HIP_CHECK(ctx, hipMemcpy(T->src, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    A->h_shared->tr = T;
    return LSA_OK;
}

int k_spmv_transpose(lsa_ctx* ctx, const lsa_mat* A, int conj, int xdtype, const void* x, void* y) {
    if (A->row0 != 0 || A->n != A->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "spmv_transpose: sharded matrices are not supported");
    LSA_CHECK(ensure_transpose(ctx, A));
    const TransposeIndex& T = *A->h_shared->tr;
    const int threads = 256;
    int64_t want = ((int64_t)A->ncols * 16 + threads - 1) / threads;
    int64_t cap = (int64_t)ctx->num_cu * 64;
    int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    if (A->dtype == LSA_F64 && xdtype == LSA_F64)
        hipLaunchKernelGGL((spmv_transpose_pull_kernel<double, double>), dim3(blocks), dim3(threads), 0, ctx->stream, A->ncols, conj, T.rp, T.ci, T.src,
                           (const double*)A->val, (const double*)x, (double*)y);
    else if (A->dtype == LSA_F64 && xdtype == LSA_C128)
        hipLaunchKernelGGL((spmv_transpose_pull_kernel<double, cplx>), dim3(blocks), dim3(threads), 0, ctx->stream, A->ncols, conj, T.rp, T.ci, T.src,
                           (const double*)A->val, (const cplx*)x, (cplx*)y);
    else if (A->dtype == LSA_C128 && xdtype == LSA_C128)
        hipLaunchKernelGGL((spmv_transpose_pull_kernel<cplx, cplx>), dim3(blocks), dim3(threads), 0, ctx->stream, A->ncols, conj, T.rp, T.ci, T.src,
                           (const cplx*)A->val, (const cplx*)x, (cplx*)y);
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

int lsa_spmv_info(lsa_ctx* ctx, const lsa_mat* A, int xdtype, char* kernel, int32_t kernel_len, int64_t* bytes_moved) {
    if (!ctx || !A) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_spmv_info: null argument");
    const int variant = spmv_variant();
    const int lpr = spmv_lanes_per_row(A, variant);
    const bool grouped = !(variant & 0x100) && spmv_wants_groups(A, variant) && ensure_groups(A);
    const bool c16 = grouped ? ((variant & 0x800) && ensure_ci16(A)) : (spmv_wants_ci16(A, variant) && ensure_ci16(A));
    const char* mt = A->dtype == LSA_C128 ? "cplx" : "double";
    const char* vt = xdtype == LSA_C128 ? "cplx" : "double";
    const char* base = grouped ? "spmv_group_kernel" : c16 ? "spmv_subwave16_kernel" : (variant & 0x200) ? "spmv_xcd_kernel" : (variant & 0x400) ? "spmv_subwave2_kernel" : "spmv_subwave_kernel";
    if (kernel && kernel_len > 0) {
        if (grouped) snprintf(kernel, (size_t)kernel_len, "%s<%s,%s,%d,%s,%s>", base, mt, vt, lpr, c16 ? "true" : "false", (variant & 0x8000) && !c16 ? "true" : "false");
        else snprintf(kernel, (size_t)kernel_len, "%s<%s,%s,%d>", base, mt, vt, lpr);
    }
    if (bytes_moved) {
        const int64_t ms = A->dtype == LSA_C128 ? 16 : 8, vs = xdtype == LSA_C128 ? 16 : 8, is = c16 ? 2 : 4;
        // per entry: value + column index (2 bytes when compressed); per row: row pointer (+ the row's first column when
        // compressed), one x entry read and one y entry written.  Grouped: the indices of the first row of a group only,
        // two row pointers, a group pointer (and the first column) per group.
        if (grouped) {
            const double gmean = (double)A->n / (double)A->ngroups;
            *bytes_moved = A->nnz * ms + (int64_t)((double)A->nnz * (double)is / gmean) + (int64_t)A->ngroups * (8 + 4 + (c16 ? 4 : 0)) + (int64_t)A->n * 2 * vs;
        } else
            *bytes_moved = A->nnz * (ms + is) + (int64_t)A->n * (4 + (c16 ? 4 : 0) + 2 * vs);
    }
    return LSA_OK;
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
