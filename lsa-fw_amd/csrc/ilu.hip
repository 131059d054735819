// ILU(k) preconditioner: host symbolic analysis, device numeric factorisation, and the two sparse triangular
// solves (PCSetUp / MatSolve of the ST's inner KSP in the reference path, Solver/utils.py:261-266).
//
// Data layout in HBM: one CSR holds both factors (L strictly below the diagonal with an implied unit diagonal,
// U on and above it); `diag[i]` is the position of u_ii, `dinv[i]` = 1/u_ii.  Rows are scheduled by dependency
// level: order_l / order_u list the rows sorted by level of the lower / upper triangular DAG.
//
// SpTRSV, sync-free form (default): ONE launch per triangular solve.  A wavefront owns a row; rows are dealt to
// wavefronts round-robin in level order, so the row a wavefront waits for always belongs to a wavefront that is
// already past all of its own earlier rows (no deadlock as long as every wavefront of the grid is resident: the
// grid is sized well below residency).  The solution vector itself carries the hand-off: it is pre-filled with
// an all-ones NaN bit pattern and a consumer polls the 8-byte granules of x[col] (agent-scope relaxed atomic
// loads, which bypass the per-CU L1) until they change; the producer publishes with agent-scope relaxed atomic
// 8-byte stores.  This is the "data is the flag" granule hand-off of the CDNA4 guide; no separate flag, no
// fences.  Every spin is bounded and watches a global abort word.
//
// SpTRSV, level form (algo 0): one launch per level; kept as the cross-check of the sync-free kernel.
#include <algorithm>
#include <chrono>

#include "lsa_internal.h"

namespace {

constexpr unsigned long long kSentinel = 0xFFFFFFFFFFFFFFFFull;  // a quiet-NaN pattern no finite result produces
constexpr int kSolveThreads = 1024;                              // 16 wavefronts per workgroup
constexpr long long kMaxSpins = 1ll << 22;

// ---------------------------------------------------------------------------------------------------------------
// host analysis
// ---------------------------------------------------------------------------------------------------------------

// level-of-fill symbolic factorisation; rows of the input must have sorted columns
int symbolic_iluk(lsa_ctx* ctx, int32_t n, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci, int levels,
                  std::vector<int32_t>& orp, std::vector<int32_t>& oci, std::vector<int32_t>& odiag) {
    orp.assign((size_t)n + 1, 0);
    odiag.assign((size_t)n, -1);
    oci.clear();
    if (levels <= 0) {
        orp = rp;
        oci = ci;
        for (int32_t i = 0; i < n; ++i) {
            auto b = oci.begin() + orp[i], e = oci.begin() + orp[i + 1];
            auto it = std::lower_bound(b, e, i);
            if (it == e || *it != i) return lsa_set_error(ctx, LSA_ERR_ZERO_PIVOT, "ILU: row %d has no diagonal entry", i);
            odiag[i] = (int32_t)(it - oci.begin());
        }
        return LSA_OK;
    }
    std::vector<int32_t> olev;
    oci.reserve(ci.size() * 2);
    olev.reserve(ci.size() * 2);
    std::vector<int32_t> next((size_t)n + 1), lev((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        int32_t head = n, prev = -1;
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
            const int32_t c = ci[p];
            lev[c] = 0;
            if (prev < 0) head = c;
            else next[prev] = c;
            prev = c;
        }
        if (prev >= 0) next[prev] = n;
        for (int32_t k = head; k < i; k = next[k]) {
            const int32_t lk = lev[k];
            int32_t ins = k;
            for (int64_t q = (int64_t)odiag[k] + 1; q < orp[k + 1]; ++q) {
                const int32_t l = lk + olev[q] + 1;
                if (l > levels) continue;
                const int32_t jcol = oci[q];
                while (next[ins] < jcol) ins = next[ins];
                if (next[ins] == jcol) {
                    if (lev[jcol] > l) lev[jcol] = l;
                } else {
                    next[jcol] = next[ins];
                    next[ins] = jcol;
                    lev[jcol] = l;
                }
                ins = jcol;
            }
        }
        for (int32_t c = head; c < n; c = next[c]) {
            if (c == i) odiag[i] = (int32_t)oci.size();
            oci.push_back(c);
            olev.push_back(lev[c]);
        }
        if (odiag[i] < 0) return lsa_set_error(ctx, LSA_ERR_ZERO_PIVOT, "ILU: row %d has no diagonal entry", i);
        if (oci.size() >= (size_t)INT32_MAX) return lsa_set_error(ctx, LSA_ERR_ARG, "ILU(k): factor pattern exceeds 2^31 entries");
        orp[i + 1] = (int32_t)oci.size();
    }
    return LSA_OK;
}

// dependency levels of the lower (rows depend on smaller columns) and upper DAGs -> row lists sorted by level
void level_schedule(int32_t n, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci,
                    const std::vector<int32_t>& diag, bool lower, std::vector<int32_t>& order,
                    std::vector<int32_t>& lvl_ptr) {
    std::vector<int32_t> lev((size_t)n, 0);
    int32_t maxlev = 0;
    if (lower) {
        for (int32_t i = 0; i < n; ++i) {
            int32_t l = 0;
            for (int32_t p = rp[i]; p < diag[i]; ++p) l = std::max(l, lev[ci[p]] + 1);
            lev[i] = l;
            maxlev = std::max(maxlev, l);
        }
    } else {
        for (int32_t i = n - 1; i >= 0; --i) {
            int32_t l = 0;
            for (int32_t p = diag[i] + 1; p < rp[i + 1]; ++p) l = std::max(l, lev[ci[p]] + 1);
            lev[i] = l;
            maxlev = std::max(maxlev, l);
        }
    }
    lvl_ptr.assign((size_t)maxlev + 2, 0);
    for (int32_t i = 0; i < n; ++i) ++lvl_ptr[(size_t)lev[i] + 1];
    for (size_t l = 0; l + 1 < lvl_ptr.size(); ++l) lvl_ptr[l + 1] += lvl_ptr[l];
    order.assign((size_t)n, 0);
    std::vector<int32_t> cur(lvl_ptr.begin(), lvl_ptr.end() - 1);
    if (lower) {
        for (int32_t i = 0; i < n; ++i) order[(size_t)cur[lev[i]]++] = i;
    } else {
        for (int32_t i = n - 1; i >= 0; --i) order[(size_t)cur[lev[i]]++] = i;
    }
    if (n == 0) lvl_ptr.assign(1, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// device: scatter the matrix values into the (possibly larger) factor pattern
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void scatter_values_kernel(int32_t n, const int32_t* __restrict__ arp, const int32_t* __restrict__ aci,
                                      const T* __restrict__ aval, const int32_t* __restrict__ frp,
                                      const int32_t* __restrict__ fci, T* __restrict__ fval) {
    // one 16-lane group per row; the factor row is a superset of the matrix row, both sorted
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 15);
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t i = gid >> 4; i < n; i += stride) {
        const int32_t f0 = frp[i], f1 = frp[i + 1];
        for (int32_t p = arp[i] + lane; p < arp[i + 1]; p += 16) {
            const int32_t c = aci[p];
            int32_t lo = f0, hi = f1 - 1;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (fci[mid] < c) lo = mid + 1;
                else hi = mid;
            }
            fval[lo] = aval[p];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// device: numeric ILU, one launch per dependency level of the lower DAG, one wavefront per row.
// The row being eliminated lives in LDS; rows of earlier levels are final and read from global memory.
// flag[1] = error (1 + row of the first zero pivot), flag[2] = number of shifted pivots.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void ilu_level_kernel(int32_t first, int32_t last, const int32_t* __restrict__ order,
                                 const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                 const int32_t* __restrict__ diag, T* __restrict__ val, T* __restrict__ dinv,
                                 double shift_tol, int32_t max_row, int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int waves_per_block = blockDim.x >> 6;
    T* row = (T*)dyn + (size_t)wave_in_block * max_row;
    const int32_t r = first + blockIdx.x * waves_per_block + wave_in_block;
    if (r >= last) return;
    const int32_t i = order[r];
    const int32_t p0 = rp[i], p1 = rp[i + 1], pd = diag[i];
    const int32_t len = p1 - p0;
    for (int32_t t = lane; t < len; t += 64) row[t] = val[p0 + t];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int32_t p = p0; p < pd; ++p) {
        const int32_t k = ci[p];
        T lik = s_mul(dinv[k], row[p - p0]);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) row[p - p0] = lik;
        // row_i[j] -= l_ik * u_kj for the entries of U's row k that exist in row i
        for (int32_t q = diag[k] + 1 + lane; q < rp[k + 1]; q += 64) {
            const int32_t c = ci[q];
            // binary search c among ci[p+1 .. p1)
            int32_t lo = p + 1, hi = p1;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (ci[mid] < c) lo = mid + 1;
                else hi = mid;
            }
            if (lo < p1 && ci[lo] == c) {
                T acc = row[lo - p0];
                T prod = s_mul(lik, val[q]);
                row[lo - p0] = s_sub(acc, prod);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        T piv = row[pd - p0];
        const double mag = sqrt(s_abs2(piv));
        if (!(mag >= shift_tol) || mag == 0.0) {
            if (shift_tol > 0.0) {
                if (mag > 0.0) piv = s_mul(shift_tol / mag, piv);
                else s_from(piv, shift_tol, 0.0);
                row[pd - p0] = piv;
                atomicAdd(&flag[2], 1);
            } else {
                atomicCAS(&flag[1], 0, i + 1);
                s_from(piv, 1.0, 0.0);  // keep going with a finite value; the host reports the error
                row[pd - p0] = piv;
            }
        }
        dinv[i] = s_inv(piv);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int32_t t = lane; t < len; t += 64) val[p0 + t] = row[t];
}

// ---------------------------------------------------------------------------------------------------------------
// device: triangular solves
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum_t(T v);
template <>
__device__ __forceinline__ double wave_sum_t<double>(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <>
__device__ __forceinline__ cplx wave_sum_t<cplx>(cplx v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        v.re += __shfl_xor(v.re, m, 64);
        v.im += __shfl_xor(v.im, m, 64);
    }
    return v;
}

// level form: rows [first, last) of `order` are independent; one 16-lane group per row
template <typename MT, typename VT, bool LOWER>
__global__ __launch_bounds__(256) void sptrsv_level_kernel(int32_t first, int32_t last, const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                           const int32_t* __restrict__ diag, const MT* __restrict__ val,
                                                           const MT* __restrict__ dinv, const VT* __restrict__ b,
                                                           VT* __restrict__ x) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 15);
    const int64_t r = first + (gid >> 4);
    if (r >= last) return;
    const int32_t i = order[r];
    const int32_t q0 = LOWER ? rp[i] : diag[i] + 1;
    const int32_t q1 = LOWER ? diag[i] : rp[i + 1];
    VT acc = scalar_traits<VT>::zero();
    for (int32_t p = q0 + lane; p < q1; p += 16) fma_acc(acc, val[p], x[ci[p]]);
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) {
        if constexpr (sizeof(VT) == 16) {
            acc.re += __shfl_xor(acc.re, m, 64);
            acc.im += __shfl_xor(acc.im, m, 64);
        } else {
            acc += __shfl_xor(acc, m, 64);
        }
    }
    if (lane == 0) {
        VT s = s_sub(b[i], acc);
        if (!LOWER) s = s_mul(dinv[i], s);
        x[i] = s;
    }
}

__device__ __forceinline__ unsigned long long load_granule(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_granule(unsigned long long* p, unsigned long long v) {
    // a NaN that happens to carry the sentinel bits would look "not yet written": publish another NaN instead
    if (v == kSentinel) v = 0x7FF8000000000000ull;
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// poll x[c] until every granule differs from the sentinel; returns false on abort / timeout (wave-uniform)
template <typename VT>
__device__ __forceinline__ bool wait_value(const VT* x, int32_t c, bool active, VT& out, int32_t* flag) {
    constexpr int G = sizeof(VT) / 8;
    const unsigned long long* g = (const unsigned long long*)(x + (active ? c : 0));
    unsigned long long bits[G];
    bool ok = !active;
    long long spins = 0;
    for (;;) {
        if (!ok) {
            bool all = true;
#pragma unroll
            for (int t = 0; t < G; ++t) {
                bits[t] = load_granule(g + t);
                all = all && (bits[t] != kSentinel);
            }
            ok = all;
        }
        if (__all(ok)) break;
        ++spins;
        if ((spins & 1023) == 0) {
            if (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > kMaxSpins) {
                __hip_atomic_store(&flag[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (active) {
        if constexpr (G == 2) out = cplx{__longlong_as_double((long long)bits[0]), __longlong_as_double((long long)bits[1])};
        else out = __longlong_as_double((long long)bits[0]);
    }
    return true;
}

template <typename VT>
__device__ __forceinline__ void publish_value(VT* x, int32_t i, VT v) {
    unsigned long long* g = (unsigned long long*)(x + i);
    if constexpr (sizeof(VT) == 16) {
        store_granule(g, (unsigned long long)__double_as_longlong(v.re));
        store_granule(g + 1, (unsigned long long)__double_as_longlong(v.im));
    } else {
        store_granule(g, (unsigned long long)__double_as_longlong(v));
    }
}

// sync-free form: x must be pre-filled with the sentinel pattern
template <typename MT, typename VT, bool LOWER>
__global__ __launch_bounds__(kSolveThreads) void sptrsv_syncfree_kernel(int32_t n, const int32_t* __restrict__ order,
                                                                        const int32_t* __restrict__ rp,
                                                                        const int32_t* __restrict__ ci,
                                                                        const int32_t* __restrict__ diag,
                                                                        const MT* __restrict__ val,
                                                                        const MT* __restrict__ dinv,
                                                                        const VT* __restrict__ b, VT* x, int32_t* flag) {
    const int lane = threadIdx.x & 63;
    const int waves_per_block = blockDim.x >> 6;
    const int wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * waves_per_block;
    for (int32_t r = wave; r < n; r += nwaves) {
        const int32_t i = order[r];
        const int32_t q0 = LOWER ? rp[i] : diag[i] + 1;
        const int32_t q1 = LOWER ? diag[i] : rp[i + 1];
        VT acc = scalar_traits<VT>::zero();
        for (int32_t base = q0; base < q1; base += 64) {
            const int32_t p = base + lane;
            const bool active = p < q1;
            int32_t c = 0;
            MT a = scalar_traits<MT>::zero();
            if (active) {
                c = ci[p];
                a = val[p];
            }
            VT xv = scalar_traits<VT>::zero();
            if (!wait_value<VT>(x, c, active, xv, flag)) return;
            if (active) fma_acc(acc, a, xv);
        }
        acc = wave_sum_t<VT>(acc);
        if (lane == 0) {
            VT s = s_sub(b[i], acc);
            if (!LOWER) s = s_mul(dinv[i], s);
            publish_value<VT>(x, i, s);
        }
    }
}

template <typename T>
int numeric_factor(lsa_ctx* ctx, lsa_ilu* pc, const lsa_mat* C, double shift_tol) {
    // 1. values into the factor pattern
    LSA_HIP_CHECK(ctx, hipMemsetAsync(pc->val, 0, (size_t)pc->nnz * sizeof(T), ctx->stream));
    {
        const int threads = 256;
        int64_t want = ((int64_t)pc->n * 16 + threads - 1) / threads;
        int blocks = (int)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)ctx->num_cu * 32);
        hipLaunchKernelGGL((scatter_values_kernel<T>), dim3(blocks), dim3(threads), 0, ctx->stream, pc->n, C->rp, C->ci,
                           (const T*)C->val, pc->rp, pc->ci, (T*)pc->val);
    }
    // 2. level-by-level elimination
    int32_t max_row = 1;
    for (int32_t i = 0; i < pc->n; ++i) max_row = std::max(max_row, pc->h_rp[i + 1] - pc->h_rp[i]);
    int waves_per_block = 4;
    while (waves_per_block > 1 && (size_t)waves_per_block * max_row * sizeof(T) > 64 * 1024) waves_per_block >>= 1;
    const size_t lds = (size_t)waves_per_block * max_row * sizeof(T);
    if (lds > 160 * 1024) return lsa_set_error(ctx, LSA_ERR_ARG, "ILU: a factor row with %d entries does not fit LDS", max_row);
    if (lds > 64 * 1024) {
        LSA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ilu_level_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    LSA_HIP_CHECK(ctx, hipMemsetAsync(pc->flag, 0, 4 * sizeof(int32_t), ctx->stream));
    const size_t nlev = pc->lvl_ptr_l.size() - 1;
    for (size_t l = 0; l < nlev; ++l) {
        const int32_t first = pc->lvl_ptr_l[l], last = pc->lvl_ptr_l[l + 1];
        const int rows = last - first;
        if (rows <= 0) continue;
        const int blocks = (rows + waves_per_block - 1) / waves_per_block;
        hipLaunchKernelGGL((ilu_level_kernel<T>), dim3(blocks), dim3(64 * waves_per_block), lds, ctx->stream, first, last,
                           pc->order_l, pc->rp, pc->ci, pc->diag, (T*)pc->val, (T*)pc->dinv, shift_tol, max_row, pc->flag);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "ILU numeric launch failed: %s", hipGetErrorString(e));
    int32_t hflag[4];
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(hflag, pc->flag, sizeof hflag, hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    pc->nshift = hflag[2];
    if (hflag[1] != 0) return lsa_set_error(ctx, LSA_ERR_ZERO_PIVOT, "ILU(k): zero pivot in row %d", hflag[1] - 1);
    return LSA_OK;
}

template <typename MT, typename VT, bool LOWER>
int launch_solve(lsa_ctx* ctx, lsa_ilu* pc, const void* b, void* x) {
    const int32_t* order = LOWER ? pc->order_l : pc->order_u;
    if (pc->algo == 0) {
        const std::vector<int32_t>& lp = LOWER ? pc->lvl_ptr_l : pc->lvl_ptr_u;
        for (size_t l = 0; l + 1 < lp.size(); ++l) {
            const int rows = lp[l + 1] - lp[l];
            if (rows <= 0) continue;
            const int blocks = (rows * 16 + 255) / 256;
            hipLaunchKernelGGL((sptrsv_level_kernel<MT, VT, LOWER>), dim3(blocks), dim3(256), 0, ctx->stream, lp[l], lp[l + 1], order,
                               pc->rp, pc->ci, pc->diag, (const MT*)pc->val, (const MT*)pc->dinv, (const VT*)b, (VT*)x);
        }
    } else {
        LSA_HIP_CHECK(ctx, hipMemsetAsync(x, 0xFF, (size_t)pc->n * sizeof(VT), ctx->stream));
        const int blocks = LOWER ? pc->sptrsv_blocks_l : pc->sptrsv_blocks_u;
        hipLaunchKernelGGL((sptrsv_syncfree_kernel<MT, VT, LOWER>), dim3(blocks), dim3(kSolveThreads), 0, ctx->stream, pc->n, order,
                           pc->rp, pc->ci, pc->diag, (const MT*)pc->val, (const MT*)pc->dinv, (const VT*)b, (VT*)x, pc->flag);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "SpTRSV launch failed: %s", hipGetErrorString(e));
    return LSA_OK;
}

template <typename MT, typename VT>
int solve_typed(lsa_ctx* ctx, lsa_ilu* pc, int which, const void* b, void* x) {
    if (which == 0) return launch_solve<MT, VT, true>(ctx, pc, b, x);
    if (which == 1) return launch_solve<MT, VT, false>(ctx, pc, b, x);
    const int vdtype = scalar_traits<VT>::dtype;
    if (!pc->tmp || pc->tmp_dtype != vdtype) {
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (pc->tmp) (void)hipFree(pc->tmp);
        pc->tmp = nullptr;
        LSA_HIP_CHECK(ctx, hipMalloc(&pc->tmp, (size_t)std::max<int32_t>(pc->n, 1) * sizeof(VT)));
        pc->tmp_dtype = vdtype;
    }
    LSA_CHECK((launch_solve<MT, VT, true>(ctx, pc, b, pc->tmp)));
    return launch_solve<MT, VT, false>(ctx, pc, pc->tmp, x);
}

int pick_blocks(lsa_ctx* ctx, int32_t n, size_t nlevels) {
    // mean rows per level decides how many workgroups (16 wavefronts each) can be kept busy; small FEM factors
    // have < 16 independent rows per level and run fastest on ONE workgroup (hand-offs stay inside a CU's reach)
    const double rows_per_level = nlevels > 0 ? (double)n / (double)nlevels : (double)n;
    int blocks = (int)(rows_per_level / 16.0 + 0.999);
    blocks = std::max(1, std::min(blocks, ctx->num_cu));  // one 1024-thread workgroup per CU is always resident
    return blocks;
}

}  // namespace

int ilu_solve_dev(lsa_ctx* ctx, lsa_ilu* pc, int which, int vdtype, const void* b, void* x) {
    if (which < 0 || which > 2) return lsa_set_error(ctx, LSA_ERR_ARG, "ilu_solve: which must be 0, 1 or 2");
    if (b == x) return lsa_set_error(ctx, LSA_ERR_ARG, "ilu_solve: b and x must not alias");
    if (pc->algo == 2) return blk_solve(ctx, pc, which, vdtype, b, x);
    if (pc->dtype == LSA_F64 && vdtype == LSA_F64) return solve_typed<double, double>(ctx, pc, which, b, x);
    if (pc->dtype == LSA_F64 && vdtype == LSA_C128) return solve_typed<double, cplx>(ctx, pc, which, b, x);
    if (pc->dtype == LSA_C128 && vdtype == LSA_C128) return solve_typed<cplx, cplx>(ctx, pc, which, b, x);
    return lsa_set_error(ctx, LSA_ERR_ARG, "ilu_solve: complex factors need complex vectors");
}

// check the abort word of the sync-free kernels (call after a synchronisation point)
int ilu_check_abort(lsa_ctx* ctx, lsa_ilu* pc) {
    int32_t h = 0;
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(&h, pc->flag, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h != 0) {
        LSA_HIP_CHECK(ctx, hipMemsetAsync(pc->flag, 0, sizeof(int32_t), ctx->stream));
        return lsa_set_error(ctx, LSA_ERR_TIMEOUT, "SpTRSV: a dependency wait expired (non-finite values in the factors or the right-hand side?)");
    }
    return LSA_OK;
}

extern "C" {

int lsa_ilu_create(lsa_ctx* ctx, const lsa_mat* C, int levels, double shift_tol, lsa_ilu** out) {
    if (!ctx || !C || !out || levels < 0 || shift_tol < 0.0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_create: bad argument");
    if (C->n != C->ncols) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_create: the matrix must be square (use the diagonal block of a shard)");
    lsa_ilu* pc = new lsa_ilu();
    pc->ctx = ctx;
    pc->n = C->n;
    pc->dtype = C->dtype;
    pc->rp = pc->ci = pc->diag = pc->order_l = pc->order_u = pc->flag = nullptr;
    pc->val = pc->dinv = pc->tmp = nullptr;
    pc->tmp_dtype = -1;
    pc->nshift = 0;
    pc->algo = 1;
    int rc = symbolic_iluk(ctx, C->n, C->h_rp.vec(), C->h_ci.vec(), levels, pc->h_rp, pc->h_ci, pc->h_diag);
    if (rc != LSA_OK) {
        delete pc;
        return rc;
    }
    pc->nnz = (int64_t)pc->h_ci.size();
    std::vector<int32_t> order_l, order_u;
    level_schedule(pc->n, pc->h_rp, pc->h_ci, pc->h_diag, true, order_l, pc->lvl_ptr_l);
    level_schedule(pc->n, pc->h_rp, pc->h_ci, pc->h_diag, false, order_u, pc->lvl_ptr_u);
    pc->sptrsv_blocks_l = pick_blocks(ctx, pc->n, pc->lvl_ptr_l.size() - 1);
    pc->sptrsv_blocks_u = pick_blocks(ctx, pc->n, pc->lvl_ptr_u.size() - 1);
    const size_t esz = pc->dtype == LSA_C128 ? 16 : 8;
    const size_t n1 = (size_t)std::max<int32_t>(pc->n, 1), z1 = (size_t)std::max<int64_t>(pc->nnz, 1);
    bool ok = hipMalloc(&pc->rp, sizeof(int32_t) * (n1 + 1)) == hipSuccess && hipMalloc(&pc->ci, sizeof(int32_t) * z1) == hipSuccess &&
              hipMalloc(&pc->diag, sizeof(int32_t) * n1) == hipSuccess && hipMalloc(&pc->order_l, sizeof(int32_t) * n1) == hipSuccess &&
              hipMalloc(&pc->order_u, sizeof(int32_t) * n1) == hipSuccess && hipMalloc(&pc->val, esz * z1) == hipSuccess &&
              hipMalloc(&pc->dinv, esz * n1) == hipSuccess && hipMalloc(&pc->flag, 4 * sizeof(int32_t)) == hipSuccess;
    if (!ok) {
        lsa_ilu_destroy(pc);
        return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_ilu_create: out of device memory (factor nnz %lld)", (long long)pc->nnz);
    }
    hipStream_t s = ctx->stream;
    ok = hipMemcpyAsync(pc->rp, pc->h_rp.data(), sizeof(int32_t) * ((size_t)pc->n + 1), hipMemcpyHostToDevice, s) == hipSuccess &&
         (pc->nnz == 0 || hipMemcpyAsync(pc->ci, pc->h_ci.data(), sizeof(int32_t) * (size_t)pc->nnz, hipMemcpyHostToDevice, s) == hipSuccess) &&
         (pc->n == 0 || (hipMemcpyAsync(pc->diag, pc->h_diag.data(), sizeof(int32_t) * (size_t)pc->n, hipMemcpyHostToDevice, s) == hipSuccess &&
                         hipMemcpyAsync(pc->order_l, order_l.data(), sizeof(int32_t) * (size_t)pc->n, hipMemcpyHostToDevice, s) == hipSuccess &&
                         hipMemcpyAsync(pc->order_u, order_u.data(), sizeof(int32_t) * (size_t)pc->n, hipMemcpyHostToDevice, s) == hipSuccess)) &&
         hipMemsetAsync(pc->flag, 0, 4 * sizeof(int32_t), s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) {
        lsa_ilu_destroy(pc);
        return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_ilu_create: upload of the factor pattern failed");
    }
    rc = pc->dtype == LSA_C128 ? numeric_factor<cplx>(ctx, pc, C, shift_tol) : numeric_factor<double>(ctx, pc, C, shift_tol);
    if (rc != LSA_OK) {
        lsa_ilu_destroy(pc);
        return rc;
    }
    // Default triangular-solve algorithm: when the dependency DAG is narrow (few rows per level, the 2D FEM case)
    // the blocked form wins by more than an order of magnitude; wide DAGs (3D, > 64 rows per level) keep the
    // sync-free row-parallel kernel, which needs no extra memory.
    const double rows_per_level = (double)pc->n / (double)std::max<size_t>(1, pc->lvl_ptr_l.size() - 1);
    if (pc->n >= 512 && rows_per_level < 64.0) {
        const char* env = getenv("LSA_SPTRSV_BLOCK");
        int32_t B = env ? atoi(env) : 1024;
        if (B > 0 && blk_setup(ctx, pc, B) == LSA_OK) pc->algo = 2;  // on failure (memory) stay with sync-free
    }
    *out = pc;
    return LSA_OK;
}

int lsa_ilu_set_algorithm(lsa_ctx* ctx, lsa_ilu* pc, int algo, int32_t block_size) {
    if (!ctx || !pc || algo < 0 || algo > 2) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_set_algorithm: algo must be 0, 1 or 2");
    if (algo == 2) {
        if (block_size <= 0) block_size = 1024;
        if (pc->blk_B != std::max(256, ((block_size + 255) / 256) * 256)) LSA_CHECK(blk_setup(ctx, pc, block_size));
    }
    pc->algo = algo;
    return LSA_OK;
}

void lsa_ilu_destroy(lsa_ilu* pc) {
    if (!pc) return;
    if (pc->ctx && pc->ctx->stream) (void)hipStreamSynchronize(pc->ctx->stream);
    blk_release(pc);
    void* ptrs[] = {pc->rp, pc->ci, pc->diag, pc->order_l, pc->order_u, pc->val, pc->dinv, pc->flag, pc->tmp};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete pc;
}

int lsa_ilu_solve(lsa_ctx* ctx, lsa_ilu* pc, int which, const lsa_vec* b, lsa_vec* x) {
    if (!ctx || !pc || !b || !x) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_solve: null argument");
    if (b->n != pc->n || x->n != pc->n || b->dtype != x->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_solve: vector shape/dtype mismatch");
    LSA_CHECK(ilu_solve_dev(ctx, pc, which, b->dtype, b->d, x->d));
    return ilu_check_abort(ctx, pc);
}

int lsa_ilu_solve_time(lsa_ctx* ctx, lsa_ilu* pc, int which, const lsa_vec* b, lsa_vec* x, int iters, double* avg_ms) {
    if (!ctx || !pc || !b || !x || !avg_ms || iters <= 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_solve_time: bad argument");
    if (b->n != pc->n || x->n != pc->n || b->dtype != x->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_solve_time: vector shape/dtype mismatch");
    LSA_CHECK(ilu_solve_dev(ctx, pc, which, b->dtype, b->d, x->d));  // warm-up (captures the graph of the blocked form)
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; ++i) LSA_CHECK(ilu_solve_dev(ctx, pc, which, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    LSA_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    LSA_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / iters;
    return ilu_check_abort(ctx, pc);
}

int lsa_ilu_info(const lsa_ilu* pc, int64_t* nnz, int32_t* levels_lower, int32_t* levels_upper, int32_t* nshift) {
    if (!pc) return LSA_ERR_ARG;
    if (nnz) *nnz = pc->nnz;
    if (levels_lower) *levels_lower = (int32_t)pc->lvl_ptr_l.size() - 1;
    if (levels_upper) *levels_upper = (int32_t)pc->lvl_ptr_u.size() - 1;
    if (nshift) *nshift = pc->nshift;
    return LSA_OK;
}

int lsa_ilu_download(lsa_ctx* ctx, const lsa_ilu* pc, int32_t* rowptr, int32_t* col, void* val) {
    if (!ctx || !pc) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ilu_download: null argument");
    if (rowptr) memcpy(rowptr, pc->h_rp.data(), sizeof(int32_t) * ((size_t)pc->n + 1));
    if (col) memcpy(col, pc->h_ci.data(), sizeof(int32_t) * (size_t)pc->nnz);
    if (val) {
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(val, pc->val, (size_t)pc->nnz * (pc->dtype == LSA_C128 ? 16 : 8), hipMemcpyDeviceToHost, ctx->stream));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return LSA_OK;
}

}  // extern "C"
