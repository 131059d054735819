// Vector and tall-skinny kernels of the Krylov loops (BVOrthogonalize / VecNorm / VecAXPY of the reference's
// SLEPc path).  All are HBM-bound streams over n-long columns: reductions are two-stage (per-block partials in
// a fixed order, then one small finishing kernel) so results are bitwise reproducible run to run and identical
// on every GPU that holds a replica of the basis.
#include "lsa_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kColTile = 8;  // basis columns handled per block in the multi-dot

template <typename T>
__device__ __forceinline__ T wave_sum(T v);
template <>
__device__ __forceinline__ double wave_sum<double>(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <>
__device__ __forceinline__ cplx wave_sum<cplx>(cplx v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        v.re += __shfl_xor(v.re, m, 64);
        v.im += __shfl_xor(v.im, m, 64);
    }
    return v;
}

// sum over the 256 threads of a block; result valid in thread 0
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 4 */) {
    v = wave_sum<T>(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    T r = smem[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = s_add(r, smem[w]);
    return r;
}

template <typename T>
__global__ void copy_kernel(int64_t n, const T* __restrict__ x, T* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = x[i];
}

template <typename T>
__global__ void axpy_kernel(int64_t n, T alpha, const T* __restrict__ x, T* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T acc = y[i];
        fma_acc(acc, alpha, x[i]);
        y[i] = acc;
    }
}

// partial[(c) * nchunks + chunk] = sum over the chunk's rows of conj(V[i,c]) w[i]
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_dot_partial_kernel(int64_t n, int j, int64_t rows_per_block,
                                                                     const T* __restrict__ V, int64_t ldv,
                                                                     const T* __restrict__ w, T* __restrict__ partial,
                                                                     int nchunks) {
    __shared__ T smem[4];
    const int chunk = blockIdx.x;
    const int c0 = blockIdx.y * kColTile;
    const int64_t r0 = (int64_t)chunk * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    T acc[kColTile];
#pragma unroll
    for (int c = 0; c < kColTile; ++c) acc[c] = scalar_traits<T>::zero();
    const int nc = (j - c0 < kColTile) ? (j - c0) : kColTile;
    if (nc == kColTile) {
        for (int64_t i = r0 + threadIdx.x; i < r1; i += kThreads) {
            const T wv = w[i];
#pragma unroll
            for (int c = 0; c < kColTile; ++c) fma_conj_acc(acc[c], V[i + (int64_t)(c0 + c) * ldv], wv);
        }
    } else {
        for (int64_t i = r0 + threadIdx.x; i < r1; i += kThreads) {
            const T wv = w[i];
#pragma unroll
            for (int c = 0; c < kColTile; ++c)
                if (c < nc) fma_conj_acc(acc[c], V[i + (int64_t)(c0 + c) * ldv], wv);
        }
    }
#pragma unroll
    for (int c = 0; c < kColTile; ++c) {
        T s = block_sum<T>(acc[c], smem);
        if (threadIdx.x == 0 && c < nc) partial[(int64_t)(c0 + c) * nchunks + chunk] = s;
    }
}

// h[c] = sum_chunk partial[c*nchunks + chunk]; one wave per column, fixed order
template <typename T>
__global__ __launch_bounds__(64) void multi_dot_finish_kernel(int j, const T* __restrict__ partial, int nchunks,
                                                               T* __restrict__ h) {
    const int c = blockIdx.x;
    if (c >= j) return;
    T acc = scalar_traits<T>::zero();
    for (int k = threadIdx.x; k < nchunks; k += 64) acc = s_add(acc, partial[(int64_t)c * nchunks + k]);
    acc = wave_sum<T>(acc);
    if (threadIdx.x == 0) h[c] = acc;
}

// w -= V h ; optionally partial_nrm[block] = sum |w_i|^2 over the block's rows
template <typename T>
__global__ __launch_bounds__(kThreads) void multi_axpy_kernel(int64_t n, int j, const T* __restrict__ V, int64_t ldv,
                                                              const T* __restrict__ h, T* __restrict__ w,
                                                              double* __restrict__ partial_nrm) {
    __shared__ double smem[4];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* hs = (T*)dyn;
    for (int c = threadIdx.x; c < j; c += kThreads) hs[c] = h[c];
    __syncthreads();
    double nrm = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T acc = scalar_traits<T>::zero();
        int c = 0;
        for (; c + 4 <= j; c += 4) {
            const T v0 = V[i + (int64_t)c * ldv], v1 = V[i + (int64_t)(c + 1) * ldv];
            const T v2 = V[i + (int64_t)(c + 2) * ldv], v3 = V[i + (int64_t)(c + 3) * ldv];
            fma_acc(acc, hs[c], v0);
            fma_acc(acc, hs[c + 1], v1);
            fma_acc(acc, hs[c + 2], v2);
            fma_acc(acc, hs[c + 3], v3);
        }
        for (; c < j; ++c) fma_acc(acc, hs[c], V[i + (int64_t)c * ldv]);
        const T r = s_sub(w[i], acc);
        w[i] = r;
        nrm += s_abs2(r);
    }
    if (partial_nrm) {
        double s = block_sum<double>(nrm, smem);
        if (threadIdx.x == 0) partial_nrm[blockIdx.x] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void nrm2_partial_kernel(int64_t n, const T* __restrict__ x,
                                                                double* __restrict__ partial) {
    __shared__ double smem[4];
    double nrm = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) nrm += s_abs2(x[i]);
    double s = block_sum<double>(nrm, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void nrm2_finish_kernel(int nblocks, const double* __restrict__ partial,
                                                          double* __restrict__ out) {
    double acc = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) acc += partial[k];
    acc = wave_sum<double>(acc);
    if (threadIdx.x == 0) out[0] = acc;
}

// w = b - z together with the partial sums of |w|^2 and |b|^2 (slots 2 blk, 2 blk + 1): the residual check of a direct
// inner solve in one pass instead of copy + axpy + two norms
template <typename T>
__global__ __launch_bounds__(kThreads) void residual_norms_partial_kernel(int64_t n, const T* __restrict__ b, const T* __restrict__ z,
                                                                          T* __restrict__ w, double* __restrict__ partial) {
    __shared__ double smem[4];
    double rw = 0.0, rb = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T bi = b[i];
        const T wi = s_sub(bi, z[i]);
        w[i] = wi;
        rw += s_abs2(wi);
        rb += s_abs2(bi);
    }
    const double sw = block_sum<double>(rw, smem);
    __syncthreads();
    const double sb = block_sum<double>(rb, smem);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sw;
        partial[2 * blockIdx.x + 1] = sb;
    }
}

__global__ __launch_bounds__(64) void nrm2_finish2_kernel(int nblocks, const double* __restrict__ partial, double* __restrict__ out) {
    double a0 = 0.0, a1 = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) {
        a0 += partial[2 * k];
        a1 += partial[2 * k + 1];
    }
    a0 = wave_sum<double>(a0);
    a1 = wave_sum<double>(a1);
    if (threadIdx.x == 0) {
        out[0] = a0;
        out[1] = a1;
    }
}

template <typename T>
__global__ void scale_inv_norm_kernel(int64_t n, const T* __restrict__ x, const double* __restrict__ nrm2,
                                      T* __restrict__ y) {
    const double s = 1.0 / sqrt(nrm2[0]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = s_mul(s, x[i]);
}

template <typename T>
__global__ void hess_column_kernel(int cnt, const T* __restrict__ h1, const T* __restrict__ h2,
                                   const double* __restrict__ nrm2, T* __restrict__ hout) {
    for (int c = threadIdx.x; c < cnt; c += blockDim.x) hout[c] = h2 ? s_add(h1[c], h2[c]) : h1[c];
    if (threadIdx.x == 0) s_from(hout[cnt], sqrt(nrm2[0]), 0.0);
}

// ---- CGS2 of one Arnoldi step in five launches (basis small enough for at most kFuseChunks row chunks) ----------------------
// The two-stage reductions keep their fixed order, but the small second stages ride in the prologue of the kernel that
// consumes them instead of being launches of their own (a dependent launch costs 4-6 us on this part, and an Arnoldi step of
// the 30 k-unknown case is a chain of ~35 of them):
//   cgs_dot     partial dots of w against V[:, 0:j], chunk-major; the workgroups of the first column tile also sum the
//               residual check of the inner solve (|b - z|^2, |b|^2 over their rows)
//   cgs_axpy    every workgroup sums the partials in chunk order (two threads per column), then w -= V h; workgroup 0 leaves h
//               (and the check's two sums) in global memory; the second pass also leaves per-workgroup sums of |w|^2
//   cgs_scale   every workgroup sums those, v_next = w / ||w||; workgroup 0 writes the Hessenberg column h1 + h2, ||w||
constexpr int kFuseChunks = 64;
constexpr int kFuseCols = 128;  // two threads per column in the prologue of cgs_axpy

// sum over the wavefront by DPP moves (two 32-bit halves per double) inside rows of 16 lanes and readlane across the four rows:
// a __shfl_xor butterfly is a chain of six ~100-cycle ds_bpermute steps per scalar, and the reductions below sum 16 scalars
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bits & 0xFFFFFFFFull), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bits >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);  // row_half_mirror
    v += dpp_mov_f64<0x140>(v);  // row_mirror: every lane of a row of 16 holds the row's sum
    const long long bits = __double_as_longlong(v);
    const int lo = (int)(unsigned)((unsigned long long)bits & 0xFFFFFFFFull), hi = (int)(unsigned)((unsigned long long)bits >> 32);
    double tot = 0.0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 16 * row), h = (unsigned)__builtin_amdgcn_readlane(hi, 16 * row);
        tot += __longlong_as_double((long long)(((unsigned long long)h << 32) | l));
    }
    return tot;
}
__device__ __forceinline__ cplx wave_sum_dpp(cplx v) { return cplx{wave_sum_dpp(v.re), wave_sum_dpp(v.im)}; }

template <typename T>
__global__ __launch_bounds__(kThreads) void cgs_dot_kernel(int64_t n, int j, int64_t rows_per_block, const T* __restrict__ V, int64_t ldv,
                                                           const T* __restrict__ w, T* __restrict__ part, int ldp,
                                                           const T* __restrict__ chk_b, const T* __restrict__ chk_z, double* __restrict__ chk_part) {
    __shared__ T wsum[4][kColTile];
    __shared__ double dsm[4][2];
    const int chunk = blockIdx.x;
    const int c0 = blockIdx.y * kColTile;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r0 = (int64_t)chunk * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    T acc[kColTile];
#pragma unroll
    for (int c = 0; c < kColTile; ++c) acc[c] = scalar_traits<T>::zero();
    const int nc = (j - c0 < kColTile) ? (j - c0) : kColTile;
    if (nc == kColTile) {
        int64_t i = r0 + threadIdx.x;
        {
            // two rows per thread and trip: 2 x kColTile loads in flight per lane (12.5 -> 10.0 us per launch at 30 k rows, j
            // averaged over a restart cycle; the order of the additions into acc[c] is that of a one-row loop)
            for (; i + kThreads < r1; i += 2 * kThreads) {
                const T wa = w[i], wb = w[i + kThreads];
                T va[kColTile], vb[kColTile];
#pragma unroll
                for (int c = 0; c < kColTile; ++c) {
                    va[c] = V[i + (int64_t)(c0 + c) * ldv];
                    vb[c] = V[i + kThreads + (int64_t)(c0 + c) * ldv];
                }
#pragma unroll
                for (int c = 0; c < kColTile; ++c) {
                    fma_conj_acc(acc[c], va[c], wa);
                    fma_conj_acc(acc[c], vb[c], wb);
                }
            }
        }
        for (; i < r1; i += kThreads) {
            const T wv = w[i];
#pragma unroll
            for (int c = 0; c < kColTile; ++c) fma_conj_acc(acc[c], V[i + (int64_t)(c0 + c) * ldv], wv);
        }
    } else {
        for (int64_t i = r0 + threadIdx.x; i < r1; i += kThreads) {
            const T wv = w[i];
#pragma unroll
            for (int c = 0; c < kColTile; ++c)
                if (c < nc) fma_conj_acc(acc[c], V[i + (int64_t)(c0 + c) * ldv], wv);
        }
    }
    // one barrier for all columns: wave sums by DPP, the four wave sums through LDS, added in wave order
#pragma unroll
    for (int c = 0; c < kColTile; ++c) {
        const T s = wave_sum_dpp(acc[c]);
        if (lane == 0) wsum[wave][c] = s;
    }
    double rw = 0.0, rb = 0.0;
    const bool chk = chk_part != nullptr && blockIdx.y == 0;
    if (chk) {
        for (int64_t i = r0 + threadIdx.x; i < r1; i += kThreads) {
            const T bi = chk_b[i];
            rw += s_abs2(s_sub(bi, chk_z[i]));
            rb += s_abs2(bi);
        }
        rw = wave_sum_dpp(rw);
        rb = wave_sum_dpp(rb);
        if (lane == 0) {
            dsm[wave][0] = rw;
            dsm[wave][1] = rb;
        }
    }
    __syncthreads();
    if (threadIdx.x < nc) {
        const int c = threadIdx.x;
        part[(int64_t)chunk * ldp + c0 + c] = s_add(s_add(wsum[0][c], wsum[1][c]), s_add(wsum[2][c], wsum[3][c]));
    }
    if (chk && threadIdx.x >= 64 && threadIdx.x < 66) {
        const int q = threadIdx.x - 64;
        chk_part[2 * chunk + q] = (dsm[0][q] + dsm[1][q]) + (dsm[2][q] + dsm[3][q]);
    }
}

// w -= V h.  Workgroup = 64 rows; four lanes share a row (lane q of the four takes the columns c = q mod 4) so that a thread
// keeps a quarter of the row's loads in flight at once: with a thread per row the j loads of a row are a chain of j / 4
// dependent batches (17.8 us per launch at 30 k rows and j = 60, on half of the CUs).  What a launch of this size costs is its
// chain of dependent round trips (round 4: 12.6 -> 9.4 us): the first eight basis entries of a thread are requested before the
// coefficients exist, the partial sums of the coefficients are requested sixteen at a time instead of one after the other, and
// the last columns of a row are one more batch of predicated loads instead of a loop of single ones.  The additions keep their
// order.
template <typename T>
__global__ __launch_bounds__(kThreads) void cgs_axpy_kernel(int64_t n, int j, const T* __restrict__ V, int64_t ldv, const T* __restrict__ part,
                                                            int nchunks, int ldp, const T* w_in, T* w, T* __restrict__ h_out,
                                                            double* __restrict__ nrm_part, const double* __restrict__ chk_part,
                                                            double* __restrict__ chk_out) {
    __shared__ double wn[4];
    __shared__ T hs[kFuseCols];
    // thread -> (row, column class): 16 consecutive lanes are 16 consecutive rows (one 256-byte segment per column), the four
    // groups of 16 lanes of a wave are the four column classes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4;
    const int64_t i = (int64_t)blockIdx.x * 64 + wave * 16 + (lane & 15);
    T v0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v0[u] = (i < n && q + 4 * u < j) ? V[i + (int64_t)(q + 4 * u) * ldv] : scalar_traits<T>::zero();
    {
        // h[c] = sum over the chunks, in chunk order inside each of two interleaved halves, then half 0 + half 1
        const int c = threadIdx.x >> 1, half = threadIdx.x & 1;
        T acc = scalar_traits<T>::zero();
        if (c < j) {
            for (int k0 = half; k0 < nchunks; k0 += 32) {
                T pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pv[u] = (k0 + 2 * u < nchunks) ? part[(int64_t)(k0 + 2 * u) * ldp + c] : scalar_traits<T>::zero();
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (k0 + 2 * u < nchunks) acc = s_add(acc, pv[u]);
            }
        }
        T other;
        if constexpr (sizeof(T) == 16) other = cplx{__shfl_xor(acc.re, 1), __shfl_xor(acc.im, 1)};
        else other = __shfl_xor(acc, 1);
        const T h = half == 0 ? s_add(acc, other) : s_add(other, acc);
        if (half == 0 && c < j) {
            hs[c] = h;
            if (blockIdx.x == 0) h_out[c] = h;
        }
        if (chk_out && blockIdx.x == 0 && threadIdx.x < 2) {
            double a = 0.0;
            for (int k = 0; k < nchunks; ++k) a += chk_part[2 * k + threadIdx.x];
            chk_out[threadIdx.x] = a;
        }
    }
    __syncthreads();
    T acc = scalar_traits<T>::zero();
    if (i < n) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q + 4 * u < j) fma_acc(acc, hs[q + 4 * u], v0[u]);
        for (int c = q + 32; c < j; c += 32) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (c + 4 * u < j) ? V[i + (int64_t)(c + 4 * u) * ldv] : scalar_traits<T>::zero();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + 4 * u < j) fma_acc(acc, hs[c + 4 * u], v[u]);
        }
    }
    // the four partial sums of a row sit 16 lanes apart: bring them together in class order (fixed order of additions)
    T a1, a2, a3;
    if constexpr (sizeof(T) == 16) {
        a1 = cplx{__shfl(acc.re, (lane & 15) + 16), __shfl(acc.im, (lane & 15) + 16)};
        a2 = cplx{__shfl(acc.re, (lane & 15) + 32), __shfl(acc.im, (lane & 15) + 32)};
        a3 = cplx{__shfl(acc.re, (lane & 15) + 48), __shfl(acc.im, (lane & 15) + 48)};
    } else {
        a1 = __shfl(acc, (lane & 15) + 16);
        a2 = __shfl(acc, (lane & 15) + 32);
        a3 = __shfl(acc, (lane & 15) + 48);
    }
    double nrm = 0.0;
    if (q == 0 && i < n) {
        const T tot = s_add(s_add(acc, a1), s_add(a2, a3));
        const T r = s_sub(w_in[i], tot);  // (w_in == w: in place; the tail form keeps the solve's result for its check)
        w[i] = r;
        nrm = s_abs2(r);
    }
    if (nrm_part) {
        nrm = wave_sum_dpp(nrm);
        if (lane == 0) wn[wave] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) nrm_part[blockIdx.x] = (wn[0] + wn[1]) + (wn[2] + wn[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void cgs_scale_kernel(int64_t n, const T* __restrict__ x, const double* __restrict__ nrm_part, int nparts,
                                                             T* __restrict__ y, int j, const T* __restrict__ h1, const T* __restrict__ h2,
                                                             T* __restrict__ hout) {
    __shared__ double smem[4];
    __shared__ double tot_s;
    double a = 0.0;
    for (int k = threadIdx.x; k < nparts; k += kThreads) a += nrm_part[k];
    const double t = block_sum<double>(a, smem);
    if (threadIdx.x == 0) tot_s = t;
    __syncthreads();
    const double tot = tot_s;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < j; c += kThreads) hout[c] = s_add(h1[c], h2[c]);
        if (threadIdx.x == 0) s_from(hout[j], sqrt(tot), 0.0);
    }
    const double s = 1.0 / sqrt(tot);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = s_mul(s, x[i]);
}

// The last launch of an Arnoldi step of OP = C^-1 M in the tail form: what cgs_scale_kernel does (||w|| from the partials of the
// second update, v_next = w / ||w||, the Hessenberg column), and in the same walk over the rows of the pattern M and C share
//   t[r] <- (M v_next)[r]                       the next step's right-hand side,
//   |t[r] - (C y)[r]|^2, |t[r]|^2               this step's check of the inner solve (t[r] read before it is replaced),
// a sub-wave of LPR lanes per row exactly as spmv_subwave_kernel walks it: entries lane, lane + LPR, ... in order, butterfly over
// the sub-wave -- the products are bit for bit those of k_spmv on v_next and y.  Up to three trips of a row are requested at once.
// (Requesting the row's entries ahead of the norm's partial sums changed nothing: 12.4 / 13.2 us; the two gathers per entry bound it.)
template <typename MT, int LPR>
__global__ __launch_bounds__(kThreads) void cgs_tail_kernel(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                            const MT* __restrict__ mval, const cplx* __restrict__ cval,
                                                            const cplx* __restrict__ w, const cplx* __restrict__ y,
                                                            const double* __restrict__ nrm_part, int nparts, cplx* __restrict__ vnext, cplx* t,
                                                            int j, const cplx* __restrict__ h1, const cplx* __restrict__ h2,
                                                            cplx* __restrict__ hout, double* __restrict__ chk_part) {
    __shared__ double smem[4];
    __shared__ double tot_s;
    __shared__ double csum[4][2];
    double a = 0.0;
    for (int k = threadIdx.x; k < nparts; k += kThreads) a += nrm_part[k];
    const double tsum = block_sum<double>(a, smem);
    if (threadIdx.x == 0) tot_s = tsum;
    __syncthreads();
    const double tot = tot_s;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < j; c += kThreads) hout[c] = s_add(h1[c], h2[c]);
        if (threadIdx.x == 0) s_from(hout[j], sqrt(tot), 0.0);
    }
    const double s = 1.0 / sqrt(tot);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t lane = (int32_t)(gid % LPR);
    const int64_t row_stride = ((int64_t)gridDim.x * blockDim.x) / LPR;
    double rw = 0.0, rb = 0.0;
    for (int64_t row = gid / LPR; row < n; row += row_stride) {
        const int32_t p0 = rp[row], p1 = rp[row + 1];
        cplx am = cplx{0.0, 0.0}, ac = cplx{0.0, 0.0};
        for (int32_t p = p0 + lane; p < p1; p += 3 * LPR) {
            int32_t c[3];
            MT vm[3];
            cplx vc[3], xw[3], xy[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool in = p + u * LPR < p1;
                c[u] = in ? ci[p + u * LPR] : 0;
                vm[u] = in ? mval[p + u * LPR] : scalar_traits<MT>::zero();
                vc[u] = in ? cval[p + u * LPR] : cplx{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool in = p + u * LPR < p1;
                xw[u] = in ? w[c[u]] : cplx{0.0, 0.0};
                xy[u] = in ? y[c[u]] : cplx{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (p + u * LPR < p1) {
                    fma_acc(am, vm[u], s_mul(s, xw[u]));
                    fma_acc(ac, vc[u], xy[u]);
                }
        }
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) {
            am = s_add(am, cplx{__shfl_xor(am.re, m, 64), __shfl_xor(am.im, m, 64)});
            ac = s_add(ac, cplx{__shfl_xor(ac.re, m, 64), __shfl_xor(ac.im, m, 64)});
        }
        if (lane == 0) {
            const cplx told = t[row];
            rw += s_abs2(s_sub(told, ac));
            rb += s_abs2(told);
            t[row] = am;
            vnext[row] = s_mul(s, w[row]);
        }
    }
    rw = wave_sum_dpp(rw);
    rb = wave_sum_dpp(rb);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        csum[wave][0] = rw;
        csum[wave][1] = rb;
    }
    __syncthreads();
    if (threadIdx.x < 2) chk_part[2 * blockIdx.x + threadIdx.x] = (csum[0][threadIdx.x] + csum[1][threadIdx.x]) + (csum[2][threadIdx.x] + csum[3][threadIdx.x]);
}

// checks[2 s], checks[2 s + 1] = the sums of slot s's pairs (one workgroup per slot, fixed order)
__global__ __launch_bounds__(kThreads) void cgs_tail_checks_kernel(int nparts, const double* __restrict__ parts, double* __restrict__ checks) {
    __shared__ double smem[4];
    const double* p = parts + (size_t)blockIdx.x * 2 * (size_t)nparts;
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nparts; k += kThreads) {
        a += p[2 * k];
        b += p[2 * k + 1];
    }
    const double sa = block_sum<double>(a, smem);
    const double sb = block_sum<double>(b, smem);
    if (threadIdx.x == 0) {
        checks[2 * blockIdx.x] = sa;
        checks[2 * blockIdx.x + 1] = sb;
    }
}

// ---- Ritz vectors: unit norm and a canonical phase for all columns in two launches --------------------------------------------
// per (block, column): sum |x|^2 over the block's rows and the entry of largest magnitude (lowest row among equals)
__global__ __launch_bounds__(kThreads) void col_stats_kernel(int64_t n, const cplx* __restrict__ X, int64_t ldx, double* __restrict__ nrm_part,
                                                             double* __restrict__ mag_part, long long* __restrict__ idx_part) {
    __shared__ double ssum[4];
    __shared__ double smag[kThreads];
    __shared__ long long sidx[kThreads];
    const cplx* x = X + (size_t)blockIdx.y * (size_t)ldx;
    double s = 0.0, best = -1.0;
    long long bi = -1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double a = s_abs2(x[i]);
        s += a;
        if (a > best) {
            best = a;
            bi = i;
        }
    }
    const double tot = block_sum<double>(s, ssum);
    smag[threadIdx.x] = best;
    sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double m2 = smag[threadIdx.x + o];
            const long long i2 = sidx[threadIdx.x + o];
            if (m2 > smag[threadIdx.x] || (m2 == smag[threadIdx.x] && i2 >= 0 && (sidx[threadIdx.x] < 0 || i2 < sidx[threadIdx.x]))) {
                smag[threadIdx.x] = m2;
                sidx[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const size_t at = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        nrm_part[at] = tot;
        mag_part[at] = smag[0];
        idx_part[at] = sidx[0];
    }
}

// every block combines its column's partials in the same fixed order, then scales its rows:  x <- x * conj(x_k) / |x_k|
// (/ ||x|| when `unit`), x_k the entry of largest magnitude;  imag_part[(column, block)] = sum Im(x)^2 afterwards
__global__ __launch_bounds__(kThreads) void col_phase_kernel(int64_t n, cplx* __restrict__ X, int64_t ldx, const double* __restrict__ nrm_part,
                                                             const double* __restrict__ mag_part, const long long* __restrict__ idx_part, int unit,
                                                             double* __restrict__ imag_part) {
    __shared__ double ssum[4];
    __shared__ double smag[kThreads];
    __shared__ long long sidx[kThreads];
    cplx* x = X + (size_t)blockIdx.y * (size_t)ldx;
    const int nb = gridDim.x;
    const size_t base = (size_t)blockIdx.y * nb;
    double s = 0.0, best = -1.0;
    long long bi = -1;
    for (int q = threadIdx.x; q < nb; q += kThreads) {
        s += nrm_part[base + q];
        const double m2 = mag_part[base + q];
        const long long i2 = idx_part[base + q];
        if (m2 > best || (m2 == best && i2 >= 0 && (bi < 0 || i2 < bi))) {
            best = m2;
            bi = i2;
        }
    }
    const double tot = block_sum<double>(s, ssum);
    smag[threadIdx.x] = best;
    sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double m2 = smag[threadIdx.x + o];
            const long long i2 = sidx[threadIdx.x + o];
            if (m2 > smag[threadIdx.x] || (m2 == smag[threadIdx.x] && i2 >= 0 && (sidx[threadIdx.x] < 0 || i2 < sidx[threadIdx.x]))) {
                smag[threadIdx.x] = m2;
                sidx[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    const long long k = sidx[0];
    cplx f{1.0, 0.0};
    if (k >= 0 && smag[0] > 0.0) {
        const cplx xk = x[k];
        const double inv = 1.0 / sqrt(smag[0]);
        f = cplx{xk.re * inv, -xk.im * inv};
    }
    if (unit && tot > 0.0) {
        const double inv = 1.0 / sqrt(tot);
        f.re *= inv;
        f.im *= inv;
    }
    __syncthreads();  // (every thread has read x[k] before any block-mate overwrites it; other blocks: x[k] is rewritten only by
                      //  the block that owns row k, and a value read after that rewrite would differ -- so row k is scaled LAST, below)
    double im2 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (i == k) continue;
        const cplx v = s_mul(x[i], f);
        x[i] = v;
        im2 += v.im * v.im;
    }
    const double it = block_sum<double>(im2, ssum);
    if (threadIdx.x == 0) imag_part[base + blockIdx.x] = it;
}

// the pivot entries themselves (skipped above so that every block reads the original value), and the column sums of Im^2
__global__ __launch_bounds__(64) void col_phase_finish_kernel(cplx* __restrict__ X, int64_t ldx, int nb, const double* __restrict__ nrm_part,
                                                              const double* __restrict__ mag_part, const long long* __restrict__ idx_part, int unit,
                                                              const double* __restrict__ imag_part, double* __restrict__ imag2) {
    const size_t base = (size_t)blockIdx.x * nb;
    double s = 0.0, im = 0.0, best = -1.0;
    long long bi = -1;
    for (int q = threadIdx.x; q < nb; q += 64) {
        s += nrm_part[base + q];
        im += imag_part[base + q];
        const double m2 = mag_part[base + q];
        const long long i2 = idx_part[base + q];
        if (m2 > best || (m2 == best && i2 >= 0 && (bi < 0 || i2 < bi))) {
            best = m2;
            bi = i2;
        }
    }
    s = wave_sum<double>(s);
    im = wave_sum<double>(im);
    for (int o = 32; o > 0; o >>= 1) {
        const double m2 = __shfl_xor(best, o);
        const long long i2 = __shfl_xor(bi, o);
        if (m2 > best || (m2 == best && i2 >= 0 && (bi < 0 || i2 < bi))) {
            best = m2;
            bi = i2;
        }
    }
    if (threadIdx.x == 0) {
        if (bi >= 0 && best > 0.0) {
            double mag = sqrt(best);
            if (unit && s > 0.0) mag /= sqrt(s);
            X[(size_t)blockIdx.x * (size_t)ldx + (size_t)bi] = cplx{mag, 0.0};  // x_k conj(x_k) / |x_k| (/ ||x||): real, positive
        }
        imag2[blockIdx.x] = im;
    }
}

// Out[i, k0+t] = sum_c V[i, c] Q[c, k0+t], t < kOutTile
constexpr int kOutTile = 8;  // (round 4: 4 -> 8 halves the passes over a basis that does not fit the caches)
template <typename T>
__global__ __launch_bounds__(kThreads) void basis_gemm_kernel(int64_t n, int m, int k, const T* __restrict__ V, int64_t ldv,
                                                              const T* __restrict__ Q, int ldq, T* __restrict__ Out,
                                                              int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* qs = (T*)dyn;  // m x kOutTile, column-major
    const int k0 = blockIdx.y * kOutTile;
    const int nk = (k - k0 < kOutTile) ? (k - k0) : kOutTile;
    for (int idx = threadIdx.x; idx < m * kOutTile; idx += kThreads) {
        const int t = idx / m, c = idx % m;
        qs[idx] = (t < nk) ? Q[c + (int64_t)(k0 + t) * ldq] : scalar_traits<T>::zero();
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T acc[kOutTile];
#pragma unroll
        for (int t = 0; t < kOutTile; ++t) acc[t] = scalar_traits<T>::zero();
        int c = 0;
        for (; c + 7 < m; c += 8) {  // eight basis entries requested at once (a loop of single loads is a chain of m round trips)
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = V[i + (int64_t)(c + u) * ldv];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int t = 0; t < kOutTile; ++t) fma_acc(acc[t], qs[c + u + t * m], v[u]);
            }
        }
        for (; c < m; ++c) {
            const T v = V[i + (int64_t)c * ldv];
#pragma unroll
            for (int t = 0; t < kOutTile; ++t) fma_acc(acc[t], qs[c + t * m], v);
        }
#pragma unroll
        for (int t = 0; t < kOutTile; ++t)
            if (t < nk) Out[i + (int64_t)(k0 + t) * ldo] = acc[t];
    }
}

inline int stream_blocks(lsa_ctx* ctx, int64_t n) {
    int64_t want = (n + kThreads - 1) / kThreads;
    int64_t cap = (int64_t)ctx->num_cu * 8;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

inline int check_launch(lsa_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return LSA_OK;
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                 \
    do {                                        \
        if ((dtype) == LSA_C128) { using T = cplx; CALL; } \
        else { using T = double; CALL; }        \
    } while (0)

int k_copy(lsa_ctx* ctx, int dtype, int64_t n, const void* x, void* y) {
    LSA_HIP_CHECK(ctx, hipMemcpyAsync(y, x, (size_t)n * (dtype == LSA_C128 ? 16 : 8), hipMemcpyDeviceToDevice, ctx->stream));
    return LSA_OK;
}

int k_set_zero(lsa_ctx* ctx, int dtype, int64_t n, void* x) {
    LSA_HIP_CHECK(ctx, hipMemsetAsync(x, 0, (size_t)n * (dtype == LSA_C128 ? 16 : 8), ctx->stream));
    return LSA_OK;
}

int k_axpy(lsa_ctx* ctx, int dtype, int64_t n, const double alpha[2], const void* x, void* y) {
    if (dtype == LSA_C128)
        hipLaunchKernelGGL((axpy_kernel<cplx>), dim3(stream_blocks(ctx, n)), dim3(kThreads), 0, ctx->stream, n,
                           cplx{alpha[0], alpha[1]}, (const cplx*)x, (cplx*)y);
    else
        hipLaunchKernelGGL((axpy_kernel<double>), dim3(stream_blocks(ctx, n)), dim3(kThreads), 0, ctx->stream, n, alpha[0],
                           (const double*)x, (double*)y);
    return check_launch(ctx, "axpy");
}

static int64_t dot_rows_per_block(int64_t n) {
    int64_t r = (n + 2047) / 2048;          // at most 2048 chunks
    r = ((r + kThreads - 1) / kThreads) * kThreads;
    return r < 512 ? 512 : r;
}

int k_multi_dot(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, const void* w, void* h_dev) {
    if (j <= 0) return LSA_OK;
    const int64_t rpb = dot_rows_per_block(n);
    const int nchunks = (int)((n + rpb - 1) / rpb);
    const size_t esz = dtype == LSA_C128 ? 16 : 8;
    LSA_CHECK(lsa_ensure_scratch(ctx, (size_t)j * nchunks * esz, 0));
    dim3 grid(nchunks, (j + kColTile - 1) / kColTile);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((multi_dot_partial_kernel<T>), grid, dim3(kThreads), 0, ctx->stream, n, j, rpb, (const T*)V, ldv,
                           (const T*)w, (T*)ctx->dscratch, nchunks);
        hipLaunchKernelGGL((multi_dot_finish_kernel<T>), dim3(j), dim3(64), 0, ctx->stream, j, (const T*)ctx->dscratch, nchunks,
                           (T*)h_dev);
    });
    return check_launch(ctx, "multi_dot");
}

int k_multi_axpy(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, const void* h_dev, void* w,
                 double* nrm2_dev) {
    const int blocks = stream_blocks(ctx, n);
    const size_t esz = dtype == LSA_C128 ? 16 : 8;
    double* partial = nullptr;
    if (nrm2_dev) {
        LSA_CHECK(lsa_ensure_scratch(ctx, sizeof(double) * (size_t)blocks, 0));
        partial = (double*)ctx->dscratch;
    }
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((multi_axpy_kernel<T>), dim3(blocks), dim3(kThreads), (size_t)(j > 0 ? j : 1) * esz, ctx->stream, n, j,
                           (const T*)V, ldv, (const T*)h_dev, (T*)w, partial);
    });
    if (nrm2_dev) hipLaunchKernelGGL(nrm2_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, nrm2_dev);
    return check_launch(ctx, "multi_axpy");
}

// First projection and second dot product of CGS2 in ONE pass over the basis (long vectors: the basis does not fit the caches,
// every pass streams j n scalars from HBM -- 0.43 ms of a 1.35 ms Arnoldi step at 500 k unknowns with four passes):
//     w <- w - V h,      partial[c * ntiles + tile] = sum over the tile's 64 rows of conj(V[i, c]) w_i (updated).
// Workgroup = 64 rows; wavefront q holds the columns q, q + 4, ... of those rows in registers (lane = row: 1 KB per load
// instruction), adds up its share of V h, the four shares meet in LDS, and the columns still in registers give the dot products.
template <typename T, int KMAX>
__global__ __launch_bounds__(kThreads) void cgs_axpy_dot_kernel(int64_t n, int j, const T* __restrict__ V, int64_t ldv, const T* __restrict__ h,
                                                                T* __restrict__ w, T* __restrict__ partial, int64_t ntiles) {
    __shared__ T accs[4][64];
    __shared__ T hs[4 * KMAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < j; c += kThreads) hs[c] = h[c];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool in = i < n;
    T v[KMAX];
    T acc = scalar_traits<T>::zero();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = wave + 4 * k;
        v[k] = (in && c < j) ? V[i + (int64_t)c * ldv] : scalar_traits<T>::zero();
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = wave + 4 * k;
        if (c < j) fma_acc(acc, hs[c], v[k]);
    }
    // Every wavefront needs the OLD w[i] and wavefront 0 writes the new one: the old value is read in front of the barrier.
    // (Round 3 read it behind the barrier in all four wavefronts -- a wavefront that fell behind wavefront 0 then saw the new
    //  value and subtracted the projection twice for its columns.  Met in round 4 with four ranks time-sharing one GPU: one
    //  tile in some step of some rank, an error of 1e-9 .. 1e-5 in a Hessenberg column, ranks out of step.)
    const T w_old = in ? w[i] : scalar_traits<T>::zero();
    accs[wave][lane] = acc;
    __syncthreads();
    T wi = scalar_traits<T>::zero();
    if (in) wi = s_sub(w_old, s_add(s_add(accs[0][lane], accs[1][lane]), s_add(accs[2][lane], accs[3][lane])));
    if (wave == 0 && in) w[i] = wi;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = wave + 4 * k;
        if (c >= j) continue;  // (uniform over the wavefront)
        T p = scalar_traits<T>::zero();
        fma_conj_acc(p, v[k], wi);
        p = wave_sum_dpp(p);
        if (lane == 0) partial[(int64_t)c * ntiles + blockIdx.x] = p;
    }
}

// h[c] = sum over thousands of tile partials: one workgroup of 1 024 threads per column (the one-wavefront form above walks
// 7 900 partials of a 500 k-row vector in 123 dependent trips: 45 us), fixed order
template <typename T>
__global__ __launch_bounds__(1024) void multi_dot_finish_wide_kernel(int j, const T* __restrict__ partial, int64_t nchunks, T* __restrict__ h) {
    __shared__ T sm[16];
    const int c = blockIdx.x;
    if (c >= j) return;
    T acc = scalar_traits<T>::zero();
    for (int64_t k = threadIdx.x; k < nchunks; k += 1024) acc = s_add(acc, partial[(int64_t)c * nchunks + k]);
    acc = wave_sum_dpp(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = sm[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) t = s_add(t, sm[q]);
        h[c] = t;
    }
}

// w -= V h1 and h2 = V^H w (of the updated w) with one pass over V; returns 1 when the shape is not handled (nothing launched)
int k_multi_axpy_dot(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, const void* h1_dev, void* w, void* h2_dev) {
    if (j <= 0 || j > 128 || n <= 0) return 1;
    const int64_t ntiles = (n + 63) / 64;
    if (ntiles > 0x7FFFFFFF) return 1;
    const size_t esz = dtype == LSA_C128 ? 16 : 8;
    LSA_CHECK(lsa_ensure_scratch(ctx, (size_t)j * (size_t)ntiles * esz, 0));
    DISPATCH_T(dtype, {
        if (j <= 80)
            hipLaunchKernelGGL((cgs_axpy_dot_kernel<T, 20>), dim3((unsigned)ntiles), dim3(kThreads), 0, ctx->stream, n, j, (const T*)V, ldv, (const T*)h1_dev,
                               (T*)w, (T*)ctx->dscratch, ntiles);
        else
            hipLaunchKernelGGL((cgs_axpy_dot_kernel<T, 32>), dim3((unsigned)ntiles), dim3(kThreads), 0, ctx->stream, n, j, (const T*)V, ldv, (const T*)h1_dev,
                               (T*)w, (T*)ctx->dscratch, ntiles);
        hipLaunchKernelGGL((multi_dot_finish_wide_kernel<T>), dim3(j), dim3(1024), 0, ctx->stream, j, (const T*)ctx->dscratch, ntiles, (T*)h2_dev);
    });
    return check_launch(ctx, "multi_axpy_dot");
}

// columns of X (n x ncols, complex, column-major): canonical phase (largest entry real positive), unit 2-norm when `unit`;
// imag2_dev[c] = sum_i Im(X[i, c])^2 afterwards.  Three launches for all columns.
int k_columns_canonical(lsa_ctx* ctx, int64_t n, int ncols, void* X, int64_t ldx, int unit, double* imag2_dev) {
    if (ncols <= 0 || n <= 0) return LSA_OK;
    const int nb = stream_blocks(ctx, n);
    const size_t per = (size_t)ncols * nb;
    LSA_CHECK(lsa_ensure_scratch(ctx, per * (3 * sizeof(double) + sizeof(long long)), 0));
    double* nrm = (double*)ctx->dscratch;
    double* mag = nrm + per;
    double* imp = mag + per;
    long long* idx = (long long*)(imp + per);
    hipLaunchKernelGGL(col_stats_kernel, dim3(nb, ncols), dim3(kThreads), 0, ctx->stream, n, (const cplx*)X, ldx, nrm, mag, idx);
    hipLaunchKernelGGL(col_phase_kernel, dim3(nb, ncols), dim3(kThreads), 0, ctx->stream, n, (cplx*)X, ldx, (const double*)nrm, (const double*)mag,
                       (const long long*)idx, unit, imp);
    hipLaunchKernelGGL(col_phase_finish_kernel, dim3(ncols), dim3(64), 0, ctx->stream, (cplx*)X, ldx, nb, (const double*)nrm, (const double*)mag,
                       (const long long*)idx, unit, (const double*)imp, imag2_dev);
    return check_launch(ctx, "columns_canonical");
}

int k_nrm2(lsa_ctx* ctx, int dtype, int64_t n, const void* x, double* nrm2_dev) {
    const int blocks = stream_blocks(ctx, n);
    LSA_CHECK(lsa_ensure_scratch(ctx, sizeof(double) * (size_t)blocks, 0));
    double* partial = (double*)ctx->dscratch;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((nrm2_partial_kernel<T>), dim3(blocks), dim3(kThreads), 0, ctx->stream, n, (const T*)x, partial);
    });
    hipLaunchKernelGGL(nrm2_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, nrm2_dev);
    return check_launch(ctx, "nrm2");
}

template <typename T>
__global__ void mask_mul_kernel(int64_t n, const double* __restrict__ keep, T* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (keep[i] == 0.0) y[i] = scalar_traits<T>::zero();
}

// y[i] = 0 where keep[i] == 0
int k_mask(lsa_ctx* ctx, int dtype, int64_t n, const double* keep_dev, void* y) {
    DISPATCH_T(dtype, { hipLaunchKernelGGL((mask_mul_kernel<T>), dim3(stream_blocks(ctx, n)), dim3(kThreads), 0, ctx->stream, n, keep_dev, (T*)y); });
    return check_launch(ctx, "mask");
}

// w = b - z;  nrm2_dev[0] = ||w||^2,  nrm2_dev[1] = ||b||^2  (fixed-order two-stage sums)
int k_residual_norms(lsa_ctx* ctx, int dtype, int64_t n, const void* b, const void* z, void* w, double* nrm2_dev) {
    const int blocks = stream_blocks(ctx, n);
    LSA_CHECK(lsa_ensure_scratch(ctx, sizeof(double) * 2 * (size_t)blocks, 0));
    double* partial = (double*)ctx->dscratch;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((residual_norms_partial_kernel<T>), dim3(blocks), dim3(kThreads), 0, ctx->stream, n, (const T*)b, (const T*)z, (T*)w,
                           partial);
    });
    hipLaunchKernelGGL(nrm2_finish2_kernel, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, nrm2_dev);
    return check_launch(ctx, "residual_norms");
}

int k_scale_by_inv_norm(lsa_ctx* ctx, int dtype, int64_t n, const void* x, const double* nrm2_dev, void* y) {
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((scale_inv_norm_kernel<T>), dim3(stream_blocks(ctx, n)), dim3(kThreads), 0, ctx->stream, n, (const T*)x,
                           nrm2_dev, (T*)y);
    });
    return check_launch(ctx, "scale");
}

int k_hess_column(lsa_ctx* ctx, int dtype, int cnt, const void* h1, const void* h2, const double* nrm2_dev, void* hout) {
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((hess_column_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, cnt, (const T*)h1, (const T*)h2, nrm2_dev,
                           (T*)hout);
    });
    return check_launch(ctx, "hess_column");
}

int k_basis_gemm(lsa_ctx* ctx, int dtype, int64_t n, int m, int k, const void* V, int64_t ldv, const void* Q, int ldq,
                 void* Out, int64_t ldo) {
    if (k <= 0 || m <= 0) return LSA_OK;
    const size_t esz = dtype == LSA_C128 ? 16 : 8;
    int bx = stream_blocks(ctx, n);
    dim3 grid(bx, (k + kOutTile - 1) / kOutTile);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((basis_gemm_kernel<T>), grid, dim3(kThreads), (size_t)m * kOutTile * esz, ctx->stream, n, m, k,
                           (const T*)V, ldv, (const T*)Q, ldq, (T*)Out, ldo);
    });
    return check_launch(ctx, "basis_gemm");
}

// CGS2 of w against V[:, 0:j] and v_next = w / ||w|| in five launches (see cgs_dot_kernel), with the inner solve's residual
// check (chk_b, chk_z -> chk_out[2] = |b - z|^2, |b|^2; all three null: none) folded in.  `work` holds at least
// k_cgs2_fused_work_bytes(ctx, n, j) bytes.  Returns 1 when the shape is outside what the fused form handles (nothing was
// launched: the caller takes the kernel-per-stage path), LSA_OK or a negative status otherwise.
size_t k_cgs2_fused_work_bytes(lsa_ctx* ctx, int64_t n, int jmax) {
    (void)ctx;
    (void)jmax;
    return (size_t)16 * (size_t)(2 * kFuseChunks * kFuseCols + 2 * kFuseCols) + sizeof(double) * (size_t)(2 * kFuseChunks + (n + 63) / 64 + 8);
}

int k_cgs2_fused(lsa_ctx* ctx, int dtype, int64_t n, int j, const void* V, int64_t ldv, void* w, void* vnext, void* hcol_dev, void* work,
                 const void* chk_b, const void* chk_z, double* chk_out) {
    if (j <= 0 || j > kFuseCols || n <= 0) return 1;
    int64_t rpb = (n + kFuseChunks - 1) / kFuseChunks;
    rpb = ((rpb + kThreads - 1) / kThreads) * kThreads;
    if (rpb < 1024) rpb = 1024;  // (every workgroup of the axpy sums the partials of all chunks: few, fat chunks)
    if (rpb > 4096) return 1;    // long vectors: the second stages are noise there, and wider grids stream better
    const int nchunks = (int)((n + rpb - 1) / rpb);
    const int blocks = (int)((n + 63) / 64);  // workgroups of the axpy: 64 rows each
    const size_t esz = dtype == LSA_C128 ? 16 : 8;
    char* p = (char*)work;
    void* part1 = p;
    p += esz * (size_t)kFuseChunks * kFuseCols;
    void* part2 = p;
    p += esz * (size_t)kFuseChunks * kFuseCols;
    void* h1 = p;
    p += esz * kFuseCols;
    void* h2 = p;
    p += esz * kFuseCols;
    double* chk_part = (double*)p;
    double* nrm_part = chk_part + 2 * kFuseChunks;
    const dim3 dgrid(nchunks, (j + kColTile - 1) / kColTile);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((cgs_dot_kernel<T>), dgrid, dim3(kThreads), 0, ctx->stream, n, j, rpb, (const T*)V, ldv, (const T*)w, (T*)part1, kFuseCols,
                           (const T*)chk_b, (const T*)chk_z, chk_out ? chk_part : (double*)nullptr);
        hipLaunchKernelGGL((cgs_axpy_kernel<T>), dim3(blocks), dim3(kThreads), 0, ctx->stream, n, j, (const T*)V, ldv, (const T*)part1, nchunks, kFuseCols,
                           (const T*)w, (T*)w, (T*)h1, (double*)nullptr, (const double*)chk_part, chk_out);
        hipLaunchKernelGGL((cgs_dot_kernel<T>), dgrid, dim3(kThreads), 0, ctx->stream, n, j, rpb, (const T*)V, ldv, (const T*)w, (T*)part2, kFuseCols,
                           (const T*)nullptr, (const T*)nullptr, (double*)nullptr);
        hipLaunchKernelGGL((cgs_axpy_kernel<T>), dim3(blocks), dim3(kThreads), 0, ctx->stream, n, j, (const T*)V, ldv, (const T*)part2, nchunks, kFuseCols,
                           (const T*)w, (T*)w, (T*)h2, nrm_part, (const double*)nullptr, (double*)nullptr);
        hipLaunchKernelGGL((cgs_scale_kernel<T>), dim3(stream_blocks(ctx, n)), dim3(kThreads), 0, ctx->stream, n, (const T*)w, (const double*)nrm_part, blocks, (T*)vnext,
                           j, (const T*)h1, (const T*)h2, (T*)hcol_dev);
    });
    return check_launch(ctx, "cgs2_fused");
}

static bool cgs2_fused_shape(int64_t n, int j, int64_t* rpb_out) {
    if (j <= 0 || j > kFuseCols || n <= 0) return false;
    int64_t rpb = (n + kFuseChunks - 1) / kFuseChunks;
    rpb = ((rpb + kThreads - 1) / kThreads) * kThreads;
    if (rpb < 1024) rpb = 1024;
    if (rpb > 4096) return false;
    if (rpb_out) *rpb_out = rpb;
    return true;
}

bool k_cgs2_tail_fits(lsa_ctx* ctx, int64_t n, int jmax, const lsa_mat* M, const lsa_mat* C) {
    (void)ctx;
    if (!M || !C || !cgs2_fused_shape(n, jmax, nullptr)) return false;
    if (C->dtype != LSA_C128 || M->n != n || C->n != n || M->ncols != n || C->ncols != n || M->row0 != 0 || C->row0 != 0) return false;
    // one pattern: the same device index arrays, or host copies that lsa_csr_axpby has found equal and made one object
    const bool same_dev = M->rp == C->rp && M->ci == C->ci;
    const bool same_host = M->h_rp.size() > 0 && M->h_rp.same_object(C->h_rp) && M->h_ci.same_object(C->h_ci);
    if (M->nnz != C->nnz || !(same_dev || same_host)) return false;
    return k_spmv_plain_subwave_lanes(M) == 16 && k_spmv_plain_subwave_lanes(C) == 16;
}

int k_cgs2_tail_parts(int64_t n) { return (int)((n * 16 + kThreads - 1) / kThreads); }

int k_cgs2_fused_tail(lsa_ctx* ctx, int64_t n, int j, const void* V, int64_t ldv, const void* y, void* w, void* vnext, void* hcol_dev, void* work,
                      const lsa_mat* M, const lsa_mat* C, void* t, double* tail_part) {
    typedef cplx T;
    int64_t rpb = 0;
    if (!cgs2_fused_shape(n, j, &rpb)) return lsa_set_error(ctx, LSA_ERR_ARG, "k_cgs2_fused_tail: shape outside the fused form");
    const int nchunks = (int)((n + rpb - 1) / rpb);
    const int blocks = (int)((n + 63) / 64);
    const size_t esz = 16;
    char* p = (char*)work;
    void* part1 = p;
    p += esz * (size_t)kFuseChunks * kFuseCols;
    void* part2 = p;
    p += esz * (size_t)kFuseChunks * kFuseCols;
    void* h1 = p;
    p += esz * kFuseCols;
    void* h2 = p;
    p += esz * kFuseCols;
    double* chk_part = (double*)p;
    double* nrm_part = chk_part + 2 * kFuseChunks;
    const dim3 dgrid(nchunks, (j + kColTile - 1) / kColTile);
    hipLaunchKernelGGL((cgs_dot_kernel<T>), dgrid, dim3(kThreads), 0, ctx->stream, n, j, rpb, (const T*)V, ldv, (const T*)y, (T*)part1, kFuseCols,
                       (const T*)nullptr, (const T*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL((cgs_axpy_kernel<T>), dim3(blocks), dim3(kThreads), 0, ctx->stream, n, j, (const T*)V, ldv, (const T*)part1, nchunks, kFuseCols,
                       (const T*)y, (T*)w, (T*)h1, (double*)nullptr, (const double*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL((cgs_dot_kernel<T>), dgrid, dim3(kThreads), 0, ctx->stream, n, j, rpb, (const T*)V, ldv, (const T*)w, (T*)part2, kFuseCols,
                       (const T*)nullptr, (const T*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL((cgs_axpy_kernel<T>), dim3(blocks), dim3(kThreads), 0, ctx->stream, n, j, (const T*)V, ldv, (const T*)part2, nchunks, kFuseCols,
                       (const T*)w, (T*)w, (T*)h2, nrm_part, (const double*)nullptr, (double*)nullptr);
    const int tparts = k_cgs2_tail_parts(n);
    if (M->dtype == LSA_C128)
        hipLaunchKernelGGL((cgs_tail_kernel<cplx, 16>), dim3(tparts), dim3(kThreads), 0, ctx->stream, (int32_t)n, C->rp, C->ci, (const cplx*)M->val,
                           (const cplx*)C->val, (const T*)w, (const T*)y, (const double*)nrm_part, blocks, (T*)vnext, (T*)t, j, (const T*)h1, (const T*)h2,
                           (T*)hcol_dev, tail_part);
    else
        hipLaunchKernelGGL((cgs_tail_kernel<double, 16>), dim3(tparts), dim3(kThreads), 0, ctx->stream, (int32_t)n, C->rp, C->ci, (const double*)M->val,
                           (const cplx*)C->val, (const T*)w, (const T*)y, (const double*)nrm_part, blocks, (T*)vnext, (T*)t, j, (const T*)h1, (const T*)h2,
                           (T*)hcol_dev, tail_part);
    return check_launch(ctx, "cgs2_fused_tail");
}

int k_cgs2_tail_checks(lsa_ctx* ctx, int nslots, int nparts, const double* tail_parts, double* checks) {
    if (nslots <= 0) return LSA_OK;
    hipLaunchKernelGGL(cgs_tail_checks_kernel, dim3(nslots), dim3(kThreads), 0, ctx->stream, nparts, tail_parts, checks);
    return check_launch(ctx, "cgs2_tail_checks");
}
