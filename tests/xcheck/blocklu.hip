// CROSS-CHECK LIBRARY (liblsa_xcheck.so, tests only -- not part of liblsa_hip.so): round 1's exact solver, superseded in the
// product by the nested-dissection multifrontal LU (lsa-fw_amd/csrc/ndlu.hip) and kept as an independent direct solver the
// tests compare it with.
//
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
// in-place Gauss-Jordan elimination with partial pivoting; the off-diagonal blocks stay the sparse rows of C.
//
// Both the factorisation and the solve are TWISTED: one chain of blocks runs down from block 0, a second one up from
// the last block, and they meet at the middle block, which receives both Schur corrections.  The factorisation runs
// the two chains on two streams (one launch per panel of 8 pivot columns, gj_fused_kernel); a solve advances one
// block of each chain per launch (2 launches per pair of blocks and sweep, replayed from a hipGraph), and its
// substitution sweep reads only the columns of the Schur inverses that meet a non-zero.  It is a *direct* solve
// (residual ~1e-14): the GMRES around it checks b - C x and only iterates if that check fails.
// Not usable when the band does not fit (3D meshes) or a Schur block is singular (no pivoting across blocks):
// lsa_blu_create then fails.
#include <algorithm>
#include <chrono>

#include "lsa_internal.h"
#include "lsa_xcheck.h"

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
    // wide blocks (> 2048 rows): per chain a side stream for the panel workgroup, which runs beside the previous panel's
    // bulk update, and the two events that order them
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_panel[2] = {nullptr, nullptr}, ev_update[2] = {nullptr, nullptr};
    int32_t mid = 0;                        // middle block: chains run 0 .. mid-1 and nb-1 .. mid+1
    void *t[2] = {nullptr, nullptr}, *y[2] = {nullptr, nullptr}, *z[2] = {nullptr, nullptr};
    void *in[2] = {nullptr, nullptr}, *out[2] = {nullptr, nullptr};
    void* graph[2] = {nullptr, nullptr};
    double seconds = 0.0;
    std::vector<int32_t> uwin_lo, lwin_hi;  // per block: first row with entries right of the block / end of the rows with entries left of it
    // absorbed couplings: G_b = Sinv_b C_{b,b-1} and K_b = Sinv_b C_{b,b+1} restricted to the columns the sparse blocks
    // touch ([gcol0[b], bs) and [be, kcol1[b])); n x ldg / n x ldk dense, row-major.  With them a sweep step is ONE dense
    // launch (no sparse half-step).  Null when they do not fit: the sweeps then use Sinv + the sparse rows of C.
    void *G = nullptr, *K = nullptr;
    int32_t ldg = 0, ldk = 0;
    bool want_absorb = false;
    std::vector<int32_t> gcol0, kcol1;
    uint64_t pattern_hash = 0;  // of C's host row pointers and column indices (key of the per-context cache)
    int64_t nnz = 0;
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

// max of a 64-bit key over the wavefront, returned to every lane: DPP steps inside each row of 16 lanes (a ds_bpermute
// butterfly costs ~100 cycles per step on the pivot search's critical path), then the four row maxima through SGPRs
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long v, unsigned long long o) { return o > v ? o : v; }
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_mov_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v & 0xFFFFFFFFull), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = dpp_max_step(v, dpp_mov_u64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = dpp_max_step(v, dpp_mov_u64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = dpp_max_step(v, dpp_mov_u64<0x141>(v));  // row_half_mirror
    v = dpp_max_step(v, dpp_mov_u64<0x140>(v));  // row_mirror: every lane of a row now holds the row's max
    unsigned long long best = 0ull;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), 16 * row);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * row);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        best = o > best ? o : best;
    }
    return best;
}

// ---- blocked form with look-ahead: ONE launch per panel of w <= W pivot columns -------------------------------------------
// Eliminating a panel leaves W (m x w) in its columns; the other columns then need the accumulated rank-w update
//     A[:, J] <- (P A)[:, J] with the pivot rows zeroed  +  W * (P A)[K, J]          (K = the panel's pivot rows),
// the same elimination in the same order as the unblocked form.  Launch p of a block does both halves at once:
//   * workgroup 0 applies the update of panel p to the columns of panel p + 1 ONLY and then eliminates panel p + 1
//     (the critical path: the next launch needs its W and its row interchanges),
//   * workgroup 1 + t applies the update of panel p to column tile t (W columns, all m rows); the tiles of panels p and
//     p + 1 are skipped.
// A column tile is owned by one workgroup, so the few rows touched by the interchanges are staged in LDS by their only
// reader and nothing has to be ordered between workgroups: m / w + 1 launches per block instead of 2 m (a dependent
// launch costs ~9 us on this part; measured per panel: 3 launches 54 us -> 1 launch).
//
// Registers hold the tile: thread t owns rows t, t + NT, ... (RPT of them) x W columns; (NT, RPT, W) is picked per block
// size so that nothing spills (a kernel with scratch also pays a per-dispatch scratch set-up).  Per pivot the panel
// workgroup does a shuffle/LDS arg-max, the two owner threads publish rows k and p through LDS, and every thread updates
// its own rows from the broadcast pivot row: three barriers per pivot.
//
// Interchange lists (perm): [0] = count of touched rows, [1 + q] = row, [33 + q] = slot of the row its new content comes
// from, [50 + j] = that slot for pivot row k0 + j; written by workgroup 0 for panel p + 1 into pnext while everybody reads pprev (two buffers, alternating).
constexpr int kPanelW = 8;
template <typename T, int NT, int RPT, int W>
__global__ __launch_bounds__(NT) void gj_fused_kernel(T* __restrict__ a, int32_t ld, int32_t m, int32_t kprev, int32_t wprev, int32_t knext,
                                                      int32_t wnext, int32_t* __restrict__ ipiv, const int32_t* __restrict__ pprev,
                                                      int32_t* __restrict__ pnext, int32_t* __restrict__ flag, double tiny2) {
    __shared__ T Zs[2 * kPanelW][W];  // staged rows of this tile (update phase)
    __shared__ int32_t zrow[2 * kPanelW], zsrc[2 * kPanelW], yslot[kPanelW];
    __shared__ T rowk[2][kPanelW], rowp[2][kPanelW];  // double-buffered: pivot jj + 1 publishes while stragglers read jj
    __shared__ unsigned long long skey[kPanelW];
    __shared__ int32_t spiv[kPanelW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool panel_wg = blockIdx.x == 0;
    if (tid < kPanelW) skey[tid] = 0ull;
    int32_t c0;
    if (panel_wg) {
        if (knext < 0) return;
        c0 = knext;
    } else {
        c0 = ((int32_t)blockIdx.x - 1) * W;
        if (kprev < 0 || c0 == kprev || c0 == knext) return;
    }
    const int32_t cw = min(W, m - c0);
    T r[RPT][W];
    if (kprev >= 0) {
        // ---- rank-wprev update of this tile ----
        // (the list loads are independent of each other: one memory latency, then one more for the staged rows; the
        // look-ahead workgroup also issues the loads of its own rows before waiting for either)
        const int32_t cnt = pprev[0];
        if (tid < 2 * kPanelW) {
            zrow[tid] = pprev[1 + tid];
            zsrc[tid] = pprev[33 + tid];
        }
        if (tid >= 64 && tid < 64 + kPanelW) yslot[tid - 64] = pprev[50 + (tid - 64)];
        if (tid >= 128 && tid < 128 + 2 * kPanelW * W) {
            const int q = (tid - 128) / W, c = (tid - 128) % W;
            const int32_t zr = min(max(pprev[1 + q], 0), m - 1);  // entries beyond cnt are stale: keep the address in range
            if (q < cnt && c < cw) Zs[q][c] = a[(size_t)zr * ld + c0 + c];
        }
        // (its W values too while RPT * W is small enough to keep them in registers next to the tile)
        constexpr bool kPrefetchW = RPT * W <= 8;
        T wv[kPrefetchW ? RPT : 1][W];
        if (panel_wg) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int32_t i = min(tid + NT * q, m - 1);
                const T* ai = a + (size_t)i * ld;
#pragma unroll
                for (int c = 0; c < W; ++c) r[q][c] = ai[c0 + c];  // ld - B = 16 columns of padding keep c0 + c in range
                if constexpr (kPrefetchW) {
#pragma unroll
                    for (int j = 0; j < W; ++j) wv[q][j] = ai[kprev + j];
                }
            }
        }
        __syncthreads();
        if (!panel_wg) {
            // update-only tile: W consecutive lanes walk one row segment (coalesced), NT / W rows per step.  Every row is
            // first treated as untouched; the <= 2 wprev rows the interchanges touch are redone after a barrier.
            constexpr int RS = NT / W;
            const int c = tid % W;
            const bool live = c < cw;
            T y[W];
#pragma unroll
            for (int j = 0; j < W; ++j) y[j] = (j < wprev) ? Zs[yslot[j]][c] : scalar_traits<T>::zero();
            // rows in flight per thread: the panel path fixes the register allocation of the whole kernel, so the fewer
            // waves fit a SIMD the more loads each must keep in flight (256 threads: 1 wave per SIMD, 512: 2, 1024: 4)
            constexpr int kRowsInFlight = NT <= 256 ? 12 : NT <= 512 ? 8 : 4;
#pragma unroll kRowsInFlight
            for (int32_t i = tid / W; i < m; i += RS) {
                T* ai = a + (size_t)i * ld;
                if (live) {
                    T acc = ai[c0 + c];
#pragma unroll
                    for (int j = 0; j < W; ++j)
                        if (j < wprev) fma_acc(acc, ai[kprev + j], y[j]);
                    ai[c0 + c] = acc;
                }
            }
            __syncthreads();
            if (tid < cnt * W && live) {
                const int q = tid / W;
                const int32_t i = zrow[q];
                T* ai = a + (size_t)i * ld;
                T acc = (i >= kprev && i < kprev + wprev) ? scalar_traits<T>::zero() : Zs[zsrc[q]][c];
#pragma unroll
                for (int j = 0; j < W; ++j)
                    if (j < wprev) fma_acc(acc, ai[kprev + j], y[j]);
                ai[c0 + c] = acc;
            }
            return;
        }
        // look-ahead tile (the next panel's columns): thread per row, the rows stay in registers for the elimination
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            int32_t sl = -1;
#pragma unroll
            for (int t = 0; t < 2 * kPanelW; ++t)
                if (t < cnt && zrow[t] == i) sl = zsrc[t];
            const bool pivot_row = i >= kprev && i < kprev + wprev;
#pragma unroll
            for (int c = 0; c < W; ++c) {
                if (pivot_row || c >= cw || i >= m) r[q][c] = scalar_traits<T>::zero();
                else if (sl >= 0) r[q][c] = Zs[sl][c];
            }
            if constexpr (kPrefetchW) {
                if (i < m) {
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        if (j < wprev) {
                            const int ys = yslot[j];
#pragma unroll
                            for (int c = 0; c < W; ++c) fma_acc(r[q][c], wv[q][j], Zs[ys][c]);
                        }
                    }
                }
            }
        }
        if constexpr (!kPrefetchW) {
            // many rows per thread: one column of W at a time for all of them (RPT independent loads in flight instead of
            // a load-use chain per row), the staged pivot row read from LDS once per column instead of once per row
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (j < wprev) {
                    T wj[RPT];
#pragma unroll
                    for (int q = 0; q < RPT; ++q) wj[q] = a[(size_t)min(tid + NT * q, m - 1) * ld + kprev + j];
                    const int ys = yslot[j];
                    T yj[W];
#pragma unroll
                    for (int c = 0; c < W; ++c) yj[c] = Zs[ys][c];
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        if (tid + NT * q < m) {
#pragma unroll
                            for (int c = 0; c < W; ++c) fma_acc(r[q][c], wj[q], yj[c]);
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
#pragma unroll
            for (int c = 0; c < W; ++c) r[q][c] = (i < m && c < cw) ? a[(size_t)i * ld + c0 + c] : scalar_traits<T>::zero();
        }
        __syncthreads();  // skey is initialised
    }
    // ---- workgroup 0: eliminate the panel [knext, knext + wnext) held in r ----
    // Pivot search: one 64-bit key per candidate row, |a|^2 with its low 13 mantissa bits replaced by (4096 - row), so that
    // a single unsigned max picks the largest magnitude and, among magnitudes equal to 2^-39 relative, the lowest row;
    // wave shuffle-reduce, then one LDS atomic max per wave: two barriers per pivot.  The pivot loop is fully unrolled so
    // that "column jj of my row" is a register, not a select chain.
    const int32_t k0 = knext, w = wnext;
#pragma unroll
    for (int jj = 0; jj < W; ++jj) {
        if (jj >= w) break;
        const int32_t k = k0 + jj;
        unsigned long long key = 0ull;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            if (i >= k && i < m) {
                const double mag = s_abs2(r[q][jj]);
                const unsigned long long kq = ((unsigned long long)__double_as_longlong(mag) & ~0x1FFFull) | (unsigned long long)(4096 - i);
                key = kq > key ? kq : key;
            }
        }
        key = wave_max_u64(key);
        if (lane == 0) atomicMax(&skey[jj], key);
        __syncthreads();
        key = skey[jj];
        const int32_t p = 4096 - (int32_t)(key & 0x1FFFull);
        if (tid == 0) {
            ipiv[k] = p;
            if (!(__longlong_as_double((long long)(key & ~0x1FFFull)) > tiny2)) atomicCAS(&flag[1], 0, k + 1);
        }
        if (lane == 0 && wave == 0) spiv[jj] = p;
        // the owner of row k publishes it; the owner of row p publishes the SCALED pivot row (one complex division per
        // pivot instead of one per thread)
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            if (i == k) {
#pragma unroll
                for (int j = 0; j < W; ++j) rowk[jj & 1][j] = r[q][j];
            }
            if (i == p) {
                T piv = r[q][jj];
                if (s_abs2(piv) == 0.0) s_from(piv, 1.0, 0.0);
                const T pinv = s_inv(piv);
#pragma unroll
                for (int j = 0; j < W; ++j) rowp[jj & 1][j] = (j == jj) ? pinv : s_mul(pinv, r[q][j]);
            }
        }
        __syncthreads();
        T prow[W];
#pragma unroll
        for (int j = 0; j < W; ++j) prow[j] = rowp[jj & 1][j];
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
                for (int j = 0; j < W; ++j) r[q][j] = rowk[jj & 1][j];
            }
            const T fm = r[q][jj];
            if (s_abs2(fm) == 0.0) continue;
            const T nfm = s_sub(scalar_traits<T>::zero(), fm);
            r[q][jj] = scalar_traits<T>::zero();
#pragma unroll
            for (int j = 0; j < W; ++j) fma_acc(r[q][j], nfm, prow[j]);
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
    // the panel's interchanges as lists for the next launch, built by wave 0 in registers (a single thread walking LDS
    // lists cost ~25 us of serial LDS latency per panel): lane l < 8 stands for row k0 + l, lane 8 + l for pivot row
    // spiv[l]; every lane tracks which original row's content its row holds while the w swaps are replayed.
    if (wave == 0) {
        int32_t row = -1;
        if (lane < kPanelW && lane < w) row = k0 + lane;
        else if (lane >= kPanelW && lane < kPanelW + w) row = spiv[lane - kPanelW];
        int32_t content = row;
#pragma unroll
        for (int jj = 0; jj < W; ++jj) {
            if (jj < w) {
                const int32_t ra = k0 + jj, rb = __builtin_amdgcn_readlane(row, kPanelW + jj);
                const int32_t ca = __builtin_amdgcn_readlane(content, jj), cb = __builtin_amdgcn_readlane(content, kPanelW + jj);
                if (row == ra) content = cb;
                else if (row == rb) content = ca;
            }
        }
        bool primary = row >= 0;
#pragma unroll
        for (int t = 0; t < 2 * kPanelW; ++t) {
            const int32_t rt = __builtin_amdgcn_readlane(row, t);
            if (t < lane && rt == row) primary = false;
        }
        const unsigned long long pm = __ballot(primary);
        const int32_t myslot = __popcll(pm & ((1ull << lane) - 1ull));
        int32_t qs = 0;
#pragma unroll
        for (int t = 0; t < 2 * kPanelW; ++t) {
            const int32_t rt = __builtin_amdgcn_readlane(row, t), stt = __builtin_amdgcn_readlane(myslot, t);
            if (((pm >> t) & 1ull) && rt == content) qs = stt;
        }
        if (lane == 0) pnext[0] = __popcll(pm);
        if (primary) {
            pnext[1 + myslot] = row;
            pnext[33 + myslot] = qs;
        }
        if (lane < w) pnext[50 + lane] = qs;  // where pivot row k0 + lane's new content is staged (rows k0 + l are primary)
    }
}

// ---- blocks of more than 2048 rows: the update of the other columns as its own, row-tiled launch ------------------------
// gj_fused_kernel's column-tile workgroups inherit the register allocation of its panel path (466 VGPRs when a thread
// holds 12 rows: one wave per SIMD) and read 128 B per row: measured 0.75 TB/s on 151 MB blocks (S500k).  Here the
// panel workgroup runs alone (grid = 1) and the rank-w update of all other columns streams whole rows: the <= 16 rows
// the interchanges touch are staged first (other workgroups overwrite them), then a workgroup owns kUpdRows rows and
// walks the columns.
template <typename T>
__global__ __launch_bounds__(256) void gj_stage_kernel(const T* __restrict__ a, int32_t ld, int32_t m, const int32_t* __restrict__ perm,
                                                       T* __restrict__ Z) {
    const int32_t q = blockIdx.y;
    if (q >= perm[0]) return;
    const int32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < m) Z[(size_t)q * m + c] = a[(size_t)perm[1 + q] * ld + c];
}

// a[i, c] = base(i, c) + sum_j W[i, j] Y[j, c] for every column c outside [k0, k0 + w) and [kskip, kskip + wskip) (the
// look-ahead tile, which the panel workgroup updates itself): Y[j, :] is the staged row perm[50 + j], base is 0 for the
// pivot rows, the staged interchanged content for the other touched rows, a[i, c] itself elsewhere.
constexpr int kUpdRows = 16;
constexpr int kUpdCols = 512;  // columns per workgroup (grid.y): enough workgroups to fill 256 CUs at m = 3072
template <typename T>
__global__ __launch_bounds__(256) void gj_panel_update_kernel(T* __restrict__ a, int32_t ld, int32_t m, int32_t k0, int32_t w, int32_t kskip,
                                                              int32_t wskip, const int32_t* __restrict__ perm, const T* __restrict__ Z) {
    __shared__ T Ws[kUpdRows][kPanelW];
    __shared__ int32_t yslot[kPanelW];   // staged row holding pivot row k0 + j
    __shared__ int32_t rslot[kUpdRows];  // staged row holding the new content of my row r, or -1
    const int32_t i0 = blockIdx.x * kUpdRows;
    if (threadIdx.x < kUpdRows * kPanelW) {
        const int r = threadIdx.x / kPanelW, j = threadIdx.x % kPanelW;
        Ws[r][j] = (i0 + r < m && j < w) ? a[(size_t)(i0 + r) * ld + k0 + j] : scalar_traits<T>::zero();
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kPanelW) {
        const int j = threadIdx.x - 64;
        yslot[j] = (j < w) ? perm[50 + j] : 0;  // W is zero beyond w; slot 0 is always valid
    }
    if (threadIdx.x >= 128 && threadIdx.x < 128 + kUpdRows) {
        const int32_t na = perm[0], row = i0 + ((int32_t)threadIdx.x - 128);
        int32_t sl = -1;
        for (int32_t q = 0; q < na; ++q)
            if (perm[1 + q] == row) sl = perm[33 + q];
        rslot[threadIdx.x - 128] = sl;
    }
    __syncthreads();
    // y (the w pivot-row values of my column) stays in registers for all kUpdRows rows, so the staged rows are re-read
    // once per kUpdRows rows of the block; W comes from LDS two rows at a time (unrolling further would hoist the whole
    // W tile into registers: 312 VGPRs at 8 rows)
    const int32_t cend = min(m, ((int32_t)blockIdx.y + 1) * kUpdCols);
#pragma unroll 1
    for (int32_t c = blockIdx.y * kUpdCols + threadIdx.x; c < cend; c += 256) {
        if ((c >= k0 && c < k0 + w) || (c >= kskip && c < kskip + wskip)) continue;
        T y[kPanelW];
#pragma unroll
        for (int j = 0; j < kPanelW; ++j) y[j] = Z[(size_t)yslot[j] * m + c];
#pragma unroll 4
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

// Undo the row interchanges on the columns of the inverse: every row r needs  for k = m-1 .. 0: swap(r[k], r[ipiv[k]]).
// The swaps are the same for all rows, so the column each entry ends up in is computed once (thread j follows the entry
// that starts in column j through the m swaps, ipiv broadcast from LDS), then every row is permuted in one pass.
__global__ __launch_bounds__(256) void gj_colperm_kernel(int32_t m, const int32_t* __restrict__ ipiv, int32_t* __restrict__ pos) {
    extern __shared__ int32_t spv[];
    for (int32_t k = threadIdx.x; k < m; k += 256) spv[k] = ipiv[k];
    __syncthreads();
    const int32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    int32_t at = j;
    for (int32_t k = m - 1; k >= 0; --k) {
        const int32_t p = spv[k];
        at = (at == k) ? p : (at == p ? k : at);
    }
    pos[j] = at;
}

template <typename T>
__global__ __launch_bounds__(256) void gj_unpivot_kernel(T* __restrict__ a, int32_t ld, int32_t m, const int32_t* __restrict__ pos) {
    T* row = a + (size_t)blockIdx.x * ld;
    T v[16];  // m <= 4096
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int32_t j = threadIdx.x + 256 * q;
        if (j < m) v[q] = row[j];
    }
    __syncthreads();  // the whole row is in registers before any of it is overwritten
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int32_t j = threadIdx.x + 256 * q;
        if (j < m) row[pos[j]] = v[q];
    }
}

// (blocks of more than 4096 rows: thread per row, swaps replayed in place)
template <typename T>
__global__ void gj_unpivot_serial_kernel(T* __restrict__ a, int32_t ld, int32_t m, const int32_t* __restrict__ ipiv) {
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
// The twisted sweeps advance two independent chains (one from each end of the block tridiagonal matrix).  Every launch
// serves one block of EACH chain -- "A" in the first workgroups, "B" in the rest -- so a sweep costs half the dependent
// launches and each launch moves twice the bytes; a chain that has run out passes an empty range.
struct BluRange {
    int32_t bs, be;  // rows [bs, be) of one diagonal block
    int32_t mode;    // sparse kernel: 1 = entries left of the block, 2 = right of it, 3 = both; +4 = window form
    int32_t cs, ce;  // window form (substitution sweep): only rows [cs, ce) of the block couple to the neighbour, so the
                     // sparse kernel forms u = C_{b,b+-1} x there and the dense kernel needs only those COLUMNS of Sinv_b
};

// rows the sparse kernel visits for one range
__host__ __device__ inline int32_t blu_rows(const BluRange& r) { return (r.mode & 4) ? r.ce - r.cs : r.be - r.bs; }

// out[r] = rhs[r] - sum over the selected off-block entries of row r of val * x[col]  (window form: out[r] = + the sum,
// rows [cs, ce) only); 16 lanes per row
template <typename MT, typename VT>
__global__ __launch_bounds__(256) void blu_sparse_kernel(BluRange ra, BluRange rb, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                         const int32_t* __restrict__ lsplit, const int32_t* __restrict__ usplit,
                                                         const MT* __restrict__ val, const VT* __restrict__ rhs, const VT* __restrict__ x,
                                                         VT* __restrict__ out) {
    const int32_t ga = ((blu_rows(ra)) * 16 + 255) / 256;
    const bool second = (int32_t)blockIdx.x >= ga;
    const BluRange rg = second ? rb : ra;
    const bool window = (rg.mode & 4) != 0;
    const int32_t gid = ((int32_t)blockIdx.x - (second ? ga : 0)) * 256 + threadIdx.x;
    const int lane = gid & 15;
    const int32_t r = (window ? rg.cs : rg.bs) + (gid >> 4);
    if (r >= (window ? rg.ce : rg.be)) return;
    VT acc = scalar_traits<VT>::zero();
    if (rg.mode & 1)
        for (int32_t p = rp[r] + lane; p < lsplit[r]; p += 16) fma_acc(acc, val[p], x[ci[p]]);
    if (rg.mode & 2)
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
    if (lane == 0) out[r] = window ? acc : s_sub(rhs[r], acc);
}

// out[r] = sum_s Sinv[r, s] in[s] over the whole diagonal block; one wavefront per row, 8 loads in flight per lane
template <typename MT, typename VT>
__global__ __launch_bounds__(256) void blu_dense_kernel(BluRange ra, BluRange rb, int32_t ld, const MT* __restrict__ sinv,
                                                        const VT* __restrict__ in, const VT* __restrict__ add, VT* __restrict__ out) {
    const int32_t ga = (ra.be - ra.bs + 3) / 4;
    const bool second = (int32_t)blockIdx.x >= ga;
    const BluRange rg = second ? rb : ra;
    const int32_t bs = rg.bs, be = rg.be;
    const bool window = (rg.mode & 4) != 0;
    const int32_t lo = window ? rg.cs : bs, hi = window ? rg.ce : be;  // columns of Sinv_b that meet a non-zero of `in`
    const int lane = threadIdx.x & 63;
    const int32_t r = bs + ((int32_t)blockIdx.x - (second ? ga : 0)) * 4 + (threadIdx.x >> 6);
    if (r >= be) return;
    const MT* row = sinv + (size_t)r * ld - bs;
    VT acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = scalar_traits<VT>::zero();
    int32_t s = lo + lane;
    for (; s + 7 * 64 < hi; s += 8 * 64) {
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
    for (; s < hi; s += 64) fma_acc(acc[0], row[s], in[s]);
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
    if (lane == 0) out[r] = window ? s_sub(add[r], v) : v;
}

// G_b[r, t] = sum_s Sinv_b[r, s] C[bs + s, col0 + t]  for the columns [col0, col0 + wcols) of a neighbouring block that the
// sparse coupling block touches (left neighbour: col0 + wcols = bs; right neighbour: col0 = be).  One workgroup per row
// r of the block: the row of Sinv_b is staged in LDS, a thread walks the CSC column col0 + t and keeps the entries
// whose row lies in the block.
template <typename T>
__global__ __launch_bounds__(256) void blu_absorb_kernel(int32_t bs, int32_t be, int32_t ld, int32_t col0, int32_t wcols, int32_t ldo,
                                                         const T* __restrict__ val, const int32_t* __restrict__ cptr,
                                                         const int32_t* __restrict__ crow, const int32_t* __restrict__ cpos,
                                                         const T* __restrict__ sinv, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* srow = (T*)dyn;
    const int32_t r = bs + blockIdx.x, m = be - bs;
    const T* src = sinv + (size_t)r * ld;
    for (int32_t k = threadIdx.x; k < m; k += 256) srow[k] = src[k];
    __syncthreads();
    T* dst = out + (size_t)r * ldo;
    for (int32_t t = threadIdx.x; t < wcols; t += 256) {
        const int32_t q0 = cptr[col0 + t], q1 = cptr[col0 + t + 1];
        T acc = scalar_traits<T>::zero();
        for (int32_t q = q0; q < q1; ++q) {
            const int32_t k = crow[q];
            if (k >= bs && k < be) fma_acc(acc, srow[k - bs], val[cpos[q]]);
        }
        dst[t] = acc;
    }
}

// One sweep step on the absorbed form, one block of each chain per launch (wavefront per row):
//   out[r] = add[r] + [mode & 1] sum_s Sinv[r, s] vfull[bs + s] - [mode & 2] sum_t G[r, t] x[gc0 + t] - [mode & 4] sum_t K[r, t] x[be + t]
struct BluStep {
    int32_t bs, be, mode;
    int32_t gc0, gw;  // columns [gc0, gc0 + gw) of the left neighbour (G)
    int32_t kw;       // columns [be, be + kw) of the right neighbour (K)
};

template <typename MT, typename VT>
__device__ __forceinline__ void wave_dot_sub(VT (&acc)[4], const MT* __restrict__ row, const VT* __restrict__ x, int32_t cnt, int lane, bool subtract) {
    VT part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k] = scalar_traits<VT>::zero();
    int32_t s = lane;
    for (; s + 7 * 64 < cnt; s += 8 * 64) {
        MT a[8];
        VT tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = row[s + k * 64];
            tv[k] = x[s + k * 64];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) fma_acc(part[k & 3], a[k], tv[k]);
    }
    for (; s < cnt; s += 64) fma_acc(part[0], row[s], x[s]);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = subtract ? s_sub(acc[k], part[k]) : s_add(acc[k], part[k]);
}

template <typename MT, typename VT>
__global__ __launch_bounds__(256) void blu_step_kernel(BluStep ra, BluStep rb, int32_t ld, int32_t ldg, int32_t ldk, const MT* __restrict__ sinv,
                                                       const MT* __restrict__ G, const MT* __restrict__ K, const VT* __restrict__ vfull,
                                                       const VT* __restrict__ x, const VT* __restrict__ add, VT* __restrict__ out) {
    const int32_t ga = (ra.be - ra.bs + 3) / 4;
    const bool second = (int32_t)blockIdx.x >= ga;
    const BluStep rg = second ? rb : ra;
    const int lane = threadIdx.x & 63;
    const int32_t r = rg.bs + ((int32_t)blockIdx.x - (second ? ga : 0)) * 4 + (threadIdx.x >> 6);
    if (r >= rg.be) return;
    VT acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = scalar_traits<VT>::zero();
    if (rg.mode & 1) wave_dot_sub<MT, VT>(acc, sinv + (size_t)r * ld, vfull + rg.bs, rg.be - rg.bs, lane, false);
    if (rg.mode & 2) wave_dot_sub<MT, VT>(acc, G + (size_t)r * ldg, x + rg.gc0, rg.gw, lane, true);
    if (rg.mode & 4) wave_dot_sub<MT, VT>(acc, K + (size_t)r * ldk, x + rg.be, rg.kw, lane, true);
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
    if (lane == 0) out[r] = add ? s_add(add[r], v) : v;
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
        LSA_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blu_absorb_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // panel width of the blocked Gauss-Jordan = the W of the gj_fused_kernel instance used below: 8 columns while a
    // thread holds at most 4 rows, 4 columns up to 8 rows (blocks of 4096 rows); LSA_GJ_PANEL=1 selects the unblocked
    // form (two launches per pivot), which is also what larger blocks get
    int32_t panel_w = B <= 3072 ? 8 : B <= 4096 ? 4 : 1;
    if (const char* e = getenv("LSA_GJ_PANEL"))
        if (atoi(e) == 1) panel_w = 1;
    // blocks of at least split_min rows use the split form (lone panel workgroup beside a row-tiled bulk update)
    const int32_t split_min = getenv("LSA_GJ_SPLIT_MIN") ? atoi(getenv("LSA_GJ_SPLIT_MIN")) : 2049;
    auto factor_block = [&](hipStream_t st, int chain, int32_t b, bool corr_left, bool corr_right) {
        const int32_t bs = b * B, be = std::min(n, bs + B), m = be - bs;
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
        int32_t* ipiv = f->ipiv[chain];
        int32_t* perm = ipiv + B;  // interchange lists of the current panel (80 ints behind the pivot indices)
        T* ws = (T*)f->colbuf[chain];
        if (panel_w >= 2) {
            // launch 0 eliminates panel 0; launch 1 + p updates with panel p and eliminates panel p + 1
            const int32_t np = (m + panel_w - 1) / panel_w;
            for (int32_t pn = -1; pn < np; ++pn) {
                const int32_t kprev = pn >= 0 ? pn * panel_w : -1, wprev = pn >= 0 ? std::min(panel_w, m - kprev) : 0;
                const int32_t knext = pn + 1 < np ? (pn + 1) * panel_w : -1, wnext = knext >= 0 ? std::min(panel_w, m - knext) : 0;
                const int32_t* pprev = perm + 64 * (pn & 1);
                int32_t* pnext = perm + 64 * ((pn + 1) & 1);
                const dim3 grid(pn >= 0 ? 1 + np : 1);
                if (B <= 1024)
                    hipLaunchKernelGGL((gj_fused_kernel<T, 1024, 1, 8>), grid, dim3(1024), 0, st, S, ld, m, kprev, wprev, knext, wnext, ipiv, pprev, pnext,
                                       f->flag, tiny2);
                else if (B <= 2048 && B < split_min)
                    hipLaunchKernelGGL((gj_fused_kernel<T, 512, 4, 8>), grid, dim3(512), 0, st, S, ld, m, kprev, wprev, knext, wnext, ipiv, pprev, pnext,
                                       f->flag, tiny2);
                else {
                    // wide blocks: the bulk update of panel p (row-tiled, full rows) on the chain's stream, the panel
                    // workgroup of p + 1 beside it on the side stream.  Panel p + 1 needs panel p (same stream) and the
                    // bulk update of p - 1 (ev_update as recorded so far); the bulk update of p needs panel p (ev_panel as
                    // recorded so far).  They touch disjoint columns and alternate interchange-list buffers.
                    hipStream_t sd = f->side[chain];
                    if (pn >= 0) {
                        (void)hipStreamWaitEvent(st, f->ev_panel[chain], 0);
                        hipLaunchKernelGGL((gj_stage_kernel<T>), dim3((m + 255) / 256, 2 * kPanelW), dim3(256), 0, st, (const T*)S, ld, m, pprev, ws);
                        hipLaunchKernelGGL((gj_panel_update_kernel<T>), dim3((m + kUpdRows - 1) / kUpdRows, (m + kUpdCols - 1) / kUpdCols), dim3(256), 0,
                                           st, S, ld, m, kprev, wprev, knext, wnext, pprev, (const T*)ws);
                    } else {
                        (void)hipEventRecord(f->ev_update[chain], st);  // scatter and Schur corrections of this block are enqueued
                    }
                    if (knext >= 0) {
                        (void)hipStreamWaitEvent(sd, f->ev_update[chain], 0);
                        if (B <= 2048)
                            hipLaunchKernelGGL((gj_fused_kernel<T, 512, 4, 8>), dim3(1), dim3(512), 0, sd, S, ld, m, kprev, wprev, knext, wnext, ipiv, pprev,
                                               pnext, f->flag, tiny2);
                        else if (B <= 3072)
                            hipLaunchKernelGGL((gj_fused_kernel<T, 256, 12, 8>), dim3(1), dim3(256), 0, sd, S, ld, m, kprev, wprev, knext, wnext, ipiv, pprev,
                                               pnext, f->flag, tiny2);
                        else
                            hipLaunchKernelGGL((gj_fused_kernel<T, 256, 16, 4>), dim3(1), dim3(256), 0, sd, S, ld, m, kprev, wprev, knext, wnext, ipiv, pprev,
                                               pnext, f->flag, tiny2);
                        (void)hipEventRecord(f->ev_panel[chain], sd);
                    }
                    if (pn >= 0) (void)hipEventRecord(f->ev_update[chain], st);
                }
            }
        } else {
            for (int32_t k = 0; k < m; ++k) {
                hipLaunchKernelGGL((gj_pivot_kernel<T>), dim3(1), dim3(1024), 0, st, S, ld, m, k, ipiv, ws, f->flag, tiny2);
                hipLaunchKernelGGL((gj_update_kernel<T>), dim3((m + 3) / 4), dim3(256), 0, st, S, ld, m, k, (const T*)ws);
            }
        }
        if (m <= 4096) {
            int32_t* pos = perm + 128;  // behind the two interchange-list buffers
            hipLaunchKernelGGL(gj_colperm_kernel, dim3((m + 255) / 256), dim3(256), (size_t)m * sizeof(int32_t), st, m, (const int32_t*)ipiv, pos);
            hipLaunchKernelGGL((gj_unpivot_kernel<T>), dim3(m), dim3(256), 0, st, S, ld, m, (const int32_t*)pos);
        } else {
            hipLaunchKernelGGL((gj_unpivot_serial_kernel<T>), dim3((m + 255) / 256), dim3(256), 0, st, S, ld, m, (const int32_t*)ipiv);
        }
        if (f->G) {  // absorb the couplings to both neighbours: G_b = Sinv_b C_{b,b-1}, K_b = Sinv_b C_{b,b+1}
            const int32_t gw = bs - f->gcol0[b], kw = f->kcol1[b] - be;
            if (gw > 0)
                hipLaunchKernelGGL((blu_absorb_kernel<T>), dim3(m), dim3(256), lds, st, bs, be, ld, f->gcol0[b], gw, f->ldg, (const T*)C->val, f->cptr,
                                   f->crow, f->cpos, (const T*)f->sinv, (T*)f->G);
            if (kw > 0)
                hipLaunchKernelGGL((blu_absorb_kernel<T>), dim3(m), dim3(256), lds, st, bs, be, ld, be, kw, f->ldk, (const T*)C->val, f->cptr, f->crow,
                                   f->cpos, (const T*)f->sinv, (T*)f->K);
        }
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
    hipStream_t st = ctx->stream;
    if (f->G) {
        // absorbed form: every sweep step is one dense launch (both chains):
        //   elimination   z_b = Sinv_b v_b - G_b z_{b-1}   (chain from the top)     z_b = Sinv_b v_b - K_b z_{b+1}  (from the bottom)
        //   middle        x_m = Sinv_m v_m - G_m z_{m-1} - K_m z_{m+1}
        //   substitution  x_b = z_b - K_b x_{b+1}          (top chain)              x_b = z_b - G_b x_{b-1}         (bottom chain)
        auto stepd = [&](int32_t b, int32_t mode) {
            if (b < 0 || b >= nb) return BluStep{0, 0, 0, 0, 0, 0};
            const int32_t bs = b * B, be = std::min(n, bs + B);
            if (b == 0) mode &= ~2;
            if (b == nb - 1) mode &= ~4;
            return BluStep{bs, be, mode, f->gcol0[b], bs - f->gcol0[b], f->kcol1[b] - be};
        };
        auto launch = [&](BluStep ra, BluStep rb, const VT* vfull, const VT* xin, const VT* add, VT* out) {
            const int32_t gd = (ra.be - ra.bs + 3) / 4 + (rb.be - rb.bs + 3) / 4;
            if (gd == 0) return;
            hipLaunchKernelGGL((blu_step_kernel<MT, VT>), dim3(gd), dim3(256), 0, st, ra, rb, f->ld, f->ldg, f->ldk, (const MT*)f->sinv, (const MT*)f->G,
                               (const MT*)f->K, vfull, xin, add, out);
        };
        for (int32_t k = 0; k < std::max(mid, nb - 1 - mid); ++k) {
            const int32_t bt = k, bb = nb - 1 - k;
            launch(stepd(bt < mid ? bt : -1, 1 | 2), stepd(bb > mid ? bb : -1, 1 | 4), v, (const VT*)z, nullptr, z);
        }
        launch(stepd(mid, 1 | 2 | 4), stepd(-1, 0), v, (const VT*)z, nullptr, x);
        for (int32_t k = 1; k <= std::max(mid, nb - 1 - mid); ++k) launch(stepd(mid - k, 4), stepd(mid + k, 2), nullptr, (const VT*)x, (const VT*)z, x);
        return LSA_OK;
    }
    auto range = [&](int32_t b, int32_t mode) {
        if (b < 0 || b >= nb) return BluRange{0, 0, 0, 0, 0};
        return BluRange{b * B, std::min(n, b * B + B), mode, 0, 0};
    };
    // substitution sweep: rows of block b that couple to the block it is solved after
    auto window = [&](int32_t b, int32_t mode) {
        if (b < 0 || b >= nb) return BluRange{0, 0, 0, 0, 0};
        const int32_t bs = b * B, be = std::min(n, bs + B);
        if (mode == 2) return BluRange{bs, be, 2 | 4, f->uwin_lo[b], be};  // entries right of the block: the last rows
        return BluRange{bs, be, 1 | 4, bs, f->lwin_hi[b]};                 // entries left of the block: the first rows
    };
    auto step = [&](BluRange ra, BluRange rb, const VT* rhs, const VT* xin, VT* tmp, const VT* add, VT* out) {
        const int32_t gs = (blu_rows(ra) * 16 + 255) / 256 + (blu_rows(rb) * 16 + 255) / 256;
        const int32_t gd = (ra.be - ra.bs + 3) / 4 + (rb.be - rb.bs + 3) / 4;
        if (gd == 0) return;
        if (gs > 0)
            hipLaunchKernelGGL((blu_sparse_kernel<MT, VT>), dim3(gs), dim3(256), 0, st, ra, rb, C->rp, C->ci, f->lsplit, f->usplit, (const MT*)C->val,
                               rhs, xin, tmp);
        hipLaunchKernelGGL((blu_dense_kernel<MT, VT>), dim3(gd), dim3(256), 0, st, ra, rb, f->ld, (const MT*)f->sinv, (const VT*)tmp, add, out);
    };
    // elimination towards the middle, both chains per launch:  y_b = v_b - C_{b,b-+1} z_{b-+1},  z_b = Sinv_b y_b
    for (int32_t k = 0; k < std::max(mid, nb - 1 - mid); ++k) {
        const int32_t bt = k, bb = nb - 1 - k;
        step(range(bt < mid ? bt : -1, 1), range(bb > mid ? bb : -1, 2), v, (const VT*)z, y, nullptr, z);
    }
    // middle block: x_m = Sinv_m (v_m - C_{m,m-1} z_{m-1} - C_{m,m+1} z_{m+1})
    step(range(mid, 3), range(-1, 0), v, (const VT*)z, t, nullptr, x);
    // substitution outwards, both chains per launch:  x_b = z_b - Sinv_b (C_{b,b+-1} x_{b+-1});  the product is non-zero
    // only on the rows within the bandwidth of the neighbour, so only those columns of Sinv_b are read (~70 % at S30k)
    for (int32_t k = 1; k <= std::max(mid, nb - 1 - mid); ++k) step(window(mid - k, 2), window(mid + k, 1), nullptr, (const VT*)x, t, (const VT*)z, x);
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

static void blu_free(lsa_blu* f) {
    if (!f) return;
    if (f->ctx && f->ctx->stream) (void)hipStreamSynchronize(f->ctx->stream);
    for (int vd = 0; vd < 2; ++vd) {
        if (f->graph[vd]) (void)hipGraphExecDestroy((hipGraphExec_t)f->graph[vd]);
        for (void* p : {f->t[vd], f->y[vd], f->z[vd], f->in[vd], f->out[vd]})
            if (p) (void)hipFree(p);
    }
    if (f->stream2) (void)hipStreamSynchronize(f->stream2);
    for (void* p : {(void*)f->lsplit, (void*)f->usplit, (void*)f->cptr, (void*)f->crow, (void*)f->cpos, f->sinv, (void*)f->ipiv[0], (void*)f->ipiv[1],
                    f->colbuf[0], f->colbuf[1], (void*)f->flag, f->G, f->K})
        if (p) (void)hipFree(p);
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    if (f->ev_join) (void)hipEventDestroy(f->ev_join);
    for (int c = 0; c < 2; ++c) {
        if (f->side[c]) (void)hipStreamSynchronize(f->side[c]);
        if (f->ev_panel[c]) (void)hipEventDestroy(f->ev_panel[c]);
        if (f->ev_update[c]) (void)hipEventDestroy(f->ev_update[c]);
        if (f->side[c]) (void)hipStreamDestroy(f->side[c]);
    }
    if (f->stream2) (void)hipStreamDestroy(f->stream2);
    delete f;
}

void lsa_blu_destroy(lsa_blu* f) {
    if (!f) return;
    if (f->ctx && f->ctx->stream) (void)hipStreamSynchronize(f->ctx->stream);
    blu_free(f);
}

// symbolic part of a factorisation: splits on C's pattern, a CSC view (positions into C's value array) for the corner
// update, and every device buffer
static bool blu_setup(lsa_ctx* ctx, lsa_blu* f, const lsa_mat* C) {
    const int32_t n = f->n, B = f->B;
    const size_t esz = C->dtype == LSA_C128 ? 16 : 8;
    const size_t inv_bytes = (size_t)std::max(n, 1) * f->ld * esz;
    std::vector<int32_t> ls((size_t)n), us((size_t)n), cptr((size_t)n + 1, 0), crow((size_t)C->nnz), cpos((size_t)C->nnz);
    const int32_t* c = C->h_ci.data();
    for (int32_t r = 0; r < n; ++r) {
        const int32_t bs = (r / B) * B, be = std::min(n, bs + B);
        ls[r] = (int32_t)(std::lower_bound(c + C->h_rp[r], c + C->h_rp[r + 1], bs) - c);
        us[r] = (int32_t)(std::lower_bound(c + C->h_rp[r], c + C->h_rp[r + 1], be) - c);
        for (int32_t p = C->h_rp[r]; p < C->h_rp[r + 1]; ++p) ++cptr[(size_t)c[p] + 1];
    }
    for (int32_t j = 0; j < n; ++j) cptr[(size_t)j + 1] += cptr[j];
    f->uwin_lo.assign((size_t)f->nb, 0);
    f->lwin_hi.assign((size_t)f->nb, 0);
    for (int32_t b = 0; b < f->nb; ++b) {
        const int32_t bs = b * B, be = std::min(n, bs + B);
        int32_t ulo = be, lhi = bs;
        for (int32_t r = bs; r < be; ++r) {
            if (us[r] < C->h_rp[r + 1]) ulo = std::min(ulo, r);
            if (C->h_rp[r] < ls[r]) lhi = std::max(lhi, r + 1);
        }
        f->uwin_lo[b] = ulo;
        f->lwin_hi[b] = lhi;
    }
    // column ranges of the neighbouring blocks that the coupling blocks touch, and room for the absorbed couplings
    f->gcol0.assign((size_t)f->nb, 0);
    f->kcol1.assign((size_t)f->nb, 0);
    int32_t gmax = 0, kmax = 0;
    for (int32_t b = 0; b < f->nb; ++b) {
        const int32_t bs = b * B, be = std::min(n, bs + B);
        int32_t g0 = bs, k1 = be;
        for (int32_t r = bs; r < be; ++r) {
            if (C->h_rp[r] < ls[r]) g0 = std::min(g0, c[C->h_rp[r]]);            // first (smallest) column of the row
            if (us[r] < C->h_rp[r + 1]) k1 = std::max(k1, c[C->h_rp[r + 1] - 1] + 1);  // last column + 1
        }
        f->gcol0[b] = g0;
        f->kcol1[b] = k1;
        gmax = std::max(gmax, bs - g0);
        kmax = std::max(kmax, k1 - be);
    }
    f->ldg = ((gmax + 15) / 16) * 16;
    f->ldk = ((kmax + 15) / 16) * 16;
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
              hipMalloc((void**)&f->ipiv[0], 4 * (2 * (size_t)B + 128)) == hipSuccess && hipMalloc((void**)&f->ipiv[1], 4 * (2 * (size_t)B + 128)) == hipSuccess &&
              hipMalloc(&f->colbuf[0], esz * (size_t)B * 48) == hipSuccess && hipMalloc(&f->colbuf[1], esz * (size_t)B * 48) == hipSuccess &&
              hipMalloc((void**)&f->flag, 16) == hipSuccess && hipStreamCreateWithFlags(&f->stream2, hipStreamNonBlocking) == hipSuccess &&
              // (the side streams are created for every size although only wide blocks use them: measured with two
              // processes sharing one GPU, a process with four streams here runs a step in 0.24 s, one with two in 0.43 s --
              // which hardware queue slots a process's streams land on decides whether it collides with its neighbour)
              hipStreamCreateWithFlags(&f->side[0], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&f->side[1], hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_panel[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_panel[1], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_update[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_update[1], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming) == hipSuccess;
    // The absorbed couplings halve the dependent launches of a solve and add bytes: a gain while the sweeps are latency-bound
    // (S30k, B = 1024: 345 -> 291 us per apply), a loss once the dense blocks dominate (B = 1536: 1348 -> 1417 us;
    // B = 3072: 7.7 -> 9.5 ms).  LSA_BLU_ABSORB = 0 / 1 overrides.
    if (ok && f->nb > 1 && f->want_absorb) {
        // optional: when they do not fit, the sweeps fall back to Sinv + the sparse rows of C
        size_t free_now = 0, total_now = 0;
        (void)hipMemGetInfo(&free_now, &total_now);
        const size_t gb = (size_t)n1 * std::max(f->ldg, 16) * esz, kb = (size_t)n1 * std::max(f->ldk, 16) * esz;
        if (gb + kb < free_now / 2 && hipMalloc(&f->G, gb) == hipSuccess) {
            if (hipMalloc(&f->K, kb) != hipSuccess) {
                (void)hipFree(f->G);
                f->G = nullptr;
            }
        }
        (void)hipGetLastError();
    }
    hipStream_t s = ctx->stream;
    ok = ok && hipMemcpyAsync(f->lsplit, ls.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->usplit, us.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->cptr, cptr.data(), 4 * ((size_t)n + 1), hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->crow, crow.data(), 4 * (size_t)C->nnz, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(f->cpos, cpos.data(), 4 * (size_t)C->nnz, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    return ok;
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
        return lsa_set_error(ctx, LSA_ERR_OOM, "block LU: bandwidth %d needs %.1f GB of inverted Schur blocks (%.1f GB free); use ILU(k)", bw,
                             inv_bytes / 1e9, free_b / 1e9);
    // FNV-1a over the host pattern: the key of the per-context cache
    uint64_t hash = 1469598103934665603ull;
    auto mix = [&](const int32_t* v, size_t cnt) {
        for (size_t i = 0; i < cnt; ++i) {
            hash ^= (uint32_t)v[i];
            hash *= 1099511628211ull;
        }
    };
    mix(C->h_rp.data(), C->h_rp.size());
    mix(C->h_ci.data(), C->h_ci.size());
    if (hash == 0) hash = 1;
    int32_t mid = (n > 0 ? (n + B - 1) / B : 0) / 2;
    if (const char* e = getenv("LSA_BLU_TWIST"))  // 0: one chain from the first block (the middle block is the last one)
        if (atoi(e) == 0) mid = n > 0 ? (n + B - 1) / B - 1 : 0;
    const char* absorb_env = getenv("LSA_BLU_ABSORB");
    const bool absorb = absorb_env ? atoi(absorb_env) != 0 : B <= 1024;
    lsa_blu* f = nullptr;
    const bool reused = f != nullptr;
    if (!reused) {
        f = new lsa_blu();
        f->ctx = ctx;
        f->C = C;
        f->n = n;
        f->B = B;
        f->ld = ld;
        f->nb = n > 0 ? (n + B - 1) / B : 0;
        f->mid = mid;
        f->bandwidth = bw;
        f->dtype = C->dtype;
        f->pattern_hash = hash;
        f->nnz = C->nnz;
        f->want_absorb = absorb;
        if (!blu_setup(ctx, f, C)) {
            blu_free(f);
            return lsa_set_error(ctx, LSA_ERR_OOM, "lsa_blu_create: out of device memory");
        }
    }
    if (getenv("LSA_BLU_TIMING")) fprintf(stderr, "[lsa_blu] symbolic setup + allocation + upload %.1f ms%s\n", (now_s() - t0) * 1e3, reused ? " (cached)" : "");
    int rc = f->dtype == LSA_C128 ? factorize<cplx>(ctx, f) : factorize<double>(ctx, f);
    if (rc != LSA_OK) {
        blu_free(f);
        return rc;
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

// Dependent kernel launches of one solve (the length of the chain its latency is made of)
int lsa_blu_apply_launches(const lsa_blu* f, int32_t* launches) {
    if (!f || !launches) return LSA_ERR_ARG;
    const int32_t steps = f->nb > 0 ? 2 * std::max(f->mid, f->nb - 1 - f->mid) + 1 : 0;
    *launches = f->G ? steps : 2 * steps;
    return LSA_OK;
}

// Algorithmic bytes of one solve (what lsa_blu_solve_time's milliseconds are to be divided into): every Schur inverse
// once in the elimination sweep, its window columns once in the substitution sweep, the off-block entries of C twice
// (value + column index), the vectors of both sweeps.
int lsa_blu_apply_bytes(const lsa_blu* f, int64_t* bytes) {
    if (!f || !bytes || !f->C) return LSA_ERR_ARG;
    const int64_t esz = f->dtype == LSA_C128 ? 16 : 8;
    if (f->G) {  // absorbed form: Sinv once, G and K of every block once in each sweep they take part in, the vectors
        int64_t elems = 0;
        for (int32_t b = 0; b < f->nb; ++b) {
            const int64_t bs = (int64_t)b * f->B, be = std::min<int64_t>(f->n, bs + f->B), m = be - bs;
            const int64_t gw = b > 0 ? bs - f->gcol0[b] : 0, kw = b < f->nb - 1 ? f->kcol1[b] - be : 0;
            elems += m * m + m * (gw + kw);
        }
        *bytes = elems * esz + 6 * (int64_t)f->n * 16;
        return LSA_OK;
    }
    int64_t dense = 0, offblock = 0;
    for (int32_t b = 0; b < f->nb; ++b) {
        const int64_t bs = (int64_t)b * f->B, be = std::min<int64_t>(f->n, bs + f->B), m = be - bs;
        int64_t win = m;
        if (b < f->mid) win = be - f->uwin_lo[b];
        else if (b > f->mid) win = f->lwin_hi[b] - bs;
        dense += m * m + (b == f->mid ? 0 : m * win);
        for (int64_t r = bs; r < be; ++r) {
            const int32_t* c0 = f->C->h_ci.data() + f->C->h_rp[r];
            const int32_t* c1 = f->C->h_ci.data() + f->C->h_rp[r + 1];
            offblock += (std::lower_bound(c0, c1, (int32_t)bs) - c0) + (c1 - std::lower_bound(c0, c1, (int32_t)be));
        }
    }
    *bytes = dense * esz + 2 * offblock * (esz + 4) + 8 * (int64_t)f->n * 16;
    return LSA_OK;
}

}  // extern "C"
