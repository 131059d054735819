// Nested-dissection multifrontal LU on the device: the exact factorisation behind PreconditionerType.LU, the setting of
// the reference's cylinder runs for the ST's KSP (.examples/eigenvalues.py:100; Sensitivity/__init__.py:182,260).
//
// Layout (analysis: nd_symbolic.hip).  Tree node t owns m unknowns and has a boundary of b unknowns of its ancestors; its
// front F_t is (m + b)^2.  What stays resident in HBM after the factorisation is PACKED, m^2 + 2 m b scalars per node:
//     L_t = [ F11^-1 ; -F21 F11^-1 ]   (f x m, row-major)        U_t = F11^-1 F12   (m x b, row-major)
// so that a solve is two sweeps over the tree in which every node is ONE dense mat-vec per sweep:
//     up   (leaves -> roots):  v = b[own] + children's updates;  [y[own]; update_t - carried] = L_t v
//     down (roots -> leaves):  x[own] = y[own] - U_t x[boundary]
// The full front exists only while its node is being factored: the nodes of a tree level are factored in CHUNKS whose
// working fronts (f^2 each) share one arena; the update matrix F22 - F21 F11^-1 F12 (b^2) a node hands to its parent lives
// in a second arena from the node's chunk to its parent's (offsets from a first-fit allocator run over the chunk order at
// analysis time).  Device memory is therefore sum(m^2 + 2 m b) + one chunk's fronts + the live update matrices, not
// sum(f^2): what lets the 3D cases beyond a million unknowns fit (round 2 kept every front whole: 40-60 % dead storage).
// All nodes of a tree level run in one launch per sweep: 2 * (levels) - 1 dependent launches per solve.
//
// Sweeps without index chasing: the iteration's vectors are kept in the elimination order (a node's own unknowns are
// contiguous: the caller orders the matrix by lsa_nd_order and hands the tree back), children PUSH their update entries
// into per-child slot rows of the parent (fixed slots: bitwise repeatable, no atomics) and parents push the solution
// entries a child's boundary needs into the child's boundary vector, so every gather on a kernel's critical path is a
// contiguous load whose address depends on the node record only; index tables (cmap, gell) feed stores.
//
// Pivoting: rows are chosen by magnitude inside the pivot block of each front and never physically interchanged (the
// permutation is undone once, when the inverse is gathered).  A pivot below 1e-15 * max|C| is reported as
// LSA_ERR_ZERO_PIVOT; the operator layer (solver.hip) verifies every solve against b - C x.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <new>
#include <string>

#include "lsa_internal.h"
#include "nd_internal.h"

int k_allgather_inplace(lsa_ctx* ctx, void* vec, size_t bytes_per_rank);  // comm.hip

namespace {

constexpr int kRT = 32;      // front rows per solve tile
constexpr int kCH = 1024;    // vector entries staged in LDS per pass of a solve tile
constexpr int kGT = 64;      // GEMM tile edge

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline size_t esize(int dtype) { return dtype == LSA_C128 ? 16 : 8; }

struct NdNodeDev {
    int64_t front_off;  // scalars into the working arena (f x f, row-major) while the node's chunk is being factored
    int64_t lfac_off;   // packed factors: [F11^-1; -F21 F11^-1], f x m row-major
    int64_t ufac_off;   // packed factors: F11^-1 F12, m x b row-major
    int64_t upd_off;    // the node's update matrix (b x b, row-major) in the update arena
    int64_t u_off;      // into the update-vector / boundary-vector buffers (b entries)
    int64_t ge_off;     // into gell (nchild * f entries)
    int64_t acc_off;    // upward sweep: this node's slot rows (nchild x f entries, row c written by child c), or -1: pull through gell
    int64_t pacc_off;   // upward sweep: this node's slot row in its parent's accumulation buffer, or -1: writes its update vector
    int32_t idx_off;    // into idx (f entries)
    int32_t cmap_off;   // into cmap (b entries)
    int32_t piv_off;    // into ipiv / rowq (m entries)
    int32_t own0;       // first own unknown when the vectors are in elimination order
    int32_t m, f, parent;
    int32_t nchild;     // rows of the node's gather table
    // distributed top nodes (NdSymbolic: kind 4): this rank's slice of the boundary rows / of the own rows of U.  Everywhere else
    // brow0 = 0, brow = f - m, orow0 = 0, orows = m.  The working front of a node is (m + brow) x f, its packed L (m + brow) x m,
    // its packed U orows x (f - m), its update matrix brow x (f - m).
    int32_t brow0, brow, orow0, orows;
    int32_t flags;      // bit 0: distributed node (its downward pushes are done by nd_dist_unpack_kernel, after the exchange of its own rows)
    int32_t pad0;
    int64_t xg_base, xg_stride;  // distributed node: own row j lies at xg_base + (j / s) * xg_stride + j % s of the own-row exchange buffer, s = ceil(m / ranks)
    int64_t inv_off;    // distributed node: the whole inverse of its pivot block in the working arena while the node is factored (-1: the
                        // first m rows of the packed L are the inverse)
};
static_assert(sizeof(NdNodeDev) == 144, "node record layout");

// what the sweeps read of a node (in level order: one record per workgroup and launch, loaded first thing -- kept at 96 bytes)
struct NdSweepNode {
    int64_t lfac_off, ufac_off, u_off, ge_off, acc_off, pacc_off;
    int32_t idx_off, cmap_off, own0, m, f, nchild;
    int32_t brow0, brow, orow0, orows, flags, pad0;
    NdSweepNode() = default;
    explicit NdSweepNode(const NdNodeDev& n)
        : lfac_off(n.lfac_off), ufac_off(n.ufac_off), u_off(n.u_off), ge_off(n.ge_off), acc_off(n.acc_off), pacc_off(n.pacc_off), idx_off(n.idx_off),
          cmap_off(n.cmap_off), own0(n.own0), m(n.m), f(n.f), nchild(n.nchild), brow0(n.brow0), brow(n.brow), orow0(n.orow0), orows(n.orows),
          flags(n.flags), pad0(0) {}
};
static_assert(sizeof(NdSweepNode) == 96, "sweep record layout");

struct TileList {
    int64_t off = 0;  // pairs of int32 into the tile buffer
    int32_t count = 0;
};

// factorisation work unit: nodes of ONE tree level whose working fronts share the arena
struct NdChunk {
    int32_t node_begin = 0, node_count = 0, max_m = 0, max_f = 0;  // range of the chunk-ordered node list (own size descending)
    std::vector<int32_t> sorted_m;
    int64_t work_entries = 0;             // sum of f^2: scalars of the arena this chunk uses
    int64_t asm_begin = 0, asm_count = 0;  // its range of the assembly lists
    TileList unperm, gemm[3], save;
    std::vector<TileList> ext;  // one per child rank
    bool exchange_before = false;  // subtree-parallel: the ranks' subtree-root update matrices are all-gathered before this chunk
    // distributed top nodes in this chunk: their children's update matrices arrive in row chunks through the staging buffer, one
    // in-place all-gather per step; slot r of a step holds rows [row0, row0 + nrows) of child `child` (a node id; nrows = 0: empty)
    struct XPiece {
        int32_t child = -1, row0 = 0, nrows = 0;
    };
    std::vector<std::vector<XPiece>> xsteps;  // [step][rank]
};

// one launch of each sweep: the nodes of a tree level
struct NdLevel {
    int32_t node_begin = 0, node_count = 0, max_m = 0, max_f = 0;
    int32_t fwd_tiles = 0, bwd_tiles = 0;  // grid.y of the sweep kernels: tiles of the tallest node (0 = nothing to do)
    int32_t sweep_rows = 32;               // rows per upward-sweep tile: 32; 8 on levels with few tiles (both sweeps); 128 on thin levels
    // distributed top nodes of the level: their range of the list d_dist_nodes, the level's exchange regions (entries per rank)
    int32_t dist_begin = 0, dist_count = 0, dist_children = 0, dist_rows = 0;  // ... most children / most entries (own rows, a child's boundary) of one of them
    int64_t ux_base = 0, ux_slot = 0, xg_base = 0, xg_slot = 0;
};

template <typename T>
__global__ __launch_bounds__(256) void nd_maxabs2_kernel(int64_t nnz, const T* __restrict__ v, unsigned long long* __restrict__ out) {
    // one atomic per workgroup (an atomic per wavefront on one address serialised: 165 us for 0.9 M entries)
    __shared__ double wmax[4];
    double best = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const double a = s_abs2(v[i]);
        if (a == a && a > best) best = a;
        else if (a != a) best = a;  // a NaN must reach the host (NaN > x is false: keep it by hand)
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(best, o);
        best = (best != best) ? best : (other != other) ? other : fmax(best, other);
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = (best != best) ? best : (wmax[w] != wmax[w]) ? wmax[w] : fmax(best, wmax[w]);
        // non-negative doubles order like their bit patterns; a NaN's pattern (0x7ff8...) is above every finite value's and infinity's
        atomicMax(out, (unsigned long long)__double_as_longlong(best));
    }
}

// test aid (LSA_ND_TEST_PERTURB): every stored factor scalar times (1 + eps), so that a solve is wrong by about eps and the
// operator layer's iterative refinement has something to do (tests/test_gpu_3d.py)
template <typename T>
__global__ void nd_scale_kernel(int64_t count, T* __restrict__ v, double factor) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) v[i] = s_mul(factor, v[i]);
}

template <typename T>
__global__ void nd_assemble_kernel(int64_t count, const T* __restrict__ val, const int32_t* __restrict__ src, const int64_t* __restrict__ dst,
                                   T* __restrict__ front) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) front[dst[e]] = val[src[e]];
}

// parent front += child's update matrix (tile = 16 rows of the child's b x b update matrix in the update arena)
template <typename T>
__global__ __launch_bounds__(256) void nd_extend_add_kernel(const int32_t* __restrict__ tiles, const NdNodeDev* __restrict__ nodes,
                                                            const int32_t* __restrict__ cmap, T* __restrict__ front, const T* __restrict__ upd) {
    const int32_t c = tiles[2 * blockIdx.x], i0 = tiles[2 * blockIdx.x + 1];
    const NdNodeDev nc = nodes[c];
    const NdNodeDev np = nodes[nc.parent];
    const int32_t b = nc.f - nc.m;
    const int32_t* map = cmap + nc.cmap_off;
    const int32_t i = i0 + (threadIdx.x >> 4);
    if (i >= b) return;
    const T* src = upd + nc.upd_off + (int64_t)i * b;
    int32_t pr = map[i];
    if (pr >= np.m) {  // a boundary row of the parent: a distributed parent keeps only its own slice of them
        pr -= np.brow0;
        if (pr < np.m || pr >= np.m + np.brow) return;
    }
    T* dst = front + np.front_off + (int64_t)pr * np.f;
    for (int32_t j = threadIdx.x & 15; j < b; j += 16) {
        T* d = dst + map[j];
        *d = s_add(*d, src[j]);
    }
}

// the same for rows [row0, row0 + nrows) of child c's update matrix that arrived in the staging buffer (`src`: nrows x b,
// row-major): a distributed parent receives its children's update matrices in row chunks, one rank's chunk per launch (the
// chunks of one step may belong to different children of one parent: launches in slot order keep the sums in a fixed order)
template <typename T>
__global__ __launch_bounds__(256) void nd_extend_add_staged_kernel(const NdNodeDev* __restrict__ nodes, const int32_t* __restrict__ cmap, T* __restrict__ front,
                                                                   const T* __restrict__ src, int32_t c, int32_t row0, int32_t nrows) {
    const NdNodeDev nc = nodes[c];
    const NdNodeDev np = nodes[nc.parent];
    const int32_t b = nc.f - nc.m;
    const int32_t* map = cmap + nc.cmap_off;
    const int32_t k = (int32_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (k >= nrows) return;
    int32_t pr = map[row0 + k];
    if (pr >= np.m) {
        pr -= np.brow0;
        if (pr < np.m || pr >= np.m + np.brow) return;
    }
    const T* s = src + (int64_t)k * b;
    T* dst = front + np.front_off + (int64_t)pr * np.f;
    for (int32_t j = threadIdx.x & 15; j < b; j += 16) {
        T* d = dst + map[j];
        *d = s_add(*d, s[j]);
    }
}

// max of a 64-bit key over the wavefront, returned to every lane: DPP steps inside each row of 16 lanes (a ds_bpermute
// butterfly costs ~100 cycles per step, and the pivot searches are chains of them), then the four row maxima through SGPRs
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_mov_key(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v & 0xFFFFFFFFull), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long v) {
    unsigned long long o;
    o = dpp_mov_key<0xB1>(v);  // quad_perm [1,0,3,2]
    v = o > v ? o : v;
    o = dpp_mov_key<0x4E>(v);  // quad_perm [2,3,0,1]
    v = o > v ? o : v;
    o = dpp_mov_key<0x141>(v);  // row_half_mirror
    v = o > v ? o : v;
    o = dpp_mov_key<0x140>(v);  // row_mirror: every lane of a row now holds the row's max
    v = o > v ? o : v;
    unsigned long long best = 0ull;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), 16 * row);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * row);
        const unsigned long long r = ((unsigned long long)hi << 32) | lo;
        best = r > best ? r : best;
    }
    return best;
}

__device__ __forceinline__ unsigned long long pivot_key(double mag2, int32_t row) {
    // |a|^2 with its low 16 mantissa bits replaced by (65535 - row): one unsigned max picks the largest magnitude and,
    // among magnitudes equal to 2^-36 relative, the lowest row (deterministic)
    return ((unsigned long long)__double_as_longlong(mag2) & ~0xFFFFull) | (unsigned long long)(65535 - row);
}

// ---- blocked Gauss-Jordan inversion of the pivot blocks of a level ---------------------------------------------------------
// Columns are eliminated in blocks of kNB; inside a block in panels of W columns (W = 8 for pivot blocks of up to 2048 rows,
// narrower for taller ones so that a thread's rows of the panel stay in registers).  Rows are never interchanged: a row
// that has served as a pivot is excluded from later searches (rowq), the permutation is undone by nd_unperm_kernel.
// ONE launch per panel (nd_gj_fused_kernel), grid (node, 1 + kNB / 16):
//   workgroup y = 0   thread per row (RPT rows per thread): first brings the panel's W columns up to date with the rank-W
//                     update of the PREVIOUS panel of the block, then eliminates them;
//   workgroups y >= 1 apply that previous panel's rank-W update to a tile of 16 of the block's other columns, all rows:
//                     A[i, c] = (i was a pivot row of the previous panel ? 0 : A[i, c]) + W_prev[i, :] Y_prev[:, c],  Y_prev =
//                     the previous panel's pivot rows in these columns, read before the tile is touched.
// A workgroup owns its columns for all rows, and the columns of the previous panel are read-only in the launch: no staging
// buffer, no second launch per panel (round 2: panel launch + block-update launch, 2 x 157 launches of ~12 us in the
// factorisation of the 30 k-unknown case).  After the block's last panel one launch of the tiles alone finishes the block;
// then the pivot rows' values in all other columns are staged (nd_gj_stage_kernel) and one rank-kNB product
//                   A[:, J] = (pivot row ? 0 : A[:, J]) + Wb Yb  updates the rest (nd_gj_gemm_kernel, 64 x 64 tiles):
// the columns outside a block are touched once per kNB pivots instead of once per 8 (the update of a 6 000-row pivot block
// streamed 1.1 GB per 8 pivots, and its panel did not fit the registers of one workgroup at W = 8).
constexpr int kNB = 32;

// k0 < 0: no panel in this launch (the tiles finish the block);  kprev < 0: no previous panel to apply
template <typename T, int NT, int RPT, int W>
__global__ __launch_bounds__(NT) void nd_gj_fused_kernel(const int32_t* __restrict__ lvl_nodes, const NdNodeDev* __restrict__ nodes,
                                                         T* __restrict__ front, int32_t* __restrict__ ipiv, int32_t* __restrict__ rowq,
                                                         int32_t kb, int32_t k0, int32_t kprev, int32_t* __restrict__ flag, double tiny2) {
    __shared__ unsigned long long skey[W];
    __shared__ T prow_s[2][W];
    __shared__ int32_t prows[W];
    __shared__ T yprev[W][16];
    __shared__ int32_t pprev[W];
    const int32_t t = lvl_nodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    T* a = front + nd.front_off;
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t wp = kprev >= 0 ? min(W, m - kprev) : 0;  // columns of the previous panel in this node
    if (blockIdx.y > 0) {
        // ---- tile of 16 block columns: the previous panel's rank-W update, all rows ----
        if (wp <= 0) return;
        const int32_t cb = ((int32_t)blockIdx.y - 1) * 16 + (tid & 15);
        const int32_t c = kb + cb;
        const bool mine = cb < kNB && c < m && !(c >= kprev && c < kprev + wp) && !(k0 >= 0 && c >= k0 && c < k0 + W);
        if (tid < W) pprev[tid] = tid < wp ? ipiv[nd.piv_off + kprev + tid] : -1;
        __syncthreads();
        for (int e = tid; e < 16 * W; e += NT) {  // (the column of entry e is that of thread e & 15 = tid & 15: NT is a multiple of 16)
            const int j = e >> 4;
            yprev[j][e & 15] = (mine && j < wp) ? a[(size_t)pprev[j] * ld + c] : scalar_traits<T>::zero();
        }
        __syncthreads();
        if (!mine) return;
        T y[W];
#pragma unroll
        for (int j = 0; j < W; ++j) y[j] = yprev[j][tid & 15];
        // four rows per trip, their loads issued together (a trip is a chain of dependent loads; with 64 threads a tile of a
        // 64-row pivot block would otherwise walk 16 of them one after the other)
        constexpr int RL = NT / 16;
        for (int32_t i0 = tid >> 4; i0 < m; i0 += 4 * RL) {
            T cur[4], mult[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t i = i0 + u * RL;
                const T* ai = a + (size_t)min(i, m - 1) * ld;
                cur[u] = ai[c];
#pragma unroll
                for (int j = 0; j < W; ++j) mult[u][j] = j < wp ? ai[kprev + j] : scalar_traits<T>::zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t i = i0 + u * RL;
                if (i >= m) break;
                bool is_piv = false;
#pragma unroll
                for (int j = 0; j < W; ++j) is_piv |= (i == pprev[j]);
                T acc = is_piv ? scalar_traits<T>::zero() : cur[u];
#pragma unroll
                for (int j = 0; j < W; ++j) fma_acc(acc, mult[u][j], y[j]);
                a[(size_t)i * ld + c] = acc;
            }
        }
        return;
    }
    // ---- the panel ----
    if (k0 < 0) return;
    const int32_t w = min(W, m - k0);
    if (w <= 0) return;
    int32_t* piv = ipiv + nd.piv_off;
    int32_t* rq = rowq + nd.piv_off;
    if (tid < W) {
        skey[tid] = 0ull;
        pprev[tid] = tid < wp ? piv[kprev + tid] : -1;
    }
    __syncthreads();
    if (wp > 0 && tid < W * W) {  // the previous panel's pivot rows in this panel's columns, before anything is overwritten
        const int j = tid / W, cc = tid % W;
        yprev[j][cc] = (j < wp && cc < w) ? a[(size_t)pprev[j] * ld + k0 + cc] : scalar_traits<T>::zero();
    }
    T r[RPT][W];
    bool used[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int32_t i = tid + NT * q;
        used[q] = i >= m || rq[min(i, m - 1)] >= 0;
#pragma unroll
        for (int c = 0; c < W; ++c) r[q][c] = (i < m && c < w) ? a[(size_t)i * ld + k0 + c] : scalar_traits<T>::zero();
    }
    __syncthreads();
    if (wp > 0) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            if (i >= m) continue;
            bool is_piv = false;
#pragma unroll
            for (int j = 0; j < W; ++j) is_piv |= (i == pprev[j]);
            T mult[W];
#pragma unroll
            for (int j = 0; j < W; ++j) mult[j] = j < wp ? a[(size_t)i * ld + kprev + j] : scalar_traits<T>::zero();
#pragma unroll
            for (int c = 0; c < W; ++c) {
                T acc = is_piv ? scalar_traits<T>::zero() : r[q][c];
#pragma unroll
                for (int j = 0; j < W; ++j) fma_acc(acc, mult[j], yprev[j][c]);
                r[q][c] = c < w ? acc : scalar_traits<T>::zero();
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < W; ++jj) {
        if (jj >= w) break;
        unsigned long long key = 0ull;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (!used[q]) {
                const unsigned long long kq = pivot_key(s_abs2(r[q][jj]), tid + NT * q);
                key = kq > key ? kq : key;
            }
        }
        key = wave_max_key(key);
        if (lane == 0) atomicMax(&skey[jj], key);
        __syncthreads();
        key = skey[jj];
        const int32_t p = 65535 - (int32_t)(key & 0xFFFFull);
        if (tid == 0) {
            piv[k0 + jj] = p;
            prows[jj] = p;
            if (!(__longlong_as_double((long long)(key & ~0xFFFFull)) > tiny2) && atomicCAS(&flag[1], 0, t + 1) == 0) {
                flag[2] = k0 + jj;
                flag[3] = (int32_t)(key >> 32);  // high word of |pivot|^2
            }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (tid + NT * q == p) {
                T pv = r[q][jj];
                if (s_abs2(pv) == 0.0) s_from(pv, 1.0, 0.0);
                const T pinv = s_inv(pv);
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const T v = (j == jj) ? pinv : s_mul(pinv, r[q][j]);
                    prow_s[jj & 1][j] = v;
                    r[q][j] = v;
                }
                used[q] = true;
                rq[p] = k0 + jj;
            }
        }
        __syncthreads();
        T prow[W];
#pragma unroll
        for (int j = 0; j < W; ++j) prow[j] = prow_s[jj & 1][j];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int32_t i = tid + NT * q;
            if (i >= m || i == p) continue;
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
}

// Yb[j][c] = A[pivot row of column kb + j][c] for the kw columns of a finished block (or super-block, tournament path), columns
// c in [c_lo, c_hi): the rows the product below needs, staged because that product overwrites them.
// grid: (node, 256-column chunk of the window);  ycap = rows of staging space per unknown
template <typename T>
__global__ __launch_bounds__(256) void nd_gj_stage_kernel(const int32_t* __restrict__ lvl_nodes, const NdNodeDev* __restrict__ nodes,
                                                          const T* __restrict__ front, const int32_t* __restrict__ ipiv, int32_t kb, int32_t kw,
                                                          int32_t ycap, int32_t c_lo, int32_t c_hi, T* __restrict__ ybuf) {
    const int32_t t = lvl_nodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    const int32_t nb = min(kw, m - kb);
    const int32_t c = c_lo + (int32_t)blockIdx.y * 256 + threadIdx.x;
    if (nb <= 0 || m <= kNB || c >= min(m, c_hi)) return;
    const T* a = front + nd.front_off;
    T* yb = ybuf + (size_t)ycap * nd.piv_off;
    const int32_t* pv = ipiv + nd.piv_off + kb;
    int32_t j = 0;
    for (; j + 4 <= nb; j += 4) {  // (independent loads, issued together)
        const T v0 = a[(size_t)pv[j] * ld + c], v1 = a[(size_t)pv[j + 1] * ld + c], v2 = a[(size_t)pv[j + 2] * ld + c], v3 = a[(size_t)pv[j + 3] * ld + c];
        yb[(size_t)j * m + c] = v0;
        yb[(size_t)(j + 1) * m + c] = v1;
        yb[(size_t)(j + 2) * m + c] = v2;
        yb[(size_t)(j + 3) * m + c] = v3;
    }
    for (; j < nb; ++j) yb[(size_t)j * m + c] = a[(size_t)pv[j] * ld + c];
}

// the columns outside the finished block: A[i, c] = (i is a pivot row of the block ? 0 : A[i, c]) + sum_j Wb[i, j] Yb[j, c]
// grid: (node, 64-row tile, 64-column tile); 4 x 4 per thread; the whole K = nb <= 32 extent in one pass through LDS
// Column windows (tournament path with look-ahead): `only` non-empty = update just the columns [only_lo, only_hi) (the next
// block's, so that its pivot search can start while the rest is updated); `skip` = leave [skip_lo, skip_hi) alone (done
// already).  ztile0 = first 64-column tile of the grid.
// INVARIANT the look-ahead relies on (launch_level_tp): while this product for block kb runs, the side stream's tournament
// for block kb + kNB may write rowq[r] for rows r that have not been pivots yet.  Such a row goes from -1 to a value
// >= kb + kNB; both read as "not a pivot row of block kb" in the test below (q >= kb && q < kb + nb), so the race cannot
// change a result.  rowq is therefore read through a plain pointer here (no __restrict__ / read-only cache path that a
// future compiler could use to assume the array does not change), and any change to rowq's encoding or to that test must
// keep the two values on the same side of it.  tests/test_gpu_ndlu.py runs the two-stream path (LSA_ND_LOOKAHEAD_MIN lowered).
template <typename T>
__global__ __launch_bounds__(256) void nd_gj_gemm_kernel(const int32_t* __restrict__ lvl_nodes, const NdNodeDev* __restrict__ nodes,
                                                         T* __restrict__ front, const int32_t* rowq, int32_t kb,
                                                         const T* __restrict__ ybuf, int32_t only_lo, int32_t only_hi, int32_t skip_lo,
                                                         int32_t skip_hi, int32_t ztile0) {
    __shared__ T Ws[kNB][kGT + 1];
    __shared__ T Ys[kNB][kGT + 1];
    const int32_t t = lvl_nodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    const int32_t nb = min(kNB, m - kb);
    const int32_t row0 = (int32_t)blockIdx.y * kGT, col0 = ((int32_t)blockIdx.z + ztile0) * kGT;
    if (nb <= 0 || row0 >= m || col0 >= m) return;
    if (col0 >= kb && col0 + kGT <= kb + nb) return;  // tile inside the block
    if (only_hi > only_lo && (col0 >= only_hi || col0 + kGT <= only_lo)) return;
    if (col0 >= skip_lo && col0 + kGT <= skip_hi) return;
    T* a = front + nd.front_off;
    const T* yb = ybuf + (size_t)kNB * nd.piv_off;
    const int32_t* rq = rowq + nd.piv_off;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    for (int e = tid; e < kNB * kGT; e += 256) {
        // Ws[j][q] = Wb[row0 + q][j]: lanes along j (a row's block columns are contiguous);  Ys[j][q] = Yb[j][col0 + q]: lanes along q
        const int qa = e / kNB, ja = e - qa * kNB;
        const int32_t gr = row0 + qa;
        Ws[ja][qa] = (ja < nb && gr < m) ? a[(size_t)gr * ld + kb + ja] : scalar_traits<T>::zero();
        const int jb = e / kGT, qb = e - jb * kGT;
        const int32_t gc = col0 + qb;
        Ys[jb][qb] = (jb < nb && gc < m) ? yb[(size_t)jb * m + gc] : scalar_traits<T>::zero();
    }
    __syncthreads();
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = scalar_traits<T>::zero();
    for (int k = 0; k < nb; ++k) {
        T av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = Ws[k][ty * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = Ys[k][tx + 16 * j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) fma_acc(acc[i][j], av[i], bv[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int32_t gr = row0 + ty * 4 + i;
        if (gr >= m) continue;
        const int32_t q = rq[gr];
        const bool is_piv = q >= kb && q < kb + nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t gc = col0 + tx + 16 * j;
            if (gc >= m || (gc >= kb && gc < kb + nb)) continue;
            if ((gc >= skip_lo && gc < skip_hi) || (only_hi > only_lo && (gc < only_lo || gc >= only_hi))) continue;
            T* cptr = a + (size_t)gr * ld + gc;
            *cptr = is_piv ? acc[i][j] : s_add(*cptr, acc[i][j]);
        }
    }
}

// ---- tournament pivoting for tall pivot blocks --------------------------------------------------------------------------
// The panel launches above search a whole column in ONE workgroup: for a 6 000-row pivot block that is 16 dependent launch
// pairs per 32 columns, each reading its column with a stride of a front row.  Levels whose tallest pivot block has
// LSA_ND_TP_MIN rows or more choose the 32 pivot rows of a block by a tournament instead (communication-avoiding LU,
// Grigori, Demmel, Xiang 2011): every 256 rows pick their 32 best rows by Gaussian elimination with partial pivoting on
// their slice of the block's columns (thread per row, the row in registers); winners meet four sets at a time until one
// set is left; the last workgroup inverts the 32 x 32 pivot tile.  The block's columns then are one small product per row
// (nd_tp_colblock_kernel), and the staged rank-32 product above does the rest: 5-7 launches per 32 columns, all of them
// wide.  The pivot rows reach ipiv / rowq as with the panel launches, so everything downstream is unchanged.
constexpr int kTA = 8;  // candidate sets per workgroup in the later rounds (8 x 32 rows, thread per row)

// rows per workgroup in the first round: thread per row, or two rows per thread where 64 more registers are to be had
template <typename T>
struct tp_first {
    static constexpr int RPT = sizeof(T) == 16 ? 1 : 2;
    static constexpr int rows = 256 * RPT;
};
constexpr int kTRmin = 256;  // smallest first-round chunk: sizes the candidate buffers

__host__ __device__ inline int32_t tp_sets(int32_t m, int32_t first_rows, int32_t round) {  // candidate sets of a pivot block before merge round `round`
    int32_t n = (m + first_rows - 1) / first_rows;
    for (int32_t r = 0; r < round; ++r) n = (n + kTA - 1) / kTA;
    return n;
}

template <typename T, bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void nd_tp_round_kernel(const int32_t* __restrict__ lvl_nodes, const NdNodeDev* __restrict__ nodes,
                                                          const T* __restrict__ front, int32_t* __restrict__ ipiv, int32_t* __restrict__ rowq, int32_t kb,
                                                          int32_t round, const int32_t* __restrict__ cand_in, int32_t* __restrict__ cand_out,
                                                          T* __restrict__ dinv, int32_t* __restrict__ flag, double tiny2) {
    constexpr int RPT = FIRST ? tp_first<T>::RPT : 1;
    __shared__ unsigned long long skey[2];
    __shared__ T prow_s[2][kNB];
    __shared__ int32_t sel_s[kNB];
    __shared__ T Ds[LAST ? kNB : 1][kNB + 1];
    const int32_t t = lvl_nodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    const int32_t w = min(kNB, m - kb);
    if (w <= 0) return;
    const int32_t g = (int32_t)blockIdx.y;  // output set
    const int32_t nin = tp_sets(m, tp_first<T>::rows, FIRST ? 0 : round);
    if (FIRST ? g >= nin : g * kTA >= nin) return;
    const T* a = front + nd.front_off;
    const int32_t* rq = rowq + nd.piv_off;
    const int64_t coff = ((int64_t)(nd.piv_off / kTRmin) + t) * kNB;
    const int tid = threadIdx.x, lane = tid & 63;
    // (two plain arrays, not v[RPT][kNB]: the two-dimensional form ends up in scratch memory)
    int32_t row0 = -1, row1 = -1;
    T v0[kNB], v1[kNB];
    if (FIRST) {
        const int32_t i0 = g * tp_first<T>::rows + tid, i1 = i0 + 256;
        if (i0 < m && rq[i0] < 0) row0 = i0;
        if (RPT == 2 && i1 < m && rq[i1] < 0) row1 = i1;
    } else {
        const int32_t s = g * kTA + (tid >> 5);
        if (s < nin) row0 = cand_in[coff + (int64_t)s * kNB + (tid & 31)];
    }
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
        v0[c] = (row0 >= 0 && c < w) ? a[(size_t)row0 * ld + kb + c] : scalar_traits<T>::zero();
        if constexpr (RPT == 2) v1[c] = (row1 >= 0 && c < w) ? a[(size_t)row1 * ld + kb + c] : scalar_traits<T>::zero();
    }
    bool alive0 = row0 >= 0, alive1 = RPT == 2 && row1 >= 0;
    if (tid < 2) skey[tid] = 0ull;
    if (tid < kNB) sel_s[tid] = -1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kNB; ++j) {  // (no early exit: the loop must unroll for the rows to stay in registers; w is uniform)
        unsigned long long key = (alive0 && j < w) ? pivot_key(s_abs2(v0[j]), tid) : 0ull;
        if constexpr (RPT == 2) {
            const unsigned long long k1 = (alive1 && j < w) ? pivot_key(s_abs2(v1[j]), 256 + tid) : 0ull;
            key = k1 > key ? k1 : key;
        }
        key = wave_max_key(key);
        if (lane == 0 && key) atomicMax(&skey[j & 1], key);
        __syncthreads();
        key = skey[j & 1];
        if (tid == 0) skey[(j + 1) & 1] = 0ull;
        const int32_t win = 65535 - (int32_t)(key & 0xFFFFull);  // (256 *) second row + tid of the winning row
        if (key != 0ull && win == tid) {
#pragma unroll
            for (int c = 0; c < kNB; ++c) prow_s[j & 1][c] = v0[c];
            sel_s[j] = row0;
            alive0 = false;
        }
        if constexpr (RPT == 2) {
            if (key != 0ull && win == 256 + tid) {
#pragma unroll
                for (int c = 0; c < kNB; ++c) prow_s[j & 1][c] = v1[c];
                sel_s[j] = row1;
                alive1 = false;
            }
        }
        __syncthreads();
        if (key != 0ull) {
            const T pv = prow_s[j & 1][j];
            if (s_abs2(pv) > 0.0) {
                const T pinv = s_inv(pv);
                if (alive0) {
                    const T nf = s_sub(scalar_traits<T>::zero(), s_mul(v0[j], pinv));
#pragma unroll
                    for (int c = 0; c < kNB; ++c)
                        if (c > j) fma_acc(v0[c], nf, prow_s[j & 1][c]);
                }
                if constexpr (RPT == 2) {
                    if (alive1) {
                        const T nf = s_sub(scalar_traits<T>::zero(), s_mul(v1[j], pinv));
#pragma unroll
                        for (int c = 0; c < kNB; ++c)
                            if (c > j) fma_acc(v1[c], nf, prow_s[j & 1][c]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!LAST) {
        if (tid < kNB) cand_out[coff + (int64_t)g * kNB + tid] = sel_s[tid];
        return;
    }
    // the winners are the pivot rows of columns kb .. kb + w - 1, in the order the elimination took them: invert their tile
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int j = e / kNB, c = e - j * kNB;
        const int32_t pr = sel_s[j];
        T d = scalar_traits<T>::zero();
        if (j < w && c < w && pr >= 0) d = a[(size_t)pr * ld + kb + c];
        if (j == c && (j >= w || pr < 0)) s_from(d, 1.0, 0.0);
        Ds[j][c] = d;
    }
    __syncthreads();
    if (tid < w && sel_s[tid] < 0 && atomicCAS(&flag[1], 0, t + 1) == 0) {  // fewer rows left than columns: cannot happen for a square block
        flag[2] = kb + tid;
        flag[3] = 0;
    }
    for (int k = 0; k < w; ++k) {
        T rk[4], fi[4], cur[4];
        const T pv0 = Ds[k][k];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, i = e / kNB, c = e - i * kNB;
            rk[q] = Ds[k][c];
            fi[q] = Ds[i][k];
            cur[q] = Ds[i][c];
        }
        __syncthreads();
        const double mag2 = s_abs2(pv0);
        if (tid == 0 && !(mag2 > tiny2) && atomicCAS(&flag[1], 0, t + 1) == 0) {
            flag[2] = kb + k;
            flag[3] = (int32_t)((unsigned long long)__double_as_longlong(mag2) >> 32);
        }
        T pv = pv0;
        if (mag2 == 0.0) s_from(pv, 1.0, 0.0);
        const T pinv = s_inv(pv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, i = e / kNB, c = e - i * kNB;
            T out;
            if (i == k) out = (c == k) ? pinv : s_mul(pinv, rk[q]);
            else if (c == k) out = s_sub(scalar_traits<T>::zero(), s_mul(fi[q], pinv));
            else out = s_sub(cur[q], s_mul(fi[q], s_mul(pinv, rk[q])));
            Ds[i][c] = out;
        }
        __syncthreads();
    }
    T* dv = dinv + (size_t)blockIdx.x * (kNB * kNB);
    for (int e = tid; e < kNB * kNB; e += 256) dv[e] = Ds[e / kNB][e % kNB];
    if (tid < w && sel_s[tid] >= 0) {
        ipiv[nd.piv_off + kb + tid] = sel_s[tid];
        rowq[nd.piv_off + sel_s[tid]] = kb + tid;
    }
}

// the block's own columns after its pivot tile is inverted:  pivot row j <- row j of D^-1,  any other row <- -A[row, K] D^-1
// grid: (node, 256-row tile), thread per row
template <typename T>
__global__ __launch_bounds__(256) void nd_tp_colblock_kernel(const int32_t* __restrict__ lvl_nodes, const NdNodeDev* __restrict__ nodes,
                                                             T* __restrict__ front, const int32_t* __restrict__ rowq, int32_t kb,
                                                             const T* __restrict__ dinv) {
    __shared__ T Ds[kNB][kNB + 1];
    const int32_t t = lvl_nodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    const int32_t w = min(kNB, m - kb);
    const int32_t r0 = (int32_t)blockIdx.y * 256;
    if (w <= 0 || r0 >= m) return;
    const T* dv = dinv + (size_t)blockIdx.x * (kNB * kNB);
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) Ds[e / kNB][e % kNB] = dv[e];
    __syncthreads();
    const int32_t i = r0 + threadIdx.x;
    if (i >= m) return;
    T* ai = front + nd.front_off + (size_t)i * ld + kb;
    const int32_t q = rowq[nd.piv_off + i];
    if (q >= kb && q < kb + w) {
#pragma unroll
        for (int c = 0; c < kNB; ++c)
            if (c < w) ai[c] = Ds[q - kb][c];
        return;
    }
    T x[kNB];
#pragma unroll
    for (int c = 0; c < kNB; ++c) x[c] = c < w ? ai[c] : scalar_traits<T>::zero();
#pragma unroll 4
    for (int c = 0; c < kNB; ++c) {
        if (c >= w) break;
        T acc = scalar_traits<T>::zero();
#pragma unroll
        for (int j = 0; j < kNB; ++j) fma_acc(acc, x[j], Ds[j][c]);
        ai[c] = s_sub(scalar_traits<T>::zero(), acc);
    }
}

// inverse gathered out of the eliminated block, straight into the packed factors: L[a][b] = S[p_a][q_b]  (p = pivot row of
// column a, q = its inverse)
template <typename T>
__global__ __launch_bounds__(256) void nd_unperm_kernel(const int32_t* __restrict__ tiles, const NdNodeDev* __restrict__ nodes,
                                                        T* front, const int32_t* __restrict__ ipiv,
                                                        const int32_t* __restrict__ rowq, T* __restrict__ lfac) {
    const int32_t t = tiles[2 * blockIdx.x], r0 = tiles[2 * blockIdx.x + 1];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    const int32_t ra = r0 + (threadIdx.x >> 4);
    if (ra >= m) return;
    const T* src = front + nd.front_off + (size_t)ipiv[nd.piv_off + ra] * ld;
    const int32_t* q = rowq + nd.piv_off;
    if (nd.inv_off < 0) {
        T* dst = lfac + nd.lfac_off + (size_t)ra * m;
        for (int32_t cb = threadIdx.x & 15; cb < m; cb += 16) dst[cb] = src[q[cb]];
        return;
    }
    // a distributed node: the whole inverse into the working arena (operand of L = -F21 inv), this rank's rows also into the factors
    T* dst = front + nd.inv_off + (size_t)ra * m;
    const bool mine = ra >= nd.orow0 && ra < nd.orow0 + nd.orows;
    T* keep = lfac + nd.lfac_off + (size_t)(mine ? ra - nd.orow0 : 0) * m;
    for (int32_t cb = threadIdx.x & 15; cb < m; cb += 16) {
        const T v = src[q[cb]];
        dst[cb] = v;
        if (mine) keep[cb] = v;
    }
}

// Batched dense products of a chunk (row-major operands, 64 x 64 tiles); inv = the inverse of the node's pivot block:
//   KIND 0:  L[m:] = -F21 inv     (b x m)      KIND 1:  F22 += L[m:] F12   (b x b, in the working front)      KIND 2:  U = inv F12   (m x b)
// (a distributed top node: this rank's rows of each, see NdNodeDev)
// On the matrix cores: v_mfma_f64_16x16x4_f64, one wavefront per 32 x 32 quarter of the 64 x 64 tile
// (2 x 2 instruction tiles; complex scalars as real and imaginary planes, four instructions per complex tile product).
// A 4 x 4-per-thread FMA kernel (round 2's) reads 8 LDS values per 16 multiply-adds and is bound by the LDS array at about a third
// of the FP64 rate; here a k-step of 4 costs a wavefront 4 LDS reads for 4 (real) or 16 (complex) instructions of 64 cycles each.
// Operand maps (cdna_hip_programming.md, "Fragment layout"): lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15]; result
// register r of lane l is C[(l >> 4) + 4 r][l & 15].
// LDS images: A row-major with a row of BK + 1 doubles (16 rows x 2 k per half-wave: 32 distinct bank pairs), B k-major with a
// row of 64 + 16 doubles (two k-rows of a half-wave land 32 banks apart).  The next K-chunk's global loads are issued into
// registers before the current chunk's products (one LDS buffer, two barriers per chunk).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double plane_of(double v, int) { return v; }
__device__ __forceinline__ double plane_of(cplx v, int p) { return p == 0 ? v.re : v.im; }

template <typename T>
struct MfmaTile {
    static constexpr int BK = 16, LDAS = BK + 1, LDBS = kGT + 16, NPL = (int)(sizeof(T) / sizeof(double));
    double As[NPL][kGT * LDAS];
    double Bs[NPL][BK * LDBS];
};

// acc += A B over k in [0, K) for the 64 x 64 tile of a 256-thread workgroup: loadA(r, k) = A[tile row r][k], loadB(k, c) =
// B[k][tile column c], both zero outside their matrix.  On return wavefront w holds rows 32 (w >> 1) .., columns 32 (w & 1) ..:
// acc[plane][i][j][r] = C[32 (w >> 1) + 16 i + (lane >> 4) + 4 r][32 (w & 1) + 16 j + (lane & 15)]
template <typename T, typename FA, typename FB>
__device__ __forceinline__ void mfma_tile_product(int32_t K, FA loadA, FB loadB, MfmaTile<T>& sm, mfma_d4 (&acc)[MfmaTile<T>::NPL][2][2]) {
    constexpr int BK = MfmaTile<T>::BK, LDAS = MfmaTile<T>::LDAS, LDBS = MfmaTile<T>::LDBS, NPL = MfmaTile<T>::NPL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1), l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[p][i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    // chunk staging: thread e = tid + 256 s;  A element (row e >> 4, k e & 15): 16 lanes along a row;  B element (k e >> 6, column e & 63)
    T pa[4], pb[4];
    auto gload = [&](int32_t kk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = tid + 256 * s;
            pa[s] = loadA(e >> 4, kk + (e & 15));
            pb[s] = loadB(kk + (e >> 6), e & 63);
        }
    };
    gload(0);
    for (int32_t kk = 0; kk < K; kk += BK) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = tid + 256 * s;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                sm.As[p][(e >> 4) * LDAS + (e & 15)] = plane_of(pa[s], p);
                sm.Bs[p][(e >> 6) * LDBS + (e & 63)] = plane_of(pb[s], p);
            }
        }
        __syncthreads();
        if (kk + BK < K) gload(kk + BK);
#pragma unroll
        for (int k4 = 0; k4 < BK; k4 += 4) {
            if (kk + k4 >= K) break;  // (zero-filled beyond K: skipping is only cheaper)
            double a[NPL][2], bb[NPL][2];
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = sm.As[p][(wr + 16 * i + l15) * LDAS + k4 + l4];
                    bb[p][i] = sm.Bs[p][(k4 + l4) * LDBS + wc + 16 * i + l15];
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], bb[0][j], acc[0][i][j], 0, 0, 0);
                    if constexpr (NPL == 2) {
                        acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[NPL - 1][i], bb[NPL - 1][j], acc[0][i][j], 0, 0, 0);
                        acc[NPL - 1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], bb[NPL - 1][j], acc[NPL - 1][i][j], 0, 0, 0);
                        acc[NPL - 1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[NPL - 1][i], bb[0][j], acc[NPL - 1][i][j], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }
}

template <typename T, int KIND>
__global__ __launch_bounds__(256) void nd_gemm_mfma_kernel(const int32_t* __restrict__ tiles, const NdNodeDev* __restrict__ nodes,
                                                           T* __restrict__ front, T* __restrict__ lfac, T* __restrict__ ufac) {
    constexpr int NPL = MfmaTile<T>::NPL;
    __shared__ MfmaTile<T> sm;
    const int32_t t = tiles[2 * blockIdx.x], packed = tiles[2 * blockIdx.x + 1];
    const int32_t tm = packed >> 16, tn = packed & 0xFFFF;
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, f = nd.f, b = f - m;
    T* F = front + nd.front_off;
    T* inv = nd.inv_off < 0 ? lfac + nd.lfac_off : front + nd.inv_off;  // the whole inverse (a distributed node: in the working arena)
    T* invrows = lfac + nd.lfac_off;                                    // this rank's own rows of it (all of them unless distributed)
    T* S1 = invrows + (size_t)nd.orows * m;
    T* S2 = ufac + nd.ufac_off;
    const T *A, *B;
    T* C;
    int32_t M, N, K, lda, ldb, ldc;
    // (a distributed top node: this rank's boundary rows of F21 / F22 and its own rows of U; F12 is whole on every rank)
    if (KIND == 0) {
        A = F + (size_t)m * f, lda = f, B = inv, ldb = m, C = S1, ldc = m, M = nd.brow, N = m, K = m;
    } else if (KIND == 1) {
        A = S1, lda = m, B = F + m, ldb = f, C = F + (size_t)m * f + m, ldc = f, M = nd.brow, N = b, K = m;
    } else {
        A = invrows, lda = m, B = F + m, ldb = f, C = S2, ldc = b, M = nd.orows, N = b, K = m;
    }
    const int32_t row0 = tm * kGT, col0 = tn * kGT;
    mfma_d4 acc[NPL][2][2];
    mfma_tile_product<T>(
        K,
        [&](int r, int32_t k) {
            const int32_t gr = row0 + r;
            return (gr < M && k < K) ? A[(size_t)gr * lda + k] : scalar_traits<T>::zero();
        },
        [&](int32_t k, int c) {
            const int32_t gc = col0 + c;
            return (k < K && gc < N) ? B[(size_t)k * ldb + gc] : scalar_traits<T>::zero();
        },
        sm, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1), l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int32_t gr = row0 + wr + 16 * i + l4 + 4 * r;
            if (gr >= M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int32_t gc = col0 + wc + 16 * j + l15;
                if (gc >= N) continue;
                T v;
                s_from(v, acc[0][i][j][r], acc[NPL - 1][i][j][r]);
                T* c = C + (size_t)gr * ldc + gc;
                if (KIND == 0) *c = s_sub(scalar_traits<T>::zero(), v);
                else if (KIND == 1) *c = s_add(*c, v);
                else *c = v;
            }
        }
}

// Gauss-Jordan, tournament path: the columns outside a finished group of kw <= kSB pivot columns [kb, kb + kw) of every node,
//          A[i, c] = (i is a pivot row of the group ? 0 : A[i, c]) + sum_j W[i, j] Y[j, c],
// W = the group's own columns (final), Y[j, :] = the row that was the pivot of column kb + j, staged BEFORE this launch
// (nd_gj_stage_kernel; stride ycap rows per unknown).  The group is one block of kNB columns -- then the window is the rest of
// its super-block -- or a whole super-block of kSB: the elimination of a block multiplies the matrix from the left by a
// matrix that differs from the identity only in the columns of its pivot rows, so does the product over the blocks of a
// super-block, and the super-block's own columns hold exactly those columns once its blocks have updated one another.  The
// rank-kNB update of a 6 700-row pivot block streamed the block through HBM once per 32 pivots (4 flops per byte: 9 TFLOP/s);
// at rank kSB = 128 the product is bound by the matrix cores.
// grid: (node, 64-row tile, 64-column tile from ztile0).  Column windows as in nd_gj_gemm_kernel, whose invariant on rowq
// holds here unchanged (rows that become pivots of a LATER block while this runs read as "not a pivot row of the group").
constexpr int kSB = 128;
template <typename T>
__global__ __launch_bounds__(256) void nd_gj_update_kernel(const int32_t* __restrict__ lvl_nodes, const NdNodeDev* __restrict__ nodes,
                                                           T* __restrict__ front, const int32_t* rowq, int32_t kb, int32_t kw,
                                                           const T* __restrict__ ybuf, int32_t ycap, int32_t only_lo, int32_t only_hi,
                                                           int32_t skip_lo, int32_t skip_hi, int32_t ztile0) {
    constexpr int NPL = MfmaTile<T>::NPL;
    __shared__ MfmaTile<T> sm;
    const int32_t t = lvl_nodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, ld = nd.f;
    const int32_t nb = min(kw, m - kb);
    const int32_t row0 = (int32_t)blockIdx.y * kGT, col0 = ((int32_t)blockIdx.z + ztile0) * kGT;
    if (nb <= 0 || row0 >= m || col0 >= m) return;
    if (col0 >= kb && col0 + kGT <= kb + nb) return;  // tile inside the group
    if (only_hi > only_lo && (col0 >= only_hi || col0 + kGT <= only_lo)) return;
    if (col0 >= skip_lo && col0 + kGT <= skip_hi) return;
    T* a = front + nd.front_off;
    const T* yb = ybuf + (size_t)ycap * nd.piv_off;
    const int32_t* rq = rowq + nd.piv_off;
    mfma_d4 acc[NPL][2][2];
    mfma_tile_product<T>(
        nb,
        [&](int r, int32_t k) {
            const int32_t gr = row0 + r;
            return (gr < m && k < nb) ? a[(size_t)gr * ld + kb + k] : scalar_traits<T>::zero();
        },
        [&](int32_t k, int c) {
            const int32_t gc = col0 + c;
            return (k < nb && gc < m) ? yb[(size_t)k * m + gc] : scalar_traits<T>::zero();
        },
        sm, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1), l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int32_t gr = row0 + wr + 16 * i + l4 + 4 * r;
            if (gr >= m) continue;
            const int32_t q = rq[gr];
            const bool is_piv = q >= kb && q < kb + nb;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int32_t gc = col0 + wc + 16 * j + l15;
                if (gc >= m || (gc >= kb && gc < kb + nb)) continue;
                if ((gc >= skip_lo && gc < skip_hi) || (only_hi > only_lo && (gc < only_lo || gc >= only_hi))) continue;
                T v;
                s_from(v, acc[0][i][j][r], acc[NPL - 1][i][j][r]);
                T* cptr = a + (size_t)gr * ld + gc;
                *cptr = is_piv ? v : s_add(*cptr, v);
            }
        }
}

// the update matrix leaves the working front for the update arena (tile = 16 rows of the b x b block)
template <typename T>
__global__ __launch_bounds__(256) void nd_save_update_kernel(const int32_t* __restrict__ tiles, const NdNodeDev* __restrict__ nodes,
                                                             const T* __restrict__ front, T* __restrict__ upd) {
    const int32_t t = tiles[2 * blockIdx.x], r0 = tiles[2 * blockIdx.x + 1];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, f = nd.f, b = f - m;
    const int32_t r = r0 + (threadIdx.x >> 4);
    if (r >= nd.brow) return;  // (this rank's rows of the update matrix: all b of them unless the node is distributed)
    const T* src = front + nd.front_off + (size_t)(m + r) * f + m;
    T* dst = upd + nd.upd_off + (size_t)r * b;
    for (int32_t c = threadIdx.x & 15; c < b; c += 16) dst[c] = src[c];
}

// sum over LPR consecutive lanes (4, 16 or 64), returned to every one of them.  DPP moves inside a row of 16 lanes (two 32-bit
// halves per double), the four row sums of a wavefront through SGPRs: a ds_bpermute butterfly is a chain of ~100-cycle steps,
// and the sweeps are chains of short kernels that end in exactly this reduction.  Fixed order: bitwise repeatable.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bits & 0xFFFFFFFFull), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)bits >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <int LPR>
__device__ __forceinline__ double lanes_sum(double v) {
    static_assert(LPR == 4 || LPR == 16 || LPR == 64, "sub-wave width");
    v += dpp_mov_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);  // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
    if constexpr (LPR >= 16) {
        v += dpp_mov_f64<0x141>(v);  // row_half_mirror
        v += dpp_mov_f64<0x140>(v);  // row_mirror: every lane of a row of 16 holds the row's sum
    }
    if constexpr (LPR == 64) {
        const long long bits = __double_as_longlong(v);
        const int lo = (int)(unsigned)((unsigned long long)bits & 0xFFFFFFFFull), hi = (int)(unsigned)((unsigned long long)bits >> 32);
        double tot = 0.0;
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 16 * row), h = (unsigned)__builtin_amdgcn_readlane(hi, 16 * row);
            tot += __longlong_as_double((long long)(((unsigned long long)h << 32) | l));
        }
        v = tot;
    }
    return v;
}
template <int LPR>
__device__ __forceinline__ cplx lanes_sum(cplx v) {
    return cplx{lanes_sum<LPR>(v.re), lanes_sum<LPR>(v.im)};
}

// acc0 += Fa[0:cn] . vs, acc1 += Fb[0:cn] . vs over the LPR lanes of a sub-wave; eight row loads in flight per lane (the
// sweeps are chains of short kernels: what they wait for is memory latency, not bandwidth)
template <int LPR = 16, typename MT, typename VT>
__device__ __forceinline__ void two_row_dot(const MT* __restrict__ Fa, const MT* __restrict__ Fb, const VT* vs, int32_t cn, int sl, VT& acc0,
                                            VT& acc1) {
    int32_t k = sl;
    for (; k + 3 * LPR < cn; k += 4 * LPR) {
        const MT a0 = Fa[k], a1 = Fa[k + LPR], a2 = Fa[k + 2 * LPR], a3 = Fa[k + 3 * LPR];
        const MT b0 = Fb[k], b1 = Fb[k + LPR], b2 = Fb[k + 2 * LPR], b3 = Fb[k + 3 * LPR];
        fma_acc(acc0, a0, vs[k]);
        fma_acc(acc1, b0, vs[k]);
        fma_acc(acc0, a1, vs[k + LPR]);
        fma_acc(acc1, b1, vs[k + LPR]);
        fma_acc(acc0, a2, vs[k + 2 * LPR]);
        fma_acc(acc1, b2, vs[k + 2 * LPR]);
        fma_acc(acc0, a3, vs[k + 3 * LPR]);
        fma_acc(acc1, b3, vs[k + 3 * LPR]);
    }
    for (; k < cn; k += LPR) {
        const MT a0 = Fa[k], b0 = Fb[k];
        fma_acc(acc0, a0, vs[k]);
        fma_acc(acc1, b0, vs[k]);
    }
}

// the first 4 * LPR columns of a row pair, loaded before the vector they multiply is ready (they depend on the node record only)
template <int LPR, typename MT>
__device__ __forceinline__ void row_pair_prefetch(const MT* __restrict__ Fa, const MT* __restrict__ Fb, int32_t cn, int sl, MT (&pa)[4], MT (&pb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t k = sl + q * LPR;
        pa[q] = k < cn ? Fa[k] : scalar_traits<MT>::zero();
        pb[q] = k < cn ? Fb[k] : scalar_traits<MT>::zero();
    }
}

template <int LPR, typename MT, typename VT>
__device__ __forceinline__ void two_row_dot_prefetched(const MT* __restrict__ Fa, const MT* __restrict__ Fb, const VT* vs, int32_t cn, int sl, VT& acc0,
                                                       VT& acc1, const MT (&pa)[4], const MT (&pb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t k = sl + q * LPR;
        if (k < cn) {
            fma_acc(acc0, pa[q], vs[k]);
            fma_acc(acc1, pb[q], vs[k]);
        }
    }
    if (cn > 4 * LPR) two_row_dot<LPR>(Fa + 4 * LPR, Fb + 4 * LPR, vs + 4 * LPR, cn - 4 * LPR, sl, acc0, acc1);
}

// PULL form: sum of the children's update-vector entries that land on front position j, through the per-child gather rows
// (fixed order: child rank).  Used where a child's vector arrives by all-gather (the replicated top of a forest cut over ranks)
// and by the transposed sweeps.
template <typename VT>
__device__ __forceinline__ VT gather_updates(const int32_t* __restrict__ ge, int32_t nchild, int32_t f, int32_t j, const VT* __restrict__ ubuf,
                                             VT v) {
    int32_t c = 0;
    for (; c + 3 < nchild; c += 4) {
        const int32_t g0 = ge[(size_t)c * f + j], g1 = ge[(size_t)(c + 1) * f + j], g2 = ge[(size_t)(c + 2) * f + j], g3 = ge[(size_t)(c + 3) * f + j];
        const VT u0 = g0 >= 0 ? ubuf[g0] : scalar_traits<VT>::zero(), u1 = g1 >= 0 ? ubuf[g1] : scalar_traits<VT>::zero();
        const VT u2 = g2 >= 0 ? ubuf[g2] : scalar_traits<VT>::zero(), u3 = g3 >= 0 ? ubuf[g3] : scalar_traits<VT>::zero();
        v = s_add(s_add(s_add(s_add(v, u0), u1), u2), u3);
    }
    int32_t g[3] = {-1, -1, -1};
    for (int q = 0; q < 3; ++q)
        if (c + q < nchild) g[q] = ge[(size_t)(c + q) * f + j];
    VT u[3];
    for (int q = 0; q < 3; ++q) u[q] = g[q] >= 0 ? ubuf[g[q]] : scalar_traits<VT>::zero();
    for (int q = 0; q < 3; ++q)
        if (c + q < nchild) v = s_add(v, u[q]);
    return v;
}

// PUSH form: the same sum from the node's slot rows (row c = what child c added to every front position; slots no child maps
// to were zeroed once and are never written): contiguous loads, no index in between.  Same order of additions as the pull form.
template <typename VT>
__device__ __forceinline__ VT slot_sum(const VT* __restrict__ slots, int32_t nchild, int32_t f, int32_t j, VT v) {
    int32_t c = 0;
    for (; c + 3 < nchild; c += 4) {
        const VT u0 = slots[(size_t)c * f + j], u1 = slots[(size_t)(c + 1) * f + j], u2 = slots[(size_t)(c + 2) * f + j],
                 u3 = slots[(size_t)(c + 3) * f + j];
        v = s_add(s_add(s_add(s_add(v, u0), u1), u2), u3);
    }
    VT u[3];
    for (int q = 0; q < 3; ++q) u[q] = c + q < nchild ? slots[(size_t)(c + q) * f + j] : scalar_traits<VT>::zero();
    for (int q = 0; q < 3; ++q)
        if (c + q < nchild) v = s_add(v, u[q]);
    return v;
}

// downward sweep: the value of front position j goes into the boundary vector of every child that has j in its boundary
template <typename VT>
__device__ __forceinline__ void push_down(const int32_t* __restrict__ ge, int32_t nchild, int32_t f, int32_t j, VT* __restrict__ xb, VT val) {
    for (int32_t c = 0; c < nchild; ++c) {
        const int32_t g = ge[(size_t)c * f + j];
        if (g >= 0) xb[g] = val;
    }
}

// One tile of the upward sweep: 512 / LPR rows from r0 of node nd's packed L block.  LPR lanes run along a pair of rows: 16 (32
// rows per workgroup) where the level has many tiles, 64 (8 rows) near the top of the tree, where a few tall fronts must still be
// spread over the whole chip, 4 (128 rows) on levels of thin separators.
// ORDERED: the vectors are in elimination order (own unknown r of the node = own0 + r), else through idx.
// A root (no boundary) also starts the downward sweep: its rows are final, they go to its children's boundary vectors.
template <typename MT, typename VT, int LPR, bool ORDERED>
__device__ __forceinline__ void nd_fwd_tile(const NdSweepNode& nd, int32_t r0, VT* vs, const MT* __restrict__ lfac, const int32_t* __restrict__ idx,
                                            const int32_t* __restrict__ gell, const int32_t* __restrict__ cmap, const VT* __restrict__ rhs,
                                            VT* __restrict__ x, VT* __restrict__ ubuf, VT* __restrict__ acc, VT* __restrict__ xb) {
    const int32_t m = nd.m, f = nd.f;
    // rows of the packed L on this rank: its own rows of the inverse, then its boundary rows (m and f - m of them unless the node
    // is distributed: then orows rows from orow0 and brow rows from brow0)
    const int32_t mr = nd.orows, floc = mr + nd.brow;
    const int32_t* ix = idx + nd.idx_off;
    const int32_t* ge = gell + nd.ge_off;
    const MT* L = lfac + nd.lfac_off;
    const int tid = threadIdx.x, sw = tid / LPR, sl = tid % LPR;
    const int32_t ra = r0 + sw, rb = ra + 256 / LPR;
    const MT* La = L + (size_t)min(ra, floc - 1) * m;
    const MT* Lb = L + (size_t)min(rb, floc - 1) * m;
    VT acc0 = scalar_traits<VT>::zero(), acc1 = scalar_traits<VT>::zero();
    // everything that depends only on the node record is requested first: the head of the rows, where the update entries go
    MT pa[4], pb[4];
    row_pair_prefetch<LPR>(La, Lb, min(kCH, m), sl, pa, pb);
    const bool push = nd.acc_off >= 0;
    const VT* slots = acc + (push ? nd.acc_off : 0);
    int32_t ca = 0, cb = 0;
    if (sl == 0 && nd.pacc_off >= 0) {
        if (ra >= mr && ra < floc) ca = cmap[nd.cmap_off + ra - mr];
        if (rb >= mr && rb < floc) cb = cmap[nd.cmap_off + rb - mr];
    }
    // ... then what the children added to the update entries these rows produce (front position of local row r >= mr: m + brow0 + r - mr)
    VT ua = scalar_traits<VT>::zero(), ub = scalar_traits<VT>::zero();
    if (sl == 0) {
        if (ra >= mr && ra < floc) {
            const int32_t jg = m + nd.brow0 + ra - mr;
            ua = push ? slot_sum(slots, nd.nchild, f, jg, ua) : gather_updates(ge, nd.nchild, f, jg, ubuf, ua);
        }
        if (rb >= mr && rb < floc) {
            const int32_t jg = m + nd.brow0 + rb - mr;
            ub = push ? slot_sum(slots, nd.nchild, f, jg, ub) : gather_updates(ge, nd.nchild, f, jg, ubuf, ub);
        }
    }
    for (int32_t c0 = 0; c0 < m; c0 += kCH) {
        const int32_t cn = min(kCH, m - c0);
        for (int32_t j = tid; j < cn; j += 256) {
            const VT v = rhs[ORDERED ? nd.own0 + c0 + j : ix[c0 + j]];
            vs[j] = push ? slot_sum(slots, nd.nchild, f, c0 + j, v) : gather_updates(ge, nd.nchild, f, c0 + j, ubuf, v);
        }
        __syncthreads();
        if (c0 == 0) two_row_dot_prefetched<LPR>(La, Lb, vs, cn, sl, acc0, acc1, pa, pb);
        else two_row_dot<LPR>(La + c0, Lb + c0, vs, cn, sl, acc0, acc1);
        __syncthreads();
    }
    const VT s0 = lanes_sum<LPR>(acc0), s1 = lanes_sum<LPR>(acc1);
    if (sl == 0) {
        const bool root_push = f == m && !(nd.flags & 1);
        if (ra < mr) {
            x[ORDERED ? nd.own0 + nd.orow0 + ra : ix[nd.orow0 + ra]] = s0;
            if (root_push) push_down(ge, nd.nchild, f, ra, xb, s0);
        } else if (ra < floc) {
            const VT u = s_add(ua, s0);
            if (nd.pacc_off >= 0) acc[nd.pacc_off + ca] = u;
            else ubuf[nd.u_off + (ra - mr)] = u;
        }
        if (rb < mr) {
            x[ORDERED ? nd.own0 + nd.orow0 + rb : ix[nd.orow0 + rb]] = s1;
            if (root_push) push_down(ge, nd.nchild, f, rb, xb, s1);
        } else if (rb < floc) {
            const VT u = s_add(ub, s1);
            if (nd.pacc_off >= 0) acc[nd.pacc_off + cb] = u;
            else ubuf[nd.u_off + (rb - mr)] = u;
        }
    }
}

// upward sweep, one tree level: workgroup (x = node of the level, y = tile of 512 / LPR rows of its packed L block)
template <typename MT, typename VT, int LPR, bool ORDERED>
__global__ __launch_bounds__(256) void nd_fwd_kernel(const NdSweepNode* __restrict__ lnodes, const MT* __restrict__ lfac,
                                                     const int32_t* __restrict__ idx, const int32_t* __restrict__ gell,
                                                     const int32_t* __restrict__ cmap, const VT* __restrict__ rhs, VT* __restrict__ x,
                                                     VT* __restrict__ ubuf, VT* __restrict__ acc, VT* __restrict__ xb) {
    __shared__ VT vs[kCH];
    const NdSweepNode nd = lnodes[blockIdx.x];
    const int32_t r0 = (int32_t)blockIdx.y * (512 / LPR);
    if (r0 >= nd.orows + nd.brow) return;
    nd_fwd_tile<MT, VT, LPR, ORDERED>(nd, r0, vs, lfac, idx, gell, cmap, rhs, x, ubuf, acc, xb);
}

// One tile of the downward sweep: x[own] -= U x[boundary] for 512 / LPR own rows from r0; the boundary vector was filled by the
// ancestors, and this tile fills the children's: the rows it finishes, and (tile `ty` of the node's `ntile`) its share of the
// boundary entries the node received.
template <typename MT, typename VT, int LPR, bool ORDERED>
__device__ __forceinline__ void nd_bwd_tile(const NdSweepNode& nd, int32_t r0, int32_t ty, VT* vs, const MT* __restrict__ ufac,
                                            const int32_t* __restrict__ idx, const int32_t* __restrict__ gell, VT* __restrict__ x,
                                            VT* __restrict__ xb) {
    constexpr int ROWS = 512 / LPR;
    const int32_t m = nd.m, f = nd.f, b = f - m;
    const int32_t* ix = idx + nd.idx_off;
    const int32_t* ge = gell + nd.ge_off;
    const MT* U = ufac + nd.ufac_off;
    const VT* bv = xb + nd.u_off;
    const int tid = threadIdx.x, sw = tid / LPR, sl = tid % LPR;
    const int32_t ra = r0 + sw, rb = r0 + sw + 256 / LPR;
    const MT* Ua = U + (size_t)min(ra, m - 1) * b;
    const MT* Ub = U + (size_t)min(rb, m - 1) * b;
    const int32_t ia = ORDERED ? nd.own0 + min(ra, m - 1) : ix[min(ra, m - 1)], ib = ORDERED ? nd.own0 + min(rb, m - 1) : ix[min(rb, m - 1)];
    MT pa[4], pb[4];
    row_pair_prefetch<LPR>(Ua, Ub, min(kCH, b), sl, pa, pb);
    // the rows' own entries are needed only at the end: issue their loads before the sweep over the boundary
    const VT xa = x[ia], xc = x[ib];
    VT acc0 = scalar_traits<VT>::zero(), acc1 = scalar_traits<VT>::zero();
    for (int32_t c0 = 0; c0 < b; c0 += kCH) {
        const int32_t cn = min(kCH, b - c0);
        for (int32_t j = tid; j < cn; j += 256) vs[j] = bv[c0 + j];
        __syncthreads();
        if (c0 == 0) two_row_dot_prefetched<LPR>(Ua, Ub, vs, cn, sl, acc0, acc1, pa, pb);
        else two_row_dot<LPR>(Ua + c0, Ub + c0, vs, cn, sl, acc0, acc1);
        __syncthreads();
    }
    acc0 = lanes_sum<LPR>(acc0);
    acc1 = lanes_sum<LPR>(acc1);
    if (sl == 0) {
        if (ra < m) {
            const VT v = s_sub(xa, acc0);
            x[ia] = v;
            push_down(ge, nd.nchild, f, ra, xb, v);
        }
        if (rb < m) {
            const VT v = s_sub(xc, acc1);
            x[ib] = v;
            push_down(ge, nd.nchild, f, rb, xb, v);
        }
    }
    if (nd.nchild > 0) {  // the boundary entries this node received, handed on to the children whose boundaries hold them
        const int32_t ntile = (m + ROWS - 1) / ROWS;
        const int64_t total = (int64_t)nd.nchild * b;
        for (int64_t e = (int64_t)ty * 256 + tid; e < total; e += (int64_t)ntile * 256) {
            const int32_t c = (int32_t)(e / b), j = (int32_t)(e - (int64_t)c * b);
            const int32_t g = ge[(size_t)c * f + m + j];
            if (g >= 0) xb[g] = bv[j];
        }
    }
}

// downward sweep, one tree level
template <typename MT, typename VT, int LPR, bool ORDERED>
__global__ __launch_bounds__(256) void nd_bwd_kernel(const NdSweepNode* __restrict__ lnodes, const MT* __restrict__ ufac,
                                                     const int32_t* __restrict__ idx, const int32_t* __restrict__ gell, VT* __restrict__ x,
                                                     VT* __restrict__ xb) {
    __shared__ VT vs[kCH];
    const NdSweepNode nd = lnodes[blockIdx.x];
    const int32_t r0 = (int32_t)blockIdx.y * (512 / LPR);
    if (r0 >= nd.m || nd.f == nd.m) return;
    nd_bwd_tile<MT, VT, LPR, ORDERED>(nd, r0, (int32_t)blockIdx.y, vs, ufac, idx, gell, x, xb);
}

// ---- downward sweep of DISTRIBUTED top nodes: a rank finishes its slice of the node's own rows (nd_bwd_kernel on a record
// that describes the slice), the slices are exchanged through the own-row buffer (pack, one in-place all-gather per level), and
// this kernel completes x and fills the boundary vectors of the node's children: entry k of child c's boundary is the parent's
// front position cmap_c[k] -- one of the parent's own rows (from the exchange buffer) or one of its boundary entries.
// grid: (distributed node of the level, 0 = the node's own rows / 1 + child, tile of 256 entries)
template <typename VT, bool ORDERED>
__global__ __launch_bounds__(256) void nd_dist_pack_kernel(const int32_t* __restrict__ dnodes, const NdNodeDev* __restrict__ nodes,
                                                           const int32_t* __restrict__ idx, int32_t rank, const VT* __restrict__ x, VT* __restrict__ xg) {
    const NdNodeDev nd = nodes[dnodes[blockIdx.x]];
    const int32_t r = (int32_t)blockIdx.y * 256 + threadIdx.x;
    if (r >= nd.orows) return;
    const int32_t j = nd.orow0 + r;
    xg[nd.xg_base + (int64_t)rank * nd.xg_stride + r] = x[ORDERED ? nd.own0 + j : idx[nd.idx_off + j]];
}

template <typename VT, bool ORDERED>
__global__ __launch_bounds__(256) void nd_dist_unpack_kernel(const int32_t* __restrict__ dnodes, const NdNodeDev* __restrict__ nodes,
                                                             const int32_t* __restrict__ child_ptr, const int32_t* __restrict__ child_idx,
                                                             const int32_t* __restrict__ cmap, const int32_t* __restrict__ idx, int32_t nranks,
                                                             VT* x, const VT* __restrict__ xg, VT* xb) {
    const int32_t t = dnodes[blockIdx.x];
    const NdNodeDev nd = nodes[t];
    const int32_t m = nd.m, b = nd.f - m;
    const int32_t ms = (m + nranks - 1) / nranks;
    const int32_t i = (int32_t)blockIdx.z * 256 + threadIdx.x;
    (void)b;
    auto own_val = [&](int32_t j) -> VT { return xg[nd.xg_base + (int64_t)(j / ms) * nd.xg_stride + j % ms]; };
    if (blockIdx.y == 0) {
        if (i >= m) return;
        x[ORDERED ? nd.own0 + i : idx[nd.idx_off + i]] = own_val(i);
        return;
    }
    const int32_t cp = child_ptr[t] + (int32_t)blockIdx.y - 1;
    if (cp >= child_ptr[t + 1]) return;
    const NdNodeDev nc = nodes[child_idx[cp]];
    if (i >= nc.f - nc.m) return;
    const int32_t p = cmap[nc.cmap_off + i];
    xb[nc.u_off + i] = p < m ? own_val(p) : xb[nd.u_off + (p - m)];
}

// ---- sweeps of the transposed / conjugate-transposed system on the same factors (the adjoint eigenproblem of
// Sensitivity/__init__.py:230-311 needs (A - sigma M)^-H without a second factorisation) --------------------------------------
// C^T has the fronts F^T, so with the stored blocks  inv = F11^-1, S1 = -F21 inv, S2 = inv F12:
//     up:    z = [inv | S2]^T v   (the columns of the packed inv and U blocks);  y[own] = z[:m];  update = v_B - z[m:]
//     down:  x[own] = y[own] + S1^T x[boundary]
// Column access of row-major blocks: 64 lanes run along a row (coalesced), four slices of the rows per workgroup.  These
// sweeps keep the pull form (gather rows + update vectors) and address the vectors through idx.
template <bool CONJ, typename MT>
__device__ __forceinline__ MT maybe_conj(MT a) {
    if constexpr (CONJ) return s_conj(a);
    else return a;
}

template <typename MT, typename VT, bool CONJ, bool DOWN>
__global__ __launch_bounds__(256) void nd_sweepT_kernel(const NdSweepNode* __restrict__ lnodes, const MT* __restrict__ lfac, const MT* __restrict__ ufac,
                                                        const int32_t* __restrict__ idx, const int32_t* __restrict__ gell,
                                                        const VT* __restrict__ rhs, VT* __restrict__ x, VT* __restrict__ ubuf,
                                                        const int64_t* __restrict__ tgoff, VT* __restrict__ pz, int64_t tg_slot, int32_t rank) {
    __shared__ VT vs[kCH];
    __shared__ VT part[4][64];
    const NdSweepNode nd = lnodes[blockIdx.x];
    const int32_t m = nd.m, f = nd.f, b = f - m;
    const bool dist = (nd.flags & 1) != 0;  // a distributed node: this rank's rows only, the sums go to its slot of the partial buffer
    const int32_t ncols = DOWN ? m : f;                         // outputs of this sweep
    const int32_t klo = DOWN ? nd.brow0 : nd.orow0;             // rows summed over: this rank's (all of them unless the node is distributed)
    const int32_t K = DOWN ? nd.brow : nd.orows;
    const int32_t c0 = (int32_t)blockIdx.y * 64;
    if (c0 >= ncols || (DOWN && b == 0)) return;
    const int32_t* ix = idx + nd.idx_off;
    const int32_t* ge = gell + nd.ge_off;
    const int tid = threadIdx.x, lane = tid & 63, sl = tid >> 6;
    const int32_t col = min(c0 + lane, ncols - 1);
    // this lane's column: base pointer and row stride inside the packed blocks (rows = the rank's rows, local numbering)
    const MT* Fc;
    int32_t ld;
    if (DOWN) Fc = lfac + nd.lfac_off + (size_t)nd.orows * m + col, ld = m;
    else if (col < m) Fc = lfac + nd.lfac_off + col, ld = m;
    else Fc = ufac + nd.ufac_off + (col - m), ld = b;
    VT acc = scalar_traits<VT>::zero();
    for (int32_t k0 = 0; k0 < K; k0 += kCH) {
        const int32_t kn = min(kCH, K - k0);
        for (int32_t j = tid; j < kn; j += 256) {
            if (DOWN) vs[j] = x[ix[m + klo + k0 + j]];
            else vs[j] = gather_updates(ge, nd.nchild, f, klo + k0 + j, ubuf, rhs[ix[klo + k0 + j]]);
        }
        __syncthreads();
        const MT* Fk = Fc + (size_t)k0 * ld;
        int32_t k = sl;
        for (; k + 12 < kn; k += 16) {
            const MT a0 = Fk[(size_t)k * ld], a1 = Fk[(size_t)(k + 4) * ld], a2 = Fk[(size_t)(k + 8) * ld], a3 = Fk[(size_t)(k + 12) * ld];
            fma_acc(acc, maybe_conj<CONJ>(a0), vs[k]);
            fma_acc(acc, maybe_conj<CONJ>(a1), vs[k + 4]);
            fma_acc(acc, maybe_conj<CONJ>(a2), vs[k + 8]);
            fma_acc(acc, maybe_conj<CONJ>(a3), vs[k + 12]);
        }
        for (; k < kn; k += 4) fma_acc(acc, maybe_conj<CONJ>(Fk[(size_t)k * ld]), vs[k]);
        __syncthreads();
    }
    part[sl][lane] = acc;
    __syncthreads();
    if (sl == 0 && c0 + lane < ncols) {
        const VT z = s_add(s_add(part[0][lane], part[1][lane]), s_add(part[2][lane], part[3][lane]));
        const int32_t r = c0 + lane;
        if (dist) pz[(int64_t)rank * tg_slot + tgoff[blockIdx.x] + r] = z;  // summed over the ranks by nd_distT_finish_kernel
        else if (DOWN) x[ix[r]] = s_add(x[ix[r]], z);
        else if (r < m) x[ix[r]] = z;
        else ubuf[nd.u_off + (r - m)] = s_sub(gather_updates(ge, nd.nchild, f, r, ubuf, scalar_traits<VT>::zero()), z);
    }
}

// The distributed nodes of a level after the exchange of their partial sums: z = the ranks' partials added in rank order (the same
// bits on every rank), then what nd_sweepT_kernel does for an ordinary node.  Upwards the node's update vector is written in the
// layout its parent's gather rows expect: entry j in the slot of the rank that owns boundary row j in the forward sweeps.
template <typename VT, bool DOWN>
__global__ __launch_bounds__(256) void nd_distT_finish_kernel(const NdSweepNode* __restrict__ lnodes, const int64_t* __restrict__ tgoff,
                                                              const int32_t* __restrict__ idx, const int32_t* __restrict__ gell,
                                                              const VT* __restrict__ pz, int64_t tg_slot, int32_t nranks, int32_t rank, int64_t ux_slot,
                                                              VT* __restrict__ x, VT* __restrict__ ubuf) {
    const NdSweepNode nd = lnodes[blockIdx.x];
    if (!(nd.flags & 1)) return;
    const int32_t m = nd.m, f = nd.f, b = f - m;
    const int32_t ncols = DOWN ? m : f;
    const int32_t r = (int32_t)blockIdx.y * 256 + threadIdx.x;
    if (r >= ncols || (DOWN && b == 0)) return;
    const int64_t off = tgoff[blockIdx.x] + r;
    VT z = scalar_traits<VT>::zero();
    for (int32_t p = 0; p < nranks; ++p) z = s_add(z, pz[(int64_t)p * tg_slot + off]);
    const int32_t* ix = idx + nd.idx_off;
    if (DOWN) {
        x[ix[r]] = s_add(x[ix[r]], z);
    } else if (r < m) {
        x[ix[r]] = z;
    } else {
        const int32_t j = r - m, w = (b + nranks - 1) / nranks, owner = j / w;
        const VT u = s_sub(gather_updates(gell + nd.ge_off, nd.nchild, f, r, ubuf, scalar_traits<VT>::zero()), z);
        ubuf[nd.u_off + (int64_t)(owner - rank) * ux_slot + (j - owner * w)] = u;
    }
}

}  // namespace

struct lsa_ndlu {
    lsa_ctx* ctx = nullptr;
    NdSymbolic S;
    int dtype = LSA_C128;
    bool ordered = false;           // the matrix came in elimination order: own unknown r of a node is own0 + r
    std::vector<NdChunk> chunks;    // factorisation order
    std::vector<NdLevel> levels;    // sweep order
    NdNodeDev* d_nodes = nullptr;                        // by node id (factorisation, exchange kernels of the sweeps)
    NdSweepNode *d_lnodes = nullptr, *d_lnodes_bwd = nullptr;  // in lvl_nodes order: the sweeps' records (downwards a distributed node appears as its slice of own rows)
    int32_t *d_dist_nodes = nullptr, *d_child_ptr = nullptr, *d_child_idx = nullptr;  // distributed nodes by level; children of every node
    void *d_xstage = nullptr, *d_xg = nullptr;          // staging of update rows on their way to distributed parents; own-row exchange buffer of the sweeps
    // transposed sweeps with distributed nodes (built by the first adjoint solve): per sweep record its offset in a rank's slot of the
    // partial-sum buffer (-1: not distributed), the buffer (nranks slots of tg_slot_max vector scalars), the slot size of every level
    int64_t* d_tgoff = nullptr;
    void* d_tg = nullptr;
    std::vector<int64_t> tg_slot;
    int64_t tg_slot_max = 0;
    int64_t xstage_slot = 0;                            // scalars per rank of d_xstage
    std::vector<int64_t> h_upd_off, h_lfac_off;         // per node: its update matrix in the update arena, its packed L (host copies of the plan)
    int64_t chunk_node_upd_off(int32_t t) const { return h_upd_off[(size_t)t]; }
    int32_t* d_gell = nullptr;
    int32_t *d_idx = nullptr, *d_cmap = nullptr, *d_tiles = nullptr, *d_chunk_nodes = nullptr;
    int64_t* d_asm_dst = nullptr;
    int32_t* d_asm_src = nullptr;
    int64_t asm_count = 0;
    int32_t *d_ipiv = nullptr, *d_rowq = nullptr, *d_flag = nullptr, *d_xflag = nullptr;
    unsigned long long* d_maxabs = nullptr;
    void *d_lfac = nullptr, *d_ufac = nullptr;  // packed factors (resident)
    void *d_work = nullptr, *d_upd = nullptr;   // working fronts of one chunk; live update matrices
    void *d_ubuf = nullptr, *d_xb = nullptr, *d_acc = nullptr;  // sweeps: update vectors (pull form), boundary vectors, slot rows (push form)
    void *d_tmp = nullptr, *d_ybuf = nullptr;
    int64_t lfac_entries = 0, ufac_entries = 0, work_entries = 0, upd_entries = 0, acc_entries = 0;
    int64_t xupd_slot = 0;                     // subtree-parallel: scalars per rank in the exchange region at the start of the update arena
    int32_t *d_cand[2] = {nullptr, nullptr};  // tournament pivoting: candidate rows, two buffers used in turn
    void* d_dinv = nullptr;                    // ... the inverted pivot tile of every node of the chunk being eliminated
    int32_t tp_min = 1 << 30;                  // chunks whose tallest pivot block has at least this many rows use it
    int32_t sb_min = 1 << 30;                  // ... and from this many rows on, with super-blocks of kSB columns (nd_gj_update_kernel)
    int32_t sb_cols = 128;                     // columns of a super-block (kSB; LSA_ND_SB_COLS, a multiple of 64, for measurements)
    int32_t ycap = 32;                         // rows of d_ybuf per unknown: kNB, or sb_cols when a chunk works in super-blocks
    hipStream_t side = nullptr;                // ... the next block's tournament runs here, under the current block's update
    hipEvent_t ev_panel = nullptr, ev_pivots = nullptr;
    double seconds_analyse = 0.0, seconds_numeric = 0.0;
    int32_t solve_launches = 0;
    int acc_vbytes = 0;  // scalar size of the vectors the slot rows were last used with (their never-written entries must read zero)
};

namespace {

void nd_free(lsa_ndlu* f) {
    if (!f) return;
    for (void* p : {(void*)f->d_lnodes_bwd, (void*)f->d_dist_nodes, (void*)f->d_child_ptr, (void*)f->d_child_idx, f->d_xstage, f->d_xg, (void*)f->d_tgoff, f->d_tg})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)f->d_nodes, (void*)f->d_lnodes, (void*)f->d_gell, (void*)f->d_idx, (void*)f->d_cmap, (void*)f->d_tiles,
                    (void*)f->d_chunk_nodes, (void*)f->d_asm_dst, (void*)f->d_asm_src, (void*)f->d_ipiv, (void*)f->d_rowq, (void*)f->d_flag, (void*)f->d_xflag, (void*)f->d_maxabs,
                    f->d_lfac, f->d_ufac, f->d_work, f->d_upd, f->d_ubuf, f->d_xb, f->d_acc, f->d_tmp, f->d_ybuf, (void*)f->d_cand[0], (void*)f->d_cand[1], f->d_dinv})
        if (p) (void)hipFree(p);
    if (f->ev_panel) (void)hipEventDestroy(f->ev_panel);
    if (f->ev_pivots) (void)hipEventDestroy(f->ev_pivots);
    if (f->side) (void)hipStreamDestroy(f->side);
    delete f;
}

template <typename U>
int upload(lsa_ctx* ctx, const std::vector<U>& h, U** d) {
    const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(U);
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)d, bytes));
    if (!h.empty()) LSA_HIP_CHECK(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice));
    return LSA_OK;
}

// device tables, the memory plan (packed factors, chunks of working fronts, update arena) and tile lists from the analysis
// free_agreed: > 0 = the device memory every rank of a forest cut over ranks has free (the smallest of them): with distributed
// top nodes the chunks of the top levels carry collectives and must come out alike on every rank
int nd_setup(lsa_ctx* ctx, lsa_ndlu* f, int64_t free_agreed = 0) {
    NdSymbolic& S = f->S;
    const int32_t nt = S.nt;
    const size_t es = esize(f->dtype);
    const bool dist = S.nranks > 1;
    {
        const char* e = getenv("LSA_ND_TP_MIN");
        f->tp_min = e && *e ? std::max(1, atoi(e)) : 512;  // (measured: S500k 46.4 ms at 384, 44.5 at 512, 47.2 at 640, 62.6 at 1024; 3D cases indifferent)
        const char* sbm = getenv("LSA_ND_SB_MIN");
        f->sb_min = std::max(f->tp_min, sbm && *sbm ? std::max(1, atoi(sbm)) : 1024);
        f->ycap = kNB;
        const char* sbc = getenv("LSA_ND_SB_COLS");
        f->sb_cols = sbc && *sbc ? std::min(1024, std::max(kGT, atoi(sbc) / kGT * kGT)) : kSB;
    }
    // (the status word of the subtree-parallel form first: ranks agree on a failed set-up through it, lsa_ndlu_create_tree)
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_xflag, 4 * sizeof(int32_t) * (size_t)std::max(1, S.nranks)));
    f->ordered = true;
    for (size_t k = 0; k < S.perm.size() && f->ordered; ++k) f->ordered = S.perm[k] == (int32_t)k || dist;
    if (dist)  // a tree given by the caller owns contiguous index ranges: own0 = the first of them (the padded layout has holes)
        for (int32_t t = 0; t < nt && f->ordered; ++t)
            for (int32_t r = 1; r < S.m[(size_t)t]; ++r)
                if (S.idx[(size_t)S.idx_off[(size_t)t] + r] != S.idx[(size_t)S.idx_off[(size_t)t]] + r) {
                    f->ordered = false;
                    break;
                }
    std::vector<NdNodeDev> nodes((size_t)nt);
    std::vector<int32_t> tiles;
    auto begin_list = [&](TileList& tl) { tl.off = (int64_t)tiles.size() / 2; tl.count = 0; };
    auto push = [&](TileList& tl, int32_t a, int32_t b) {
        tiles.push_back(a);
        tiles.push_back(b);
        ++tl.count;
    };
    // ---- the memory plan (nd_symbolic.hip: host arithmetic on the analysis, also what lsa_nd_sym_memory reports) ----
    int64_t budget = 0;
    {
        // a level is factored in one go while its fronts fit a quarter of what the packed factors leave free (LSA_ND_WORK_MB
        // overrides); a single front always has to fit
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)16 << 30;
        const int64_t after = (int64_t)free_b - S.factor_entries * (int64_t)es;
        budget = std::max<int64_t>(after / 4, (int64_t)256 << 20) / (int64_t)es;
        // (distributed top nodes: a figure every rank computes alike -- a sixth of the smallest free memory of all ranks)
        if (S.has_dist) budget = std::max<int64_t>((free_agreed > 0 ? free_agreed : (int64_t)free_b) / 6, (int64_t)256 << 20) / (int64_t)es;
        if (const char* e = getenv("LSA_ND_WORK_MB")) budget = std::max<int64_t>(atoll(e), 1) * (1 << 20) / (int64_t)es;
    }
    if (getenv("LSA_ND_TEST_OOM"))  // test aid: rehearses the one failure the operator layer answers by a leaner method
        return lsa_set_error(ctx, LSA_ERR_OOM, "lsa_ndlu: out of device memory (forced by LSA_ND_TEST_OOM)");
    NdMemoryPlan P;
    nd_memory_plan(S, budget, P);
    for (int32_t t = 0; t < nt; ++t) {  // every kept node, the other ranks' subtree roots included (they are children here)
        NdNodeDev& nd = nodes[(size_t)t];
        nd.front_off = P.work_off[(size_t)t];
        nd.lfac_off = P.lfac_off[(size_t)t];
        nd.ufac_off = P.ufac_off[(size_t)t];
        nd.upd_off = P.upd_off[(size_t)t];
        nd.u_off = S.u_off[(size_t)t];
        nd.ge_off = S.ge_off[(size_t)t];
        nd.acc_off = P.acc_off[(size_t)t];
        nd.pacc_off = P.pacc_off[(size_t)t];
        nd.idx_off = (int32_t)S.idx_off[(size_t)t];
        nd.cmap_off = S.cmap_off[(size_t)t];
        nd.piv_off = S.piv_off[(size_t)t];
        nd.own0 = S.m[(size_t)t] > 0 ? S.idx[(size_t)S.idx_off[(size_t)t]] : 0;
        nd.m = S.m[(size_t)t];
        nd.f = S.f[(size_t)t];
        nd.parent = S.parent[(size_t)t];
        nd.nchild = S.child_ptr[(size_t)t + 1] - S.child_ptr[(size_t)t];
        nd.brow0 = S.brow0[(size_t)t];
        nd.brow = S.brow[(size_t)t];
        nd.orow0 = S.orow0[(size_t)t];
        nd.orows = S.orows[(size_t)t];
        nd.flags = S.kind[(size_t)t] == 4 ? 1 : 0;
        nd.pad0 = 0;
        nd.xg_base = S.xg_base[(size_t)t];
        nd.xg_stride = S.xg_stride[(size_t)t];
        // (the plan's working block of a distributed node: its slice of the front at the widest slice of any rank, then the inverse)
        nd.inv_off = S.kind[(size_t)t] == 4
                         ? nd.front_off + (int64_t)(S.m[(size_t)t] + nd_slice_width(S.f[(size_t)t] - S.m[(size_t)t], S.nranks)) * S.f[(size_t)t]
                         : -1;
    }
    f->xstage_slot = P.xstage_slot;
    f->h_upd_off = P.upd_off;
    f->h_lfac_off = P.lfac_off;
    f->lfac_entries = P.lfac_entries;
    f->ufac_entries = P.ufac_entries;
    f->acc_entries = P.acc_entries;
    f->work_entries = P.work_entries;
    f->upd_entries = P.upd_entries;
    f->xupd_slot = P.xupd_slot;
    f->chunks.clear();
    const std::vector<int32_t>& chunk_nodes = S.lvl_nodes;  // node ids in chunk order: the chunks cut the levels' lists
    std::vector<int32_t> chunk_of((size_t)nt, -1);
    for (size_t c = 0; c + 1 < P.chunk_begin.size(); ++c) {
        NdChunk ch;
        ch.node_begin = P.chunk_begin[c];
        ch.node_count = P.chunk_begin[c + 1] - P.chunk_begin[c];
        ch.work_entries = P.chunk_work[c];
        ch.exchange_before = P.chunk_exchange_before[c] != 0;
        for (int32_t q = 0; q < ch.node_count; ++q) {
            const int32_t t = chunk_nodes[(size_t)ch.node_begin + q];
            ch.max_m = std::max(ch.max_m, S.m[(size_t)t]);
            ch.max_f = std::max(ch.max_f, S.f[(size_t)t]);
            ch.sorted_m.push_back(S.m[(size_t)t]);
            chunk_of[(size_t)t] = (int32_t)c;
        }
        f->chunks.push_back(std::move(ch));
    }
    // ---- assembly lists by chunk: front_buffer[work offset] = values[asm_src] ----
    {
        std::vector<std::pair<int64_t, int32_t>> by_off;  // (offset of the node's front in the analysis' logical layout, node)
        by_off.reserve((size_t)nt);
        for (int32_t t = 0; t < nt; ++t) by_off.emplace_back(S.front_off[(size_t)t], t);
        std::sort(by_off.begin(), by_off.end());
        const size_t ne = S.asm_src.size();
        std::vector<int32_t> node_of_entry(ne);
        std::vector<int64_t> count(f->chunks.size() + 1, 0);
        for (size_t e = 0; e < ne; ++e) {
            auto it = std::upper_bound(by_off.begin(), by_off.end(), std::make_pair(S.asm_dst[e], (int32_t)0x7fffffff));
            const int32_t t = (it - 1)->second;
            node_of_entry[e] = t;
            ++count[(size_t)chunk_of[(size_t)t] + 1];
        }
        for (size_t c = 0; c < f->chunks.size(); ++c) {
            f->chunks[c].asm_begin = count[c];
            f->chunks[c].asm_count = count[c + 1];
            count[c + 1] += count[c];
        }
        std::vector<int32_t> src(ne);
        std::vector<int64_t> dst(ne);
        std::vector<int64_t> fill(count.begin(), count.end() - 1);
        for (size_t e = 0; e < ne; ++e) {
            const int32_t t = node_of_entry[e];
            const int64_t at = fill[(size_t)chunk_of[(size_t)t]]++;
            src[(size_t)at] = S.asm_src[e];
            dst[(size_t)at] = nodes[(size_t)t].front_off + (S.asm_dst[e] - S.front_off[(size_t)t]);
        }
        f->asm_count = (int64_t)ne;
        LSA_CHECK(upload(ctx, dst, &f->d_asm_dst));
        LSA_CHECK(upload(ctx, src, &f->d_asm_src));
        // (the analysis' own copies are not needed again: the refactorisations walk the device lists)
        std::vector<int32_t>().swap(S.asm_src);
        std::vector<int64_t>().swap(S.asm_dst);
    }
    // ---- tile lists of the chunks ----
    int32_t widest_tp = 0;
    for (NdChunk& c : f->chunks) {
        auto node = [&](int32_t q) { return chunk_nodes[(size_t)c.node_begin + q]; };
        int32_t lvl_children = 0;
        for (int32_t q = 0; q < c.node_count; ++q) lvl_children = std::max(lvl_children, S.child_ptr[(size_t)node(q) + 1] - S.child_ptr[(size_t)node(q)]);
        // extend-add: one list per child rank (children of one parent never share a launch)
        c.ext.assign((size_t)lvl_children, TileList());
        for (int32_t r = 0; r < lvl_children; ++r) {
            begin_list(c.ext[(size_t)r]);
            for (int32_t q = 0; q < c.node_count; ++q) {
                const int32_t t = node(q);
                if (S.child_ptr[(size_t)t] + r >= S.child_ptr[(size_t)t + 1]) continue;
                const int32_t ch = S.child_idx[(size_t)S.child_ptr[(size_t)t] + r];
                // (the update matrices of a distributed node's children travel through the staging buffer, below -- except a
                //  replicated child's, which every rank holds whole)
                if (S.kind[(size_t)t] == 4 && S.kind[(size_t)ch] != 2) continue;
                const int32_t bc = S.f[(size_t)ch] - S.m[(size_t)ch];
                for (int32_t i0 = 0; i0 < bc; i0 += 16) push(c.ext[(size_t)r], ch, i0);
            }
        }
        begin_list(c.unperm);
        for (int32_t q = 0; q < c.node_count; ++q)
            for (int32_t r0 = 0; r0 < S.m[(size_t)node(q)]; r0 += 16) push(c.unperm, node(q), r0);
        static const bool xcd_order = !(getenv("LSA_ND_XCD_ORDER") && atoi(getenv("LSA_ND_XCD_ORDER")) == 0);  // (A/B measurement aid)
        for (int kind = 0; kind < 3; ++kind) {
            begin_list(c.gemm[kind]);
            for (int32_t q = 0; q < c.node_count; ++q) {
                const int32_t t = node(q);
                const int32_t m = S.m[(size_t)t], b = S.f[(size_t)t] - m;
                if (b == 0) continue;
                // (a distributed node: this rank's boundary rows of L and of the update matrix, its own rows of U)
                const int32_t M = kind == 2 ? S.orows[(size_t)t] : S.brow[(size_t)t], N = kind == 0 ? m : b;
                if (M == 0) continue;
                const int32_t TM = (M + kGT - 1) / kGT, TN = (N + kGT - 1) / kGT;
                if ((int64_t)TM * TN < 512 || !xcd_order) {
                    for (int32_t tm = 0; tm < TM; ++tm)
                        for (int32_t tn = 0; tn < TN; ++tn) push(c.gemm[kind], t, (tm << 16) | tn);
                    continue;
                }
                // A large product: consecutive workgroups go to the 8 XCDs in turn (launch index mod 8 labels the workgroups
                // that share an L2), and row-major order would hand every L2 one tile in eight of a dozen tile rows -- each
                // workgroup then streams its own 64 x K and K x 64 panels, 8 flops per byte from beyond the L2.  Here the
                // workgroups of one label stay inside one band of tile rows and walk it in super-tiles of 8 x 16 tiles (about
                // what an XCD holds in flight): at any k they share 8 + 16 panel chunks instead of reading 2 x 128.
                std::vector<int32_t> seq[8];
                for (int x = 0; x < 8; ++x) {
                    const int32_t r0 = (int32_t)((int64_t)TM * x / 8), r1 = (int32_t)((int64_t)TM * (x + 1) / 8);
                    if (r1 <= r0) continue;
                    const int32_t sh = std::min(r1 - r0, 8), sw = std::max(1, 128 / sh);
                    for (int32_t rb = r0; rb < r1; rb += sh)
                        for (int32_t cb = 0; cb < TN; cb += sw)
                            for (int32_t r = rb; r < std::min(rb + sh, r1); ++r)
                                for (int32_t cc = cb; cc < std::min(cb + sw, TN); ++cc) seq[x].push_back((r << 16) | cc);
                }
                size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int64_t left = (int64_t)TM * TN; left > 0; --left) {
                    int x = (int)(c.gemm[kind].count % 8);
                    for (int tries = 0; tries < 8 && pos[x] >= seq[x].size(); ++tries) x = (x + 1) % 8;  // (a band ran out: help the next)
                    push(c.gemm[kind], t, seq[x][pos[x]++]);
                }
            }
        }
        begin_list(c.save);
        for (int32_t q = 0; q < c.node_count; ++q) {
            const int32_t t = node(q);
            for (int32_t r0 = 0; r0 < S.brow[(size_t)t]; r0 += 16) push(c.save, t, r0);
        }
        // the row chunks that reach this chunk's distributed nodes through the staging buffer: every rank works off its own
        // queue (its slice of the rows of a distributed child, all rows of a subtree root it owns), one piece per step
        {
            std::vector<std::vector<NdChunk::XPiece>> queue((size_t)S.nranks);
            for (int32_t q = 0; q < c.node_count; ++q) {
                const int32_t t = node(q);
                if (S.kind[(size_t)t] != 4) continue;
                for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp) {
                    const int32_t ch = S.child_idx[(size_t)cp];
                    if (S.kind[(size_t)ch] == 2) continue;
                    const int32_t bc = S.f[(size_t)ch] - S.m[(size_t)ch];
                    if (bc == 0) continue;
                    const int32_t per = (int32_t)std::max<int64_t>(1, std::min<int64_t>(P.xstage_slot / bc, 1 << 30));
                    for (int r = 0; r < S.nranks; ++r) {
                        int32_t lo = 0, cnt = 0;
                        if (S.kind[(size_t)ch] == 4) nd_slice(bc, S.nranks, r, &lo, &cnt);
                        else if (S.owner[(size_t)ch] == r) cnt = bc;
                        for (int32_t r0 = lo; r0 < lo + cnt; r0 += per) {
                            NdChunk::XPiece pc;
                            pc.child = ch, pc.row0 = r0, pc.nrows = std::min(per, lo + cnt - r0);
                            queue[(size_t)r].push_back(pc);
                        }
                    }
                }
            }
            size_t nsteps = 0;
            for (const auto& qv : queue) nsteps = std::max(nsteps, qv.size());
            c.xsteps.assign(nsteps, std::vector<NdChunk::XPiece>((size_t)S.nranks));
            for (size_t st = 0; st < nsteps; ++st)
                for (int r = 0; r < S.nranks; ++r)
                    if (st < queue[(size_t)r].size()) c.xsteps[st][(size_t)r] = queue[(size_t)r][st];
        }
        if (c.max_m > 16384 && c.max_m < f->tp_min)  // (the tournament path has no such limit)
            return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu: a pivot block of %d rows exceeds the 16 384 the panel kernels hold in registers (LSA_ND_TP_MIN = %d)",
                                 c.max_m, f->tp_min);
        if (c.max_m >= f->tp_min) widest_tp = std::max(widest_tp, c.node_count);
        if (c.max_m >= f->sb_min) f->ycap = f->sb_cols;
    }
    // ---- sweep levels ----
    f->levels.assign((size_t)S.nlevels, NdLevel());
    for (int32_t l = 0; l < S.nlevels; ++l) {
        NdLevel& L = f->levels[(size_t)l];
        L.node_begin = S.lvl_ptr[(size_t)l];
        L.node_count = S.lvl_ptr[(size_t)l + 1] - L.node_begin;
        // the sweeps wait for memory: a level whose fronts make fewer 32-row tiles than a few per CU gets 8-row tiles
        int64_t tiles32 = 0;
        for (int32_t q = 0; q < L.node_count; ++q) {
            const int32_t t = S.lvl_nodes[(size_t)L.node_begin + q];
            tiles32 += (S.orows[(size_t)t] + S.brow[(size_t)t] + kRT - 1) / kRT;
            L.max_m = std::max(L.max_m, S.m[(size_t)t]);
            L.max_f = std::max(L.max_f, S.f[(size_t)t]);
        }
        static const int64_t few = getenv("LSA_ND_SWEEP_FEW") ? atoll(getenv("LSA_ND_SWEEP_FEW")) : 4 * (int64_t)ctx->num_cu;
        // thin separators (pivot blocks of a few dozen unknowns under fronts of a few hundred rows): the upward sweep reads
        // f short rows per node; four lanes per row pair and 128 rows per workgroup gather the node's vector 1/4 as often
        static const int32_t thin = getenv("LSA_ND_SWEEP_THIN") ? atoi(getenv("LSA_ND_SWEEP_THIN")) : 64;
        // (64-row tiles -- NP = 2 of nd_fwd_kernel -- on the leaf level of the 500 k-unknown forest, 42 000 tiles: 900 us per solve
        //  against 886 with 32-row tiles: no gain from sharing the gather, measured round 3)
        L.sweep_rows = tiles32 <= few ? 8 : (L.max_m <= thin && l > 0) ? 128 : kRT;
        for (int32_t q = 0; q < L.node_count; ++q) {
            const int32_t t = S.lvl_nodes[(size_t)L.node_begin + q];
            L.fwd_tiles = std::max(L.fwd_tiles, (S.orows[(size_t)t] + S.brow[(size_t)t] + L.sweep_rows - 1) / L.sweep_rows);
            const int32_t bwd_rows = L.sweep_rows == 8 ? 8 : kRT;  // (the downward sweep of a thin level has few, long rows: 32-row tiles; 32 as well where the upward sweep takes 64)
            if (S.f[(size_t)t] > S.m[(size_t)t]) L.bwd_tiles = std::max(L.bwd_tiles, (S.orows[(size_t)t] + bwd_rows - 1) / bwd_rows);
        }
        if (L.fwd_tiles > 65535) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu: a front of more than %d rows is not supported", 65535 * 8);
    }
    LSA_CHECK(upload(ctx, nodes, &f->d_nodes));
    {
        // the sweeps' records, in level order.  A distributed node differs from its factorisation record: upwards its update
        // entries go to its rank's slot of the level's exchange region; downwards it appears as its slice of the own rows (the
        // pushes to its children are nd_dist_unpack_kernel's, after the exchange)
        std::vector<NdSweepNode> lnodes(S.lvl_nodes.size()), bnodes(S.lvl_nodes.size());
        std::vector<int32_t> dist_nodes;
        for (size_t q = 0; q < S.lvl_nodes.size(); ++q) {
            const int32_t t = S.lvl_nodes[q];
            lnodes[q] = bnodes[q] = NdSweepNode(nodes[(size_t)t]);
            if (S.kind[(size_t)t] != 4) continue;
            lnodes[q].u_off = S.ux_base[(size_t)t] + (int64_t)S.rank * S.ux_stride[(size_t)t];
            NdSweepNode& bn = bnodes[q];
            const int32_t b = bn.f - bn.m;
            bn.idx_off += bn.orow0;
            bn.own0 += bn.orow0;
            bn.m = bn.orows;
            bn.f = bn.orows + b;
            bn.nchild = 0;
        }
        for (int32_t l = 0; l < S.nlevels; ++l) {
            NdLevel& L = f->levels[(size_t)l];
            L.dist_begin = (int32_t)dist_nodes.size();
            for (int32_t q = 0; q < L.node_count; ++q) {
                const int32_t t = S.lvl_nodes[(size_t)L.node_begin + q];
                if (S.kind[(size_t)t] != 4) continue;
                if (L.dist_count == 0) {
                    L.ux_base = S.ux_base[(size_t)t], L.ux_slot = S.ux_stride[(size_t)t];
                    L.xg_base = S.xg_base[(size_t)t], L.xg_slot = S.xg_stride[(size_t)t];
                }
                L.ux_base = std::min(L.ux_base, S.ux_base[(size_t)t]);
                L.xg_base = std::min(L.xg_base, S.xg_base[(size_t)t]);
                ++L.dist_count;
                dist_nodes.push_back(t);
                L.dist_children = std::max(L.dist_children, S.child_ptr[(size_t)t + 1] - S.child_ptr[(size_t)t]);
                L.dist_rows = std::max(L.dist_rows, S.m[(size_t)t]);
                for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp) {
                    const int32_t ch = S.child_idx[(size_t)cp];
                    L.dist_rows = std::max(L.dist_rows, S.f[(size_t)ch] - S.m[(size_t)ch]);
                }
            }
        }
        LSA_CHECK(upload(ctx, lnodes, &f->d_lnodes));
        LSA_CHECK(upload(ctx, bnodes, &f->d_lnodes_bwd));
        LSA_CHECK(upload(ctx, dist_nodes, &f->d_dist_nodes));
        LSA_CHECK(upload(ctx, S.child_ptr, &f->d_child_ptr));
        LSA_CHECK(upload(ctx, S.child_idx, &f->d_child_idx));
        if (f->xstage_slot > 0) LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_xstage, (size_t)f->xstage_slot * (size_t)S.nranks * es));
        if (S.xg_entries > 0) LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_xg, (size_t)S.xg_entries * 16));
    }
    LSA_CHECK(upload(ctx, S.gell, &f->d_gell));
    LSA_CHECK(upload(ctx, S.idx, &f->d_idx));
    LSA_CHECK(upload(ctx, S.cmap, &f->d_cmap));
    LSA_CHECK(upload(ctx, chunk_nodes, &f->d_chunk_nodes));
    LSA_CHECK(upload(ctx, tiles, &f->d_tiles));
    const size_t nn = (size_t)std::max<int32_t>(S.n, 1);
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_ipiv, nn * sizeof(int32_t)));
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_rowq, nn * sizeof(int32_t)));
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_flag, 4 * sizeof(int32_t)));
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_maxabs, sizeof(unsigned long long)));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_lfac, (size_t)std::max<int64_t>(f->lfac_entries, 1) * es));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_ufac, (size_t)std::max<int64_t>(f->ufac_entries, 1) * es));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_work, (size_t)f->work_entries * es));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_upd, (size_t)f->upd_entries * es));
    const size_t ub = (size_t)std::max<int64_t>(S.u_off[(size_t)nt], 1) * 16;
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_ubuf, ub));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_xb, ub));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_acc, (size_t)std::max<int64_t>(f->acc_entries, 1) * 16));
    // slots no child maps to are read by every solve and written by none: zero, once
    LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_acc, 0, (size_t)std::max<int64_t>(f->acc_entries, 1) * 16, ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_ubuf, 0, ub, ctx->stream));
    LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_xb, 0, ub, ctx->stream));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_tmp, nn * 16));
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_ybuf, nn * (size_t)f->ycap * es));
    if (widest_tp > 0) {
        const size_t cand = ((size_t)S.n / kTRmin + (size_t)nt + 1) * kNB * sizeof(int32_t);
        LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_cand[0], cand));
        LSA_HIP_ALLOC(ctx, hipMalloc((void**)&f->d_cand[1], cand));
        LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_dinv, (size_t)widest_tp * kNB * kNB * es));
        const char* la = getenv("LSA_ND_LOOKAHEAD");
        if (!(la && *la && atoi(la) == 0)) {
            // highest priority: the tournament's few workgroups must not queue behind the thousands of the product they
            // run under
            int lo_pri = 0, hi_pri = 0;
            LSA_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
            LSA_HIP_CHECK(ctx, hipStreamCreateWithPriority(&f->side, hipStreamNonBlocking, hi_pri));
            LSA_HIP_CHECK(ctx, hipEventCreateWithFlags(&f->ev_panel, hipEventDisableTiming));
            LSA_HIP_CHECK(ctx, hipEventCreateWithFlags(&f->ev_pivots, hipEventDisableTiming));
        }
    }
    f->solve_launches = 0;
    for (const NdLevel& L : f->levels) f->solve_launches += (L.fwd_tiles > 0) + (L.bwd_tiles > 0);
    return LSA_OK;
}

template <typename T, int NT, int RPT, int W>
void launch_block(lsa_ctx* ctx, lsa_ndlu* f, const NdChunk& L, int32_t kb, double tiny2) {
    hipStream_t st = ctx->stream;
    const int32_t* lv = f->d_chunk_nodes + L.node_begin;
    T* front = (T*)f->d_work;
    auto active_at = [&](int32_t k) {  // nodes are sorted by own size: those that still have column k form a prefix
        return (int32_t)(std::lower_bound(L.sorted_m.begin(), L.sorted_m.end(), k, std::greater<int32_t>()) - L.sorted_m.begin());
    };
    const int32_t kend = std::min(kb + kNB, L.max_m);
    const bool others = std::min(kNB, L.max_m - kb) > W;  // the block has columns besides one panel
    int32_t kprev = -1, active_prev = 0;
    for (int32_t k0 = kb; k0 < kend; k0 += W) {
        const int32_t active = active_at(k0);
        if (active == 0) break;
        // (the tiles of this launch serve the nodes that had the previous panel: a superset of those that have this one)
        hipLaunchKernelGGL((nd_gj_fused_kernel<T, NT, RPT, W>), dim3(std::max(active, active_prev), others && kprev >= 0 ? 1 + kNB / 16 : 1), dim3(NT), 0, st, lv,
                           f->d_nodes, front, f->d_ipiv, f->d_rowq, kb, k0, kprev, f->d_flag, tiny2);
        kprev = k0;
        active_prev = active;
    }
    if (others && kprev >= 0)  // the last panel's update of the block's other columns
        hipLaunchKernelGGL((nd_gj_fused_kernel<T, NT, RPT, W>), dim3(active_prev, 1 + kNB / 16), dim3(NT), 0, st, lv, f->d_nodes, front, f->d_ipiv, f->d_rowq, kb,
                           -1, kprev, f->d_flag, tiny2);
    if (L.max_m > kNB) {  // columns outside the block exist (in the larger nodes)
        const int32_t active = active_at(kb);
        hipLaunchKernelGGL((nd_gj_stage_kernel<T>), dim3(active, (L.max_m + 255) / 256), dim3(256), 0, st, lv, f->d_nodes, (const T*)front, f->d_ipiv, kb,
                           kNB, kNB, 0, L.max_m, (T*)f->d_ybuf);
        const int32_t tiles = (L.max_m + kGT - 1) / kGT;
        hipLaunchKernelGGL((nd_gj_gemm_kernel<T>), dim3(active, tiles, tiles), dim3(256), 0, st, lv, f->d_nodes, front, f->d_rowq, kb, (const T*)f->d_ybuf, 0, 0,
                           0, 0, 0);
    }
}

// the tournament for the block of columns starting at kb, on stream `st`: leaves D^-1 per node in d_dinv and the pivot
// rows in ipiv / rowq
template <typename T>
void launch_tournament(lsa_ndlu* f, const NdChunk& L, int32_t kb, double tiny2, hipStream_t st) {
    const int32_t* lv = f->d_chunk_nodes + L.node_begin;
    const T* front = (const T*)f->d_work;
    const int32_t active = (int32_t)(std::lower_bound(L.sorted_m.begin(), L.sorted_m.end(), kb, std::greater<int32_t>()) - L.sorted_m.begin());
    if (active == 0) return;
    int32_t sets = (L.max_m + tp_first<T>::rows - 1) / tp_first<T>::rows;
    hipLaunchKernelGGL((nd_tp_round_kernel<T, true, false>), dim3(active, sets), dim3(256), 0, st, lv, f->d_nodes, front, f->d_ipiv, f->d_rowq, kb, 0,
                       (const int32_t*)nullptr, f->d_cand[0], (T*)nullptr, f->d_flag, tiny2);
    int src = 0;
    for (int32_t round = 0;; ++round) {
        const int32_t groups = (sets + kTA - 1) / kTA;
        if (groups == 1) {
            hipLaunchKernelGGL((nd_tp_round_kernel<T, false, true>), dim3(active, 1), dim3(256), 0, st, lv, f->d_nodes, front, f->d_ipiv, f->d_rowq, kb, round,
                               (const int32_t*)f->d_cand[src], (int32_t*)nullptr, (T*)f->d_dinv, f->d_flag, tiny2);
            break;
        }
        hipLaunchKernelGGL((nd_tp_round_kernel<T, false, false>), dim3(active, groups), dim3(256), 0, st, lv, f->d_nodes, front, f->d_ipiv, f->d_rowq, kb, round,
                           (const int32_t*)f->d_cand[src], f->d_cand[src ^ 1], (T*)nullptr, f->d_flag, tiny2);
        src ^= 1;
        sets = groups;
    }
}

// Gauss-Jordan of all pivot blocks of a level by tournament pivoting, with look-ahead: once block k's own columns are done
// the next block's 32 columns are updated first, and its tournament (a chain of single-workgroup launches) runs on a
// second stream underneath the rank-32 product that updates everything else.
template <typename T>
int launch_level_tp(lsa_ctx* ctx, lsa_ndlu* f, const NdChunk& L, double tiny2) {
    hipStream_t st = ctx->stream, side = f->side;
    const int32_t* lv = f->d_chunk_nodes + L.node_begin;
    T* front = (T*)f->d_work;
    auto active_at = [&](int32_t k) {
        return (int32_t)(std::lower_bound(L.sorted_m.begin(), L.sorted_m.end(), k, std::greater<int32_t>()) - L.sorted_m.begin());
    };
    // (two cross-stream hand-offs per block cost ~15 us: worth it only where the product they hide behind is long.
    // Measured: C300k 472 -> 438 ms, C160k 186 -> 183 ms; S500k, tallest pivot block 838 rows, 56 -> 59 ms without this limit)
    const int32_t ahead_min = getenv("LSA_ND_LOOKAHEAD_MIN") ? atoi(getenv("LSA_ND_LOOKAHEAD_MIN")) : 1024;
    const bool ahead = side != nullptr && L.max_m >= std::max(ahead_min, 2 * kNB + 1);
    // super-blocks of kSB columns where the pivot blocks are large (see nd_gj_update_kernel); elsewhere a "super-block" is one block
    const bool wide = L.max_m >= f->sb_min;
    const int32_t sbw = wide ? f->sb_cols : kNB, ycap = f->ycap;
    const int32_t tiles = (L.max_m + kGT - 1) / kGT;
    auto stage = [&](int32_t active, int32_t k0, int32_t kw, int32_t c_lo, int32_t c_hi) {
        hipLaunchKernelGGL((nd_gj_stage_kernel<T>), dim3(active, (c_hi - c_lo + 255) / 256), dim3(256), 0, st, lv, f->d_nodes, (const T*)front, f->d_ipiv, k0, kw,
                           ycap, c_lo, c_hi, (T*)f->d_ybuf);
    };
    auto update = [&](int32_t active, int32_t k0, int32_t kw, int32_t zt0, int32_t ztn, int32_t only_lo, int32_t only_hi, int32_t skip_lo, int32_t skip_hi) {
        hipLaunchKernelGGL((nd_gj_update_kernel<T>), dim3(active, tiles, ztn), dim3(256), 0, st, lv, f->d_nodes, front, f->d_rowq, k0, kw, (const T*)f->d_ybuf,
                           ycap, only_lo, only_hi, skip_lo, skip_hi, zt0);
    };
    launch_tournament<T>(f, L, 0, tiny2, st);
    for (int32_t sb0 = 0; sb0 < L.max_m; sb0 += sbw) {
        const int32_t sb1 = std::min(sb0 + sbw, L.max_m);
        if (active_at(sb0) == 0) break;
        for (int32_t kb = sb0; kb < sb1; kb += kNB) {
            const int32_t active = active_at(kb);
            if (active == 0) break;
            hipLaunchKernelGGL((nd_tp_colblock_kernel<T>), dim3(active, (L.max_m + 255) / 256), dim3(256), 0, st, lv, f->d_nodes, front, f->d_rowq, kb,
                               (const T*)f->d_dinv);
            if (sb1 - sb0 > kNB) {  // the super-block's other columns (earlier blocks' included), so that its next block can be searched
                stage(active, kb, kNB, sb0, sb1);
                update(active, kb, kNB, sb0 / kGT, (sb1 + kGT - 1) / kGT - sb0 / kGT, sb0, sb1, 0, 0);
                if (kb + kNB < sb1 && active_at(kb + kNB) > 0) launch_tournament<T>(f, L, kb + kNB, tiny2, st);
            }
        }
        const int32_t active = active_at(sb0), next = sb1;
        const bool has_next = next < L.max_m && active_at(next) > 0;
        if (L.max_m > sb1 - sb0) {  // columns outside the super-block exist (in the larger nodes)
            stage(active, sb0, sb1 - sb0, 0, L.max_m);
            if (has_next && ahead) {
                update(active, sb0, sb1 - sb0, next / kGT, 1, next, next + kNB, 0, 0);
                LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_panel, st));
                LSA_HIP_CHECK(ctx, hipStreamWaitEvent(side, f->ev_panel, 0));
                launch_tournament<T>(f, L, next, tiny2, side);
                LSA_HIP_CHECK(ctx, hipEventRecord(f->ev_pivots, side));
                update(active, sb0, sb1 - sb0, 0, tiles, 0, 0, next, next + kNB);
                LSA_HIP_CHECK(ctx, hipStreamWaitEvent(st, f->ev_pivots, 0));
                continue;
            }
            update(active, sb0, sb1 - sb0, 0, tiles, 0, 0, 0, 0);
        }
        if (has_next) launch_tournament<T>(f, L, next, tiny2, st);
    }
    return LSA_OK;
}

template <typename T>
int nd_numeric(lsa_ctx* ctx, lsa_ndlu* f, const lsa_mat* C) {
    const NdSymbolic& S = f->S;
    hipStream_t st = ctx->stream;
    T* front = (T*)f->d_work;
    T* lfac = (T*)f->d_lfac;
    T* ufac = (T*)f->d_ufac;
    T* upd = (T*)f->d_upd;
    int rc0 = LSA_OK;
    double max2 = 0.0;
    {
        // (a local failure up to here -- non-finite input, a HIP error -- is agreed on by all ranks before the first exchange)
        auto head = [&]() -> int {
            LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_rowq, 0xFF, (size_t)std::max<int32_t>(S.n, 1) * sizeof(int32_t), st));
            LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_flag, 0, 4 * sizeof(int32_t), st));
            LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_maxabs, 0, sizeof(unsigned long long), st));
            if (S.nnz > 0) {
                const int blocks = (int)std::min<int64_t>((S.nnz + 255) / 256, (int64_t)ctx->num_cu * 2);
                hipLaunchKernelGGL((nd_maxabs2_kernel<T>), dim3(blocks), dim3(256), 0, st, S.nnz, (const T*)C->val, f->d_maxabs);
            }
            unsigned long long mbits = 0;
            LSA_HIP_CHECK(ctx, hipMemcpyAsync(&mbits, f->d_maxabs, sizeof mbits, hipMemcpyDeviceToHost, st));
            LSA_HIP_CHECK(ctx, hipStreamSynchronize(st));
            memcpy(&max2, &mbits, sizeof max2);
            if (!std::isfinite(max2)) return lsa_set_error(ctx, LSA_ERR_NONFINITE, "lsa_ndlu: the matrix holds non-finite values");
            return LSA_OK;
        };
        // (agreed on over the ranks THIS factorisation is split over: a rank-local factorisation on a multi-rank context --
        //  a rank's diagonal block, a retry only one rank takes -- must not enter a collective the others are not in)
        rc0 = S.nranks > 1 ? k_agree_status(ctx, head()) : head();
        if (rc0 != LSA_OK) return rc0;
    }
    // (1e-15 * max|C|)^2: rounding level.  A shift next to an eigenvalue (the adjoint problem of the reference is shifted exactly at
    // a converged eigenvalue) gives legitimate pivots of 1e-12 max|C|; those solves are judged by their backward error.
    const double tiny2 = 1e-30 * max2;
    const int32_t* tl = f->d_tiles;
    bool exchanged = false;
    for (size_t ci = 0; ci < f->chunks.size(); ++ci) {
        const NdChunk& L = f->chunks[ci];
        // subtree-parallel: the ranks' subtree roots are done; every rank receives all of their update matrices
        if (L.exchange_before && f->xupd_slot > 0) {
            LSA_CHECK(k_allgather_inplace(ctx, f->d_upd, (size_t)f->xupd_slot * sizeof(T)));
            exchanged = true;
        }
        LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_work, 0, (size_t)L.work_entries * sizeof(T), st));
        if (L.asm_count > 0) {
            const int blocks = (int)std::min<int64_t>((L.asm_count + 255) / 256, (int64_t)ctx->num_cu * 16);
            hipLaunchKernelGGL((nd_assemble_kernel<T>), dim3(blocks), dim3(256), 0, st, L.asm_count, (const T*)C->val, f->d_asm_src + L.asm_begin,
                               f->d_asm_dst + L.asm_begin, front);
        }
        for (const TileList& e : L.ext)
            if (e.count > 0)
                hipLaunchKernelGGL((nd_extend_add_kernel<T>), dim3(e.count), dim3(256), 0, st, tl + 2 * e.off, f->d_nodes, f->d_cmap, front, (const T*)upd);
        // distributed top nodes: their children's update matrices, in row chunks.  Per step: every rank copies its piece (rows of
        // a distributed child it holds, or of a subtree root it owns) into its slot of the staging buffer, one in-place
        // all-gather, then every rank adds the rows it keeps -- the pivot block and F12 rows on every rank, boundary rows on
        // their owner -- slot by slot (fixed order of the sums: the replicated pivot blocks stay bitwise alike).
        for (const auto& step : L.xsteps) {
            T* stage = (T*)f->d_xstage;
            // (the slots of a step are as wide as its largest piece, not as the buffer allows: the exchange moves what travels)
            int64_t stride = 0;
            for (int r = 0; r < S.nranks; ++r) {
                const NdChunk::XPiece& pc = step[(size_t)r];
                if (pc.nrows > 0) stride = std::max(stride, (int64_t)pc.nrows * (S.f[(size_t)pc.child] - S.m[(size_t)pc.child]));
            }
            if (stride == 0) continue;
            const NdChunk::XPiece& mine = step[(size_t)S.rank];
            if (mine.nrows > 0) {
                const int32_t bc = S.f[(size_t)mine.child] - S.m[(size_t)mine.child];
                const int64_t src_off = f->chunk_node_upd_off(mine.child) + (int64_t)(mine.row0 - S.brow0[(size_t)mine.child]) * bc;
                LSA_HIP_CHECK(ctx, hipMemcpyAsync(stage + (size_t)S.rank * (size_t)stride, upd + src_off, (size_t)mine.nrows * (size_t)bc * sizeof(T),
                                                  hipMemcpyDeviceToDevice, st));
            }
            LSA_CHECK(k_allgather_inplace(ctx, stage, (size_t)stride * sizeof(T)));
            for (int r = 0; r < S.nranks; ++r) {
                const NdChunk::XPiece& pc = step[(size_t)r];
                if (pc.nrows <= 0) continue;
                hipLaunchKernelGGL((nd_extend_add_staged_kernel<T>), dim3((pc.nrows + 15) / 16), dim3(256), 0, st, f->d_nodes, f->d_cmap, front,
                                   (const T*)(stage + (size_t)r * (size_t)stride), pc.child, pc.row0, pc.nrows);
            }
        }
        if (L.max_m >= f->tp_min) LSA_CHECK(launch_level_tp<T>(ctx, f, L, tiny2));
        for (int32_t kb = 0; kb < L.max_m && L.max_m < f->tp_min; kb += kNB) {
            if (L.max_m <= 64) launch_block<T, 64, 1, 8>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 128) launch_block<T, 128, 1, 8>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 256) launch_block<T, 256, 1, 8>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 512) launch_block<T, 512, 1, 8>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 1024) launch_block<T, 1024, 1, 8>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 2048) launch_block<T, 1024, 2, 8>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 4096) launch_block<T, 1024, 4, 4>(ctx, f, L, kb, tiny2);
            else if (L.max_m <= 8192) launch_block<T, 1024, 8, 2>(ctx, f, L, kb, tiny2);
            else launch_block<T, 1024, 16, 1>(ctx, f, L, kb, tiny2);
        }
        if (L.unperm.count > 0)
            hipLaunchKernelGGL((nd_unperm_kernel<T>), dim3(L.unperm.count), dim3(256), 0, st, tl + 2 * L.unperm.off, f->d_nodes, front, f->d_ipiv,
                               f->d_rowq, lfac);
        auto product = [&](auto kind) {
            constexpr int KIND = decltype(kind)::value;
            if (L.gemm[KIND].count == 0) return;
            hipLaunchKernelGGL((nd_gemm_mfma_kernel<T, KIND>), dim3(L.gemm[KIND].count), dim3(256), 0, st, tl + 2 * L.gemm[KIND].off, f->d_nodes, front, lfac, ufac);
        };
        product(std::integral_constant<int, 0>{});
        product(std::integral_constant<int, 1>{});
        product(std::integral_constant<int, 2>{});
        if (L.save.count > 0)
            hipLaunchKernelGGL((nd_save_update_kernel<T>), dim3(L.save.count), dim3(256), 0, st, tl + 2 * L.save.off, f->d_nodes, (const T*)front, upd);
    }
    if (!exchanged && S.nranks > 1 && f->xupd_slot > 0)  // (no replicated level: still a collective)
        LSA_CHECK(k_allgather_inplace(ctx, f->d_upd, (size_t)f->xupd_slot * sizeof(T)));
    int32_t hflag[4] = {0, 0, 0, 0};
    if (S.nranks > 1) {
        // every rank must take the same decision (a rank that returned early would leave the others in a collective):
        // the failure flags are exchanged, the first failing rank's record wins
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(f->d_xflag + 4 * S.rank, f->d_flag, 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        LSA_CHECK(k_allgather_inplace(ctx, f->d_xflag, 4 * sizeof(int32_t)));
        std::vector<int32_t> all((size_t)4 * S.nranks, 0);
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(all.data(), f->d_xflag, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(st));
        for (int r = 0; r < S.nranks; ++r)
            if (all[(size_t)4 * r + 1] != 0) {
                memcpy(hflag, &all[(size_t)4 * r], sizeof hflag);
                if (r != S.rank) hflag[1] = -(r + 1);  // another rank's node: no local record of it
                break;
            }
    } else {
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(hflag, f->d_flag, sizeof hflag, hipMemcpyDeviceToHost, st));
        LSA_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    LSA_HIP_CHECK(ctx, hipGetLastError());
    if (hflag[1] < 0)
        return lsa_set_error(ctx, LSA_ERR_ZERO_PIVOT, "lsa_ndlu: a pivot block on rank %d is singular to 1e-15 * max|C| (column %d of its node)", -hflag[1] - 1,
                             hflag[2]);
    if (hflag[1] != 0) {
        const int32_t t = hflag[1] - 1;
        const unsigned long long hi = (unsigned long long)(uint32_t)hflag[3] << 32;
        double mag2;
        memcpy(&mag2, &hi, sizeof mag2);
        return lsa_set_error(ctx, LSA_ERR_ZERO_PIVOT,
                             "lsa_ndlu: the pivot block of tree node %d (%d unknowns, front %d, level %d) is singular at its column %d: largest "
                             "candidate pivot %.3e against max|C| = %.3e (threshold 1e-15 max|C|); the matrix is singular, or needs pivoting "
                             "across fronts",
                             t, S.m[(size_t)t], S.f[(size_t)t], S.level[(size_t)t], hflag[2], std::sqrt(mag2), std::sqrt(max2));
    }
    if (const char* pe = getenv("LSA_ND_TEST_PERTURB")) {
        const double eps = atof(pe);
        if (eps != 0.0 && f->ufac_entries > 0) {
            const int blocks = (int)std::min<int64_t>((f->ufac_entries + 255) / 256, (int64_t)ctx->num_cu * 16);
            hipLaunchKernelGGL((nd_scale_kernel<T>), dim3(blocks), dim3(256), 0, st, f->ufac_entries, ufac, 1.0 + eps);
            LSA_HIP_CHECK(ctx, hipStreamSynchronize(st));
        }
    }
    return LSA_OK;
}

template <typename MT, typename VT, bool ORDERED>
int nd_apply_ordered(lsa_ctx* ctx, lsa_ndlu* f, const VT* b, VT* x) {
    hipStream_t st = ctx->stream;
    const MT* lfac = (const MT*)f->d_lfac;
    const MT* ufac = (const MT*)f->d_ufac;
    const NdSymbolic& S = f->S;
    VT *ubuf = (VT*)f->d_ubuf, *acc = (VT*)f->d_acc, *xb = (VT*)f->d_xb;
    if (f->acc_vbytes != (int)sizeof(VT)) {
        // the slot rows are indexed in units of the vector scalar: after a solve with the other scalar type the entries no
        // child writes no longer read zero
        LSA_HIP_CHECK(ctx, hipMemsetAsync(f->d_acc, 0, (size_t)std::max<int64_t>(f->acc_entries, 1) * 16, st));
        f->acc_vbytes = (int)sizeof(VT);
    }
    for (size_t li = 0; li <= f->levels.size(); ++li) {
        // subtree-parallel: the update vectors of all ranks' subtree roots, before the replicated top of the tree
        if ((int32_t)li == S.phase_b_level && S.nranks > 1 && S.xu_slot > 0) LSA_CHECK(k_allgather_inplace(ctx, f->d_ubuf, (size_t)S.xu_slot * sizeof(VT)));
        if (li == f->levels.size()) break;
        const NdLevel& L = f->levels[li];
        if (L.fwd_tiles > 0) {
            const dim3 grid(L.node_count, L.fwd_tiles);
            const NdSweepNode* ln = f->d_lnodes + L.node_begin;
            if (L.sweep_rows == 8)
                hipLaunchKernelGGL((nd_fwd_kernel<MT, VT, 64, ORDERED>), grid, dim3(256), 0, st, ln, lfac, f->d_idx, f->d_gell, f->d_cmap, b, x, ubuf, acc, xb);
            else if (L.sweep_rows == 128)
                hipLaunchKernelGGL((nd_fwd_kernel<MT, VT, 4, ORDERED>), grid, dim3(256), 0, st, ln, lfac, f->d_idx, f->d_gell, f->d_cmap, b, x, ubuf, acc, xb);
            else
                hipLaunchKernelGGL((nd_fwd_kernel<MT, VT, 16, ORDERED>), grid, dim3(256), 0, st, ln, lfac, f->d_idx, f->d_gell, f->d_cmap, b, x, ubuf, acc, xb);
        }
        // distributed top nodes of the level: every rank produced its slice of their update entries
        if (L.dist_count > 0 && L.ux_slot > 0) LSA_CHECK(k_allgather_inplace(ctx, ubuf + L.ux_base, (size_t)L.ux_slot * sizeof(VT)));
    }
    for (size_t l = f->levels.size(); l-- > 0;) {
        const NdLevel& L = f->levels[l];
        if (L.bwd_tiles > 0) {
            const dim3 grid(L.node_count, L.bwd_tiles);
            const NdSweepNode* ln = f->d_lnodes_bwd + L.node_begin;
            if (L.sweep_rows == 8) hipLaunchKernelGGL((nd_bwd_kernel<MT, VT, 64, ORDERED>), grid, dim3(256), 0, st, ln, ufac, f->d_idx, f->d_gell, x, xb);
            else hipLaunchKernelGGL((nd_bwd_kernel<MT, VT, 16, ORDERED>), grid, dim3(256), 0, st, ln, ufac, f->d_idx, f->d_gell, x, xb);
        }
        if (L.dist_count > 0) {
            // ... their own rows: slices -> exchange buffer -> all ranks; then x and the children's boundary vectors
            const int32_t* dn = f->d_dist_nodes + L.dist_begin;
            VT* xg = (VT*)f->d_xg;
            if (L.xg_slot > 0) {
                hipLaunchKernelGGL((nd_dist_pack_kernel<VT, ORDERED>), dim3(L.dist_count, (L.dist_rows + 255) / 256), dim3(256), 0, st, dn, f->d_nodes, f->d_idx, S.rank,
                                   (const VT*)x, xg);
                LSA_CHECK(k_allgather_inplace(ctx, xg + L.xg_base, (size_t)L.xg_slot * sizeof(VT)));
            }
            hipLaunchKernelGGL((nd_dist_unpack_kernel<VT, ORDERED>), dim3(L.dist_count, 1 + L.dist_children, (L.dist_rows + 255) / 256), dim3(256), 0, st, dn, f->d_nodes,
                               f->d_child_ptr, f->d_child_idx, f->d_cmap, f->d_idx, S.nranks, x, (const VT*)xg, xb);
        }
    }
    LSA_HIP_CHECK(ctx, hipGetLastError());
    return LSA_OK;
}

template <typename MT, typename VT>
int nd_apply(lsa_ctx* ctx, lsa_ndlu* f, const VT* b, VT* x) {
    return f->ordered ? nd_apply_ordered<MT, VT, true>(ctx, f, b, x) : nd_apply_ordered<MT, VT, false>(ctx, f, b, x);
}

}  // namespace

namespace {
// the partial-sum buffer of the transposed sweeps over distributed nodes, built by the first adjoint solve: a level's slot holds
// the f outputs of each of its distributed nodes (the downward sweep's m fit the same places)
int nd_ensure_transposed_dist(lsa_ctx* ctx, lsa_ndlu* f) {
    const NdSymbolic& S = f->S;
    if (!S.has_dist || f->d_tgoff) return LSA_OK;
    std::vector<int64_t> off(S.lvl_nodes.size(), -1);
    f->tg_slot.assign(f->levels.size(), 0);
    f->tg_slot_max = 0;
    for (size_t l = 0; l < f->levels.size(); ++l) {
        const NdLevel& L = f->levels[l];
        int64_t run = 0;
        for (int32_t q = 0; q < L.node_count; ++q) {
            const int32_t t = S.lvl_nodes[(size_t)L.node_begin + q];
            if (S.kind[(size_t)t] != 4) continue;
            off[(size_t)L.node_begin + q] = run;
            run += S.f[(size_t)t];
        }
        f->tg_slot[l] = run;
        f->tg_slot_max = std::max(f->tg_slot_max, run);
    }
    LSA_HIP_ALLOC(ctx, hipMalloc(&f->d_tg, (size_t)std::max<int64_t>(f->tg_slot_max, 1) * (size_t)S.nranks * 16));
    LSA_CHECK(upload(ctx, off, &f->d_tgoff));
    return LSA_OK;
}

template <typename MT, typename VT, bool CONJ>
int nd_apply_T(lsa_ctx* ctx, lsa_ndlu* f, const VT* b, VT* x) {
    hipStream_t st = ctx->stream;
    const MT* lfac = (const MT*)f->d_lfac;
    const MT* ufac = (const MT*)f->d_ufac;
    const NdSymbolic& S = f->S;
    LSA_CHECK(nd_ensure_transposed_dist(ctx, f));
    VT* pz = (VT*)f->d_tg;
    for (size_t li = 0; li <= f->levels.size(); ++li) {
        if ((int32_t)li == S.phase_b_level && S.nranks > 1 && S.xu_slot > 0) LSA_CHECK(k_allgather_inplace(ctx, f->d_ubuf, (size_t)S.xu_slot * sizeof(VT)));
        if (li == f->levels.size()) break;
        const NdLevel& L = f->levels[li];
        const int64_t slot = L.dist_count > 0 ? f->tg_slot[li] : 0;
        if (L.fwd_tiles > 0)
            hipLaunchKernelGGL((nd_sweepT_kernel<MT, VT, CONJ, false>), dim3(L.node_count, (L.max_f + 63) / 64), dim3(256), 0, st, f->d_lnodes + L.node_begin,
                               lfac, ufac, f->d_idx, f->d_gell, b, x, (VT*)f->d_ubuf, f->d_tgoff ? f->d_tgoff + L.node_begin : nullptr, pz, slot, S.rank);
        if (L.dist_count > 0) {
            // the distributed nodes of the level: every rank summed over its rows; partials to all, added in rank order
            LSA_CHECK(k_allgather_inplace(ctx, pz, (size_t)slot * sizeof(VT)));
            hipLaunchKernelGGL((nd_distT_finish_kernel<VT, false>), dim3(L.node_count, (L.max_f + 255) / 256), dim3(256), 0, st, f->d_lnodes + L.node_begin,
                               f->d_tgoff + L.node_begin, f->d_idx, f->d_gell, (const VT*)pz, slot, S.nranks, S.rank, L.ux_slot, x, (VT*)f->d_ubuf);
        }
    }
    for (size_t l = f->levels.size(); l-- > 0;) {
        const NdLevel& L = f->levels[l];
        const int64_t slot = L.dist_count > 0 ? f->tg_slot[l] : 0;
        if (L.bwd_tiles > 0 || L.dist_count > 0)
            hipLaunchKernelGGL((nd_sweepT_kernel<MT, VT, CONJ, true>), dim3(L.node_count, (L.max_m + 63) / 64), dim3(256), 0, st, f->d_lnodes + L.node_begin,
                               lfac, ufac, f->d_idx, f->d_gell, b, x, (VT*)f->d_ubuf, f->d_tgoff ? f->d_tgoff + L.node_begin : nullptr, pz, slot, S.rank);
        if (L.dist_count > 0) {
            LSA_CHECK(k_allgather_inplace(ctx, pz, (size_t)slot * sizeof(VT)));
            hipLaunchKernelGGL((nd_distT_finish_kernel<VT, true>), dim3(L.node_count, (L.max_m + 255) / 256), dim3(256), 0, st, f->d_lnodes + L.node_begin,
                               f->d_tgoff + L.node_begin, f->d_idx, f->d_gell, (const VT*)pz, slot, S.nranks, S.rank, L.ux_slot, x, (VT*)f->d_ubuf);
        }
    }
    LSA_HIP_CHECK(ctx, hipGetLastError());
    return LSA_OK;
}
}  // namespace

// x = C^-T b (conj == 0) or C^-H b (conj != 0) on the factors of C
int ndlu_solve_adjoint_dev(lsa_ctx* ctx, lsa_ndlu* f, int conj, int vdtype, const void* b, void* x) {
    if (f->dtype == LSA_C128 && vdtype != LSA_C128) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve_adjoint: complex factors need complex vectors");
    if (f->S.n == 0) return LSA_OK;
    if (b == x) {
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(f->d_tmp, b, (size_t)f->S.n * esize(vdtype), hipMemcpyDeviceToDevice, ctx->stream));
        b = f->d_tmp;
    }
    if (f->dtype == LSA_C128) return conj ? nd_apply_T<cplx, cplx, true>(ctx, f, (const cplx*)b, (cplx*)x) : nd_apply_T<cplx, cplx, false>(ctx, f, (const cplx*)b, (cplx*)x);
    if (vdtype == LSA_C128) return nd_apply_T<double, cplx, false>(ctx, f, (const cplx*)b, (cplx*)x);  // real factors: C^H = C^T
    return nd_apply_T<double, double, false>(ctx, f, (const double*)b, (double*)x);
}

// x = C^-1 b on device pointers (b and x distinct or identical: an aliased right-hand side is copied first)
int ndlu_solve_dev(lsa_ctx* ctx, lsa_ndlu* f, int vdtype, const void* b, void* x) {
    if (f->dtype == LSA_C128 && vdtype != LSA_C128) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve: complex factors need complex vectors");
    if (f->S.n == 0) return LSA_OK;
    if (b == x) {
        LSA_HIP_CHECK(ctx, hipMemcpyAsync(f->d_tmp, b, (size_t)f->S.n * esize(vdtype), hipMemcpyDeviceToDevice, ctx->stream));
        b = f->d_tmp;
    }
    if (f->dtype == LSA_C128) return nd_apply<cplx, cplx>(ctx, f, (const cplx*)b, (cplx*)x);
    if (vdtype == LSA_C128) return nd_apply<double, cplx>(ctx, f, (const cplx*)b, (cplx*)x);
    return nd_apply<double, double>(ctx, f, (const double*)b, (double*)x);
}

template <typename T>
__global__ void nd_zero_diag_kernel(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const T* __restrict__ val,
                                    int8_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int8_t z = 1;  // structurally absent counts as zero
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
            if (ci[p] == (int32_t)i) z = s_abs2(val[p]) == 0.0 ? 1 : 0;
        flags[i] = z;
    }
}

// flags[i] = 1 where the diagonal entry of C is exactly zero (or not stored)
static int nd_zero_diagonal_flags(lsa_ctx* ctx, const lsa_mat* C, std::vector<int8_t>& flags) {
    flags.assign((size_t)std::max<int32_t>(C->n, 1), 0);
    int8_t* d = nullptr;
    LSA_HIP_ALLOC(ctx, hipMalloc((void**)&d, flags.size()));
    const int blocks = std::max(1, std::min((C->n + 255) / 256, ctx->num_cu * 8));
    if (C->dtype == LSA_C128) hipLaunchKernelGGL((nd_zero_diag_kernel<cplx>), dim3(blocks), dim3(256), 0, ctx->stream, C->n, C->rp, C->ci, (const cplx*)C->val, d);
    else hipLaunchKernelGGL((nd_zero_diag_kernel<double>), dim3(blocks), dim3(256), 0, ctx->stream, C->n, C->rp, C->ci, (const double*)C->val, d);
    const hipError_t e = hipMemcpyAsync(flags.data(), d, flags.size(), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess || e2 != hipSuccess) return lsa_set_error(ctx, LSA_ERR_HIP, "lsa_ndlu: reading the diagonal failed");
    flags.resize((size_t)C->n);
    return LSA_OK;
}

// the last destroyed factorisation of a context is kept (analysis, tables, buffers): a shift sweep refactorises the
// same pattern once per sigma (.examples/eigenvalues.py:97-108)
extern "C" void lsa_ndlu_drop_cache(lsa_ctx* ctx) {
    if (ctx && ctx->nd_cache) {
        nd_free(ctx->nd_cache);
        ctx->nd_cache = nullptr;
    }
}

extern "C" {

int lsa_ndlu_refactor(lsa_ctx* ctx, lsa_ndlu* f, const lsa_mat* C) {
    if (!ctx || !f || !C) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_refactor: null argument");
    if (C->n != f->S.n || C->n != C->ncols || C->nnz != f->S.nnz || C->dtype != f->dtype)
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_refactor: the matrix does not match the analysed pattern");
    const double t0 = now_s();
    const int rc = f->dtype == LSA_C128 ? nd_numeric<cplx>(ctx, f, C) : nd_numeric<double>(ctx, f, C);
    f->seconds_numeric = now_s() - t0;
    return rc;
}

// nd_pattern_hash of a matrix's host pattern, kept with the (shared) pattern
static uint64_t mat_pattern_hash(const lsa_mat* P) {
    if (!P->h_hash) P->h_hash = std::make_shared<uint64_t>(0);
    if (*P->h_hash == 0) {
        const uint64_t h = nd_pattern_hash(P->n, P->h_rp.data(), P->h_ci.data());
        *P->h_hash = h ? h : 1;
    }
    return *P->h_hash;
}

// analysis + device tables + buffers for the pattern of P and factors of type `dtype`; taken from the context's cache
// when the parked factorisation matches
// (strict: the parked analysis must also have been made for the same set of constraint unknowns)
static int nd_symbolic_phase(lsa_ctx* ctx, const lsa_mat* P, int dtype, int32_t leaf_size, const int8_t* constraint, bool strict,
                             lsa_ndlu** out) {
    *out = nullptr;
    if (P->n != P->ncols || P->row0 != 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu: needs a square, unsharded matrix");
    if ((int64_t)P->h_rp.size() != (int64_t)P->n + 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu: the matrix has no host copy of its pattern");
    if (leaf_size <= 0) leaf_size = nd_default_leaf(P->n);
    const double t0 = now_s();
    // same pattern as the parked factorisation: only the numbers change
    if (ctx->nd_cache) {
        lsa_ndlu* c = ctx->nd_cache;
        uint64_t want = 0;
        if (strict && constraint) {
            want = 1469598103934665603ull;
            for (int32_t v = 0; v < P->n; ++v)
                if (constraint[v]) {
                    want ^= (uint64_t)(uint32_t)v;
                    want *= 1099511628211ull;
                }
            want |= 1ull;
        }
        // (an analysis parked with the caller's tree -- lsa_ndlu_prepare_tree -- stands for its pattern whatever the leaf size)
        if (c->S.n == P->n && c->S.nnz == P->nnz && c->dtype == dtype && c->S.nranks == 1 && (c->S.tree_hash != 0 || c->S.leaf_size == leaf_size) &&
            (!strict || c->S.constraint_hash == want) &&
            c->S.pattern_hash == mat_pattern_hash(P)) {
            ctx->nd_cache = nullptr;
            c->seconds_analyse = 0.0;
            *out = c;
            return LSA_OK;
        }
        lsa_ndlu_drop_cache(ctx);
    }
    lsa_ndlu* f = new lsa_ndlu();
    f->ctx = ctx;
    f->dtype = dtype;
    char buf[256] = {0};
    int rc;
    try {
        rc = nd_analyse(P->n, P->h_rp.data(), P->h_ci.data(), leaf_size, constraint, &f->S, buf, (int)sizeof buf);
    } catch (const std::bad_alloc&) {
        rc = LSA_ERR_ARG;
        snprintf(buf, sizeof buf, "lsa_ndlu: out of host memory in the analysis");
    }
    if (rc != LSA_OK) {
        nd_free(f);
        return lsa_set_error(ctx, rc, "%s", buf);
    }
    rc = nd_setup(ctx, f);
    if (rc != LSA_OK) {
        nd_free(f);
        return rc;
    }
    f->seconds_analyse = now_s() - t0;
    *out = f;
    return LSA_OK;
}

int lsa_ndlu_prepare(lsa_ctx* ctx, const lsa_mat* P, int dtype, int32_t leaf_size, const int8_t* constraint) {
    if (!ctx || !P || (dtype != LSA_F64 && dtype != LSA_C128)) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_prepare: bad argument");
    lsa_ndlu* f = nullptr;
    LSA_CHECK(nd_symbolic_phase(ctx, P, dtype, leaf_size, constraint, true, &f));
    lsa_ndlu_drop_cache(ctx);
    ctx->nd_cache = f;  // the next lsa_ndlu_create on this pattern only runs the numeric phase
    return LSA_OK;
}

int lsa_ndlu_prepare_tree(lsa_ctx* ctx, const lsa_mat* P, int dtype, int32_t ntree, const int32_t* first, const int32_t* size, const int32_t* parent) {
    if (!ctx || !P || (dtype != LSA_F64 && dtype != LSA_C128) || ntree < 0 || (ntree > 0 && (!first || !size || !parent)))
        return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_prepare_tree: bad argument");
    if (P->n != P->ncols || P->row0 != 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_prepare_tree: needs a square, unsharded matrix");
    if ((int64_t)P->h_rp.size() != (int64_t)P->n + 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_prepare_tree: the matrix has no host copy of its pattern");
    const double t0 = now_s();
    lsa_ndlu_drop_cache(ctx);
    lsa_ndlu* f = new lsa_ndlu();
    f->ctx = ctx;
    f->dtype = dtype;
    char buf[256] = {0};
    int rc;
    try {
        rc = nd_analyse_tree(P->n, P->h_rp.data(), P->h_ci.data(), ntree, first, size, parent, nullptr, 0, 1, &f->S, buf, (int)sizeof buf);
    } catch (const std::bad_alloc&) {
        rc = LSA_ERR_ARG;
        snprintf(buf, sizeof buf, "lsa_ndlu_prepare_tree: out of host memory in the analysis");
    }
    if (rc != LSA_OK) lsa_set_error(ctx, rc, "%s", buf);
    else rc = nd_setup(ctx, f);
    if (rc != LSA_OK) {
        nd_free(f);
        return rc;
    }
    f->seconds_analyse = now_s() - t0;
    ctx->nd_cache = f;
    return LSA_OK;
}

int lsa_ndlu_create(lsa_ctx* ctx, const lsa_mat* C, int32_t leaf_size, lsa_ndlu** out) {
    if (!ctx || !C || !out) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_create: null argument");
    *out = nullptr;
    lsa_ndlu* f = nullptr;
    LSA_CHECK(nd_symbolic_phase(ctx, C, C->dtype, leaf_size, nullptr, false, &f));
    int rc = lsa_ndlu_refactor(ctx, f, C);
    if (rc == LSA_ERR_ZERO_PIVOT && f->S.constraint_hash == 0) {
        // A pivot block is singular although pivots are searched over its whole columns.  On saddle-point matrices that is
        // a leaf holding more constraint (zero-diagonal) unknowns than its interior supports: analyse again with those
        // unknowns eliminated after their neighbours, once.  The new analysis replaces the old one for this pattern.
        std::vector<int8_t> flags;
        const int frc = nd_zero_diagonal_flags(ctx, C, flags);
        bool any = false;
        for (int8_t v : flags) any |= v != 0;
        if (frc == LSA_OK && any) {
            const std::string first = ctx->err;
            const int32_t leaf = f->S.leaf_size;
            nd_free(f);
            f = nullptr;
            LSA_CHECK(nd_symbolic_phase(ctx, C, C->dtype, leaf, flags.data(), true, &f));
            rc = lsa_ndlu_refactor(ctx, f, C);
            if (rc == LSA_ERR_ZERO_PIVOT) lsa_set_error(ctx, rc, "%s (also with the zero-diagonal unknowns eliminated last; first attempt: %s)", std::string(ctx->err).c_str(), first.c_str());
        }
    }
    if (rc != LSA_OK) {
        nd_free(f);
        return rc;
    }
    *out = f;
    return LSA_OK;
}

int lsa_ndlu_create_tree(lsa_ctx* ctx, const lsa_mat* C, int32_t ntree, const int32_t* first, const int32_t* size, const int32_t* parent,
                         const int32_t* owner, lsa_ndlu** out) {
    if (!ctx || !C || !out || ntree < 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_create_tree: bad argument");
    *out = nullptr;
    if (C->n != C->ncols || C->row0 != 0) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_create_tree: needs the whole square matrix (every rank holds it)");
    if ((int64_t)C->h_rp.size() != (int64_t)C->n + 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_create_tree: the matrix has no host copy of its pattern");
    const double t0 = now_s();
    lsa_ndlu* f = nullptr;
    int setup_rc = LSA_OK;
    // the free device memory the plan may count on, agreed over the ranks (collective: entered by every rank, first thing)
    int64_t free_agreed = 0;
    if (ctx->nranks > 1) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)16 << 30;
        if (ctx->nd_cache) {  // (what a parked factorisation holds comes back when it is dropped or reused)
            const lsa_ndlu* c = ctx->nd_cache;
            free_b += (size_t)(c->lfac_entries + c->ufac_entries + c->work_entries + c->upd_entries) * esize(c->dtype);
        }
        free_agreed = (int64_t)free_b;
        LSA_CHECK(k_agree_min_i64(ctx, &free_agreed));
    }
    // the parked factorisation, if it was made for this pattern, this tree and this rank
    if (ctx->nd_cache) {
        lsa_ndlu* c = ctx->nd_cache;
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](int32_t v) {
            h ^= (uint64_t)(uint32_t)v;
            h *= 1099511628211ull;
        };
        for (int32_t t = 0; t < ntree; ++t) {
            mix(first[t]);
            mix(size[t]);
            mix(parent[t]);
            mix(owner ? owner[t] : 0);
        }
        mix(ctx->rank);
        mix(ctx->nranks);
        if (c->S.tree_hash == (h | 1ull) && c->S.n == C->n && c->S.nnz == C->nnz && c->dtype == C->dtype &&
            c->S.pattern_hash == mat_pattern_hash(C)) {
            f = c;
            ctx->nd_cache = nullptr;
            f->seconds_analyse = 0.0;
        } else {
            lsa_ndlu_drop_cache(ctx);
        }
    }
    if (!f) {
        f = new lsa_ndlu();
        f->ctx = ctx;
        f->dtype = C->dtype;
        char buf[256] = {0};
        int rc;
        try {
            rc = nd_analyse_tree(C->n, C->h_rp.data(), C->h_ci.data(), ntree, first, size, parent, owner, ctx->rank, ctx->nranks, &f->S, buf, (int)sizeof buf);
        } catch (const std::bad_alloc&) {
            rc = LSA_ERR_ARG;
            snprintf(buf, sizeof buf, "lsa_ndlu_create_tree: out of host memory in the analysis");
        }
        if (rc == LSA_OK) rc = nd_setup(ctx, f, free_agreed);
        else lsa_set_error(ctx, rc, "%s", buf);
        setup_rc = rc;
        f->seconds_analyse = now_s() - t0;
    }
    // Collective agreement on the set-up (out of device memory for this rank's buffers -- their sizes differ from rank to
    // rank --, a limit of the kernels, out of host memory in the analysis): no rank enters the exchanges of the numeric
    // phase alone.
    setup_rc = k_agree_status(ctx, setup_rc);
    if (setup_rc != LSA_OK) {
        nd_free(f);
        return setup_rc;
    }
    const int rc = lsa_ndlu_refactor(ctx, f, C);
    if (rc != LSA_OK) {
        nd_free(f);
        return rc;
    }
    *out = f;
    return LSA_OK;
}

void lsa_ndlu_destroy(lsa_ndlu* f) {
    if (!f) return;
    lsa_ctx* ctx = f->ctx;
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx && !getenv("LSA_ND_NO_CACHE")) {
        lsa_ndlu_drop_cache(ctx);
        ctx->nd_cache = f;
        return;
    }
    nd_free(f);
}

int lsa_ndlu_solve(lsa_ctx* ctx, lsa_ndlu* f, const lsa_vec* b, lsa_vec* x) {
    if (!ctx || !f || !b || !x) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve: null argument");
    if (b->n != f->S.n || x->n != f->S.n || b->dtype != x->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve: shape/dtype mismatch");
    LSA_CHECK(ndlu_solve_dev(ctx, f, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

int lsa_ndlu_solve_adjoint(lsa_ctx* ctx, lsa_ndlu* f, int conj, const lsa_vec* b, lsa_vec* x) {
    if (!ctx || !f || !b || !x) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve_adjoint: null argument");
    if (b->n != f->S.n || x->n != f->S.n || b->dtype != x->dtype) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve_adjoint: shape/dtype mismatch");
    LSA_CHECK(ndlu_solve_adjoint_dev(ctx, f, conj, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
}

int lsa_ndlu_solve_time(lsa_ctx* ctx, lsa_ndlu* f, const lsa_vec* b, lsa_vec* x, int iters, double* avg_ms) {
    if (!ctx || !f || !b || !x || !avg_ms || iters < 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve_time: bad argument");
    if (b->n != f->S.n || x->n != f->S.n || b->dtype != x->dtype || b->d == x->d) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_solve_time: shape/dtype mismatch");
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; ++i) LSA_CHECK(ndlu_solve_dev(ctx, f, b->dtype, b->d, x->d));
    LSA_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    LSA_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    LSA_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / iters;
    return LSA_OK;
}

// Inertia of a REAL SYMMETRIC C from its factorisation: the multifrontal elimination is a block congruence C = L D L^T with D
// the pivot blocks (however each of them was inverted), so inertia(C) = sum over the tree nodes of the inertia of their pivot
// blocks = of the inverses the factors hold (first m rows of the packed L).  The blocks come to the host one by one
// (lsa_dense_sym_inertia: Bunch-Kaufman, O(m^3) each): meant for the interval sweep behind iEpsWhich.ALL on Hermitian
// problems (Solver/utils.py:248-254), whose pivot blocks are a few hundred rows, not for the 3D fronts of the flow problems.
int lsa_ndlu_inertia(lsa_ctx* ctx, lsa_ndlu* f, int64_t* negative, int64_t* zero, int64_t* positive) {
    if (!ctx || !f || !negative || !zero || !positive) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_inertia: null argument");
    if (f->dtype != LSA_F64) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_inertia: needs real factors (a real symmetric matrix)");
    if (f->S.nranks != 1) return lsa_set_error(ctx, LSA_ERR_ARG, "lsa_ndlu_inertia: not available for a forest cut over ranks");
    *negative = *zero = *positive = 0;
    const NdSymbolic& S = f->S;
    std::vector<double> blk;
    LSA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int32_t t = 0; t < S.nt; ++t) {
        const int32_t m = S.m[(size_t)t];
        if (m <= 0) continue;
        blk.resize((size_t)m * m);
        // (the plan's offset of the node's packed L: recomputed as nd_memory_plan lays it out -- f x m scalars per node, in node order)
        LSA_HIP_CHECK(ctx, hipMemcpy(blk.data(), (const double*)f->d_lfac + f->h_lfac_off[(size_t)t], (size_t)m * m * sizeof(double), hipMemcpyDeviceToHost));
        int64_t ng = 0, ze = 0, ps = 0;
        const int rc = lsa_dense_sym_inertia(m, blk.data(), m, 1e-13, &ng, &ze, &ps);
        if (rc != LSA_OK) return lsa_set_error(ctx, rc, "lsa_ndlu_inertia: the pivot block of tree node %d holds non-finite values", t);
        *negative += ng;
        *zero += ze;
        *positive += ps;
    }
    return LSA_OK;
}

int lsa_ndlu_info(const lsa_ndlu* f, int32_t* ntree, int32_t* nlevels, int32_t* max_front, int64_t* factor_entries, int64_t* front_entries,
                  int64_t* apply_bytes, int32_t* apply_launches, double* seconds_analyse, double* seconds_numeric) {
    if (!f) return LSA_ERR_ARG;
    const NdSymbolic& S = f->S;
    if (ntree) *ntree = S.nt;
    if (nlevels) *nlevels = S.nlevels;
    if (max_front) {
        int32_t mf = 0;
        for (int32_t v : S.f) mf = std::max(mf, v);
        *max_front = mf;
    }
    if (factor_entries) *factor_entries = S.factor_entries;
    if (front_entries) *front_entries = f->lfac_entries + f->ufac_entries + f->work_entries + f->upd_entries;  // scalars of the device buffers
    // one solve reads every factor scalar once, the right-hand side once, and reads + writes the solution and the update vectors
    if (apply_bytes) *apply_bytes = S.factor_entries * (int64_t)esize(f->dtype) + 16 * (3 * (int64_t)S.n + 2 * S.u_off[(size_t)S.nt]);
    if (apply_launches) *apply_launches = f->solve_launches;
    if (seconds_analyse) *seconds_analyse = f->seconds_analyse;
    if (seconds_numeric) *seconds_numeric = f->seconds_numeric;
    return LSA_OK;
}

}  // extern "C"
