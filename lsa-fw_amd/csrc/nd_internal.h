// Nested-dissection multifrontal LU: the analysis result shared by nd_symbolic.hip (host) and ndlu.hip (device).
//
// Elimination tree of supernodes ("tree nodes") from recursive graph bisection.  Node t owns m_t unknowns (a leaf
// subdomain or a separator) and has a boundary of b_t unknowns that belong to its ancestors; its front is the dense
// (m_t + b_t)^2 matrix over idx_t = [own | boundary].  Nodes are numbered in post-order, so children precede parents.
#pragma once
#include <cstdlib>

#include <algorithm>
#include <cstdint>
#include <vector>

// leaf size of the dissection when the caller names none: 96 unknowns.  Measured inside eigen-solves (the factors compete with
// the Krylov basis for the caches there; a loop of bare applies flatters large leaves): 30 k unknowns 45.1 / 44.2 / 46.0 / 46.4 /
// 46.2 ms per solve at 64 / 96 / 128 / 192 / 256; 121 k unknowns 159 (96) against 165 ms (128); 504 k unknowns 326 against 328 ms;
// the 3D cases indifferent.
inline int32_t nd_default_leaf(int64_t n) {
    if (const char* e = getenv("LSA_ND_LEAF"))  // (measurement aid)
        if (atoi(e) > 0) return atoi(e);
    (void)n;
    return 96;
}

struct NdSymbolic {
    int32_t n = 0;
    int64_t nnz = 0;
    uint64_t pattern_hash = 0;
    uint64_t constraint_hash = 0;  // 0: no constraint unknowns were given; else a hash of the flagged set
    int32_t leaf_size = 0;
    bool order_only = false;       // lsa_nd_order: only perm, node_start, parent, level are filled
    uint64_t tree_hash = 0;        // 0: the tree came from the library's own dissection; else a hash of the caller's tree
    int32_t nt = 0;       // tree nodes this rank keeps (all of them on one rank)
    int32_t nlevels = 0;  // work levels (entries of lvl_ptr - 1)
    int32_t nranks = 1, rank = 0;  // subtree-parallel factorisation: the ranks the forest is split over
    int32_t phase_b_level = 0;     // work levels [0, phase_b_level) are this rank's own subtrees, the rest the replicated top
    int64_t xfront_slot = 0;       // scalars per rank in the exchange region at the start of the front buffer
    int64_t xu_slot = 0;           // entries per rank in the exchange region at the start of the update-vector buffer
    std::vector<int32_t> kind;     // per kept node: 1 = factored here (own subtree), 2 = replicated top, 3 = another rank's subtree root,
                                   // 4 = DISTRIBUTED top node (below)
    // Distributed top nodes (owner -2 in the caller's forest; closed upwards: the parent of one is one).  While such a node is
    // factored every rank holds the pivot block F11 whole (the Gauss-Jordan inversion runs redundantly, bitwise alike) and F12
    // whole, and only its own slice of the boundary ROWS of F21 / F22: working front (m + brow) x f (+ m^2 for the inverse).
    // What stays: the rank's slice of the OWN rows of the inverse and of U = inv F12 (orows x m, orows x b: the rank produces
    // and finishes that slice of x[own] in the sweeps), its boundary rows of L = -F21 inv (brow x m) and of the update matrix
    // (brow x b).  Slices are equal cuts: rank r owns [r s, min((r + 1) s, count)), s = ceil(count / P).
    std::vector<int32_t> owner;           // per kept node: owning rank (kinds 1, 3), -1 (kind 2), -2 (kind 4)
    std::vector<int32_t> brow0, brow;     // per kept node: this rank's boundary rows [brow0, brow0 + brow)   (all of them unless kind 4)
    std::vector<int32_t> orow0, orows;    // per kept node: this rank's own rows of U                          (all of them unless kind 4)
    // sweeps: the update entries a distributed node produces land in ITS RANK'S SLOT of its level's exchange region of the
    // update-vector buffer (entry k of the boundary at ux_base + (k / s) * ux_stride + k % s; one in-place all-gather per level),
    // its finished own rows in the same way in a buffer of their own (xg_base, xg_stride; s = ceil(m / P))
    std::vector<int64_t> ux_base, ux_stride, xg_base, xg_stride;  // per kept node (0 unless kind 4)
    int64_t xg_entries = 0;               // entries of the own-row exchange buffer
    bool has_dist = false;                // any kind-4 node
    std::vector<int32_t> piv_off;  // per kept node: first elimination position (offset into the pivot arrays of length n)
    std::vector<int32_t> perm;        // elimination order: perm[k] = original index of the k-th eliminated unknown
    std::vector<int32_t> node_start;  // nt + 1: node t owns perm[node_start[t] .. node_start[t + 1]) (one rank; else running sums of m)
    std::vector<int32_t> parent;      // nt, -1 for roots
    std::vector<int32_t> level;       // nt: 0 for leaves, 1 + max(children) otherwise
    std::vector<int32_t> m, f;        // nt: own size, front size (boundary b = f - m)
    std::vector<int64_t> idx_off;     // nt + 1 offsets into idx
    std::vector<int32_t> idx;         // concatenated front index lists in ORIGINAL numbering: own (elimination order), then
                                      // boundary sorted by elimination position
    std::vector<int64_t> front_off;   // nt + 1 offsets (scalars) into the front buffer, f_t^2 each
    std::vector<int32_t> child_ptr, child_idx;  // children of every node
    std::vector<int32_t> cmap_off;    // nt + 1 offsets into cmap
    std::vector<int32_t> cmap;        // for node t: position in parent(t)'s front of each of t's b boundary unknowns
    std::vector<int64_t> u_off;       // nt + 1 offsets into the update-vector buffer (b_t each)
    // forward-solve gather lists: front position j of node t sums ubuf[gidx[g]] for g in [gptr[g_off[t] + j], gptr[g_off[t] + j + 1])
    std::vector<int64_t> g_off;       // nt + 1 offsets into gptr (f_t + 1 entries per node)
    std::vector<int32_t> gptr, gidx;
    // the same lists as one row per child (independent loads on the device): gell[ge_off[t] + c * f_t + j] = index into the
    // update-vector buffer that child c of node t contributes to front position j, or -1
    std::vector<int64_t> ge_off;      // nt + 1
    std::vector<int32_t> gell;
    // assembly of the matrix entries this rank's fronts need: front_buffer[asm_dst[e]] = values[asm_src[e]]
    std::vector<int32_t> asm_src;
    std::vector<int64_t> asm_dst;
    // nodes sorted by (level, own size descending): lvl_nodes[lvl_ptr[l] .. lvl_ptr[l + 1])
    std::vector<int32_t> lvl_ptr, lvl_nodes;
    int32_t max_children = 0;
    int64_t factor_entries = 0;  // sum of m^2 + 2 m b: the scalars one solve reads
    int64_t front_entries = 0;   // sum of f^2
    double flops = 0.0;          // multiply-adds (scalar) of the numeric factorisation
};

// Analysis of a square pattern (CSR, any order, need not be structurally symmetric).  Returns 0 or a negative lsa_status.
int nd_analyse_tree(int32_t n, const int32_t* rp, const int32_t* ci, int32_t nt, const int32_t* first, const int32_t* size, const int32_t* parent,
                    const int32_t* owner, int rank, int nranks, NdSymbolic* out, char* err, int errlen);
// constraint: null, or n flags marking the unknowns with a numerically zero diagonal (eliminated after their neighbours)
int nd_analyse(int32_t n, const int32_t* rp, const int32_t* ci, int32_t leaf_size, const int8_t* constraint, NdSymbolic* out, char* err,
               int errlen, bool order_only = false);
uint64_t nd_pattern_hash(int32_t n, const int32_t* rp, const int32_t* ci);

// Where everything of a factorisation lives on the device, from the analysis alone (host arithmetic; nd_symbolic.hip):
// packed factors, the chunks of working fronts that share one arena, the update arena laid out by a first-fit allocator run
// over the chunk order (a node's update matrix lives from its chunk to its parent's; the subtree roots of a forest cut
// over ranks sit in one slot per rank at its start, the exchange region of the in-place all-gather), the slot rows of the
// upward sweep.  All offsets and sizes in scalars.
struct NdMemoryPlan {
    std::vector<int64_t> work_off, upd_off, lfac_off, ufac_off, acc_off, pacc_off;  // per kept node (acc / pacc: -1 = pull form)
    std::vector<int32_t> chunk_begin;          // chunk c = lvl_nodes[chunk_begin[c] .. chunk_begin[c + 1])
    std::vector<int64_t> chunk_work;           // scalars of the working arena chunk c uses
    std::vector<char> chunk_exchange_before;   // the subtree roots' update matrices are all-gathered before this chunk
    int64_t lfac_entries = 0, ufac_entries = 0, work_entries = 1, upd_entries = 1, acc_entries = 0, xupd_slot = 0;
    int64_t xstage_slot = 0;  // distributed top nodes: scalars per rank of the staging buffer their children's update rows travel through
    int64_t max_front_entries = 0, max_level_entries = 0;
};
// budget_entries: scalars the working arena may take (a single front always fits); <= 0: every level in one chunk
void nd_memory_plan(const NdSymbolic& S, int64_t budget_entries, NdMemoryPlan& P);
// equal cut of `count` items over `nranks`: rank's [first, first + size)
inline void nd_slice(int32_t count, int nranks, int rank, int32_t* first, int32_t* size) {
    const int32_t s = (count + nranks - 1) / std::max(nranks, 1);
    const int32_t lo = std::min<int64_t>(count, (int64_t)rank * s), hi = std::min<int64_t>(count, (int64_t)(rank + 1) * s);
    *first = lo;
    *size = hi - lo;
}
inline int32_t nd_slice_width(int32_t count, int nranks) { return (count + nranks - 1) / std::max(nranks, 1); }
