// Host-side analysis of the nested-dissection multifrontal LU (no GPU needed): ordering by recursive graph bisection,
// supernodal elimination forest, front index lists and every index table the device kernels of ndlu.hip walk.
//
// Stands in for the symbolic phase of the sparse direct solver behind PETSc's PC LU, which is what the reference's
// cylinder runs select for the ST's KSP (.examples/eigenvalues.py:100; Sensitivity/__init__.py:182,260).  Only the
// matrices reach this library (no mesh, no coordinates), so the dissection works on the graph of the pattern:
// level structures from a pseudo-peripheral vertex (George's automatic nested dissection), the separator thinned to the
// vertices of the middle level that touch the next one.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>

#include "../../include/lsa_hip.h"
#include "nd_internal.h"

namespace {

struct Graph {
    int32_t n = 0;
    std::vector<int64_t> ptr;
    std::vector<int32_t> adj;
};

// pattern + transpose, no diagonal, neighbours sorted and unique
void build_graph(int32_t n, const int32_t* rp, const int32_t* ci, Graph& g) {
    g.n = n;
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
            const int32_t j = ci[p];
            if (j == i) continue;
            ++cnt[(size_t)i + 1];
            ++cnt[(size_t)j + 1];
        }
    for (int32_t i = 0; i < n; ++i) cnt[(size_t)i + 1] += cnt[i];
    std::vector<int32_t> tmp((size_t)cnt[n]);
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
            const int32_t j = ci[p];
            if (j == i) continue;
            tmp[(size_t)fill[i]++] = j;
            tmp[(size_t)fill[j]++] = i;
        }
    g.ptr.assign((size_t)n + 1, 0);
    g.adj.clear();
    g.adj.reserve(tmp.size() / 2 + 16);
    for (int32_t i = 0; i < n; ++i) {
        auto b = tmp.begin() + cnt[i], e = tmp.begin() + cnt[(size_t)i + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        g.adj.insert(g.adj.end(), b, e);
        g.ptr[(size_t)i + 1] = (int64_t)g.adj.size();
    }
}

struct Item {
    std::vector<int32_t> verts;
    int32_t parent;
};

// breadth-first level structure of the region `rid` from `start`; order receives the vertices level by level
int32_t bfs_levels(const Graph& g, const std::vector<int32_t>& region, int32_t rid, int32_t start, std::vector<int32_t>& lev,
                   std::vector<int32_t>& order) {
    order.clear();
    order.push_back(start);
    lev[start] = 0;
    int32_t nlev = 1;
    for (size_t head = 0; head < order.size(); ++head) {
        const int32_t v = order[head];
        const int32_t lv = lev[v];
        for (int64_t p = g.ptr[v]; p < g.ptr[(size_t)v + 1]; ++p) {
            const int32_t w = g.adj[(size_t)p];
            if (region[w] != rid || lev[w] >= 0) continue;
            lev[w] = lv + 1;
            nlev = std::max(nlev, lv + 2);
            order.push_back(w);
        }
    }
    return nlev;
}

void dissect(const Graph& g, int32_t leaf, std::vector<std::vector<int32_t>>& own, std::vector<int32_t>& parent) {
    const int32_t n = g.n;
    std::vector<int32_t> region((size_t)n, -1), lev((size_t)n, -1);
    int32_t next_region = 0;
    std::vector<Item> stack;
    {
        Item all;
        all.verts.resize((size_t)n);
        std::iota(all.verts.begin(), all.verts.end(), 0);
        all.parent = -1;
        stack.push_back(std::move(all));
    }
    std::vector<int32_t> order, order2, cnt;
    // The smaller part must hold at least `balance` percent of the region.  Among the candidates the smallest separator wins,
    // so a loose bound buys small separators with a lopsided, deeper forest -- and every level of the forest is one dependent
    // launch per sweep of every solve.  Planar meshes: 40 (30 k unknowns: 10 levels instead of 12 for 4 % more factor
    // entries; 120 k: 13 instead of 16); 3D meshes, whose solves stream gigabytes per level: 30.  LSA_ND_BALANCE overrides.
    const bool use_index_cut = !(getenv("LSA_ND_INDEX_CUT") && atoi(getenv("LSA_ND_INDEX_CUT")) == 0);
    const int64_t balance = getenv("LSA_ND_BALANCE") ? std::max(1, std::min(49, atoi(getenv("LSA_ND_BALANCE")))) : (g.ptr[(size_t)n] > 60 * (int64_t)n ? 30 : 40);
    auto emit = [&](std::vector<int32_t>&& verts, int32_t par) {
        own.push_back(std::move(verts));
        parent.push_back(par);
        return (int32_t)own.size() - 1;
    };
    while (!stack.empty()) {
        Item it = std::move(stack.back());
        stack.pop_back();
        if (it.verts.empty()) continue;
        const int32_t rid = next_region++;
        for (int32_t v : it.verts) {
            region[v] = rid;
            lev[v] = -1;
        }
        // connected components of the region
        std::vector<std::vector<int32_t>> comps;
        for (int32_t v : it.verts) {
            if (lev[v] >= 0) continue;
            bfs_levels(g, region, rid, v, lev, order);
            comps.emplace_back(order);
            if (comps.back().size() == it.verts.size()) break;
        }
        if (comps.size() > 1) {
            // large components are dissected on their own; the small ones are packed into leaves (a node may own
            // unknowns that are not connected to each other: its front is then block diagonal)
            std::vector<int32_t> bin;
            for (auto& c : comps) {
                if ((int32_t)c.size() > leaf) {
                    stack.push_back(Item{std::move(c), it.parent});
                    continue;
                }
                if (!bin.empty() && (int32_t)(bin.size() + c.size()) > leaf) {
                    emit(std::move(bin), it.parent);
                    bin.clear();
                }
                bin.insert(bin.end(), c.begin(), c.end());
            }
            if (!bin.empty()) emit(std::move(bin), it.parent);
            continue;
        }
        std::vector<int32_t>& verts = comps[0];
        if ((int32_t)verts.size() <= leaf) {
            emit(std::move(verts), it.parent);
            continue;
        }
        // level structure from a pseudo-peripheral vertex: hop to the far end while the structure gets deeper
        int32_t start = verts[0];
        for (int32_t v : verts) lev[v] = -1;
        int32_t nlev = bfs_levels(g, region, rid, start, lev, order);
        for (int hop = 0; hop < 4; ++hop) {
            // a vertex of minimum degree in the last level
            int32_t far = order.back();
            int64_t fdeg = g.ptr[(size_t)far + 1] - g.ptr[far];
            for (size_t q = order.size(); q-- > 0 && lev[order[q]] == nlev - 1;) {
                const int64_t d = g.ptr[(size_t)order[q] + 1] - g.ptr[order[q]];
                if (d < fdeg) {
                    fdeg = d;
                    far = order[q];
                }
            }
            for (int32_t v : verts) lev[v] = -1;
            const int32_t nlev2 = bfs_levels(g, region, rid, far, lev, order2);
            const bool deeper = nlev2 > nlev;
            start = far;
            nlev = nlev2;
            order.swap(order2);
            if (!deeper) break;
        }
        if (nlev < 3) {  // (nearly) complete graph: nothing to dissect
            emit(std::move(verts), it.parent);
            continue;
        }
        cnt.assign((size_t)nlev, 0);
        for (int32_t v : verts) ++cnt[(size_t)lev[v]];
        // middle level: balanced within 30 / 70, smallest level wins; otherwise the most balanced one
        const int64_t total = (int64_t)verts.size();
        int32_t best = -1, fallback = 1;
        int64_t below = cnt[0], best_cnt = 0, fb_gap = -1;
        for (int32_t k = 1; k <= nlev - 2; ++k) {
            const int64_t above = total - below - cnt[(size_t)k];
            const int64_t gap = below > above ? below - above : above - below;
            if (fb_gap < 0 || gap < fb_gap) {
                fb_gap = gap;
                fallback = k;
            }
            if (std::min(below, above) * 100 >= total * balance && (best < 0 || cnt[(size_t)k] < best_cnt)) {
                best = k;
                best_cnt = cnt[(size_t)k];
            }
            below += cnt[(size_t)k];
        }
        const int32_t k = best >= 0 ? best : fallback;
        std::vector<int32_t> sep, left, right;
        for (int32_t v : order) {
            const int32_t lv = lev[v];
            if (lv > k) right.push_back(v);
            else if (lv < k) left.push_back(v);
            else {
                bool touches = false;
                for (int64_t p = g.ptr[v]; p < g.ptr[(size_t)v + 1] && !touches; ++p) {
                    const int32_t w = g.adj[(size_t)p];
                    touches = region[w] == rid && lev[w] == k + 1;
                }
                (touches ? sep : left).push_back(v);
            }
        }
        if (use_index_cut) {
            // Second candidate: cut the region's unknowns at the median of their INDICES; the separator is the lower half's
            // side of the cut (its vertices with a neighbour in the upper half).  Meshes reach the library in numberings that
            // keep neighbours close (lexicographic, or bandwidth-reducing: dolfinx reorders by default), and there an index cut
            // is a plane across the shortest extent of the numbering's sweep -- while the level sets of a breadth-first search
            // from a corner of a 3D box are L-shaped shells of up to twice a cross-section's size (the root separator of the
            // 5 M-unknown cube: 82 731 unknowns against 44 500 of a coordinate plane).  The smaller separator wins, so the
            // search-based cut still takes over wherever the index cut is poor (long thin regions cut along their length).
            std::vector<int32_t> sorted(verts);
            std::sort(sorted.begin(), sorted.end());
            const int32_t midv = sorted[sorted.size() / 2];
            int64_t nsep = 0, nlow = 0;
            for (int32_t v : sorted) {
                if (v >= midv) break;
                ++nlow;
                bool touches = false;
                for (int64_t p = g.ptr[v]; p < g.ptr[(size_t)v + 1] && !touches; ++p) {
                    const int32_t w = g.adj[(size_t)p];
                    touches = region[w] == rid && w >= midv;
                }
                nsep += touches;
            }
            const int64_t nup = total - nlow;
            if (nsep > 0 && nsep < (int64_t)sep.size() && std::min(nlow - nsep, nup) * 100 >= total * balance) {
                sep.clear();
                left.clear();
                right.clear();
                for (int32_t v : sorted) {
                    if (v >= midv) {
                        right.push_back(v);
                        continue;
                    }
                    bool touches = false;
                    for (int64_t p = g.ptr[v]; p < g.ptr[(size_t)v + 1] && !touches; ++p) {
                        const int32_t w = g.adj[(size_t)p];
                        touches = region[w] == rid && w >= midv;
                    }
                    (touches ? sep : left).push_back(v);
                }
            }
        }
        const int32_t s = emit(std::move(sep), it.parent);
        stack.push_back(Item{std::move(left), s});
        stack.push_back(Item{std::move(right), s});
    }
}

}  // namespace

uint64_t nd_pattern_hash(int32_t n, const int32_t* rp, const int32_t* ci) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        h ^= v;
        h *= 1099511628211ull;
    };
    mix((uint64_t)n);
    for (int32_t i = 0; i <= n; ++i) mix((uint64_t)(uint32_t)rp[i]);
    const int64_t nnz = rp[n];
    for (int64_t p = 0; p < nnz; ++p) mix((uint64_t)(uint32_t)ci[p]);
    return h;
}

namespace {

struct Fail {
    char* err;
    int errlen;
    int operator()(int code, const char* msg) const {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", msg);
        return code;
    }
};

// Everything after the tree is known.  `own` / `par`: unknowns and parent of every tree node (any numbering in which a
// parent can be looked up); `owner`: null, or per tree node the rank whose subtree it belongs to (-1 = the replicated top
// of the tree).  With an owner map the tables are localised for `rank`: nodes of other ranks' subtrees are dropped, except
// their roots, which stay as "ghost" leaves whose fronts and update vectors arrive by all-gather.
int nd_finish(NdSymbolic& S, int32_t n, const int32_t* rp, const int32_t* ci, Graph& g, std::vector<std::vector<int32_t>>& own,
              std::vector<int32_t>& par, const int32_t* owner, int rank, int nranks, const Fail& fail) {
    const int32_t nt = (int32_t)own.size();
    // post-order numbering (children before parents)
    std::vector<std::vector<int32_t>> kids((size_t)nt);
    std::vector<int32_t> roots;
    for (int32_t t = 0; t < nt; ++t) (par[t] >= 0 ? kids[(size_t)par[t]] : roots).push_back(t);
    std::vector<int32_t> newid((size_t)nt, -1), seq;
    seq.reserve((size_t)nt);
    {
        std::vector<std::pair<int32_t, size_t>> st;
        for (int32_t r : roots) {
            st.emplace_back(r, 0);
            while (!st.empty()) {
                auto& top = st.back();
                if (top.second < kids[(size_t)top.first].size()) {
                    const int32_t c = kids[(size_t)top.first][top.second++];
                    st.emplace_back(c, 0);
                } else {
                    newid[(size_t)top.first] = (int32_t)seq.size();
                    seq.push_back(top.first);
                    st.pop_back();
                }
            }
        }
    }
    if ((int32_t)seq.size() != nt) return fail(LSA_ERR_ARG, "nd_analyse: the parent array does not describe a forest");
    // ---- the global forest ----
    std::vector<int32_t> gstart((size_t)nt + 1, 0), gparent((size_t)nt, -1), gm((size_t)nt, 0), gowner((size_t)nt, 0);
    S.perm.clear();
    S.perm.reserve((size_t)n);
    for (int32_t k = 0; k < nt; ++k) {
        const int32_t t = seq[(size_t)k];
        S.perm.insert(S.perm.end(), own[(size_t)t].begin(), own[(size_t)t].end());
        gstart[(size_t)k + 1] = (int32_t)S.perm.size();
        gm[(size_t)k] = (int32_t)own[(size_t)t].size();
        gparent[(size_t)k] = par[t] >= 0 ? newid[(size_t)par[t]] : -1;
        gowner[(size_t)k] = owner ? owner[t] : 0;
    }
    own.clear();
    const int32_t nreal = (int32_t)S.perm.size();  // unknowns that belong to a tree node (all of them unless the matrix is padded)
    std::vector<int32_t> pos((size_t)n, -1), node_of((size_t)std::max(nreal, 1));
    for (int32_t k = 0; k < nreal; ++k) {
        const int32_t v = S.perm[(size_t)k];
        if (v < 0 || v >= n || pos[(size_t)v] >= 0) return fail(LSA_ERR_ARG, "nd_analyse: the tree's unknowns are not distinct indices of the matrix");
        pos[(size_t)v] = k;
    }
    for (int32_t i = 0; i < n; ++i) {
        const bool real = pos[(size_t)i] >= 0;
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
            if (!real || pos[(size_t)ci[p]] < 0) return fail(LSA_ERR_ARG, "nd_analyse: a matrix entry lies in a row or column that belongs to no tree node");
    }
    for (int32_t t = 0; t < nt; ++t)
        for (int32_t k = gstart[(size_t)t]; k < gstart[(size_t)t + 1]; ++k) node_of[(size_t)k] = t;
    std::vector<int32_t> gchild_ptr((size_t)nt + 1, 0), gchild_idx, glevel((size_t)nt, 0);
    for (int32_t t = 0; t < nt; ++t)
        if (gparent[(size_t)t] >= 0) ++gchild_ptr[(size_t)gparent[(size_t)t] + 1];
    for (int32_t t = 0; t < nt; ++t) gchild_ptr[(size_t)t + 1] += gchild_ptr[(size_t)t];
    gchild_idx.assign((size_t)gchild_ptr[(size_t)nt], 0);
    {
        std::vector<int32_t> fillc(gchild_ptr.begin(), gchild_ptr.end() - 1);
        for (int32_t t = 0; t < nt; ++t)
            if (gparent[(size_t)t] >= 0) gchild_idx[(size_t)fillc[(size_t)gparent[(size_t)t]]++] = t;
    }
    for (int32_t t = 0; t < nt; ++t)
        if (gparent[(size_t)t] >= 0) glevel[(size_t)gparent[(size_t)t]] = std::max(glevel[(size_t)gparent[(size_t)t]], glevel[(size_t)t] + 1);
    if (S.order_only) {  // lsa_nd_order: the elimination order and the forest are all the caller wants
        S.nt = nt;
        S.nlevels = 0;
        for (int32_t t = 0; t < nt; ++t) S.nlevels = std::max(S.nlevels, glevel[(size_t)t] + 1);
        S.node_start = gstart;
        S.parent = gparent;
        S.level = glevel;
        S.m = gm;
        S.f = gm;
        return LSA_OK;
    }
    // boundary (struct) of every node in elimination positions
    std::vector<std::vector<int32_t>> bnd((size_t)nt);
    {
        std::vector<int32_t> stamp((size_t)std::max(nreal, 1), -1);
        for (int32_t t = 0; t < nt; ++t) {
            const int32_t a = gstart[(size_t)t], b = gstart[(size_t)t + 1];
            std::vector<int32_t>& L = bnd[(size_t)t];
            for (int32_t k = a; k < b; ++k) {
                const int32_t v = S.perm[(size_t)k];
                for (int64_t p = g.ptr[v]; p < g.ptr[(size_t)v + 1]; ++p) {
                    const int32_t q = pos[(size_t)g.adj[(size_t)p]];
                    if (q >= b && stamp[(size_t)q] != t) {
                        stamp[(size_t)q] = t;
                        L.push_back(q);
                    }
                }
            }
            for (int32_t cp = gchild_ptr[(size_t)t]; cp < gchild_ptr[(size_t)t + 1]; ++cp)
                for (int32_t q : bnd[(size_t)gchild_idx[(size_t)cp]])
                    if (q >= b && stamp[(size_t)q] != t) {
                        stamp[(size_t)q] = t;
                        L.push_back(q);
                    }
            std::sort(L.begin(), L.end());
            if (!L.empty() && gparent[(size_t)t] < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (root with a boundary)");
        }
    }
    g.adj.clear();
    g.adj.shrink_to_fit();
    // ---- the nodes this rank keeps ----
    const bool dist = owner != nullptr && nranks > 1;
    S.nranks = dist ? nranks : 1;
    S.rank = dist ? rank : 0;
    std::vector<int32_t> kind((size_t)nt, 1), loc((size_t)nt, -1), keep;  // 1 own, 2 top (replicated), 3 ghost root, 0 dropped
    if (dist) {
        for (int32_t t = 0; t < nt; ++t) {
            const int32_t o = gowner[(size_t)t], p = gparent[(size_t)t];
            if (o >= nranks || o < -2) return fail(LSA_ERR_ARG, "nd_analyse: owner rank out of range");
            if (o < 0) {
                kind[(size_t)t] = o == -2 ? 4 : 2;
                if (p >= 0 && gowner[(size_t)p] >= 0) return fail(LSA_ERR_ARG, "nd_analyse: a top node lies below a rank's subtree");
                if (o == -2 && p >= 0 && gowner[(size_t)p] != -2) return fail(LSA_ERR_ARG, "nd_analyse: the parent of a distributed top node must be distributed");
            } else {
                if (p >= 0 && gowner[(size_t)p] >= 0 && gowner[(size_t)p] != o) return fail(LSA_ERR_ARG, "nd_analyse: a subtree is split between ranks");
                const bool root = p < 0 || gowner[(size_t)p] < 0;
                kind[(size_t)t] = o == rank ? 1 : (root && p >= 0 ? 3 : 0);
            }
        }
    }
    for (int32_t t = 0; t < nt; ++t)
        if (kind[(size_t)t] != 0) {
            loc[(size_t)t] = (int32_t)keep.size();
            keep.push_back(t);
        }
    const int32_t nl = (int32_t)keep.size();
    S.nt = nl;
    S.m.assign((size_t)nl, 0);
    S.f.assign((size_t)nl, 0);
    S.parent.assign((size_t)nl, -1);
    S.level.assign((size_t)nl, 0);
    S.kind.assign((size_t)nl, 1);
    S.owner.assign((size_t)nl, 0);
    S.brow0.assign((size_t)nl, 0);
    S.brow.assign((size_t)nl, 0);
    S.orow0.assign((size_t)nl, 0);
    S.orows.assign((size_t)nl, 0);
    S.ux_base.assign((size_t)nl, 0);
    S.ux_stride.assign((size_t)nl, 0);
    S.xg_base.assign((size_t)nl, 0);
    S.xg_stride.assign((size_t)nl, 0);
    S.xg_entries = 0;
    S.has_dist = false;
    S.piv_off.assign((size_t)nl, 0);
    S.node_start.assign((size_t)nl + 1, 0);
    for (int32_t q = 0; q < nl; ++q) {
        const int32_t t = keep[(size_t)q];
        S.m[(size_t)q] = gm[(size_t)t];
        S.f[(size_t)q] = gm[(size_t)t] + (int32_t)bnd[(size_t)t].size();
        S.parent[(size_t)q] = gparent[(size_t)t] >= 0 ? loc[(size_t)gparent[(size_t)t]] : -1;
        if (gparent[(size_t)t] >= 0 && S.parent[(size_t)q] < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (kept node with a dropped parent)");
        S.level[(size_t)q] = glevel[(size_t)t];
        S.kind[(size_t)q] = kind[(size_t)t];
        S.owner[(size_t)q] = dist ? gowner[(size_t)t] : 0;
        S.brow[(size_t)q] = (int32_t)bnd[(size_t)t].size();
        S.orows[(size_t)q] = gm[(size_t)t];
        if (kind[(size_t)t] == 4) {
            nd_slice((int32_t)bnd[(size_t)t].size(), nranks, rank, &S.brow0[(size_t)q], &S.brow[(size_t)q]);
            nd_slice(gm[(size_t)t], nranks, rank, &S.orow0[(size_t)q], &S.orows[(size_t)q]);
            S.has_dist = true;
        }
        S.piv_off[(size_t)q] = gstart[(size_t)t];
        S.node_start[(size_t)q + 1] = S.node_start[(size_t)q] + gm[(size_t)t];
    }
    if (!dist) S.node_start.assign(gstart.begin(), gstart.end());
    // children among the kept nodes (ghost roots keep none)
    S.child_ptr.assign((size_t)nl + 1, 0);
    for (int32_t q = 0; q < nl; ++q)
        if (S.parent[(size_t)q] >= 0) ++S.child_ptr[(size_t)S.parent[(size_t)q] + 1];
    for (int32_t q = 0; q < nl; ++q) S.child_ptr[(size_t)q + 1] += S.child_ptr[(size_t)q];
    S.child_idx.assign((size_t)S.child_ptr[(size_t)nl], 0);
    {
        std::vector<int32_t> fillc(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int32_t q = 0; q < nl; ++q)
            if (S.parent[(size_t)q] >= 0) S.child_idx[(size_t)fillc[(size_t)S.parent[(size_t)q]]++] = q;
    }
    S.max_children = 0;
    S.nlevels = 0;
    for (int32_t q = 0; q < nl; ++q) {
        S.max_children = std::max(S.max_children, S.child_ptr[(size_t)q + 1] - S.child_ptr[(size_t)q]);
        S.nlevels = std::max(S.nlevels, S.level[(size_t)q] + 1);
    }
    // ---- offsets: the fronts and update vectors of the subtree roots lie in one slot per rank at the start of their
    // buffers (the exchange regions of the in-place all-gathers), everything else behind them ----
    // a subtree root below the top: its update VECTOR always travels through the exchange region of the update-vector buffer;
    // its update MATRIX through the exchange region of the front / update buffers only when its parent is replicated (to a
    // distributed parent it travels in row chunks through the staging buffer, nd_numeric)
    auto is_top = [&](int32_t k) { return k == 2 || k == 4; };
    auto is_xroot_u = [&](int32_t q) { return dist && !is_top(S.kind[(size_t)q]) && S.parent[(size_t)q] >= 0 && is_top(S.kind[(size_t)S.parent[(size_t)q]]); };
    auto is_xroot = [&](int32_t q) { return dist && !is_top(S.kind[(size_t)q]) && S.parent[(size_t)q] >= 0 && S.kind[(size_t)S.parent[(size_t)q]] == 2; };
    S.idx_off.assign((size_t)nl + 1, 0);
    S.front_off.assign((size_t)nl + 1, 0);
    S.u_off.assign((size_t)nl + 1, 0);
    S.cmap_off.assign((size_t)nl + 1, 0);
    S.g_off.assign((size_t)nl + 1, 0);
    S.ge_off.assign((size_t)nl + 1, 0);
    S.xfront_slot = S.xu_slot = 0;
    {
        std::vector<int64_t> fuse((size_t)S.nranks, 0), uuse((size_t)S.nranks, 0);
        if (dist) {
            for (int32_t t = 0; t < nt; ++t) {  // slot sizes from the GLOBAL forest: every rank must agree on them
                const int32_t o = gowner[(size_t)t], p = gparent[(size_t)t];
                if (o >= 0 && p >= 0 && gowner[(size_t)p] < 0) {
                    const int64_t f = gm[(size_t)t] + (int64_t)bnd[(size_t)t].size(), b = (int64_t)bnd[(size_t)t].size();
                    if (gowner[(size_t)p] == -1) fuse[(size_t)o] += f * f;
                    uuse[(size_t)o] += b;
                }
            }
            for (int r = 0; r < S.nranks; ++r) {
                S.xfront_slot = std::max(S.xfront_slot, fuse[(size_t)r]);
                S.xu_slot = std::max(S.xu_slot, uuse[(size_t)r]);
            }
            std::fill(fuse.begin(), fuse.end(), 0);
            std::fill(uuse.begin(), uuse.end(), 0);
        }
        int64_t frun = S.xfront_slot * S.nranks, urun = S.xu_slot * S.nranks;
        for (int32_t q = 0; q < nl; ++q) {
            const int64_t m = S.m[(size_t)q], f = S.f[(size_t)q], b = f - m;
            const int32_t nchild = S.child_ptr[(size_t)q + 1] - S.child_ptr[(size_t)q];
            S.idx_off[(size_t)q + 1] = S.idx_off[(size_t)q] + f;
            S.cmap_off[(size_t)q + 1] = S.cmap_off[(size_t)q] + (int32_t)b;
            S.g_off[(size_t)q + 1] = S.g_off[(size_t)q] + f + 1;
            S.ge_off[(size_t)q + 1] = S.ge_off[(size_t)q] + (int64_t)nchild * f;
            const int32_t o = gowner[(size_t)keep[(size_t)q]];
            if (is_xroot(q)) {
                S.front_off[(size_t)q] = S.xfront_slot * o + fuse[(size_t)o];
                fuse[(size_t)o] += f * f;
            } else {
                S.front_off[(size_t)q] = frun;
                frun += f * f;
            }
            if (is_xroot_u(q)) {
                S.u_off[(size_t)q] = S.xu_slot * o + uuse[(size_t)o];
                uuse[(size_t)o] += b;
            } else {
                S.u_off[(size_t)q] = urun;
                urun += b;
            }
            if (S.kind[(size_t)q] == 4) {  // this rank's share: its own rows of the inverse and of U, its boundary rows of L
                const int64_t br = S.brow[(size_t)q], orr = S.orows[(size_t)q];
                S.factor_entries += orr * m + br * m + orr * b;
                S.flops += (double)m * m * m + (double)m * m * br + (double)m * m * orr * (b > 0) + (double)m * br * b;
            } else if (S.kind[(size_t)q] != 3) {
                S.factor_entries += m * m + 2 * m * b;
                S.flops += (double)m * m * m + 2.0 * m * m * b + (double)m * b * b;
            }
        }
        // exchange regions of the sweeps, one per tree level that holds distributed nodes, behind everything else
        if (S.has_dist) {
            int32_t maxlvl = 0;
            for (int32_t q = 0; q < nl; ++q) maxlvl = std::max(maxlvl, S.level[(size_t)q]);
            for (int32_t l = 0; l <= maxlvl; ++l) {
                int64_t uslot = 0, gslot = 0;
                for (int32_t q = 0; q < nl; ++q)
                    if (S.kind[(size_t)q] == 4 && S.level[(size_t)q] == l) {
                        uslot += nd_slice_width(S.f[(size_t)q] - S.m[(size_t)q], S.nranks);
                        gslot += nd_slice_width(S.m[(size_t)q], S.nranks);
                    }
                if (uslot + gslot == 0) continue;
                int64_t ucur = urun, gcur = S.xg_entries;
                for (int32_t q = 0; q < nl; ++q)
                    if (S.kind[(size_t)q] == 4 && S.level[(size_t)q] == l) {
                        S.ux_base[(size_t)q] = ucur;
                        S.ux_stride[(size_t)q] = uslot;
                        S.xg_base[(size_t)q] = gcur;
                        S.xg_stride[(size_t)q] = gslot;
                        ucur += nd_slice_width(S.f[(size_t)q] - S.m[(size_t)q], S.nranks);
                        gcur += nd_slice_width(S.m[(size_t)q], S.nranks);
                    }
                urun += uslot * S.nranks;
                S.xg_entries += gslot * S.nranks;
            }
        }
        S.front_off[(size_t)nl] = frun;  // total scalars of the front buffer
        S.u_off[(size_t)nl] = urun;      // total entries of the update-vector buffer
    }
    S.front_entries = S.front_off[(size_t)nl];
    if (S.idx_off[(size_t)nl] > 0x7fffffff || S.u_off[(size_t)nl] > 0x7fffffff)
        return fail(LSA_ERR_ARG, "nd_analyse: front index lists exceed 2^31 entries");
    // front index lists in the matrix's numbering: own unknowns in elimination order, then the boundary
    S.idx.resize((size_t)S.idx_off[(size_t)nl]);
    for (int32_t q = 0; q < nl; ++q) {
        const int32_t t = keep[(size_t)q];
        int32_t* dst = S.idx.data() + S.idx_off[(size_t)q];
        for (int32_t k = gstart[(size_t)t]; k < gstart[(size_t)t + 1]; ++k) *dst++ = S.perm[(size_t)k];
        for (int32_t e : bnd[(size_t)t]) *dst++ = S.perm[(size_t)e];
    }
    // local front position of elimination position e in (global) node t
    auto local = [&](int32_t t, int32_t e) -> int32_t {
        const int32_t a = gstart[(size_t)t], b = gstart[(size_t)t + 1];
        if (e >= a && e < b) return e - a;
        const std::vector<int32_t>& L = bnd[(size_t)t];
        auto itp = std::lower_bound(L.begin(), L.end(), e);
        if (itp == L.end() || *itp != e) return -1;
        return (b - a) + (int32_t)(itp - L.begin());
    };
    // position of every boundary unknown in the parent's front
    S.cmap.resize((size_t)S.cmap_off[(size_t)nl]);
    for (int32_t q = 0; q < nl; ++q) {
        const int32_t t = keep[(size_t)q];
        int32_t* dst = S.cmap.data() + S.cmap_off[(size_t)q];
        for (int32_t e : bnd[(size_t)t]) {
            const int32_t l = local(gparent[(size_t)t], e);
            if (l < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (boundary not contained in the parent's front)");
            *dst++ = l;
        }
    }
    // gather lists of the upward sweep: CSR form (tests) and one row per child (device)
    S.gptr.assign((size_t)S.g_off[(size_t)nl], 0);
    S.gidx.resize((size_t)S.cmap_off[(size_t)nl]);
    S.gell.assign((size_t)S.ge_off[(size_t)nl], -1);
    // where entry k of child c's update vector lies in the update-vector buffer (a distributed child: in the slot of the rank
    // that produces it)
    auto upos = [&](int32_t c, int32_t k) -> int32_t {
        if (S.kind[(size_t)c] != 4) return (int32_t)(S.u_off[(size_t)c] + k);
        const int32_t w = nd_slice_width(S.f[(size_t)c] - S.m[(size_t)c], S.nranks);
        return (int32_t)(S.ux_base[(size_t)c] + (int64_t)(k / w) * S.ux_stride[(size_t)c] + k % w);
    };
    {
        int64_t run = 0;
        std::vector<int32_t> count;
        for (int32_t q = 0; q < nl; ++q) {
            const int32_t f = S.f[(size_t)q];
            count.assign((size_t)f + 1, 0);
            for (int32_t cp = S.child_ptr[(size_t)q]; cp < S.child_ptr[(size_t)q + 1]; ++cp) {
                const int32_t c = S.child_idx[(size_t)cp];
                for (int32_t k = S.cmap_off[(size_t)c]; k < S.cmap_off[(size_t)c + 1]; ++k) ++count[(size_t)S.cmap[(size_t)k] + 1];
            }
            int32_t* gp = S.gptr.data() + S.g_off[(size_t)q];
            gp[0] = (int32_t)run;
            for (int32_t j = 0; j < f; ++j) gp[j + 1] = gp[j] + count[(size_t)j + 1];
            std::vector<int32_t> cur(gp, gp + f);
            for (int32_t cp = S.child_ptr[(size_t)q]; cp < S.child_ptr[(size_t)q + 1]; ++cp) {
                const int32_t c = S.child_idx[(size_t)cp];
                int32_t* row = S.gell.data() + S.ge_off[(size_t)q] + (int64_t)(cp - S.child_ptr[(size_t)q]) * f;
                const int32_t b = S.cmap_off[(size_t)c + 1] - S.cmap_off[(size_t)c];
                for (int32_t k = 0; k < b; ++k) {
                    const int32_t j = S.cmap[(size_t)S.cmap_off[(size_t)c] + k];
                    S.gidx[(size_t)cur[(size_t)j]++] = upos(c, k);
                    row[j] = upos(c, k);
                }
            }
            run = gp[f];
        }
    }
    // assembly map: every matrix entry goes to the front of the node that eliminates its first unknown, if this rank
    // factors that node
    S.asm_src.clear();
    S.asm_dst.clear();
    S.asm_src.reserve((size_t)S.nnz / (size_t)std::max(1, S.nranks) + 16);
    S.asm_dst.reserve((size_t)S.nnz / (size_t)std::max(1, S.nranks) + 16);
    for (int32_t i = 0; i < n; ++i) {
        const int32_t pi = pos[(size_t)i];
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
            const int32_t pj = pos[(size_t)ci[p]];
            const int32_t t = node_of[(size_t)std::min(pi, pj)];
            if (kind[(size_t)t] != 1 && kind[(size_t)t] != 2 && kind[(size_t)t] != 4) continue;
            int32_t li = local(t, pi);
            const int32_t lj = local(t, pj);
            if (li < 0 || lj < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (entry outside its front)");
            const int32_t q = loc[(size_t)t];
            if (kind[(size_t)t] == 4 && li >= S.m[(size_t)q]) {  // a boundary row of a distributed front: kept by the rank that owns it
                li -= S.brow0[(size_t)q];
                if (li < S.m[(size_t)q] || li >= S.m[(size_t)q] + S.brow[(size_t)q]) continue;
            }
            S.asm_src.push_back(p);
            S.asm_dst.push_back(S.front_off[(size_t)q] + (int64_t)li * S.f[(size_t)q] + lj);
        }
    }
    // work lists: this rank's own nodes level by level (larger pivot blocks first), then -- after the exchange of the
    // subtree roots -- the replicated nodes level by level
    S.lvl_ptr.assign(1, 0);
    S.lvl_nodes.clear();
    for (int phase = 1; phase <= 2; ++phase) {
        if (phase == 2) S.phase_b_level = (int32_t)S.lvl_ptr.size() - 1;
        for (int32_t l = 0; l < S.nlevels; ++l) {
            const size_t before = S.lvl_nodes.size();
            for (int32_t q = 0; q < nl; ++q)
                if ((S.kind[(size_t)q] == phase || (phase == 2 && S.kind[(size_t)q] == 4)) && S.level[(size_t)q] == l) S.lvl_nodes.push_back(q);
            if (S.lvl_nodes.size() == before) continue;
            std::stable_sort(S.lvl_nodes.begin() + (int64_t)before, S.lvl_nodes.end(), [&](int32_t x, int32_t y) { return S.m[(size_t)x] > S.m[(size_t)y]; });
            S.lvl_ptr.push_back((int32_t)S.lvl_nodes.size());
        }
    }
    S.nlevels = (int32_t)S.lvl_ptr.size() - 1;  // work levels from here on
    if (!dist) S.phase_b_level = S.nlevels;
    return LSA_OK;
}

int check_pattern(int32_t n, const int32_t* rp, const int32_t* ci, const Fail& fail) {
    if (n < 0 || !rp || (!ci && n > 0 && rp[n] > 0)) return fail(LSA_ERR_ARG, "nd_analyse: bad argument");
    for (int32_t i = 0; i < n; ++i) {
        if (rp[i + 1] < rp[i]) return fail(LSA_ERR_ARG, "nd_analyse: row pointers decrease");
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
            if (ci[p] < 0 || ci[p] >= n) return fail(LSA_ERR_ARG, "nd_analyse: column index out of range");
    }
    return LSA_OK;
}

}  // namespace

int nd_analyse(int32_t n, const int32_t* rp, const int32_t* ci, int32_t leaf_size, const int8_t* constraint, NdSymbolic* out, char* err,
               int errlen, bool order_only) {
    const Fail fail{err, errlen};
    if (!out) return fail(LSA_ERR_ARG, "nd_analyse: bad argument");
    if (int rc = check_pattern(n, rp, ci, fail)) return rc;
    if (leaf_size <= 0) leaf_size = nd_default_leaf(n);
    NdSymbolic& S = *out;
    S = NdSymbolic();
    S.order_only = order_only;
    S.n = n;
    S.nnz = n > 0 ? rp[n] : 0;
    S.leaf_size = leaf_size;
    S.pattern_hash = nd_pattern_hash(n, rp, ci);
    Graph g;
    build_graph(n, rp, ci, g);
    std::vector<std::vector<int32_t>> own;
    std::vector<int32_t> par;
    if (n > 0) dissect(g, leaf_size, own, par);
    const int32_t nt = (int32_t)own.size();
    S.constraint_hash = 0;
    if (constraint && nt > 0) {
        // Saddle-point matrices: an unknown with a (numerically) zero diagonal -- a constraint, the pressure of a mesh
        // vertex -- is eliminated in the highest tree node that owns one of its neighbours.  All its neighbours then lie
        // in that node's subtree (they are pairwise linked through the vertex's other unknowns, so they sit on one root
        // path), every pivot block sees complete constraint rows, and the Schur complement -B F^-1 B^T it receives from
        // the eliminated neighbours is what the row is pivoted on.  Without this a leaf can hold more constraints than its
        // interior unknowns support and its pivot block is singular (met on the 3D Taylor-Hood pattern).
        std::vector<int32_t> depth((size_t)nt, 0), node_of_v((size_t)n, -1);
        for (int32_t t = 0; t < nt; ++t) depth[(size_t)t] = par[(size_t)t] >= 0 ? depth[(size_t)par[(size_t)t]] + 1 : 0;  // parents are created first
        for (int32_t t = 0; t < nt; ++t)
            for (int32_t v : own[(size_t)t]) node_of_v[(size_t)v] = t;
        std::vector<int32_t> target((size_t)n, -1), left((size_t)nt, 0);
        for (int32_t t = 0; t < nt; ++t) left[(size_t)t] = (int32_t)own[(size_t)t].size();
        uint64_t h = 1469598103934665603ull;
        for (int32_t v = 0; v < n; ++v) {
            if (!constraint[v]) continue;
            h ^= (uint64_t)(uint32_t)v;
            h *= 1099511628211ull;
            const int32_t t0 = node_of_v[(size_t)v];
            int32_t best = t0;
            for (int64_t p = g.ptr[v]; p < g.ptr[(size_t)v + 1]; ++p) {
                const int32_t t = node_of_v[(size_t)g.adj[(size_t)p]];
                if (depth[(size_t)t] < depth[(size_t)best]) best = t;
            }
            if (best == t0 || left[(size_t)t0] <= 1) continue;
            int32_t a = t0;  // best must be an ancestor of t0
            while (a >= 0 && depth[(size_t)a] > depth[(size_t)best]) a = par[(size_t)a];
            if (a != best) continue;
            target[(size_t)v] = best;
            --left[(size_t)t0];
        }
        S.constraint_hash = h | 1ull;
        std::vector<std::vector<int32_t>> moved((size_t)nt);
        for (int32_t t = 0; t < nt; ++t) {
            std::vector<int32_t>& L = own[(size_t)t];
            size_t keepn = 0;
            for (int32_t v : L) {
                if (target[(size_t)v] >= 0) moved[(size_t)target[(size_t)v]].push_back(v);
                else L[keepn++] = v;
            }
            L.resize(keepn);
        }
        for (int32_t t = 0; t < nt; ++t) own[(size_t)t].insert(own[(size_t)t].end(), moved[(size_t)t].begin(), moved[(size_t)t].end());
    }
    // The top of the forest is a chain of small pivot blocks, one dependent launch per level and sweep in the solve (6.6 us
    // each on this part) for a few hundred KB of data.  The levels under each root are merged into the root, top down, while
    // the merged pivot block stays below `top_limit` unknowns: one dense block whose inverse is applied in a single launch
    // (and has no boundary: nothing to do in the downward sweep) instead of 2 x (levels) launches.  It costs a longer
    // Gauss-Jordan chain in the factorisation.  Measured at 30 k unknowns with a limit of 1100: 23 -> 19 launches, the solve
    // 151 -> 140 us, the factorisation 8.1 -> 18.6 ms (the 980-row pivot block's panels take 85 us each): a loss, so the
    // default is off; LSA_ND_TOP=<unknowns> turns it on.
    int32_t top_limit = 0;  // off: the longer Gauss-Jordan chain costs more than the launches it saves (S30k: + 10 ms per factorisation for - 11 us per solve)
    if (const char* e = getenv("LSA_ND_TOP")) top_limit = atoi(e);
    // Pairs of levels: a separator node at an even depth takes in the separators of its children (not their leaves) while the
    // merged pivot block stays below `pair_limit` unknowns -- a four-way dissection with half as many levels, i.e. half as many
    // dependent launches in both sweeps, for denser pivot blocks and a longer panel chain per level.
    // On by default where the sweeps are bound by their launch chain (measured: 30 k unknowns, 19 -> 13 launches, 122 -> 96 us per
    // apply for + 1.3 ms of factorisation, a solve 49.1 -> 45.6 ms; 38 k unknowns in 3D - 10 % per apply; 121 k unknowns - 9 %);
    // from a few hundred thousand unknowns on the sweeps are bound by the bytes of the factors, which merging increases
    // (504 k unknowns: + 10 % bytes, + 2 % time): off.  The limit keeps merged blocks on the panel path (below LSA_ND_TP_MIN = 512).
    int32_t pair_limit = n <= 200000 ? 512 : 0;
    if (const char* e = getenv("LSA_ND_PAIR")) pair_limit = atoi(e);
    if ((top_limit > 0 || pair_limit > 0) && nt > 1) {
        std::vector<std::vector<int32_t>> kids((size_t)nt);
        std::vector<int32_t> height((size_t)nt, 0);
        for (int32_t t = nt - 1; t >= 0; --t)  // parents are created before their children
            if (par[(size_t)t] >= 0) {
                kids[(size_t)par[(size_t)t]].push_back(t);
                height[(size_t)par[(size_t)t]] = std::max(height[(size_t)par[(size_t)t]], height[(size_t)t] + 1);
            }
        std::vector<int32_t> into((size_t)nt, -1);  // node -> the root it is merged into
        if (pair_limit > 0) {
            std::vector<int32_t> depth((size_t)nt, 0);
            for (int32_t t = 0; t < nt; ++t) depth[(size_t)t] = par[(size_t)t] >= 0 ? depth[(size_t)par[(size_t)t]] + 1 : 0;
            for (int32_t t = 0; t < nt; ++t) {
                if (into[(size_t)t] >= 0 || (depth[(size_t)t] & 1)) continue;
                int64_t add = 0;
                for (int32_t c : kids[(size_t)t])
                    if (!kids[(size_t)c].empty()) add += (int64_t)own[(size_t)c].size();
                if (add == 0 || (int64_t)own[(size_t)t].size() + add > pair_limit) continue;
                for (int32_t c : kids[(size_t)t])
                    if (!kids[(size_t)c].empty()) into[(size_t)c] = t;
            }
        }
        for (int32_t r = 0; r < nt && top_limit > 0; ++r) {
            if (par[(size_t)r] >= 0) continue;
            int64_t cum = (int64_t)own[(size_t)r].size();
            std::vector<int32_t> layer{r};
            while (true) {
                std::vector<int32_t> next;
                int64_t add = 0;
                for (int32_t t : layer)
                    for (int32_t c : kids[(size_t)t])
                        if (!kids[(size_t)c].empty()) {  // leaves stay: they carry the bulk of the unknowns
                            next.push_back(c);
                            add += (int64_t)own[(size_t)c].size();
                        }
                if (next.empty() || cum + add > top_limit) break;
                for (int32_t c : next) into[(size_t)c] = r;
                cum += add;
                layer.swap(next);
            }
        }
        bool any = false;
        for (int32_t t = 0; t < nt; ++t) any |= into[(size_t)t] >= 0;
        if (any) {
            // merged nodes hand their unknowns to the root (deeper nodes first: the order inside one pivot block is free) and
            // their remaining children to it
            for (int32_t t = nt - 1; t >= 0; --t)
                if (into[(size_t)t] >= 0) {
                    std::vector<int32_t>& dst = own[(size_t)into[(size_t)t]];
                    dst.insert(dst.begin(), own[(size_t)t].begin(), own[(size_t)t].end());
                    own[(size_t)t].clear();
                }
            std::vector<int32_t> remap((size_t)nt, -1);
            std::vector<std::vector<int32_t>> own2;
            std::vector<int32_t> par2;
            for (int32_t t = 0; t < nt; ++t)
                if (into[(size_t)t] < 0) {
                    remap[(size_t)t] = (int32_t)own2.size();
                    own2.push_back(std::move(own[(size_t)t]));
                    int32_t p = par[(size_t)t];
                    while (p >= 0 && into[(size_t)p] >= 0) p = into[(size_t)p];
                    par2.push_back(p);  // still an old id: parents precede children, so remap[p] is known
                }
            for (int32_t& p : par2)
                if (p >= 0) p = remap[(size_t)p];
            own.swap(own2);
            par.swap(par2);
        }
    }
    return nd_finish(S, n, rp, ci, g, own, par, nullptr, 0, 1, fail);
}

// The same tables for a tree the caller provides (nodes in any parent-consistent order; node t owns the matrix indices
// [first[t], first[t] + size[t])), optionally split over ranks: owner[t] = rank of the subtree the node belongs to, -1 for
// the replicated top.  Matrix rows that belong to no node must be empty (the padding of the sharded block layout).
int nd_analyse_tree(int32_t n, const int32_t* rp, const int32_t* ci, int32_t nt, const int32_t* first, const int32_t* size, const int32_t* parent,
                    const int32_t* owner, int rank, int nranks, NdSymbolic* out, char* err, int errlen) {
    const Fail fail{err, errlen};
    if (!out || nt < 0 || (nt > 0 && (!first || !size || !parent))) return fail(LSA_ERR_ARG, "nd_analyse_tree: bad argument");
    if (int rc = check_pattern(n, rp, ci, fail)) return rc;
    NdSymbolic& S = *out;
    S = NdSymbolic();
    S.n = n;
    S.nnz = n > 0 ? rp[n] : 0;
    S.leaf_size = 0;
    S.pattern_hash = nd_pattern_hash(n, rp, ci);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](int32_t v) {
        h ^= (uint64_t)(uint32_t)v;
        h *= 1099511628211ull;
    };
    std::vector<std::vector<int32_t>> own((size_t)nt);
    std::vector<int32_t> par(parent, parent + nt);
    for (int32_t t = 0; t < nt; ++t) {
        if (first[t] < 0 || size[t] < 1 || (int64_t)first[t] + size[t] > n || parent[t] >= nt || parent[t] == t)
            return fail(LSA_ERR_ARG, "nd_analyse_tree: node range or parent out of bounds");
        own[(size_t)t].resize((size_t)size[t]);
        std::iota(own[(size_t)t].begin(), own[(size_t)t].end(), first[t]);
        mix(first[t]);
        mix(size[t]);
        mix(parent[t]);
        mix(owner ? owner[t] : 0);
    }
    mix(rank);
    mix(nranks);
    S.tree_hash = h | 1ull;
    Graph g;
    build_graph(n, rp, ci, g);
    return nd_finish(S, n, rp, ci, g, own, par, owner, rank, nranks, fail);
}

// ---- memory plan -------------------------------------------------------------------------------------------------------------
namespace {
// first-fit allocator over [0, inf) with a coalescing free list, run on the host over the chunk order: the offsets it hands
// out are the update arena's layout, its high-water mark the arena's size
struct ArenaPlan {
    std::vector<std::pair<int64_t, int64_t>> free_;  // (offset, size), sorted by offset
    int64_t top = 0;                                  // high-water mark
    int64_t alloc(int64_t size) {
        if (size <= 0) return 0;
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= size) {
                const int64_t off = free_[i].first;
                if (free_[i].second == size) free_.erase(free_.begin() + (int64_t)i);
                else free_[i] = {off + size, free_[i].second - size};
                return off;
            }
        if (!free_.empty() && free_.back().first + free_.back().second == top) {  // grow the block at the end
            const int64_t off = free_.back().first;
            top = off + size;
            free_.pop_back();
            return off;
        }
        const int64_t off = top;
        top += size;
        return off;
    }
    void release(int64_t off, int64_t size) {
        if (size <= 0) return;
        auto it = std::lower_bound(free_.begin(), free_.end(), std::make_pair(off, (int64_t)0));
        it = free_.insert(it, {off, size});
        if (it + 1 != free_.end() && it->first + it->second == (it + 1)->first) {
            it->second += (it + 1)->second;
            free_.erase(it + 1);
        }
        if (it != free_.begin() && (it - 1)->first + (it - 1)->second == it->first) {
            (it - 1)->second += it->second;
            free_.erase(it);
        }
    }
};
}  // namespace

void nd_memory_plan(const NdSymbolic& S, int64_t budget_entries, NdMemoryPlan& P) {
    const int32_t nt = S.nt;
    const bool dist = S.nranks > 1;
    auto is_top = [&](int32_t k) { return k == 2 || k == 4; };
    // subtree roots below the top (see nd_finish): _u = its update vector is exchanged; plain = its update matrix too, through the
    // exchange region (replicated parent); _4 = its update matrix travels in row chunks to a distributed parent
    auto is_xroot_u = [&](int32_t q) { return dist && !is_top(S.kind[(size_t)q]) && S.parent[(size_t)q] >= 0 && is_top(S.kind[(size_t)S.parent[(size_t)q]]); };
    auto is_xroot = [&](int32_t q) { return dist && !is_top(S.kind[(size_t)q]) && S.parent[(size_t)q] >= 0 && S.kind[(size_t)S.parent[(size_t)q]] == 2; };
    // scalars of a node's working front / of the update matrix it keeps on this rank
    // (a distributed node: the widest slice of any rank, so that the chunks of the top levels -- their cuts decide where the
    //  collectives of the factorisation fall -- come out alike on every rank)
    auto work_size = [&](int32_t t) -> int64_t {
        if (S.kind[(size_t)t] != 4) return (int64_t)S.f[(size_t)t] * S.f[(size_t)t];
        // ... its slice of the front, and behind it the whole inverse of the pivot block while the node is factored (the packed
        // factors keep only this rank's rows of it)
        const int64_t rows = S.m[(size_t)t] + nd_slice_width(S.f[(size_t)t] - S.m[(size_t)t], S.nranks);
        return rows * S.f[(size_t)t] + (int64_t)S.m[(size_t)t] * S.m[(size_t)t];
    };
    auto upd_size = [&](int32_t t) -> int64_t {
        if (S.kind[(size_t)t] == 3 && !is_xroot(t)) return 0;  // another rank's subtree root under a distributed parent: arrives in chunks
        return (int64_t)S.brow[(size_t)t] * (S.f[(size_t)t] - S.m[(size_t)t]);
    };
    P = NdMemoryPlan();
    P.work_off.assign((size_t)nt, 0);
    P.upd_off.assign((size_t)nt, 0);
    P.lfac_off.assign((size_t)nt, 0);
    P.ufac_off.assign((size_t)nt, 0);
    P.acc_off.assign((size_t)nt, -1);
    P.pacc_off.assign((size_t)nt, -1);
    // ---- packed factors ----
    for (int32_t t = 0; t < nt; ++t) {
        if (S.kind[(size_t)t] == 3) continue;  // another rank's subtree root: only its update matrix and vector arrive here
        const int64_t m = S.m[(size_t)t], ff = S.f[(size_t)t];
        P.lfac_off[(size_t)t] = P.lfac_entries;
        P.ufac_off[(size_t)t] = P.ufac_entries;
        P.lfac_entries += ((int64_t)S.orows[(size_t)t] + S.brow[(size_t)t]) * m;  // rows of the inverse (all m unless distributed), then of -F21 inv
        P.ufac_entries += (int64_t)S.orows[(size_t)t] * (ff - m);
    }
    // ---- slot rows of the upward sweep: push form unless a child's update vector arrives by all-gather (then the node pulls) ----
    for (int32_t t = 0; t < nt; ++t) {
        const int32_t c0 = S.child_ptr[(size_t)t], c1 = S.child_ptr[(size_t)t + 1];
        bool pull = S.kind[(size_t)t] == 4;
        for (int32_t cp = c0; cp < c1; ++cp) pull |= is_xroot_u(S.child_idx[(size_t)cp]);
        if (c1 == c0 || pull || S.kind[(size_t)t] == 3) continue;
        P.acc_off[(size_t)t] = P.acc_entries;
        for (int32_t cp = c0; cp < c1; ++cp) P.pacc_off[(size_t)S.child_idx[(size_t)cp]] = P.acc_entries + (int64_t)(cp - c0) * S.f[(size_t)t];
        P.acc_entries += (int64_t)(c1 - c0) * S.f[(size_t)t];
    }
    // ---- chunks: the nodes of a work level, larger pivot blocks first, cut where the working fronts would outgrow the arena ----
    for (int32_t l = 0; l < S.nlevels; ++l) {
        int64_t sum = 0;
        for (int32_t q = S.lvl_ptr[(size_t)l]; q < S.lvl_ptr[(size_t)l + 1]; ++q) {
            const int64_t w = work_size(S.lvl_nodes[(size_t)q]);
            sum += w;
            P.max_front_entries = std::max(P.max_front_entries, w);
        }
        P.max_level_entries = std::max(P.max_level_entries, sum);
    }
    const int64_t budget = std::max(P.max_front_entries, budget_entries > 0 ? std::min(P.max_level_entries, budget_entries) : P.max_level_entries);
    for (int32_t l = 0; l < S.nlevels; ++l) {
        int64_t used = 0;
        for (int32_t q = S.lvl_ptr[(size_t)l]; q < S.lvl_ptr[(size_t)l + 1]; ++q) {
            const int32_t t = S.lvl_nodes[(size_t)q];
            const int64_t w = work_size(t);
            if (q == S.lvl_ptr[(size_t)l] || used + w > budget) {
                P.chunk_begin.push_back(q);
                P.chunk_work.push_back(0);
                P.chunk_exchange_before.push_back(dist && l == S.phase_b_level && q == S.lvl_ptr[(size_t)l]);
                used = 0;
            }
            P.work_off[(size_t)t] = used;
            used += w;
            P.chunk_work.back() = used;
            P.work_entries = std::max(P.work_entries, used);
        }
    }
    P.chunk_begin.push_back(S.lvl_ptr.empty() ? 0 : S.lvl_ptr.back());
    // ---- update arena ----
    if (dist) {
        std::vector<int64_t> use((size_t)S.nranks, 0);
        for (int32_t q = 0; q < nt; ++q)
            if (is_xroot(q)) {
                const int64_t b = S.f[(size_t)q] - S.m[(size_t)q];
                use[(size_t)S.owner[(size_t)q]] += b * b;
            }
        for (int64_t v : use) P.xupd_slot = std::max(P.xupd_slot, v);
        std::fill(use.begin(), use.end(), 0);
        for (int32_t q = 0; q < nt; ++q)
            if (is_xroot(q)) {  // (same order, same sizes on every rank: the in-place all-gather moves slot r of rank r)
                const int64_t b = S.f[(size_t)q] - S.m[(size_t)q];
                const int32_t o = S.owner[(size_t)q];
                P.upd_off[(size_t)q] = P.xupd_slot * o + use[(size_t)o];
                use[(size_t)o] += b * b;
            }
        // staging of the update rows that travel to distributed parents: one slot per rank, a slot holds whole rows of one child
        if (S.has_dist) {
            int64_t widest = 0;
            for (int32_t q = 0; q < nt; ++q)
                if (S.parent[(size_t)q] >= 0 && S.kind[(size_t)S.parent[(size_t)q]] == 4 && S.kind[(size_t)q] != 2)
                    widest = std::max<int64_t>(widest, S.f[(size_t)q] - S.m[(size_t)q]);
            int64_t slot = (int64_t)(32u << 20) / 8;  // 32 MB of float64 per rank and step
            if (const char* e = getenv("LSA_ND_XSTAGE_KB")) slot = std::max<int64_t>(1, atoll(e)) * 1024 / 8;  // (tests: many small steps)
            P.xstage_slot = widest > 0 ? std::max(slot, widest) : 0;
        }
    }
    ArenaPlan arena;
    const int64_t base = P.xupd_slot * S.nranks;
    for (size_t c = 0; c + 1 < P.chunk_begin.size(); ++c) {
        for (int32_t q = P.chunk_begin[c]; q < P.chunk_begin[c + 1]; ++q) {  // blocks written by this chunk
            const int32_t t = S.lvl_nodes[(size_t)q];
            if (upd_size(t) > 0 && !is_xroot(t)) P.upd_off[(size_t)t] = base + arena.alloc(upd_size(t));
        }
        // blocks consumed by this chunk are free from the next chunk on.  (This chunk's own blocks were placed first: a block a
        // parent reads in this chunk's extend-add must not be handed to a node that is saved in the same chunk.)
        for (int32_t q = P.chunk_begin[c]; q < P.chunk_begin[c + 1]; ++q) {
            const int32_t t = S.lvl_nodes[(size_t)q];
            for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp) {
                const int32_t ch = S.child_idx[(size_t)cp];
                if (!is_xroot(ch) && S.kind[(size_t)ch] != 3) arena.release(P.upd_off[(size_t)ch] - base, upd_size(ch));
            }
        }
    }
    P.upd_entries = std::max<int64_t>(base + arena.top, 1);
}

// ---- C-ABI: analysis only (host) -----------------------------------------------------------------------------------------
struct lsa_nd_sym {
    NdSymbolic S;
    std::string err;
};

extern "C" {

int lsa_nd_analyse(int32_t n, const int32_t* rowptr, const int32_t* col, int32_t leaf_size, const int8_t* constraint, lsa_nd_sym** out) {
    if (!out) return LSA_ERR_ARG;
    lsa_nd_sym* h = new lsa_nd_sym();
    char buf[256] = {0};
    int rc;
    try {
        rc = nd_analyse(n, rowptr, col, leaf_size, constraint, &h->S, buf, (int)sizeof buf);
    } catch (const std::bad_alloc&) {
        rc = LSA_ERR_ARG;
        snprintf(buf, sizeof buf, "nd_analyse: out of host memory");
    }
    h->err = buf;
    *out = h;  // returned on failure too, so that the message can be read; release with lsa_nd_sym_destroy
    return rc;
}

int lsa_nd_order(int32_t n, const int32_t* rowptr, const int32_t* col, int32_t leaf_size, const int8_t* constraint, lsa_nd_sym** out) {
    if (!out) return LSA_ERR_ARG;
    lsa_nd_sym* h = new lsa_nd_sym();
    char buf[256] = {0};
    int rc;
    try {
        rc = nd_analyse(n, rowptr, col, leaf_size, constraint, &h->S, buf, (int)sizeof buf, true);
    } catch (const std::bad_alloc&) {
        rc = LSA_ERR_ARG;
        snprintf(buf, sizeof buf, "nd_order: out of host memory");
    }
    h->err = buf;
    *out = h;
    return rc;
}

int lsa_nd_analyse_tree(int32_t n, const int32_t* rowptr, const int32_t* col, int32_t ntree, const int32_t* first, const int32_t* size,
                        const int32_t* parent, const int32_t* owner, int32_t rank, int32_t nranks, lsa_nd_sym** out) {
    if (!out) return LSA_ERR_ARG;
    lsa_nd_sym* h = new lsa_nd_sym();
    char buf[256] = {0};
    int rc;
    try {
        rc = nd_analyse_tree(n, rowptr, col, ntree, first, size, parent, owner, rank, nranks, &h->S, buf, (int)sizeof buf);
    } catch (const std::bad_alloc&) {
        rc = LSA_ERR_ARG;
        snprintf(buf, sizeof buf, "nd_analyse_tree: out of host memory");
    }
    h->err = buf;
    *out = h;
    return rc;
}

const char* lsa_nd_sym_error(const lsa_nd_sym* h) { return h ? h->err.c_str() : "null handle"; }

void lsa_nd_sym_destroy(lsa_nd_sym* h) { delete h; }

int lsa_nd_sym_info(const lsa_nd_sym* h, int32_t* ntree, int32_t* nlevels, int32_t* max_front, int64_t* index_entries,
                    int64_t* factor_entries, int64_t* front_entries, double* flops) {
    if (!h) return LSA_ERR_ARG;
    const NdSymbolic& S = h->S;
    if (ntree) *ntree = S.nt;
    if (nlevels) *nlevels = S.nlevels;
    if (max_front) {
        int32_t mf = 0;
        for (int32_t v : S.f) mf = std::max(mf, v);
        *max_front = mf;
    }
    if (index_entries) *index_entries = S.idx_off.empty() ? 0 : S.idx_off.back();
    if (factor_entries) *factor_entries = S.factor_entries;
    if (front_entries) *front_entries = S.front_entries;
    if (flops) *flops = S.flops;
    return LSA_OK;
}

int lsa_nd_sym_memory(const lsa_nd_sym* h, int32_t scalar_bytes, int64_t work_budget_bytes, int64_t* out, int64_t* upd_off, int64_t* work_off, int32_t* chunk_of) {
    if (!h || !out || (scalar_bytes != 8 && scalar_bytes != 16)) return LSA_ERR_ARG;
    const NdSymbolic& S = h->S;
    if (S.order_only) return LSA_ERR_ARG;
    NdMemoryPlan P;
    nd_memory_plan(S, work_budget_bytes > 0 ? work_budget_bytes / scalar_bytes : 0, P);
    out[0] = (P.lfac_entries + P.ufac_entries) * scalar_bytes;  // packed factors
    out[1] = P.work_entries * scalar_bytes;                    // working fronts of the largest chunk
    const int64_t xstage = P.xstage_slot * S.nranks * scalar_bytes;  // staging of the update rows on their way to distributed parents
    out[2] = P.upd_entries * scalar_bytes + xstage;            // update arena (with the exchange region of a forest cut over ranks, and the staging)
    out[3] = P.xupd_slot * S.nranks * scalar_bytes + xstage;   // ... of which the exchange region and the staging
    out[4] = P.acc_entries * 16 + 2 * S.u_off[(size_t)S.nt] * 16 + S.xg_entries * 16;  // sweep buffers: slot rows, update and boundary vectors, own-row exchange (complex vectors)
    out[5] = (int64_t)P.chunk_begin.size() - 1;                // chunks
    out[6] = P.max_front_entries * scalar_bytes;               // the largest front
    out[7] = (int64_t)(S.idx.size() + S.gell.size() + S.cmap.size()) * 4 + (int64_t)S.asm_src.size() * 12 + (int64_t)S.nt * 96 * 2;  // index tables
    {
        // ... and the per-unknown staging of the elimination: pivot rows of a finished block (32 rows per unknown; 128 where the
        // pivot blocks are inverted in super-blocks, from 1 024 rows on by default), pivot maps, one vector
        int32_t max_m = 0;
        for (int32_t t = 0; t < S.nt; ++t) max_m = std::max(max_m, S.m[(size_t)t]);
        out[7] += (int64_t)S.n * ((max_m >= 1024 ? 128 : 32) * scalar_bytes + 8 + 16);
    }
    if (upd_off) std::copy(P.upd_off.begin(), P.upd_off.end(), upd_off);
    if (work_off) std::copy(P.work_off.begin(), P.work_off.end(), work_off);
    if (chunk_of)
        for (size_t c = 0; c + 1 < P.chunk_begin.size(); ++c)
            for (int32_t q = P.chunk_begin[c]; q < P.chunk_begin[c + 1]; ++q) chunk_of[(size_t)S.lvl_nodes[(size_t)q]] = (int32_t)c;
    return LSA_OK;
}

int lsa_nd_sym_export(const lsa_nd_sym* h, int32_t* perm, int32_t* node_start, int32_t* parent, int32_t* level, int32_t* front_size,
                      int32_t* idx) {
    if (!h) return LSA_ERR_ARG;
    const NdSymbolic& S = h->S;
    if (perm) std::copy(S.perm.begin(), S.perm.end(), perm);
    if (node_start) std::copy(S.node_start.begin(), S.node_start.end(), node_start);
    if (parent) std::copy(S.parent.begin(), S.parent.end(), parent);
    if (level) std::copy(S.level.begin(), S.level.end(), level);
    if (front_size) std::copy(S.f.begin(), S.f.end(), front_size);
    if (idx) std::copy(S.idx.begin(), S.idx.end(), idx);
    return LSA_OK;
}

int lsa_nd_sym_export_dist(const lsa_nd_sym* h, int32_t* kind, int64_t* front_off, int64_t* u_off, int32_t* asm_src, int32_t* children_ptr,
                           int32_t* children_idx, int64_t* scalars) {
    if (!h) return LSA_ERR_ARG;
    const NdSymbolic& S = h->S;
    if (kind) std::copy(S.kind.begin(), S.kind.end(), kind);
    if (front_off) std::copy(S.front_off.begin(), S.front_off.end(), front_off);
    if (u_off) std::copy(S.u_off.begin(), S.u_off.end(), u_off);
    if (asm_src) std::copy(S.asm_src.begin(), S.asm_src.end(), asm_src);
    if (children_ptr) std::copy(S.child_ptr.begin(), S.child_ptr.end(), children_ptr);
    if (children_idx) std::copy(S.child_idx.begin(), S.child_idx.end(), children_idx);
    if (scalars) {
        scalars[0] = S.xfront_slot;
        scalars[1] = S.xu_slot;
        scalars[2] = S.phase_b_level;
        scalars[3] = (int64_t)S.asm_src.size();
        scalars[4] = S.nranks;
        scalars[5] = S.rank;
    }
    return LSA_OK;
}

int lsa_nd_sym_export_top(const lsa_nd_sym* h, int32_t* owner, int32_t* rows, int64_t* exch, int64_t* totals) {
    if (!h) return LSA_ERR_ARG;
    const NdSymbolic& S = h->S;
    if (S.order_only) return LSA_ERR_ARG;
    for (int32_t q = 0; q < S.nt; ++q) {
        if (owner) owner[q] = S.owner[(size_t)q];
        if (rows) {
            rows[4 * q] = S.brow0[(size_t)q];
            rows[4 * q + 1] = S.brow[(size_t)q];
            rows[4 * q + 2] = S.orow0[(size_t)q];
            rows[4 * q + 3] = S.orows[(size_t)q];
        }
        if (exch) {
            exch[4 * q] = S.ux_base[(size_t)q];
            exch[4 * q + 1] = S.ux_stride[(size_t)q];
            exch[4 * q + 2] = S.xg_base[(size_t)q];
            exch[4 * q + 3] = S.xg_stride[(size_t)q];
        }
    }
    if (totals) {
        totals[0] = S.u_off[(size_t)S.nt];
        totals[1] = S.xg_entries;
    }
    return LSA_OK;
}

int lsa_nd_sym_export_tables(const lsa_nd_sym* h, int32_t* cmap, int32_t* gptr, int32_t* gidx, int64_t* asm_dst, int32_t* lvl_ptr,
                             int32_t* lvl_nodes) {
    if (!h) return LSA_ERR_ARG;
    const NdSymbolic& S = h->S;
    if (cmap) std::copy(S.cmap.begin(), S.cmap.end(), cmap);
    if (gptr) std::copy(S.gptr.begin(), S.gptr.end(), gptr);
    if (gidx) std::copy(S.gidx.begin(), S.gidx.end(), gidx);
    if (asm_dst) std::copy(S.asm_dst.begin(), S.asm_dst.end(), asm_dst);
    if (lvl_ptr) std::copy(S.lvl_ptr.begin(), S.lvl_ptr.end(), lvl_ptr);
    if (lvl_nodes) std::copy(S.lvl_nodes.begin(), S.lvl_nodes.end(), lvl_nodes);
    return LSA_OK;
}

}  // extern "C"
