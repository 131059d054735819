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
            if (std::min(below, above) * 10 >= total * 3 && (best < 0 || cnt[(size_t)k] < best_cnt)) {
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

int nd_analyse(int32_t n, const int32_t* rp, const int32_t* ci, int32_t leaf_size, const int8_t* constraint, NdSymbolic* out, char* err,
               int errlen) {
    auto fail = [&](int code, const char* msg) {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", msg);
        return code;
    };
    if (n < 0 || !rp || (!ci && n > 0 && rp[n] > 0) || !out) return fail(LSA_ERR_ARG, "nd_analyse: bad argument");
    if (leaf_size <= 0) leaf_size = 128;
    NdSymbolic& S = *out;
    S = NdSymbolic();
    S.n = n;
    S.nnz = n > 0 ? rp[n] : 0;
    S.leaf_size = leaf_size;
    for (int32_t i = 0; i < n; ++i) {
        if (rp[i + 1] < rp[i]) return fail(LSA_ERR_ARG, "nd_analyse: row pointers decrease");
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p)
            if (ci[p] < 0 || ci[p] >= n) return fail(LSA_ERR_ARG, "nd_analyse: column index out of range");
    }
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
            size_t keep = 0;
            for (int32_t v : L) {
                if (target[(size_t)v] >= 0) moved[(size_t)target[(size_t)v]].push_back(v);
                else L[keep++] = v;
            }
            L.resize(keep);
        }
        for (int32_t t = 0; t < nt; ++t) own[(size_t)t].insert(own[(size_t)t].end(), moved[(size_t)t].begin(), moved[(size_t)t].end());
    }
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
    S.nt = nt;
    S.perm.reserve((size_t)n);
    S.node_start.assign((size_t)nt + 1, 0);
    S.parent.assign((size_t)nt, -1);
    S.m.assign((size_t)nt, 0);
    for (int32_t k = 0; k < nt; ++k) {
        const int32_t t = seq[(size_t)k];
        S.perm.insert(S.perm.end(), own[(size_t)t].begin(), own[(size_t)t].end());
        S.node_start[(size_t)k + 1] = (int32_t)S.perm.size();
        S.m[(size_t)k] = (int32_t)own[(size_t)t].size();
        S.parent[(size_t)k] = par[t] >= 0 ? newid[(size_t)par[t]] : -1;
    }
    if ((int32_t)S.perm.size() != n) return fail(LSA_ERR_ARG, "nd_analyse: internal error (ordering is not a permutation)");
    own.clear();
    // children lists, levels
    S.child_ptr.assign((size_t)nt + 1, 0);
    for (int32_t t = 0; t < nt; ++t)
        if (S.parent[(size_t)t] >= 0) ++S.child_ptr[(size_t)S.parent[(size_t)t] + 1];
    for (int32_t t = 0; t < nt; ++t) S.child_ptr[(size_t)t + 1] += S.child_ptr[(size_t)t];
    S.child_idx.assign((size_t)S.child_ptr[(size_t)nt], 0);
    {
        std::vector<int32_t> fillc(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int32_t t = 0; t < nt; ++t)
            if (S.parent[(size_t)t] >= 0) S.child_idx[(size_t)fillc[(size_t)S.parent[(size_t)t]]++] = t;
    }
    S.level.assign((size_t)nt, 0);
    for (int32_t t = 0; t < nt; ++t) {
        const int32_t p = S.parent[(size_t)t];
        if (p >= 0) S.level[(size_t)p] = std::max(S.level[(size_t)p], S.level[(size_t)t] + 1);
        S.max_children = std::max(S.max_children, S.child_ptr[(size_t)t + 1] - S.child_ptr[(size_t)t]);
    }
    S.nlevels = 0;
    for (int32_t t = 0; t < nt; ++t) S.nlevels = std::max(S.nlevels, S.level[(size_t)t] + 1);
    // elimination position of every unknown, node of every position
    std::vector<int32_t> pos((size_t)n), node_of((size_t)n);
    for (int32_t k = 0; k < n; ++k) pos[(size_t)S.perm[(size_t)k]] = k;
    for (int32_t t = 0; t < nt; ++t)
        for (int32_t k = S.node_start[(size_t)t]; k < S.node_start[(size_t)t + 1]; ++k) node_of[(size_t)k] = t;
    // boundary (struct) of every node in elimination positions
    std::vector<std::vector<int32_t>> bnd((size_t)nt);
    {
        std::vector<int32_t> stamp((size_t)n, -1);
        for (int32_t t = 0; t < nt; ++t) {
            const int32_t a = S.node_start[(size_t)t], b = S.node_start[(size_t)t + 1];
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
            for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp)
                for (int32_t q : bnd[(size_t)S.child_idx[(size_t)cp]])
                    if (q >= b && stamp[(size_t)q] != t) {
                        stamp[(size_t)q] = t;
                        L.push_back(q);
                    }
            std::sort(L.begin(), L.end());
            // every boundary unknown must belong to an ancestor
            if (!L.empty() && S.parent[(size_t)t] < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (root with a boundary)");
        }
    }
    g.adj.clear();
    g.adj.shrink_to_fit();
    S.f.assign((size_t)nt, 0);
    S.idx_off.assign((size_t)nt + 1, 0);
    S.front_off.assign((size_t)nt + 1, 0);
    S.u_off.assign((size_t)nt + 1, 0);
    S.cmap_off.assign((size_t)nt + 1, 0);
    S.g_off.assign((size_t)nt + 1, 0);
    for (int32_t t = 0; t < nt; ++t) {
        const int64_t m = S.m[(size_t)t], b = (int64_t)bnd[(size_t)t].size(), f = m + b;
        S.f[(size_t)t] = (int32_t)f;
        S.idx_off[(size_t)t + 1] = S.idx_off[(size_t)t] + f;
        S.front_off[(size_t)t + 1] = S.front_off[(size_t)t] + f * f;
        S.u_off[(size_t)t + 1] = S.u_off[(size_t)t] + b;
        S.cmap_off[(size_t)t + 1] = S.cmap_off[(size_t)t] + (int32_t)b;
        S.g_off[(size_t)t + 1] = S.g_off[(size_t)t] + f + 1;
        S.factor_entries += m * m + 2 * m * b;
        S.flops += (double)m * m * m + 2.0 * m * m * b + (double)m * b * b;
    }
    S.front_entries = S.front_off[(size_t)nt];
    if (S.idx_off[(size_t)nt] > 0x7fffffff || S.u_off[(size_t)nt] > 0x7fffffff)
        return fail(LSA_ERR_ARG, "nd_analyse: front index lists exceed 2^31 entries");
    S.idx.resize((size_t)S.idx_off[(size_t)nt]);
    for (int32_t t = 0; t < nt; ++t) {
        int32_t* dst = S.idx.data() + S.idx_off[(size_t)t];
        for (int32_t k = S.node_start[(size_t)t]; k < S.node_start[(size_t)t + 1]; ++k) *dst++ = S.perm[(size_t)k];
        for (int32_t q : bnd[(size_t)t]) *dst++ = S.perm[(size_t)q];
    }
    // local front position of elimination position q in node t
    auto local = [&](int32_t t, int32_t q) -> int32_t {
        const int32_t a = S.node_start[(size_t)t], b = S.node_start[(size_t)t + 1];
        if (q >= a && q < b) return q - a;
        const std::vector<int32_t>& L = bnd[(size_t)t];
        auto itp = std::lower_bound(L.begin(), L.end(), q);
        if (itp == L.end() || *itp != q) return -1;
        return (b - a) + (int32_t)(itp - L.begin());
    };
    // position of every boundary unknown in the parent's front
    S.cmap.resize((size_t)S.cmap_off[(size_t)nt]);
    for (int32_t t = 0; t < nt; ++t) {
        const int32_t p = S.parent[(size_t)t];
        int32_t* dst = S.cmap.data() + S.cmap_off[(size_t)t];
        for (int32_t q : bnd[(size_t)t]) {
            const int32_t l = local(p, q);
            if (l < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (boundary not contained in the parent's front)");
            *dst++ = l;
        }
    }
    // gather lists of the forward solve
    S.gptr.assign((size_t)S.g_off[(size_t)nt], 0);
    S.gidx.resize((size_t)S.u_off[(size_t)nt]);
    {
        int64_t run = 0;
        std::vector<int32_t> count;
        for (int32_t t = 0; t < nt; ++t) {
            const int32_t f = S.f[(size_t)t];
            count.assign((size_t)f + 1, 0);
            for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp) {
                const int32_t c = S.child_idx[(size_t)cp];
                for (int32_t k = S.cmap_off[(size_t)c]; k < S.cmap_off[(size_t)c + 1]; ++k) ++count[(size_t)S.cmap[(size_t)k] + 1];
            }
            int32_t* gp = S.gptr.data() + S.g_off[(size_t)t];
            gp[0] = (int32_t)run;
            for (int32_t j = 0; j < f; ++j) gp[j + 1] = gp[j] + count[(size_t)j + 1];
            std::vector<int32_t> cur(gp, gp + f);
            for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp) {
                const int32_t c = S.child_idx[(size_t)cp];
                const int32_t b = S.cmap_off[(size_t)c + 1] - S.cmap_off[(size_t)c];
                for (int32_t k = 0; k < b; ++k) {
                    const int32_t j = S.cmap[(size_t)S.cmap_off[(size_t)c] + k];
                    S.gidx[(size_t)cur[(size_t)j]++] = (int32_t)(S.u_off[(size_t)c] + k);
                }
            }
            run = gp[f];
        }
    }
    // the gather lists again, one row per child
    S.ge_off.assign((size_t)nt + 1, 0);
    for (int32_t t = 0; t < nt; ++t)
        S.ge_off[(size_t)t + 1] = S.ge_off[(size_t)t] + (int64_t)(S.child_ptr[(size_t)t + 1] - S.child_ptr[(size_t)t]) * S.f[(size_t)t];
    S.gell.assign((size_t)S.ge_off[(size_t)nt], -1);
    for (int32_t t = 0; t < nt; ++t) {
        const int32_t f = S.f[(size_t)t];
        for (int32_t cp = S.child_ptr[(size_t)t]; cp < S.child_ptr[(size_t)t + 1]; ++cp) {
            const int32_t c = S.child_idx[(size_t)cp];
            int32_t* row = S.gell.data() + S.ge_off[(size_t)t] + (int64_t)(cp - S.child_ptr[(size_t)t]) * f;
            const int32_t b = S.cmap_off[(size_t)c + 1] - S.cmap_off[(size_t)c];
            for (int32_t k = 0; k < b; ++k) row[S.cmap[(size_t)S.cmap_off[(size_t)c] + k]] = (int32_t)(S.u_off[(size_t)c] + k);
        }
    }
    // assembly map of the original entries
    S.asm_src.resize((size_t)S.nnz);
    S.asm_dst.resize((size_t)S.nnz);
    for (int32_t i = 0; i < n; ++i) {
        const int32_t pi = pos[(size_t)i];
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
            const int32_t pj = pos[(size_t)ci[p]];
            const int32_t t = node_of[(size_t)std::min(pi, pj)];
            const int32_t li = local(t, pi), lj = local(t, pj);
            if (li < 0 || lj < 0) return fail(LSA_ERR_ARG, "nd_analyse: internal error (entry outside its front)");
            S.asm_src[(size_t)p] = p;
            S.asm_dst[(size_t)p] = S.front_off[(size_t)t] + (int64_t)li * S.f[(size_t)t] + lj;
        }
    }
    // nodes by level, larger pivot blocks first
    S.lvl_ptr.assign((size_t)S.nlevels + 1, 0);
    for (int32_t t = 0; t < nt; ++t) ++S.lvl_ptr[(size_t)S.level[(size_t)t] + 1];
    for (int32_t l = 0; l < S.nlevels; ++l) S.lvl_ptr[(size_t)l + 1] += S.lvl_ptr[(size_t)l];
    S.lvl_nodes.resize((size_t)nt);
    {
        std::vector<int32_t> fillp(S.lvl_ptr.begin(), S.lvl_ptr.end() - 1);
        for (int32_t t = 0; t < nt; ++t) S.lvl_nodes[(size_t)fillp[(size_t)S.level[(size_t)t]]++] = t;
        for (int32_t l = 0; l < S.nlevels; ++l)
            std::stable_sort(S.lvl_nodes.begin() + S.lvl_ptr[(size_t)l], S.lvl_nodes.begin() + S.lvl_ptr[(size_t)l + 1],
                             [&](int32_t x, int32_t y) { return S.m[(size_t)x] > S.m[(size_t)y]; });
    }
    return LSA_OK;
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
