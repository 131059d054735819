"""CPU prototype of the nested-dissection multifrontal LU with inverted pivot blocks (development tool).

Checks, on the synthetic saddle-point pairs, that pivoting restricted to the pivot block of every front is stable enough
for a direct solve, and reports the sizes that decide the device layout (fill, largest front, levels, bytes per apply).

    python tools/nd_prototype.py S30k [leaf]
"""

from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]


def bfs_levels(indptr, indices, mask_id, cur, start):
    """Level structure of the component of `start` inside the vertex set {v: mask_id[v] == cur}."""
    level = {start: 0}
    frontier = np.array([start])
    levels = [frontier]
    seen = np.zeros(0, dtype=np.int64)
    visited = set([start])
    while True:
        nb = np.concatenate([indices[indptr[v]:indptr[v + 1]] for v in frontier]) if len(frontier) else np.zeros(0, dtype=np.int64)
        nb = np.unique(nb)
        nb = nb[mask_id[nb] == cur]
        nb = np.array([v for v in nb if v not in visited], dtype=np.int64)
        if len(nb) == 0:
            break
        visited.update(nb.tolist())
        levels.append(nb)
        frontier = nb
    return levels


def nd_order(A: sp.csr_matrix, leaf: int = 128):
    """Returns (perm new->old, node_start, parent) in postorder; node t owns permuted rows [node_start[t], node_start[t+1])."""
    n = A.shape[0]
    pat = (A + A.T).tocsr()
    indptr, indices = pat.indptr.astype(np.int64), pat.indices.astype(np.int64)
    from scipy.sparse.csgraph import connected_components

    nodes = []  # (own dofs, children list)

    def recurse(verts):
        """returns list of tree-node ids that are the roots of the forest built for `verts`"""
        if len(verts) == 0:
            return []
        sub = pat[verts][:, verts]
        ncomp, lab = connected_components(sub, directed=False)
        if ncomp > 1:
            roots = []
            for c in range(ncomp):
                roots += recurse(verts[lab == c])
            return roots
        if len(verts) <= leaf:
            nodes.append((verts, []))
            return [len(nodes) - 1]
        # level structure from a pseudo-peripheral vertex (local indices)
        from scipy.sparse.csgraph import breadth_first_order

        def levels_from(s):
            order, pred = breadth_first_order(sub, s, directed=False)
            lev = np.zeros(len(verts), dtype=np.int64)
            for v in order[1:]:
                lev[v] = lev[pred[v]] + 1
            return lev, order

        s = 0
        lev, order = levels_from(s)
        for _ in range(3):
            far = order[-1]
            lev2, order2 = levels_from(far)
            if lev2.max() <= lev.max():
                break
            lev, order, s = lev2, order2, far
        cnt = np.bincount(lev)
        cum = np.cumsum(cnt)
        k = int(np.searchsorted(cum, len(verts) / 2))
        k = min(max(k, 1), len(cnt) - 2) if len(cnt) >= 3 else None
        if k is None:
            nodes.append((verts, []))
            return [len(nodes) - 1]
        # thin separator: vertices of level k with a neighbour in level k+1
        inlev = np.flatnonzero(lev == k)
        sp_ptr, sp_idx = sub.indptr, sub.indices
        is_sep = np.array([np.any(lev[sp_idx[sp_ptr[v]:sp_ptr[v + 1]]] == k + 1) for v in inlev])
        sep = inlev[is_sep]
        sepflag = np.zeros(len(verts), dtype=bool)
        sepflag[sep] = True
        left = np.flatnonzero((lev <= k) & ~sepflag)
        right = np.flatnonzero(lev > k)
        ch = recurse(verts[left]) + recurse(verts[right])
        nodes.append((verts[sep], ch))
        return [len(nodes) - 1]

    roots = recurse(np.arange(n))
    # postorder numbering
    order = []
    parent = {}

    def post(t, p):
        for c in nodes[t][1]:
            post(c, t)
        parent[t] = p
        order.append(t)

    for r in roots:
        post(r, -1)
    newid = {t: i for i, t in enumerate(order)}
    perm = np.concatenate([nodes[t][0] for t in order])
    sizes = np.array([len(nodes[t][0]) for t in order])
    node_start = np.concatenate([[0], np.cumsum(sizes)])
    par = np.array([newid[parent[t]] if parent[t] >= 0 else -1 for t in order])
    return perm, node_start, par


def symbolic(Cp: sp.csr_matrix, node_start, parent):
    """struct[t] = sorted boundary indices (permuted numbering) of node t."""
    nt = len(parent)
    pat = (Cp != 0).astype(np.int8) + (Cp.T != 0).astype(np.int8)
    pat = sp.csr_matrix(((Cp + Cp.T).data * 0 + 1, (Cp + Cp.T).indices, (Cp + Cp.T).indptr), shape=Cp.shape) if False else (abs(Cp) + abs(Cp.T)).tocsr()
    # keep explicit pattern (explicit zeros count): use the structural pattern
    S = sp.csr_matrix((np.ones(Cp.nnz), Cp.indices, Cp.indptr), shape=Cp.shape)
    S = (S + S.T).tocsr()
    children = [[] for _ in range(nt)]
    for t, p in enumerate(parent):
        if p >= 0:
            children[p].append(t)
    struct = [None] * nt
    for t in range(nt):
        a, b = node_start[t], node_start[t + 1]
        cols = S.indices[S.indptr[a]:S.indptr[b]]
        sets = [cols[cols >= b]] + [struct[c][struct[c] >= b] for c in children[t]]
        struct[t] = np.unique(np.concatenate(sets)) if sets else np.zeros(0, dtype=np.int64)
    return struct, children


def factor(Cp: sp.csr_matrix, node_start, parent, struct, children):
    nt = len(parent)
    Cc = Cp.tocsc()
    Cr = Cp.tocsr()
    Tf, G, U = [None] * nt, [None] * nt, [None] * nt
    minpiv = np.inf
    for t in range(nt):
        a, b = node_start[t], node_start[t + 1]
        m = b - a
        bd = struct[t]
        idx = np.concatenate([np.arange(a, b), bd])
        f = len(idx)
        F = np.zeros((f, f), dtype=Cp.dtype)
        # original entries: rows of I (all columns >= a in idx), columns of I for boundary rows
        F[:m, :] = Cr[a:b][:, idx].toarray()
        F[m:, :m] = Cr[bd][:, a:b].toarray()
        for c in children[t]:
            pos = np.searchsorted(idx, struct[c])
            assert np.array_equal(idx[pos], struct[c])
            F[np.ix_(pos, pos)] += U[c]
            U[c] = None
        F11 = F[:m, :m]
        import scipy.linalg as sla

        lu, piv = sla.lu_factor(F11)
        d = np.abs(np.diag(lu))
        minpiv = min(minpiv, d.min() / max(np.abs(F11).max(), 1e-300))
        inv = sla.lu_solve((lu, piv), np.eye(m, dtype=Cp.dtype))
        L21 = F[m:, :m] @ inv
        G[t] = inv @ F[:m, m:]
        U[t] = F[m:, m:] - L21 @ F[:m, m:]
        Tf[t] = np.vstack([inv, -L21])
    return Tf, G, minpiv


def solve(b, node_start, parent, struct, children, Tf, G):
    nt = len(parent)
    y = np.zeros_like(b)
    u = [None] * nt
    for t in range(nt):
        a, e = node_start[t], node_start[t + 1]
        m = e - a
        idx = np.concatenate([np.arange(a, e), struct[t]])
        v = np.zeros(len(idx), dtype=b.dtype)
        v[:m] = b[a:e]
        for c in children[t]:
            pos = np.searchsorted(idx, struct[c])
            v[pos] += u[c]
        out = Tf[t] @ v[:m]
        y[a:e] = out[:m]
        u[t] = v[m:] + out[m:]
    x = y.copy()
    for t in range(nt - 1, -1, -1):
        a, e = node_start[t], node_start[t + 1]
        if len(struct[t]):
            x[a:e] = y[a:e] - G[t] @ x[struct[t]]
    return x


def main():
    from synthetic import fem

    case = sys.argv[1] if len(sys.argv) > 1 else "S5k"
    leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    if case.startswith("C"):
        es = fem.cube_case(case)
    else:
        es = fem.cylinder_case(case)
    sigma = fem.SIGMA_RE50
    C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    n = C.shape[0]
    t0 = time.time()
    perm, node_start, parent = nd_order(C, leaf)
    t1 = time.time()
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    struct, children = symbolic(Cp, node_start, parent)
    nt = len(parent)
    m = np.diff(node_start)
    bsz = np.array([len(s) for s in struct])
    level = np.zeros(nt, dtype=int)
    for t in range(nt):
        for c in children[t]:
            level[t] = max(level[t], level[c] + 1)
    entries = int(np.sum(m * m + 2 * m * bsz))
    print(f"{case}: n={n} nnz={C.nnz} tree nodes={nt} levels={level.max() + 1} order {t1 - t0:.1f}s")
    print(f"  max own={m.max()} max boundary={bsz.max()} max front={(m + bsz).max()}  apply entries={entries} ({entries * 16 / 1e6:.1f} MB c128, {entries / C.nnz:.1f} x nnz)")
    print(f"  sum front^2 = {int(np.sum((m + bsz) ** 2)) * 16 / 1e6:.1f} MB; flops ~ {np.sum(2.0 * m**3 + 2.0 * m * m * bsz * 2 + 2.0 * m * bsz * bsz) * 4 / 1e9:.2f} GFLOP(real)")
    for lv in range(level.max() + 1):
        sel = level == lv
        print(f"   level {lv}: {sel.sum()} nodes, own {m[sel].min()}..{m[sel].max()}, boundary {bsz[sel].min()}..{bsz[sel].max()}")
    if n > 60000:
        return
    t2 = time.time()
    Tf, G, minpiv = factor(Cp, node_start, parent, struct, children)
    t3 = time.time()
    rng = np.random.default_rng(0)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x = solve(b, node_start, parent, struct, children, Tf, G)
    r = np.linalg.norm(b - Cp @ x) / np.linalg.norm(b)
    import scipy.sparse.linalg as spla

    xs = spla.splu(Cp.tocsc()).solve(b)
    print(f"  factor {t3 - t2:.1f}s  min relative pivot {minpiv:.2e}  residual {r:.2e}  vs SuperLU {np.linalg.norm(x - xs) / np.linalg.norm(xs):.2e} (SuperLU residual {np.linalg.norm(b - Cp @ xs) / np.linalg.norm(b):.2e})")


if __name__ == "__main__":
    main()
