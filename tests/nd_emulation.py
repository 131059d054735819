"""numpy walk of the nested-dissection analysis tables, in the order and with the data flow of the device kernels of
``csrc/ndlu.hip`` (test infrastructure: checks ``lsa_nd_analyse`` without a GPU; never imported by the product).

Every step reads only what the matching kernel reads: ``asm_dst`` (assembly), ``cmap`` (extend-add), ``gptr``/``gidx``
(upward sweep), ``idx`` (both sweeps); the pivot blocks are inverted with LAPACK instead of the panel Gauss-Jordan."""

from __future__ import annotations

import numpy as np


class Emulated:
    def __init__(self, tables: dict, values: np.ndarray):
        t = tables
        self.t = t
        self.nt = len(t["parent"])
        self.m = np.diff(t["node_start"]).astype(np.int64)
        self.f = t["front_size"].astype(np.int64)
        self.b = self.f - self.m
        self.front_off = np.concatenate([[0], np.cumsum(self.f * self.f)])
        self.idx_off = np.concatenate([[0], np.cumsum(self.f)])
        self.u_off = np.concatenate([[0], np.cumsum(self.b)])
        self.g_off = np.concatenate([[0], np.cumsum(self.f + 1)])
        front = np.zeros(int(self.front_off[-1]), dtype=values.dtype)
        assert len(np.unique(t["asm_dst"])) == len(t["asm_dst"]), "two matrix entries share a front slot"
        front[t["asm_dst"]] = values
        self.front = front
        self.min_rel_pivot = np.inf
        for lv in range(len(t["lvl_ptr"]) - 1):
            for node in t["lvl_nodes"][t["lvl_ptr"][lv]:t["lvl_ptr"][lv + 1]]:
                self._factor_node(int(node))

    def F(self, node):
        f = int(self.f[node])
        return self.front[self.front_off[node]:self.front_off[node + 1]].reshape(f, f)

    def _factor_node(self, node):
        t = self.t
        m, F = int(self.m[node]), self.F(node)
        for c in np.flatnonzero(t["parent"] == node):  # ascending = the order of the child ranks
            bc, mc = int(self.b[c]), int(self.m[c])
            pos = t["cmap"][self.u_off[c]:self.u_off[c] + bc]
            F[np.ix_(pos, pos)] += self.F(c)[mc:, mc:]
        import scipy.linalg as sla

        lu, piv = sla.lu_factor(F[:m, :m])
        self.min_rel_pivot = min(self.min_rel_pivot, np.abs(np.diag(lu)).min() / max(np.abs(F[:m, :m]).max(), 1e-300))
        inv = sla.lu_solve((lu, piv), np.eye(m, dtype=F.dtype))
        s1 = -F[m:, :m] @ inv
        F[m:, m:] += s1 @ F[:m, m:]
        F[:m, m:] = inv @ F[:m, m:]
        F[m:, :m] = s1
        F[:m, :m] = inv

    def solve(self, rhs: np.ndarray) -> np.ndarray:
        t = self.t
        x = np.zeros_like(rhs, dtype=np.result_type(rhs.dtype, self.front.dtype))
        ubuf = np.zeros(int(self.u_off[-1]), dtype=x.dtype)
        nl = len(t["lvl_ptr"]) - 1
        for lv in range(nl):
            for node in t["lvl_nodes"][t["lvl_ptr"][lv]:t["lvl_ptr"][lv + 1]]:
                m, f = int(self.m[node]), int(self.f[node])
                ix = t["idx"][self.idx_off[node]:self.idx_off[node + 1]]
                gp = t["gptr"][self.g_off[node]:self.g_off[node + 1]]
                gath = np.array([ubuf[t["gidx"][gp[j]:gp[j + 1]]].sum() for j in range(f)])
                v = rhs[ix[:m]] + gath[:m]
                out = self.F(node)[:, :m] @ v
                x[ix[:m]] = out[:m]
                ubuf[self.u_off[node]:self.u_off[node + 1]] = gath[m:] + out[m:]
        for lv in range(nl - 1, -1, -1):
            for node in t["lvl_nodes"][t["lvl_ptr"][lv]:t["lvl_ptr"][lv + 1]]:
                m = int(self.m[node])
                if self.b[node] == 0:
                    continue
                ix = t["idx"][self.idx_off[node]:self.idx_off[node + 1]]
                x[ix[:m]] -= self.F(node)[:m, m:] @ x[ix[m:]]
        return x


class EmulatedRanks:
    """The subtree-parallel factorisation walked rank by rank in numpy: per-rank tables of ``lsa_nd_analyse_tree``, own
    subtrees first, the in-place all-gathers of the exchange regions (subtree-root fronts, then update vectors), then the
    replicated top -- the data flow of ``nd_numeric`` / ``nd_apply`` in ``csrc/ndlu.hip`` with more than one rank."""

    def __init__(self, tables_per_rank: list[dict], values: np.ndarray):
        self.P = len(tables_per_rank)
        self.r = []
        for t in tables_per_rank:
            d = dict(t)
            d["m"] = np.diff(t["node_start"]).astype(np.int64)
            d["f"] = t["front_size"].astype(np.int64)
            d["b"] = d["f"] - d["m"]
            d["idx_off"] = np.concatenate([[0], np.cumsum(d["f"])])
            d["g_off"] = np.concatenate([[0], np.cumsum(d["f"] + 1)])
            d["cmap_off"] = np.concatenate([[0], np.cumsum(d["b"])])
            front = np.zeros(int(t["front_off"][-1]), dtype=values.dtype)
            front[t["asm_dst"]] = values[t["asm_src"]]
            d["front"] = front
            self.r.append(d)
        slot = self.r[0]["front_slot"]
        assert all(d["front_slot"] == slot and d["u_slot"] == self.r[0]["u_slot"] for d in self.r)
        for d in self.r:  # own subtrees
            self._levels(d, 0, d["phase_b_level"])
        for src in range(self.P):  # all-gather of the subtree roots' fronts
            for dst in range(self.P):
                if dst != src:
                    self.r[dst]["front"][src * slot:(src + 1) * slot] = self.r[src]["front"][src * slot:(src + 1) * slot]
        for d in self.r:  # replicated top
            self._levels(d, d["phase_b_level"], len(d["lvl_ptr"]) - 1)

    @staticmethod
    def _F(d, q):
        f = int(d["f"][q])
        return d["front"][d["front_off"][q]:d["front_off"][q] + f * f].reshape(f, f)

    def _levels(self, d, l0, l1):
        import scipy.linalg as sla

        for lv in range(l0, l1):
            for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                q = int(q)
                m, F = int(d["m"][q]), self._F(d, q)
                for c in d["child_idx"][d["child_ptr"][q]:d["child_ptr"][q + 1]]:
                    bc, mc = int(d["b"][c]), int(d["m"][c])
                    pos = d["cmap"][d["cmap_off"][c]:d["cmap_off"][c] + bc]
                    F[np.ix_(pos, pos)] += self._F(d, int(c))[mc:, mc:]
                inv = sla.inv(F[:m, :m])
                s1 = -F[m:, :m] @ inv
                F[m:, m:] += s1 @ F[:m, m:]
                F[:m, m:] = inv @ F[:m, m:]
                F[m:, :m] = s1
                F[:m, :m] = inv

    def solve(self, rhs: np.ndarray) -> np.ndarray:
        """Returns x assembled from what every rank is responsible for (own subtrees; the top from the last rank)."""
        P = self.P
        xs = [np.full(rhs.shape, np.nan, dtype=np.result_type(rhs.dtype, self.r[0]["front"].dtype)) for _ in range(P)]
        us = [np.zeros(int(d["u_off"][-1]), dtype=xs[0].dtype) for d in self.r]
        uslot = self.r[0]["u_slot"]

        def fwd(d, x, ub, l0, l1):
            for lv in range(l0, l1):
                for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                    q = int(q)
                    m, f = int(d["m"][q]), int(d["f"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    gp = d["gptr"][d["g_off"][q]:d["g_off"][q + 1]]
                    gath = np.array([ub[d["gidx"][gp[j]:gp[j + 1]]].sum() for j in range(f)])
                    v = rhs[ix[:m]] + gath[:m]
                    out = self._F(d, q)[:, :m] @ v
                    x[ix[:m]] = out[:m]
                    ub[d["u_off"][q]:d["u_off"][q] + f - m] = gath[m:] + out[m:]

        def bwd(d, x, l0, l1):
            for lv in range(l1 - 1, l0 - 1, -1):
                for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                    q = int(q)
                    m = int(d["m"][q])
                    if d["b"][q] == 0:
                        continue
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    x[ix[:m]] -= self._F(d, q)[:m, m:] @ x[ix[m:]]

        for r, d in enumerate(self.r):
            fwd(d, xs[r], us[r], 0, d["phase_b_level"])
        for src in range(P):  # all-gather of the subtree roots' update vectors
            for dst in range(P):
                if dst != src:
                    us[dst][src * uslot:(src + 1) * uslot] = us[src][src * uslot:(src + 1) * uslot]
        for r, d in enumerate(self.r):
            nl = len(d["lvl_ptr"]) - 1
            fwd(d, xs[r], us[r], d["phase_b_level"], nl)
            bwd(d, xs[r], d["phase_b_level"], nl)
            bwd(d, xs[r], 0, d["phase_b_level"])
        x = np.full(rhs.shape, np.nan, dtype=xs[0].dtype)
        for r in range(P):  # later ranks overwrite: the top is identical everywhere, own subtrees are disjoint
            ok = ~np.isnan(xs[r])
            x[ok] = xs[r][ok]
        return x


class EmulatedDistributedTop:
    """The forest cut over ranks WITH DISTRIBUTED TOP NODES (kind 4), walked in numpy with the data flow of ``nd_numeric`` /
    ``nd_apply_ordered``: a distributed node keeps on every rank the whole pivot block and F12, and the rank's slice of the
    boundary rows (front (m + brow) x f); its children's update matrices reach it row by row from the ranks that hold them;
    the inverse is formed redundantly, ``L = -F21[rows] inv`` and the update rows locally, ``U`` by slices of the own rows.
    Sweeps: a rank produces its slice of a distributed node's update entries into its slot of the level's exchange region
    (positions from ``ux_base`` / ``ux_stride``, the gather tables point there), finishes its slice of the own rows on the
    way down; both are exchanged level by level."""

    def __init__(self, tables_per_rank: list[dict], values: np.ndarray):
        import scipy.linalg as sla

        self.P = len(tables_per_rank)
        self.r = []
        for t in tables_per_rank:
            d = dict(t)
            d["m"] = np.diff(t["node_start"]).astype(np.int64)
            d["f"] = t["front_size"].astype(np.int64)
            d["b"] = d["f"] - d["m"]
            d["idx_off"] = np.concatenate([[0], np.cumsum(d["f"])])
            d["g_off"] = np.concatenate([[0], np.cumsum(d["f"] + 1)])
            d["cmap_off"] = np.concatenate([[0], np.cumsum(d["b"])])
            front = np.zeros(int(t["front_off"][-1]), dtype=values.dtype)
            front[t["asm_dst"]] = values[t["asm_src"]]
            d["front"] = front
            d["upd"] = {}   # node -> this rank's rows of its update matrix (rows x b)
            d["fac"] = {}   # node -> (inv, L rows, U rows) of a distributed node
            self.r.append(d)
        d0 = self.r[0]
        nt = len(d0["parent"])
        assert all(len(d["parent"]) == nt or True for d in self.r)
        # own subtrees (kind 1), rank by rank
        for d in self.r:
            self._levels_plain(d, 0, d["phase_b_level"])
        # the subtree roots' fronts under REPLICATED parents travel through the exchange region
        slot = d0["front_slot"]
        if slot:
            for src in range(self.P):
                for dst in range(self.P):
                    if dst != src:
                        self.r[dst]["front"][src * slot:(src + 1) * slot] = self.r[src]["front"][src * slot:(src + 1) * slot]
        # the top, level by level, all ranks in step.  Kept-node numbering differs from rank to rank: nodes are matched by their
        # position in the level lists (alike on every rank for the top levels).
        nl = len(d0["lvl_ptr"]) - 1
        for off in range(nl - d0["phase_b_level"]):
            lists = []
            for d in self.r:
                lv = d["phase_b_level"] + off
                lists.append([int(q) for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]])
            assert len({len(x) for x in lists}) == 1
            for pos in range(len(lists[0])):
                qs = [lists[r][pos] for r in range(self.P)]
                kinds = {int(self.r[r]["kind"][qs[r]]) for r in range(self.P)}
                assert len(kinds) == 1
                if kinds == {2}:
                    for r in range(self.P):
                        self._factor_plain(self.r[r], qs[r])
                else:
                    assert kinds == {4}
                    self._factor_dist(qs, sla)

    @staticmethod
    def _F(d, q, rows=None):
        f = int(d["f"][q])
        rows = f if rows is None else rows
        return d["front"][d["front_off"][q]:d["front_off"][q] + rows * f].reshape(rows, f)

    def _child_update(self, d, c):
        """child c's update matrix as this rank holds it (a plain node: the trailing block of its factored front)"""
        if int(c) in d["upd"]:
            return d["upd"][int(c)]
        mc = int(d["m"][c])
        return self._F(d, int(c))[mc:, mc:]

    def _factor_plain(self, d, q):
        import scipy.linalg as sla

        m, F = int(d["m"][q]), self._F(d, q)
        for c in d["child_idx"][d["child_ptr"][q]:d["child_ptr"][q + 1]]:
            bc = int(d["b"][c])
            pos = d["cmap"][d["cmap_off"][c]:d["cmap_off"][c] + bc]
            F[np.ix_(pos, pos)] += self._child_update(d, c)
        inv = sla.inv(F[:m, :m])
        s1 = -F[m:, :m] @ inv
        F[m:, m:] += s1 @ F[:m, m:]
        F[:m, m:] = inv @ F[:m, m:]
        F[m:, :m] = s1
        F[:m, :m] = inv

    def _levels_plain(self, d, l0, l1):
        for lv in range(l0, l1):
            for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                self._factor_plain(d, int(q))

    def _factor_dist(self, qs, sla):
        P = self.P
        d0, q0 = self.r[0], qs[0]
        m, f, b = int(d0["m"][q0]), int(d0["f"][q0]), int(d0["b"][q0])
        nchild = int(d0["child_ptr"][q0 + 1] - d0["child_ptr"][q0])
        fronts = []
        for r in range(P):
            d, q = self.r[r], qs[r]
            assert int(d["m"][q]) == m and int(d["f"][q]) == f
            fronts.append(self._F(d, q, m + int(d["brow"][q])))
        for k in range(nchild):
            cs = [int(self.r[r]["child_idx"][self.r[r]["child_ptr"][qs[r]] + k]) for r in range(P)]
            ck = int(self.r[0]["kind"][cs[0]])
            bc = int(self.r[0]["b"][cs[0]])
            if ck == 2:  # replicated child: every rank holds its update matrix whole
                full = [self._child_update(self.r[r], cs[r]) for r in range(P)]
            else:  # rows travel from the ranks that hold them
                whole = np.zeros((bc, bc), dtype=fronts[0].dtype)
                seen = np.zeros(bc, dtype=bool)
                for r in range(P):
                    d, c = self.r[r], cs[r]
                    kc = int(d["kind"][c])
                    if kc == 4:
                        lo, cnt = int(d["brow0"][c]), int(d["brow"][c])
                        whole[lo:lo + cnt] = d["upd"][c]
                        seen[lo:lo + cnt] = True
                    elif kc == 1:
                        assert int(d["owner"][c]) == r
                        whole[:] = self._child_update(d, c)
                        seen[:] = True
                    else:
                        assert kc == 3 and int(d["owner"][c]) != r
                assert seen.all()
                full = [whole] * P
            for r in range(P):
                d, q, c = self.r[r], qs[r], cs[r]
                pos = d["cmap"][d["cmap_off"][c]:d["cmap_off"][c] + bc].astype(np.int64)
                lo, cnt = int(d["brow0"][q]), int(d["brow"][q])
                lrow = np.where(pos < m, pos, pos - lo)
                keep = (pos < m) | ((pos - m >= lo) & (pos - m < lo + cnt))
                fronts[r][np.ix_(lrow[keep], pos)] += full[r][keep]
        invs = []
        for r in range(P):
            d, q, F = self.r[r], qs[r], fronts[r]
            inv = sla.inv(F[:m, :m])
            invs.append(inv)
            Lr = -F[m:, :m] @ inv
            upd = F[m:, m:] + Lr @ F[:m, m:]
            o0, on = int(d["orow0"][q]), int(d["orows"][q])
            Ur = inv[o0:o0 + on] @ F[:m, m:]
            d["fac"][q] = (inv, Lr, Ur)
            d["upd"][q] = upd
        for r in range(1, P):  # the replicated pivot blocks were assembled alike
            assert np.allclose(invs[r], invs[0], rtol=1e-12, atol=1e-12 * np.abs(invs[0]).max())

    def solve(self, rhs: np.ndarray) -> np.ndarray:
        P = self.P
        dt = np.result_type(rhs.dtype, self.r[0]["front"].dtype)
        xs = [np.full(rhs.shape, np.nan, dtype=dt) for _ in range(P)]
        us = [np.zeros(int(d["u_entries"]), dtype=dt) for d in self.r]
        uslot = self.r[0]["u_slot"]

        def gather(d, ub, q):
            f = int(d["f"][q])
            gp = d["gptr"][d["g_off"][q]:d["g_off"][q + 1]]
            return np.array([ub[d["gidx"][gp[j]:gp[j + 1]]].sum() for j in range(f)])

        def fwd_plain(d, x, ub, q):
            m, f = int(d["m"][q]), int(d["f"][q])
            ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
            gath = gather(d, ub, q)
            out = self._F(d, q)[:, :m] @ (rhs[ix[:m]] + gath[:m])
            x[ix[:m]] = out[:m]
            ub[d["u_off"][q]:d["u_off"][q] + f - m] = gath[m:] + out[m:]

        def bwd_plain(d, x, q):
            m = int(d["m"][q])
            if d["b"][q] == 0:
                return
            ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
            x[ix[:m]] -= self._F(d, q)[:m, m:] @ x[ix[m:]]

        for r, d in enumerate(self.r):
            for lv in range(d["phase_b_level"]):
                for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                    fwd_plain(d, xs[r], us[r], int(q))
        for src in range(P):  # the subtree roots' update vectors
            for dst in range(P):
                if dst != src:
                    us[dst][src * uslot:(src + 1) * uslot] = us[src][src * uslot:(src + 1) * uslot]
        d0 = self.r[0]
        nl = len(d0["lvl_ptr"]) - 1
        top_levels = []
        for off in range(nl - d0["phase_b_level"]):
            lists = [[int(q) for q in d["lvl_nodes"][d["lvl_ptr"][d["phase_b_level"] + off]:d["lvl_ptr"][d["phase_b_level"] + off + 1]]] for d in self.r]
            top_levels.append(lists)
        for lists in top_levels:
            for pos in range(len(lists[0])):
                for r, d in enumerate(self.r):
                    q = lists[r][pos]
                    if int(d["kind"][q]) == 2:
                        fwd_plain(d, xs[r], us[r], q)
                        continue
                    m = int(d["m"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    inv, Lr, _ = d["fac"][q]
                    gath = gather(d, us[r], q)
                    v = rhs[ix[:m]] + gath[:m]
                    o0, on = int(d["orow0"][q]), int(d["orows"][q])
                    xs[r][ix[o0:o0 + on]] = inv[o0:o0 + on] @ v  # (a rank keeps and applies only its own rows of the inverse)
                    lo, cnt = int(d["brow0"][q]), int(d["brow"][q])
                    at = int(d["ux_base"][q]) + r * int(d["ux_stride"][q])
                    us[r][at:at + cnt] = gath[m + lo:m + lo + cnt] + Lr @ v
            for pos in range(len(lists[0])):  # the level's exchange: slot s comes from rank s
                q0 = lists[0][pos]
                if int(d0["kind"][q0]) != 4:
                    continue
                for src in range(P):
                    ds, qs_ = self.r[src], lists[src][pos]
                    at = int(ds["ux_base"][qs_]) + src * int(ds["ux_stride"][qs_])
                    cnt = int(ds["brow"][qs_])
                    for dst in range(P):
                        dd, qd = self.r[dst], lists[dst][pos]
                        assert int(dd["ux_stride"][qd]) == int(ds["ux_stride"][qs_])  # (the bases are each rank's own)
                        to = int(dd["ux_base"][qd]) + src * int(dd["ux_stride"][qd])
                        us[dst][to:to + cnt] = us[src][at:at + cnt]
        for lists in reversed(top_levels):
            for pos in range(len(lists[0])):
                pieces = []
                for r, d in enumerate(self.r):
                    q = lists[r][pos]
                    if int(d["kind"][q]) == 2:
                        bwd_plain(d, xs[r], q)
                        continue
                    m = int(d["m"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    o0, on = int(d["orow0"][q]), int(d["orows"][q])
                    piece = xs[r][ix[o0:o0 + on]] - (d["fac"][q][2] @ xs[r][ix[m:]] if d["b"][q] > 0 else 0.0)
                    pieces.append((ix[o0:o0 + on], piece))
                for own, piece in pieces:  # exchange of the own-row slices
                    for r in range(P):
                        xs[r][own] = piece
        for r, d in enumerate(self.r):
            for lv in range(d["phase_b_level"] - 1, -1, -1):
                for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                    bwd_plain(d, xs[r], int(q))
        x = np.full(rhs.shape, np.nan, dtype=dt)
        for r in range(P):
            ok = ~np.isnan(xs[r])
            x[ok] = xs[r][ok]
        return x


    def solve_transposed(self, rhs: np.ndarray, conj: bool = False) -> np.ndarray:
        """``C^T x = rhs`` (``C^H`` with ``conj``) on the same factors, with the data flow of ``nd_apply_T``: the fronts of the
        transposed forest are the transposed fronts, so an output of a node sums over ROWS of its packed blocks -- for a
        distributed node every rank sums over the rows it holds, the partial results are added in rank order
        (``nd_sweepT_kernel`` with row ranges, ``nd_distT_finish_kernel``); a distributed node's update vector is written, whole,
        into the slots of the level's exchange region on every rank."""
        P = self.P
        cj = (lambda a: np.conj(a)) if conj else (lambda a: a)
        dt = np.result_type(rhs.dtype, self.r[0]["front"].dtype)
        xs = [np.full(rhs.shape, np.nan, dtype=dt) for _ in range(P)]
        us = [np.zeros(int(d["u_entries"]), dtype=dt) for d in self.r]
        uslot = self.r[0]["u_slot"]

        def gather(d, ub, q):
            f = int(d["f"][q])
            gp = d["gptr"][d["g_off"][q]:d["g_off"][q + 1]]
            return np.array([ub[d["gidx"][gp[j]:gp[j + 1]]].sum() for j in range(f)])

        def up_plain(d, x, ub, q):
            m, f = int(d["m"][q]), int(d["f"][q])
            ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
            gath = gather(d, ub, q)
            z = cj(self._F(d, q)[:m, :]).T @ (rhs[ix[:m]] + gath[:m])  # [inv | U]^T v
            x[ix[:m]] = z[:m]
            ub[d["u_off"][q]:d["u_off"][q] + f - m] = gath[m:] - z[m:]

        def down_plain(d, x, q):
            m = int(d["m"][q])
            if d["b"][q] == 0:
                return
            ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
            x[ix[:m]] += cj(self._F(d, q)[m:, :m]).T @ x[ix[m:]]  # L^T x_boundary

        for r, d in enumerate(self.r):
            for lv in range(d["phase_b_level"]):
                for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                    up_plain(d, xs[r], us[r], int(q))
        for src in range(P):
            for dst in range(P):
                if dst != src:
                    us[dst][src * uslot:(src + 1) * uslot] = us[src][src * uslot:(src + 1) * uslot]
        d0 = self.r[0]
        nl = len(d0["lvl_ptr"]) - 1
        top_levels = [[[int(q) for q in d["lvl_nodes"][d["lvl_ptr"][d["phase_b_level"] + off]:d["lvl_ptr"][d["phase_b_level"] + off + 1]]] for d in self.r]
                      for off in range(nl - d0["phase_b_level"])]
        for lists in top_levels:
            for pos in range(len(lists[0])):
                if int(d0["kind"][lists[0][pos]]) == 2:
                    for r, d in enumerate(self.r):
                        up_plain(d, xs[r], us[r], lists[r][pos])
                    continue
                partial = []
                for r, d in enumerate(self.r):  # every rank: the rows it holds of [inv | U]
                    q = lists[r][pos]
                    m = int(d["m"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    inv, _, Ur = d["fac"][q]
                    o0, on = int(d["orow0"][q]), int(d["orows"][q])
                    v = (rhs[ix[:m]] + gather(d, us[r], q)[:m])[o0:o0 + on]
                    partial.append(np.concatenate([cj(inv[o0:o0 + on]).T @ v, cj(Ur).T @ v if d["b"][q] > 0 else np.zeros(0, dtype=dt)]))
                z = partial[0].copy()
                for pz in partial[1:]:  # rank order
                    z = z + pz
                for r, d in enumerate(self.r):
                    q = lists[r][pos]
                    m, b = int(d["m"][q]), int(d["b"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    xs[r][ix[:m]] = z[:m]
                    if b == 0:
                        continue
                    u = gather(d, us[r], q)[m:] - z[m:]
                    w = -(-b // P)
                    for j in range(b):  # entry j in the slot of the rank that owns boundary row j in the forward sweeps
                        us[r][int(d["ux_base"][q]) + (j // w) * int(d["ux_stride"][q]) + j % w] = u[j]
        for lists in reversed(top_levels):
            for pos in range(len(lists[0])):
                if int(d0["kind"][lists[0][pos]]) == 2:
                    for r, d in enumerate(self.r):
                        down_plain(d, xs[r], lists[r][pos])
                    continue
                partial = []
                for r, d in enumerate(self.r):  # every rank: the boundary rows it holds of L
                    q = lists[r][pos]
                    m = int(d["m"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    lo, cnt = int(d["brow0"][q]), int(d["brow"][q])
                    partial.append(cj(d["fac"][q][1]).T @ xs[r][ix[m + lo:m + lo + cnt]] if cnt > 0 else np.zeros(m, dtype=dt))
                z = partial[0].copy()
                for pz in partial[1:]:
                    z = z + pz
                for r, d in enumerate(self.r):
                    q = lists[r][pos]
                    m = int(d["m"][q])
                    ix = d["idx"][d["idx_off"][q]:d["idx_off"][q + 1]]
                    xs[r][ix[:m]] = xs[r][ix[:m]] + z
        for r, d in enumerate(self.r):
            for lv in range(d["phase_b_level"] - 1, -1, -1):
                for q in d["lvl_nodes"][d["lvl_ptr"][lv]:d["lvl_ptr"][lv + 1]]:
                    down_plain(d, xs[r], int(q))
        x = np.full(rhs.shape, np.nan, dtype=dt)
        for r in range(P):
            ok = ~np.isnan(xs[r])
            x[ok] = xs[r][ok]
        return x
