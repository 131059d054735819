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
