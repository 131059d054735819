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
