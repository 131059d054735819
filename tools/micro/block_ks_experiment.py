"""CPU experiment: how many BLOCK operator applies does a block Krylov-Schur need (block size s) for the bench problem?
A block apply on the GPU costs about as much as a single one (the Schur inverses are read once for all s vectors)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.linalg as sla  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as spla  # noqa: E402
from scipy.linalg import lapack  # noqa: E402

from synthetic import fem  # noqa: E402


def block_krylov_schur(op, n, nev, ncv, s, tol, sigma, maxit=200, seed=0):
    rng = np.random.default_rng(seed)
    m = ncv - (ncv % s)
    V = np.zeros((n, m + s), dtype=complex)
    H = np.zeros((m + s, m), dtype=complex)

    def orth(w, j):
        h = np.zeros(j + 1, dtype=complex)
        for _ in range(2):
            c = V[:, :j].conj().T @ w
            w = w - V[:, :j] @ c
            h[:j] += c
        h[j] = np.linalg.norm(w)
        return w / h[j], h

    for c in range(s):
        v, _ = orth(rng.standard_normal(n) + 1j * rng.standard_normal(n), c)
        V[:, c] = v
    k, applies, blocks, restarts = 0, 0, 0, 0
    key = lambda th: np.abs(1.0 / th)  # nearest sigma = largest |theta|  (small key = wanted)
    while True:
        for jb in range(k, m, s):
            W = np.column_stack([op(V[:, jb + c]) for c in range(s)])
            applies += s
            blocks += 1
            for c in range(s):
                v, h = orth(W[:, c], jb + s + c)
                V[:, jb + s + c] = v
                H[:, jb + c] = 0
                H[: jb + s + c + 1, jb + c] = h
        Hm, B = H[:m, :m], H[m : m + s, :m]
        T, Q = sla.schur(Hm, output="complex")
        w, S = sla.eig(T)
        order = np.argsort(key(w), kind="stable")
        w, S = w[order], S[:, order] / np.linalg.norm(S[:, order], axis=0)
        est = np.linalg.norm((B @ Q) @ S, axis=0)
        rel = est / np.abs(w)
        nconv = 0
        while nconv < m and rel[nconv] <= tol:
            nconv += 1
        if nconv >= nev or restarts >= maxit:
            return sigma + 1.0 / w[:nconv], applies, blocks, restarts
        knew = nconv + (m - nconv) // 2
        knew -= (m - knew) % s  # the rest of the basis is refilled in whole blocks
        knew = max(min(knew, m - s), 1)
        kd = key(np.diag(T))
        thr = np.sort(kd)[knew - 1]
        sel = (kd <= thr).astype(np.int32)
        Ts, Qs, *_ = lapack.ztrsen(sel, T, Q, job="N", wantq=1)
        knew = int(sel.sum())
        while (m - knew) % s:
            knew -= 1
        Bt = B @ Qs
        Vn = V[:, :m] @ Qs[:, :knew]
        last = V[:, m : m + s].copy()
        V[:, :knew] = Vn
        V[:, knew : knew + s] = last
        H[:, :] = 0
        H[:knew, :knew] = Ts[:knew, :knew]
        H[knew : knew + s, :knew] = Bt[:, :knew]
        k = knew
        restarts += 1


for case in sys.argv[1:] or ["S5k"]:
    es = fem.cylinder_case(case)
    sigma = fem.SIGMA_RE50
    C = (es.A - sigma * es.M).astype(complex).tocsc()
    lu = spla.splu(C)
    Mc = es.M.astype(complex).tocsr()
    op = lambda x: lu.solve(Mc @ x)
    ref = None
    for s in (1, 2, 4, 8):
        t0 = time.time()
        lam, applies, blocks, restarts = block_krylov_schur(op, es.n, 20, 80, s, 1e-10, sigma)
        lam = lam[np.argsort(np.abs(lam - sigma))][:20]
        if ref is None:
            ref = lam
        err = max(np.min(np.abs(lam - r)) / abs(r) for r in ref)
        print(f"{case} block size {s}: {len(lam)} pairs, {applies} vector applies = {blocks} block applies, {restarts} restarts, "
              f"max rel diff vs s=1 {err:.1e}, {time.time() - t0:.1f} s", flush=True)
