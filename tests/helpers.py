"""Test infrastructure shared by the CPU and GPU suites (never imported by the product)."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "lsa-fw_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)


class NumpyKrylovBackend:
    """CPU stand-in for ``lsa_hip.KrylovBasis`` used to test the host-side Krylov-Schur logic without a GPU.

    ``op`` is a callable x -> OP x (e.g. SuperLU shift-invert from the oracle).  Orthogonalisation is CGS2 like the
    device path."""

    def __init__(self, op, n: int, ncv: int):
        self.op, self.n, self.ncv = op, n, ncv
        self.V = np.zeros((n, ncv + 1), dtype=np.complex128, order="F")
        self.applies = 0

    def _orth(self, j, w):
        h = np.zeros(j + 1, dtype=np.complex128)
        for _ in range(2):
            c = self.V[:, :j].conj().T @ w
            w = w - self.V[:, :j] @ c
            h[:j] += c
        h[j] = np.linalg.norm(w)
        return w, h

    def inject(self, j, v):
        w, h = self._orth(j, np.asarray(v, dtype=np.complex128))
        self.V[:, j] = w / h[j]

    def extend(self, j0, j1, H):
        for j in range(j0, j1):
            w = self.op(self.V[:, j])
            self.applies += 1
            w, h = self._orth(j + 1, w)
            H[:, j] = 0
            H[: j + 2, j] = h
            beta = h[j + 1].real
            if beta <= 1e-14 * max(np.abs(h[: j + 1]).max(), 1e-300):
                with np.errstate(all="ignore"):
                    self.V[:, j + 1] = w / beta if beta > 0 else 0
                return j
            self.V[:, j + 1] = w / beta
        return -1

    def restart(self, m, Q):
        k = Q.shape[1]
        new = self.V[:, :m] @ Q
        last = self.V[:, m].copy()
        self.V[:, :k] = new
        self.V[:, k] = last

    def ritz_vectors(self, m, Y, normalise=True):
        X = self.V[:, :m] @ Y
        if normalise:
            X = X / np.linalg.norm(X, axis=0)
        return np.asfortranarray(X)


def match_nearest(found: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """For every reference eigenvalue the relative distance to the nearest computed one."""
    found = np.asarray(found)
    return np.array([np.min(np.abs(found - r)) / max(abs(r), 1e-300) for r in ref])
