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


def eigenvalue_condition_numbers(es, lam: np.ndarray, V: np.ndarray, atol: float = 1e-9) -> np.ndarray:
    """kappa_i = ||a_i|| ||M v_i|| / |a_i^H M v_i| for computed eigenpairs (lam_i, v_i) of (A, M), on the GPU path.

    The left eigenvector a_i comes the way the reference gets it (``Sensitivity/__init__.py:247-274``): the eigenpair of
    (A^H, M^H) nearest conj(lam_i) by shift-invert AT conj(lam_i) -- here ``adjoint=True``, the transposed sweeps on the
    factors of A - lam_i M, one small solve per eigenvalue on one prepared solver (only the target moves).  First-order
    perturbation theory: a residual r_i = A v_i - lam_i M v_i moves the eigenvalue by a_i^H r_i / (a_i^H M v_i), so two
    solvers with relative residuals res (``Solver/eigen2.py:48-56``: ||r|| / (||A v|| + |lam| ||M v||) ~ ||r|| / (2 |lam| ||M v||))
    agree to  |d lam| / |lam| <= 2 kappa_i (res_1 + res_2)  plus second-order terms."""
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=1, atol=atol, ncv=24, max_it=200), check_hermitian=False, adjoint=True)
    s.solver.set_st_type(iSTType.SINVERT)
    s.solver.set_st_pc_type(PreconditionerType.LU)
    MV = es.M @ V
    kappa = np.full(len(lam), np.inf)
    for i, z in enumerate(lam):
        s.solver.set_target(np.conj(z))
        s.solver.solve()
        if s.solver.get_num_converged() < 1:
            continue
        a = s.solver.get_eigenvector_array(0)
        kappa[i] = np.linalg.norm(a) * np.linalg.norm(MV[:, i]) / abs(np.vdot(a, MV[:, i]))
    s.solver.release()
    return kappa
