"""SLEPc-free shift-invert eigensolver front end on the HIP path.

Drop-in for ``/root/reference/Solver/eigen2.py``: same ``ShiftInvertConfig`` fields, the
same constructor ``ArpackEigenSolver(cfg, A, M, *, dofs_u, dofs_p)`` and the same ``solve() -> (lam, V, res)``.

What the reference does there (``:109-263``): ``C = A - sigma M``, one factorisation, the operator
``x -> P C^-1 M P x`` with ``P`` zeroing the pressure dofs on both sides (``:164-201``), ARPACK ``eigs(which="LM")`` on it,
``lambda = sigma + 1/mu`` (``:209-211,239``), a sort by ``which_sort`` (``:240-242``) and the residual report of
``_compute_residuals`` (``:48-56,244-263``).  Here the operator is the device one (``lsa_op_*`` with
``lsa_op_set_projection``), the outer iteration is the device-resident Krylov-Schur of :mod:`lsa_hip.krylov_schur`
instead of ARPACK's implicitly restarted Arnoldi (both return the ``k`` eigenvalues of the same operator nearest
``sigma``), and the residuals are evaluated by ``lsa_eig_residuals``.
"""

from __future__ import annotations

import logging
import time
from dataclasses import dataclass

import numpy as np

from FEM.utils import iPETScMatrix

from .utils import PreconditionerType, iEpsProblemType, iEpsSolver, iEpsWhich, iSTType

logger = logging.getLogger(__name__)


def _sort_indices(lam: np.ndarray, which: str) -> np.ndarray:
    """Descending order of the key named by ``which`` (``Solver/eigen2.py:32-45``)."""
    keys = {"LR": lambda z: z.real, "LI": lambda z: z.imag, "SR": lambda z: -z.real, "SI": lambda z: -z.imag, "LM_abs": np.abs}
    if which not in keys:
        raise ValueError(f"Unknown which_sort = {which!r}")
    return np.argsort(-keys[which](np.asarray(lam)))


def _compute_residuals(A, M, lam: np.ndarray, V: np.ndarray) -> np.ndarray:
    """``||A v - lam M v|| / (||A v|| + |lam| ||M v|| + 1e-16)`` column-wise (``Solver/eigen2.py:48-56``), on the host:
    the formula for callers that hold scipy matrices; :class:`ArpackEigenSolver` evaluates it on the device."""
    Av = A @ V
    Mv = M @ V
    num = np.linalg.norm(Av - Mv * lam[np.newaxis, :], axis=0)
    return num / (np.linalg.norm(Av, axis=0) + np.abs(lam) * np.linalg.norm(Mv, axis=0) + 1e-16)


@dataclass
class ShiftInvertConfig:
    """Shift-invert eigensolver configuration (``Solver/eigen2.py:59-68``)."""

    sigma: complex = 0.0
    k: int = 20
    tol: float = 1e-6
    maxiter: int = 500
    ncv: int | None = None
    which_sort: iEpsWhich = iEpsWhich.LARGEST_REAL


class ArpackEigenSolver:
    """Solver for ``A x = lambda M x`` around ``cfg.sigma`` in the velocity subspace (``Solver/eigen2.py:71-265``)."""

    def __init__(self, cfg: ShiftInvertConfig, A, M, *, dofs_u: np.ndarray, dofs_p: np.ndarray, **solver_kwargs) -> None:
        A = A if isinstance(A, iPETScMatrix) else iPETScMatrix.from_matrix(A)
        M = M if isinstance(M, iPETScMatrix) else iPETScMatrix.from_matrix(M)
        nrows, ncols = A.shape
        mrows, mcols = M.shape
        if nrows != ncols or mrows != mcols or (mrows, mcols) != (nrows, ncols):
            raise ValueError(f"Operators must be square and have the same shape. Got A shape {A.shape}; and M shape {M.shape}")
        self._cfg = cfg
        self._n = nrows
        self._dofs_u = np.asarray(dofs_u, dtype=np.int32)
        self._dofs_p = np.asarray(dofs_p, dtype=np.int32)
        if self._dofs_p.size and (self._dofs_p.min() < 0 or self._dofs_p.max() >= nrows):
            raise ValueError("dofs_p holds indices outside the operator")
        solver_kwargs.setdefault("ksp_rtol", float(np.clip(cfg.tol * 1e-2, 1e-13, 1e-8)))
        self._eps = iEpsSolver(A, M, project_out=self._dofs_p, **solver_kwargs)
        self._eps.set_problem_type(iEpsProblemType.GNHEP)
        self._eps.set_st_type(iSTType.SINVERT)
        self._eps.set_target(complex(cfg.sigma))
        self._eps.set_st_pc_type(PreconditionerType.LU)  # the reference factorises C once (``:121-151``)
        self._eps.set_which_eigenpairs(iEpsWhich.TARGET_MAGNITUDE)  # eigs(which="LM") on mu = 1 / (lambda - sigma)
        # "a roomy Krylov subspace if not provided" (``:223-224``)
        ncv = cfg.ncv if cfg.ncv is not None else max(4 * cfg.k, 40)
        self._eps.set_dimensions(cfg.k, min(ncv, max(nrows - 1, 1)))
        self._eps.set_tolerances(cfg.tol, cfg.maxiter)
        logger.info("Initialized eigensolver (HIP Krylov-Schur, velocity-subspace shift-invert).")

    @property
    def solver(self) -> iEpsSolver:
        return self._eps

    @staticmethod
    def _mu_to_lambda(mu: np.ndarray, sigma: complex) -> np.ndarray:
        return sigma + 1.0 / mu

    def solve(self) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Solve EVP: ``(lam, V, res)`` with ``V`` of shape ``(n, k)``, pressure entries zero, unit 2-norm columns."""
        cfg = self._cfg
        logger.info("Started eigenvalue solve around %s: nev=%d, tol=%g, max_it=%d", cfg.sigma, cfg.k, cfg.tol, cfg.maxiter)
        t0 = time.time()
        self._eps.solve()
        nconv = self._eps.get_num_converged()
        if nconv < cfg.k:
            raise RuntimeError(f"Krylov-Schur converged {nconv} of {cfg.k} requested eigenpairs in {cfg.maxiter} restarts")
        logger.info("Solve completed in %.2f s.", time.time() - t0)
        # the k pairs nearest sigma (ARPACK returns exactly k), then the caller's sort
        lam = np.array([self._eps.get_eigenvalue(i) for i in range(cfg.k)], dtype=np.complex128)
        V = np.column_stack([self._eps.get_eigenvector_array(i) for i in range(cfg.k)])
        res_all = self._eps.residuals()
        idx = _sort_indices(lam, cfg.which_sort.to_arpack())
        lam, V, res = lam[idx], V[:, idx], np.asarray(res_all[: cfg.k])[idx]
        med_res, max_res = float(np.median(res)), float(np.max(res))
        logger.debug("Computed residuals: median=%.2e, max=%.2e, >1e-6=%d, >1e-4=%d.", med_res, max_res,
                     int(np.sum(res > 1e-6)), int(np.sum(res > 1e-4)))
        if max_res > 1e-4 or med_res > 1e-6:
            logger.warning("Poor eigenpair quality (adjust sigma and k/ncv).")
        return lam, V, res
