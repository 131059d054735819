"""Direct + adjoint eigenpair (structural-sensitivity algebra) on the HIP eigen path.

Restates the two ``EigenSolver`` calls and the linear algebra around them in
``/root/reference/Sensitivity/__init__.py``:

* ``solve_direct_mode``   (``:158-228``): shift-invert at the target, LU-class inner solves, pair nearest the target
  (``:200-201``);
* ``solve_adjoint_mode``  (``:230-311``): eigenproblem of ``(A^H, M^H)`` at ``conj(sigma)``, ``TARGET_REAL`` ordering
  (``:256-262``), pair nearest ``conj(sigma)`` (``:277-278``), scaling so that ``a^H M v = 1`` with the conjugating dot
  (``:281-287``).  The reference forms both transposes explicitly (``:47-57,247-248``); here the operator
  ``(A - sigma M)^-H M^H`` is applied on the device with transposed sweeps of the factors of ``A - sigma M`` and
  transposed SpMVs (``EigenSolver(..., adjoint=True)``): no transposed matrix exists anywhere;
* ``compute_wavemaker``   (``:404-445``) reduced to its nodal algebra on dof arrays:
  ``Sw = |a_u| |v_u| / |<a_u, M v_u>|`` per velocity node (the UFL projection to the pressure space needs dolfinx and
  is out of scope).

The dolfinx ``Function`` wrappers of the reference are replaced by numpy arrays; everything numeric runs through
``Solver.eigen.EigenSolver`` (GPU).
"""

from __future__ import annotations

import logging

import numpy as np

from FEM.utils import iPETScMatrix
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iEpsProblemType, iEpsWhich, iSTType

logger = logging.getLogger(__name__)


def _as_array(vec) -> np.ndarray:
    return vec.real.as_array() + (1j * vec.imag.as_array() if vec.imag is not None else 0)


class EigenSensitivitySolver:
    """Direct / adjoint eigenpair around a target for a pre-assembled pair ``(A, M)``."""

    def __init__(self, A, M, *, target: complex | None = None, tol_direct: float = 1e-6, tol_adjoint: float = 1e-3,
                 max_it: int = 500, max_modes: int = 5, adjoint_shift_nudge: float = 0.0, **solver_kwargs) -> None:
        self._A = A if isinstance(A, iPETScMatrix) else iPETScMatrix.from_matrix(A)
        self._M = M if isinstance(M, iPETScMatrix) else iPETScMatrix.from_matrix(M)
        self._target = target
        self._tol_direct, self._tol_adjoint = tol_direct, tol_adjoint
        self._max_it, self._max_modes = max_it, max_modes
        self._kw = solver_kwargs
        # The reference shifts the adjoint problem exactly at conj(sigma) of the converged direct mode, which makes
        # A^H - conj(sigma) M^H singular to working precision.  The exact LU handles that as the reference's does: the
        # solves are accepted on their backward error (``stats["backward_accepted"]``).  ``adjoint_shift_nudge`` > 0 moves
        # the shift by that relative amount (needed only with an iterative inner solve, which cannot reach a residual
        # tolerance on a singular system).
        self._nudge = adjoint_shift_nudge
        self._sigma: complex | None = None
        self._v: np.ndarray | None = None
        self._a: np.ndarray | None = None

    def solve_direct_mode(self, target: complex | None = None) -> tuple[complex, np.ndarray]:
        target = self._target if target is None else target
        cfg = EigensolverConfig(num_eig=self._max_modes, problem_type=iEpsProblemType.GNHEP, atol=self._tol_direct, max_it=self._max_it)
        es = EigenSolver(self._A, self._M, cfg, check_hermitian=False, **self._kw)
        if target is not None:
            es.solver.set_st_type(iSTType.SINVERT)
            es.solver.set_target(target)
            es.solver.set_st_pc_type(PreconditionerType.LU)
        else:
            es.solver.set_which_eigenpairs(iEpsWhich.LARGEST_REAL)
        if not (pairs := es.solve()):
            raise RuntimeError("No eigenpairs returned by the eigensolver.")
        if target is not None:
            sigma, vec = min(pairs, key=lambda p: abs(p[0] - target))
        else:
            sigma, vec = max(pairs, key=lambda p: p[0].real)
        self._sigma, self._v = complex(sigma), _as_array(vec)
        logger.info("Direct eigenpair: sigma = %.4e %+.4e j", self._sigma.real, self._sigma.imag)
        return self._sigma, self._v

    def solve_adjoint_mode(self, sigma: complex | None = None, v: np.ndarray | None = None) -> np.ndarray:
        if sigma is None or v is None:
            sigma, v = self._sigma, self._v
        if sigma is None or v is None:
            raise RuntimeError("Direct eigenpair must be computed before adjoint solve.")
        cfg = EigensolverConfig(num_eig=self._max_modes, problem_type=iEpsProblemType.GNHEP, atol=self._tol_adjoint, max_it=self._max_it)
        es_adj = EigenSolver(self._A, self._M, cfg, check_hermitian=False, adjoint=True, **self._kw)
        es_adj.solver.set_st_type(iSTType.SINVERT)
        es_adj.solver.set_st_pc_type(PreconditionerType.LU)
        es_adj.solver.set_target(np.conj(sigma) * (1.0 + self._nudge))
        es_adj.solver.set_which_eigenpairs(iEpsWhich.TARGET_REAL)
        if not (pairs := es_adj.solve()):
            raise RuntimeError("No eigenpairs returned by the adjoint eigensolver.")
        target_star = np.conj(sigma)
        sigma_adj, a_vec = min(pairs, key=lambda p: abs(p[0] - target_star))
        self._sigma_adj = complex(sigma_adj)
        a = _as_array(a_vec)
        prod = np.vdot(a, self._M.as_scipy_array() @ v)  # conjugating dot: a^H M v
        if prod == 0:
            raise RuntimeError("Bi-orthonormal normalization failed (a^H B v = 0).")
        # the reference scales by 1/prod; with the conjugating dot the scale that makes a^H M v = 1 is 1/conj(prod)
        self._a = a / np.conj(prod)
        return self._a

    def compute_wavemaker(self, dofs_ux: np.ndarray, dofs_uy: np.ndarray) -> np.ndarray:
        """Nodal structural sensitivity |a_u| |v_u| / |a_u^H M v_u| on the velocity nodes."""
        if self._v is None or self._a is None:
            raise RuntimeError("Compute direct and adjoint modes before Sw.")
        v, a = self._v, self._a
        mask = np.zeros(v.shape[0], dtype=bool)
        mask[dofs_ux] = mask[dofs_uy] = True
        denom = abs(np.vdot(np.where(mask, a, 0), self._M.as_scipy_array() @ np.where(mask, v, 0)))
        if denom == 0.0:
            raise RuntimeError("Denominator <u+,u> = 0; normalization issue.")
        na = np.sqrt(np.abs(a[dofs_ux]) ** 2 + np.abs(a[dofs_uy]) ** 2)
        nv = np.sqrt(np.abs(v[dofs_ux]) ** 2 + np.abs(v[dofs_uy]) ** 2)
        return na * nv / denom
