"""LSA-FW eigensolver on MI355X: drop-in for ``/root/reference/Solver/eigen.py``.

Same public surface (``EigensolverConfig``, ``EigenSolver(A, M, cfg, *, check_hermitian)``, ``.solver``,
``.config``, ``.solve()``) with the arithmetic on the GPU (see ``Solver/utils.py`` here).  Example, identical to the
reference's docstring (``Solver/eigen.py:6-23``)::

    cfg = EigensolverConfig(num_eig=6, problem_type=iEpsProblemType.GNHEP, atol=1e-8, max_it=500)
    es = EigenSolver(A, M, cfg)
    es.solver.set_st_type(iSTType.SINVERT)
    es.solver.set_target(0.018 + 0.738j)
    es.solver.set_st_pc_type(PreconditionerType.ILU)
    eigenpairs = es.solve()

The legacy argument order ``EigenSolver(cfg, A=..., M=...)`` still used by the reference's tests, CLI and docs
(``tests/unit/Solver/test_eigen.py:91``, ``Solver/cli.py:168``) is accepted as well.
"""

from __future__ import annotations

import logging
import time
from dataclasses import dataclass

from FEM.utils import iComplexPETScVector, iPETScMatrix

from .utils import iEpsProblemType, iEpsSolver

logger = logging.getLogger(__name__)

_HERMITIAN_TYPES: set[iEpsProblemType] = {
    iEpsProblemType.HEP,
    iEpsProblemType.GHEP,
    iEpsProblemType.GHIEP,
}


@dataclass(frozen=True)
class EigensolverConfig:
    """Eigensolver configuration (``Solver/eigen.py:48-61``)."""

    num_eig: int = 5
    """Number of computed eigenpairs."""
    problem_type: iEpsProblemType = iEpsProblemType.GNHEP
    """Problem type."""
    atol: float = 1e-6
    """Tolerance of the (relative) convergence test."""
    max_it: int = 500
    """Maximum number of restarts."""
    ncv: int = 80
    """Subspace dimension."""


class EigenSolver:
    """Solver for the generalized eigenvalue problem Ax = lambda Mx on the HIP path."""

    def __init__(self, *args, check_hermitian: bool = True, **solver_kwargs) -> None:
        """``EigenSolver(A, M=None, cfg=None, *, check_hermitian=True)`` (``Solver/eigen.py:67-74``); the legacy order
        ``EigenSolver(cfg, A=..., M=...)`` is recognised by type.  Extra keywords go to :class:`iEpsSolver`."""
        A, M, cfg = solver_kwargs.pop("A", None), solver_kwargs.pop("M", None), solver_kwargs.pop("cfg", None)
        for a in args:
            if isinstance(a, EigensolverConfig):
                cfg = a
            elif A is None:
                A = a
            elif M is None:
                M = a
            else:
                raise TypeError("EigenSolver takes at most the operators A, M and one EigensolverConfig")
        if A is None:
            raise ValueError("Operator A is required.")
        A = iPETScMatrix.from_matrix(A) if not isinstance(A, iPETScMatrix) else A
        if M is not None and not isinstance(M, iPETScMatrix):
            M = iPETScMatrix.from_matrix(M)
        self._cfg = cfg or EigensolverConfig()

        nrows, ncols = A.shape
        if nrows != ncols:
            raise ValueError(f"Operator A must be square, got shape ({nrows}, {ncols})")
        if M is not None:
            mrows, mcols = M.shape
            if (mrows, mcols) != (nrows, ncols):
                raise ValueError(f"Operator M shape {M.shape} does not match A's shape {A.shape}")
        if self._cfg.problem_type in _HERMITIAN_TYPES and check_hermitian:
            if not A.is_numerically_hermitian():
                logger.warning(
                    "Problem type '%s' assumes Hermitian A, but A is not (numerically) Hermitian.", self._cfg.problem_type.name
                )
            if (
                M is not None
                and self._cfg.problem_type in {iEpsProblemType.GHEP, iEpsProblemType.GHIEP}
                and not M.is_numerically_hermitian()
            ):
                # the reference dereferences the *argument* cfg here (Solver/eigen.py:106), which is None when defaulted
                logger.warning(
                    "Problem type '%s' assumes Hermitian M, but M is not (numerically) Hermitian.", self._cfg.problem_type.name
                )

        self._solver = iEpsSolver(A, M, **solver_kwargs)
        self._solver.set_problem_type(self._cfg.problem_type)
        self._solver.set_tolerances(self._cfg.atol, self._cfg.max_it)
        self._solver.set_dimensions(self._cfg.num_eig, self._cfg.ncv)

    @property
    def solver(self) -> iEpsSolver:
        """Get the solver object."""
        return self._solver

    @property
    def config(self) -> EigensolverConfig:
        """Get the solver configuration."""
        return self._cfg

    def solve(self) -> list[tuple[float | complex, iComplexPETScVector]]:
        """Run the solver and return eigenpairs."""
        logger.info(
            "Started eigenvalue solve: type=%s, nev=%d, tol=%g, max_it=%d",
            self._cfg.problem_type.name, self._cfg.num_eig, self._cfg.atol, self._cfg.max_it,
        )
        t0 = time.time()
        self._solver.solve()
        elapsed = time.time() - t0
        nconv = self._solver.get_num_converged()
        try:
            its = self._solver.raw.getST().getKSP().getIterationNumber()
        except Exception:
            its = None
        logger.info(
            "Solve completed in %.2f s; converged %d eigenpairs%s", elapsed, nconv, f"; iterations={its}" if its is not None else ""
        )
        pairs = list(self._solver.get_all_eigenpairs_up_to(self._cfg.num_eig))
        logger.info("Retrieved %d eigenpairs", len(pairs))
        return pairs
