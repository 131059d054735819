"""Front end of the MI355X eigen path with the public surface of the reference's ``Solver/eigen.py``.

What callers see is unchanged -- ``EigensolverConfig``, ``EigenSolver(A, M, cfg, *, check_hermitian)``, the ``solver`` and
``config`` properties, ``solve()`` returning ``[(eigenvalue, iComplexPETScVector), ...]`` -- so the reference's own usage
example (``Solver/eigen.py:6-23``) runs as written::

    cfg = EigensolverConfig(num_eig=6, problem_type=iEpsProblemType.GNHEP, atol=1e-8, max_it=500)
    es = EigenSolver(A, M, cfg)
    es.solver.set_st_type(iSTType.SINVERT)
    es.solver.set_target(0.018 + 0.738j)
    es.solver.set_st_pc_type(PreconditionerType.ILU)
    eigenpairs = es.solve()

Everything numerical happens behind ``es.solver`` (:class:`Solver.utils.iEpsSolver`, HIP kernels through
``liblsa_hip.so``).  Two call orders are understood, told apart by argument type: the current one and the older
``EigenSolver(cfg, A=..., M=...)`` that the reference's tests, CLI and docs still use
(``tests/unit/Solver/test_eigen.py:91``, ``Solver/cli.py:168``).
"""

from __future__ import annotations

import logging
import time
from dataclasses import dataclass

from FEM.utils import iComplexPETScVector, iPETScMatrix

from .utils import iEpsProblemType, iEpsSolver

logger = logging.getLogger(__name__)

# problem types for which SLEPc assumes a Hermitian A (and, for the generalized ones, a Hermitian M)
_NEEDS_HERMITIAN_A = frozenset({iEpsProblemType.HEP, iEpsProblemType.GHEP, iEpsProblemType.GHIEP})
_NEEDS_HERMITIAN_M = frozenset({iEpsProblemType.GHEP, iEpsProblemType.GHIEP})


@dataclass(frozen=True)
class EigensolverConfig:
    """The five knobs of ``Solver/eigen.py:48-61``, same names and defaults."""

    num_eig: int = 5  # eigenpairs to compute
    problem_type: iEpsProblemType = iEpsProblemType.GNHEP
    atol: float = 1e-6  # tolerance of the relative convergence test (EPS_CONV_REL)
    max_it: int = 500  # restarts allowed
    ncv: int = 80  # Krylov subspace dimension


def _wrap(operator):
    """Anything with a CSR view (reference wrapper, scipy matrix, ndarray) as the local matrix shim."""
    return operator if isinstance(operator, iPETScMatrix) else iPETScMatrix.from_matrix(operator)


def _sort_arguments(args, kwargs):
    """(A, M, cfg) from positional arguments in either call order plus the ``A=``, ``M=``, ``cfg=`` keywords."""
    A, M, cfg = kwargs.pop("A", None), kwargs.pop("M", None), kwargs.pop("cfg", None)
    for item in args:
        if isinstance(item, EigensolverConfig):
            cfg = item
        elif A is None:
            A = item
        elif M is None:
            M = item
        else:
            raise TypeError("EigenSolver takes at most the operators A, M and one EigensolverConfig")
    return A, M, cfg


def _require_matching_squares(A: iPETScMatrix, M: iPETScMatrix | None) -> None:
    """``ValueError`` for a non-square A or an M of another shape (``Solver/eigen.py:78-87``)."""
    rows, cols = A.shape
    if rows != cols:
        raise ValueError(f"Operator A must be square, got shape ({rows}, {cols})")
    if M is not None and tuple(M.shape) != (rows, cols):
        raise ValueError(f"Operator M shape {M.shape} does not match A's shape {A.shape}")


def _warn_about_symmetry(kind: iEpsProblemType, A: iPETScMatrix, M: iPETScMatrix | None) -> None:
    """The reference's advisory check (``:88-108``): a Hermitian problem type with a numerically non-Hermitian operator
    only earns a warning.  (There the M branch reads the *argument* ``cfg``, ``None`` when defaulted -- ``:106``; here the
    effective configuration is used.)"""
    message = "Problem type '%s' assumes Hermitian %s, but %s is not (numerically) Hermitian."
    if kind in _NEEDS_HERMITIAN_A and not A.is_numerically_hermitian():
        logger.warning(message, kind.name, "A", "A")
    if M is not None and kind in _NEEDS_HERMITIAN_M and not M.is_numerically_hermitian():
        logger.warning(message, kind.name, "M", "M")


class EigenSolver:
    """``A x = lambda M x`` (``M`` optional) on the GPU; thin shell around :class:`iEpsSolver`."""

    def __init__(self, *args, check_hermitian: bool = True, **solver_kwargs) -> None:
        """``EigenSolver(A, M=None, cfg=None, *, check_hermitian=True)`` as in ``Solver/eigen.py:67-74``, or the legacy
        ``EigenSolver(cfg, A=..., M=...)``.  Keywords the reference does not know (``device``, ``ilu_levels``,
        ``restart``, ``layout``, ...) are handed to :class:`iEpsSolver`."""
        A, M, cfg = _sort_arguments(args, solver_kwargs)
        if A is None:
            raise ValueError("Operator A is required.")
        A = _wrap(A)
        M = None if M is None else _wrap(M)
        _require_matching_squares(A, M)
        self._cfg = cfg if cfg is not None else EigensolverConfig()
        if check_hermitian:
            _warn_about_symmetry(self._cfg.problem_type, A, M)

        eps = iEpsSolver(A, M, **solver_kwargs)
        eps.set_problem_type(self._cfg.problem_type)
        eps.set_tolerances(self._cfg.atol, self._cfg.max_it)
        eps.set_dimensions(self._cfg.num_eig, self._cfg.ncv)
        self._solver = eps

    @property
    def solver(self) -> iEpsSolver:
        """The configurable solver object (``set_st_type``, ``set_target``, ``set_st_pc_type``, ...)."""
        return self._solver

    @property
    def config(self) -> EigensolverConfig:
        """The configuration this solver was built with."""
        return self._cfg

    def solve(self) -> list[tuple[float | complex, iComplexPETScVector]]:
        """Solve and hand back up to ``num_eig`` converged pairs, wanted first (``Solver/eigen.py:125-155``)."""
        cfg = self._cfg
        logger.info("Started eigenvalue solve: type=%s, nev=%d, tol=%g, max_it=%d", cfg.problem_type.name, cfg.num_eig, cfg.atol, cfg.max_it)
        started = time.time()
        self._solver.solve()
        seconds = time.time() - started
        inner = self._solver.stats.get("gmres_iters")
        logger.info("Solve completed in %.2f s; converged %d eigenpairs%s", seconds, self._solver.get_num_converged(),
                    "" if inner is None else f"; iterations={inner}")
        pairs = list(self._solver.get_all_eigenpairs_up_to(cfg.num_eig))
        logger.info("Retrieved %d eigenpairs", len(pairs))
        return pairs
