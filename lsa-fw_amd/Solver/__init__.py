"""LSA-FW Solver package, eigen path only (drop-in for ``/root/reference/Solver``'s ``eigen`` and ``utils`` modules)."""

from .eigen import EigenSolver, EigensolverConfig  # noqa: F401
from .utils import KSPType, PreconditionerType, iEpsProblemType, iEpsSolver, iEpsWhich, iSTType  # noqa: F401
