"""Drop-in for the EPS half of ``/root/reference/Solver/utils.py`` on MI355X.

Same names, argument meaning and error behaviour as the reference wrapper over ``SLEPc.EPS``
(``Solver/utils.py:27-328``); the arithmetic runs in ``liblsa_hip.so`` (HIP kernels) instead of SLEPc/PETSc:

============================  ======================================================================================
reference (slepc4py)          here
============================  ======================================================================================
``EPS`` Krylov-Schur          ``lsa_hip.krylov_schur`` (host logic) over ``lsa_hip.KrylovBasis`` (device basis, CGS2)
``ST`` SINVERT / SHIFT        ``lsa_hip.ShiftInvertOperator`` (C = A - sigma M built on the device)
``KSP`` of the ST             device GMRES (right-preconditioned, CGS2)
``PC`` ILU / LU               device ILU(k) + sync-free SpTRSV; LU/CHOLESKY map to a higher fill level
``BV`` / ``DS``               HIP tall-skinny kernels / LAPACK on the ncv x ncv Hessenberg
============================  ======================================================================================

There is no CPU fallback: ``solve()`` raises if the HIP library or a GPU is missing.
"""

from __future__ import annotations

import logging
from enum import Enum
from typing import Iterator

import numpy as np
import scipy.sparse as sp

from FEM.utils import iComplexPETScVector, iPETScMatrix, iPETScVector

logger = logging.getLogger(__name__)

Scalar = complex
"""Scalar type of the solver. The HIP path always iterates in complex128 (the reference targets are complex,
``.examples/eigenvalues.py:37-49``), so the reference's real/complex *build* switch does not exist here."""


class _LowerStrEnum(str, Enum):
    """``enum.StrEnum`` + ``auto()`` semantics of the reference (members compare equal to their lower-case name)."""

    def __str__(self) -> str:
        return str(self.value)


class iEpsProblemType(Enum):
    """EPS problem types (``Solver/utils.py:27-63``); values follow SLEPc's ``EPSProblemType`` numbering."""

    HEP = 1
    GHEP = 2
    NHEP = 3
    GNHEP = 4
    PGNHEP = 5
    GHIEP = 6

    def to_slepc(self) -> int:
        return self.value

    @classmethod
    def from_slepc(cls, problem_type) -> "iEpsProblemType":
        try:
            return cls(problem_type) if not hasattr(problem_type, "name") else cls[problem_type.name]
        except (KeyError, ValueError):
            raise ValueError(f"Unsupported SLEPc EPS ProblemType: {problem_type}")

    @classmethod
    def from_string(cls, name: str) -> "iEpsProblemType":
        try:
            return cls[name.upper()]
        except KeyError:
            raise ValueError(f"Invalid problem type: {name}. Choose from {list(cls.__members__.keys())}.")


class PreconditionerType(_LowerStrEnum):
    """Preconditioner names accepted by ``set_st_pc_type`` (``Solver/utils.py:66-93``)."""

    NONE = "none"
    JACOBI = "jacobi"
    SOR = "sor"
    ASM = "asm"
    ILU = "ilu"
    ICC = "icc"
    LU = "lu"
    CHOLESKY = "cholesky"
    GAMG = "gamg"
    HYPRE = "hypre"
    REDUNDANT = "redundant"
    SHELL = "shell"


class KSPType(_LowerStrEnum):
    """KSP names (``Solver/utils.py:96-128``)."""

    CG = "cg"
    GMRES = "gmres"
    BICG = "bicg"
    BICGSTAB = "bicgstab"
    RICHARDSON = "richardson"
    CHEBYSHEV = "chebyshev"
    PREONLY = "preonly"
    QCG = "qcg"
    CGS = "cgs"
    GCR = "gcr"
    LSQR = "lsqr"
    LGMRES = "lgmres"
    FGMRES = "fgmres"

    def to_petsc(self) -> str:
        return self.name.lower()


class iSTType(Enum):
    """Spectral transformations (``Solver/utils.py:131-149``)."""

    SHELL = "shell"
    SHIFT = "shift"
    SINVERT = "sinvert"
    CAYLEY = "cayley"
    PRECOND = "precond"
    FILTER = "filter"

    def to_slepc(self) -> str:
        return self.value


class iEpsWhich(Enum):
    """Which eigenpairs (``Solver/utils.py:152-187``). SMALLEST_MAGNITUDE is an alias of LARGEST_REAL there
    (``:157``) and therefore here."""

    ALL = 10
    LARGEST_MAGNITUDE = 1
    LARGEST_REAL = 3
    SMALLEST_MAGNITUDE = 3
    SMALLEST_REAL = 4
    LARGEST_IMAGINARY = 5
    SMALLEST_IMAGINARY = 6
    TARGET_MAGNITUDE = 7
    TARGET_REAL = 8
    TARGET_IMAGINARY = 9
    USER = 11

    def to_slepc(self) -> int:
        return self.value

    def to_arpack(self) -> str:
        table = {
            iEpsWhich.LARGEST_REAL: "LR",
            iEpsWhich.LARGEST_IMAGINARY: "LI",
            iEpsWhich.SMALLEST_REAL: "SR",
            iEpsWhich.SMALLEST_IMAGINARY: "SI",
            iEpsWhich.LARGEST_MAGNITUDE: "LM_abs",
        }
        if self not in table:
            raise ValueError(f"Unsupported type for ARPACK-based eigensolver: {self.name}.")
        return table[self]


def _lambda_rank_key(which: iEpsWhich, target: complex):
    """Sort key on eigenvalues lambda (small = wanted), following SLEPc's EPSWhich semantics."""
    keys = {
        iEpsWhich.LARGEST_MAGNITUDE: lambda lam: -np.abs(lam),
        iEpsWhich.LARGEST_REAL: lambda lam: -lam.real,
        iEpsWhich.SMALLEST_REAL: lambda lam: lam.real,
        iEpsWhich.LARGEST_IMAGINARY: lambda lam: -lam.imag,
        iEpsWhich.SMALLEST_IMAGINARY: lambda lam: lam.imag,
        iEpsWhich.TARGET_MAGNITUDE: lambda lam: np.abs(lam - target),
        iEpsWhich.TARGET_REAL: lambda lam: np.abs(lam.real - target.real),
        iEpsWhich.TARGET_IMAGINARY: lambda lam: np.abs(lam.imag - target.imag),
    }
    if which not in keys:
        raise ValueError(f"which = {which.name} is not supported by the HIP eigensolver")
    return keys[which]


def delay_zero_diagonal_rows(C: sp.csr_matrix, block_starts: np.ndarray | None = None, fraction: float = 0.5) -> np.ndarray:
    """Permutation (new -> old) that moves every zero-diagonal row (the pressure rows of the saddle-point operator)
    behind ``fraction`` of the rows it couples to, so that its ILU pivot receives fill from enough velocity rows: one
    preceding neighbour is not enough in general (two pressure rows that see the same two velocity rows leave a
    singular leading minor).  Rows that already satisfy the rule stay where they are.  With ``block_starts`` the rule
    is applied inside each contiguous row block separately (only couplings inside the block count): the form the
    block-Jacobi factors of the sharded layout need.  Rows keep their block."""
    n = C.shape[0]
    zero_diag = np.flatnonzero(C.diagonal() == 0)
    if zero_diag.size == 0:
        return np.arange(n, dtype=np.int64)
    blk = None if block_starts is None else np.searchsorted(block_starts, np.arange(n), side="right") - 1
    key = np.arange(n, dtype=np.float64)
    indptr, indices, data = C.indptr, C.indices, C.data
    # all entries of the zero-diagonal rows at once (no Python loop over rows: 3 480 of them at S30k)
    starts = indptr[zero_diag].astype(np.int64)
    lens = (indptr[zero_diag + 1] - indptr[zero_diag]).astype(np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.arange(n, dtype=np.int64)
    pos = np.repeat(starts - (np.cumsum(lens) - lens), lens) + np.arange(total)
    rid = np.repeat(np.arange(zero_diag.size), lens)  # which zero-diagonal row an entry belongs to
    cols = indices[pos].astype(np.int64)
    keep = (data[pos] != 0) & (cols != zero_diag[rid])
    if blk is not None:
        keep &= blk[cols] == blk[zero_diag[rid]]
    cols, rid = cols[keep], rid[keep]
    order = np.lexsort((cols, rid))  # by row, neighbours ascending inside a row
    cols, rid = cols[order], rid[order]
    cnt = np.bincount(rid, minlength=zero_diag.size)
    first = np.cumsum(cnt) - cnt
    has = cnt > 0
    pick = first[has] + np.maximum(np.ceil(fraction * cnt[has]).astype(np.int64) - 1, 0)
    target = cols[pick]
    z = zero_diag[has]
    later = target > z
    key[z[later]] = target[later] + 0.5
    return np.argsort(key, kind="stable").astype(np.int64)


def pivot_safe_rcm(C: sp.csr_matrix) -> np.ndarray:
    """Bandwidth-reducing symmetric permutation that keeps ILU pivots non-zero on saddle-point matrices.

    Reverse Cuthill-McKee on the pattern (the analogue of PETSc's ``-pc_factor_mat_ordering_type rcm``), then every
    row with a zero diagonal (the pressure rows: ``tests/unit/FEM/test_operators.py:209-210``) that RCM placed before
    *all* rows it couples to is moved right behind the first of them (:func:`delay_zero_diagonal_rows`).
    """
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    pattern = sp.csr_matrix((np.ones(C.nnz, dtype=np.int8), C.indices, C.indptr), shape=C.shape)
    perm = np.asarray(reverse_cuthill_mckee(pattern, symmetric_mode=False), dtype=np.int64)
    Cp = C[perm][:, perm].tocsr()
    return perm[delay_zero_diagonal_rows(Cp)]


class _RawPC:
    def __init__(self, owner: "iEpsSolver"):
        self._o = owner

    def getType(self) -> str:
        return self._o._pc_type.name.lower()


class _RawKSP:
    def __init__(self, owner: "iEpsSolver"):
        self._o = owner

    def getPC(self) -> _RawPC:
        return _RawPC(self._o)

    def getType(self) -> str:
        return self._o._ksp_type.to_petsc()

    def getIterationNumber(self) -> int:
        return int(self._o._stats.get("gmres_iters", 0))


class _RawST:
    def __init__(self, owner: "iEpsSolver"):
        self._o = owner

    def getKSP(self) -> _RawKSP:
        return _RawKSP(self._o)

    def getType(self) -> str:
        return self._o._st_type.to_slepc()

    def getShift(self) -> complex:
        return self._o._target


class _RawEPS:
    """The few ``SLEPc.EPS`` getters the reference's callers and tests poke through ``iEpsSolver.raw``
    (``Solver/eigen.py:140-144``; ``tests/unit/Solver/test_eigen.py:97-104,320-322``)."""

    def __init__(self, owner: "iEpsSolver"):
        self._o = owner

    def getTolerances(self) -> tuple[float, int]:
        return self._o._tol, self._o._max_it

    def getDimensions(self) -> tuple[int, int, int]:
        return self._o._nev, self._o._ncv, self._o._ncv

    def getProblemType(self) -> int:
        return self._o._problem_type.to_slepc()

    def getST(self) -> _RawST:
        return _RawST(self._o)

    def getConverged(self) -> int:
        return self._o.get_num_converged()

    def getIterationNumber(self) -> int:
        return int(self._o._restarts)


_HERMITIAN = {iEpsProblemType.HEP, iEpsProblemType.GHEP}


class iEpsSolver:
    """SLEPc-shaped eigensolver front end running on the HIP path (reference: ``Solver/utils.py:190-328``).

    Build-only knobs (the reference leaves them to the PETSc options database) are keyword arguments:
    ``ksp_type`` (inner Krylov method, GMRES), ``ksp_rtol``, ``restart``, ``ksp_max_it``, ``ilu_levels``,
    ``ilu_shift``, ``device``.
    """

    def __init__(self, A=None, M=None, comm=None, *, device: int = 0, ksp_type: KSPType = KSPType.GMRES,
                 ksp_rtol: float | None = None, restart: int = 1000, ksp_max_it: int = 4000, ilu_levels: int | None = None,
                 ilu_shift: float = 0.0, ordering: str = "rcm", seed: int = 0, layout: str = "single",
                 project_out: np.ndarray | None = None, lu: str = "nd", adjoint: bool = False) -> None:
        if M is not None and A is None:
            raise ValueError("Cannot set right-hand operator M without left-hand operator A.")
        self._A = self._M = None
        self._problem_type = iEpsProblemType.NHEP
        self._nev, self._ncv = 1, None
        self._tol, self._max_it = 1e-8, 100
        self._which: iEpsWhich | None = None
        self._target: complex = 0.0
        self._interval = None
        self._st_type = iSTType.SHIFT
        self._antishift: complex | None = None
        self._pc_type = PreconditionerType.LU  # SLEPc's default for the ST's KSP is preonly + LU
        self._ksp_type = ksp_type
        self._ksp_rtol, self._restart_len, self._ksp_max_it = ksp_rtol, restart, ksp_max_it
        self._ilu_levels, self._ilu_shift = ilu_levels, ilu_shift
        self._ordering = ordering
        if lu != "nd":  # ('band', round 1's block-tridiagonal LU of the RCM order, is a cross-check library of the tests now)
            raise ValueError("lu must be 'nd' (nested-dissection multifrontal LU)")
        self._lu = lu
        # adjoint=True: eigenpairs of (A^H, M^H) -- left eigenvectors of (A, M) -- through the operator
        # (A - conj(target) M)^-H M^H applied on the factors of A - conj(target) M: no transposed matrix is formed
        # (reference: Sensitivity/__init__.py:47-57,247-248 builds A^H and M^H explicitly)
        self._adjoint = bool(adjoint)
        self._device, self._seed = device, seed
        if layout not in ("single", "sharded"):
            raise ValueError("layout must be 'single' or 'sharded'")
        self._layout = layout  # 'sharded': rows of (A, M) and the ILU split over the ranks of torch.distributed
        # dof indices zeroed on both sides of every operator apply (the velocity-subspace projection of
        # ``ArpackEigenSolver``, reference ``Solver/eigen2.py:164-201``)
        self._project_out = None if project_out is None else np.unique(np.asarray(project_out, dtype=np.int64))
        self._eigenvalues = np.zeros(0, dtype=np.complex128)
        self._eigenvectors = np.zeros((0, 0), dtype=np.complex128)
        self._imag_norms = None
        self._residual_estimates = np.zeros(0)
        self._stats: dict = {}
        self._restarts = 0
        self._prepared = None
        if A is not None:
            self.set_operators(A, M)

    # ---- configuration (same names as the reference) ----------------------------------------------------------------
    @property
    def raw(self) -> _RawEPS:
        return _RawEPS(self)

    def set_operators(self, A, M=None) -> None:
        self._A = A if isinstance(A, iPETScMatrix) else iPETScMatrix.from_matrix(A)
        self._M = None if M is None else M if isinstance(M, iPETScMatrix) else iPETScMatrix.from_matrix(M)

    def set_problem_type(self, problem_type: iEpsProblemType) -> None:
        self._problem_type = problem_type

    def set_dimensions(self, number_eigenpairs: int, subspace_dimension: int | None = None) -> None:
        self._nev = int(number_eigenpairs)
        self._ncv = None if subspace_dimension is None else int(subspace_dimension)

    def set_tolerances(self, atol: float, max_it: int) -> None:
        self._tol, self._max_it = float(atol), int(max_it)

    def set_which_eigenpairs(self, which: iEpsWhich) -> None:
        self._which = which

    def set_target(self, sigma: float | complex) -> None:
        self._target = complex(Scalar(sigma))

    def set_interval(self, a: float, b: float) -> None:
        self._interval = (float(a), float(b), -np.inf, np.inf)

    def set_interval_complex(self, a: float, b: float, c: float, d: float) -> None:
        self._interval = (float(a), float(b), float(c), float(d))

    def set_st_type(self, st_type: iSTType) -> None:
        self._st_type = st_type

    def set_st_antishift(self, nu: float | complex | None) -> None:
        """Antishift of the Cayley transform ``(A - sigma M)^-1 (A + nu M)`` (``STCayleySetAntishift``; build-only:
        the reference never sets it, SLEPc then uses ``nu = sigma``)."""
        self._antishift = nu

    def set_st_pc_type(self, pc_type: PreconditionerType) -> None:
        self._pc_type = PreconditionerType(pc_type)

    # ---- solve ---------------------------------------------------------------------------------------------------------
    def _fill_level(self) -> tuple[int, int]:
        """(pc_type code, ILU fill level) for the requested PETSc PC name."""
        if self._pc_type is PreconditionerType.NONE:
            return 0, 0
        if self._ilu_levels is not None:
            return 1, int(self._ilu_levels)
        if self._pc_type in (PreconditionerType.ILU, PreconditionerType.ICC):
            return 1, 0  # PETSc's PCILU default: zero fill
        if self._pc_type in (PreconditionerType.LU, PreconditionerType.CHOLESKY):
            # exact solves in the reference: nested-dissection multifrontal LU on the device; ILU(2) + GMRES only if the
            # factors do not fit the device memory
            return 2, 2
        return 1, 2  # every other PETSc name: ILU(2) + GMRES

    def _signature(self):
        return (id(self._A), id(self._M), self._st_type, self._target, self._pc_type, self._ilu_levels, self._ordering, self._device,
                self._layout, self._lu, self._adjoint)  # (the antishift only changes the multiplied matrix, built per solve)

    def _only_the_target_moved(self, prev: dict, sig: tuple) -> bool:
        old = prev["sig"]
        if old[:3] + old[4:] != sig[:3] + sig[4:] or not prev["sinvert"] or prev["part"] is not None or prev["forest"] is not None:
            return False
        if self._target is None or old[3] is None:
            return False
        # real <-> complex factors are different scalar types on the device (lsa_ndlu_prepare)
        return (complex(old[3]).imag != 0.0) == (complex(self._target).imag != 0.0)

    def prepare(self) -> None:
        """Host-side analysis + upload: shared pattern, fill-reducing / pivot-safe ordering, CSR -> HBM.

        ``solve()`` calls this on demand; calling it first keeps file I/O, ordering and the PCIe upload out of a timed
        ``solve()`` (the metric of BASELINE.json counts factorisation + iteration, not upload)."""
        import lsa_hip

        if self._A is None:
            raise ValueError("Operators are not set.")
        prev, sig = getattr(self, "_prepared", None), self._signature()
        if prev is not None and prev["sig"] == sig:
            return
        if prev is not None and self._only_the_target_moved(prev, sig):
            # A shift sweep on one pair (A, M) -- the interval solve behind iEpsWhich.ALL, a user scanning targets: everything
            # prepared depends on the PATTERN of A - sigma M and on whether its factors are real or complex, not on the shift.
            # The context (with its cached analysis and buffers), the uploaded matrices and the ordering stay.
            prev["sigma"], prev["sig"] = self._target, sig
            return
        self.release()
        A = self._A.as_scipy_array()
        M = None if self._M is None else self._M.as_scipy_array()
        n = A.shape[0]
        if self._st_type not in (iSTType.SHIFT, iSTType.SINVERT, iSTType.CAYLEY):
            raise NotImplementedError(f"spectral transformation {self._st_type.name} is not available on the HIP path "
                                      "(SHIFT, SINVERT and CAYLEY are)")
        sinvert = self._st_type in (iSTType.SINVERT, iSTType.CAYLEY)  # both factorise A - sigma M
        sigma = self._target if sinvert else 0.0
        # one shared sparsity pattern for A and M (explicit zeros where only the other matrix has an entry)
        if M is not None and (A.nnz != M.nnz or not (np.array_equal(A.indptr, M.indptr) and np.array_equal(A.indices, M.indices))):
            ones = lambda X: sp.csr_matrix((np.ones(X.nnz), X.indices, X.indptr), shape=X.shape)  # noqa: E731
            union = (ones(A) + ones(M)).tocsr()
            union.sort_indices()
            A, M = _onto_pattern(A, union), _onto_pattern(M, union)
        # symmetric permutation for the factorisation; the whole iteration runs in permuted numbering
        pc_code, levels = self._fill_level()
        if sinvert:
            K = _combine(A, M, sigma) if M is not None else A  # the matrix that gets factorised
        else:
            K = M
        nd_tree = None
        if pc_code == 2 and K is not None and n > 8 and self._ordering != "natural" and not (self._layout == "sharded" and _dist_rank_world()[1] > 1):
            # Exact nested-dissection LU: the whole iteration runs in ITS elimination order (a tree node's own unknowns are
            # contiguous in every vector: the sweeps address them without index lists); the forest goes back to the library
            # with the permuted pattern.  Zero-diagonal (pressure) unknowns as constraints, eliminated after their
            # neighbours, on 3D-like patterns (> 60 entries per row), where a leaf subdomain can hold more pressure
            # unknowns than its interior supports; costs 20 % more factor entries in 2D, where it has not been needed (and
            # the library re-analyses by itself if it ever is).
            Kc = sp.csr_matrix(K)
            zero_diag = None
            if Kc.nnz > 60 * n:
                zd = Kc.diagonal() == 0
                zero_diag = zd if zd.any() else None
            nd_tree = lsa_hip.nd_order(Kc, 0, constraint=zero_diag)
            perm = nd_tree["perm"]
        elif self._ordering == "rcm" and n > 8 and pc_code >= 1 and K is not None:
            perm = pivot_safe_rcm(sp.csr_matrix(K))
        else:
            perm = np.arange(n)
        Ap = _permute(A, perm)
        Mp = None if M is None else _permute(M, perm)
        ctx = lsa_hip.Context(self._device)
        part = dAd = dMd = forest = None
        world = _dist_rank_world()[1] if self._layout == "sharded" else 1
        if self._layout == "sharded" and world > 1 and pc_code == 2 and sinvert:
            # Subtree-parallel exact LU (the LU-class setting of the reference, sharded): the nested-dissection forest is cut
            # over the ranks, a rank's unknowns are one row block of the padded layout, every rank holds the whole (A, M)
            # -- it factors its subtrees plus the replicated top of the forest; the sparse products run replicated (2D patterns)
            # or on its rows (3D patterns), see lsa_op_create_dist.
            from lsa_hip import sharding

            rank = _dist_rank_world()[0]
            Kc = sp.csr_matrix(K)
            zd = Kc.diagonal() == 0
            an = lsa_hip.NdAnalysis(Kc, 0, constraint=zd if zd.any() else None)  # (constraints last: no retry across ranks)
            ex = an.export()
            # (top nodes with large fronts -- the 3D cases -- are distributed over the ranks, for the adjoint problem too: the
            #  transposed sweeps sum a distributed node's partial results of all ranks in rank order)
            forest = sharding.partition_forest(ex["perm"], ex["node_start"], ex["parent"], ex["front_size"], world)
            perm, part = forest.order, forest.rows
            dA = lsa_hip.CsrMatrix.from_scipy(ctx, sharding.pad_square(_permute(A, perm), part))
            dM = None if M is None else lsa_hip.CsrMatrix.from_scipy(ctx, sharding.pad_square(_permute(M, perm), part))
            _dist_comm_init(ctx, world, rank)
        elif self._layout == "sharded":
            # one process per GPU: rows of the permuted pair go to the ranks of torch.distributed (RCCL bootstrap through it)
            from lsa_hip import sharding

            rank, world = _dist_rank_world()
            part = sharding.partition_rows(Ap.indptr, world)
            if pc_code >= 1 and K is not None and world > 1:
                # block-Jacobi factors see only their diagonal block: redo the zero-pivot rule inside each block
                q = delay_zero_diagonal_rows(_permute(sp.csr_matrix(K), perm), part.starts)
                perm = perm[q]
                Ap = _permute(A, perm)
                Mp = None if M is None else _permute(M, perm)
            dA = lsa_hip.CsrMatrix.from_scipy_shard(ctx, sharding.shard_rows(Ap, part, rank), part.n_pad, rank * part.b_pad)
            dM = None if Mp is None else lsa_hip.CsrMatrix.from_scipy_shard(ctx, sharding.shard_rows(Mp, part, rank), part.n_pad, rank * part.b_pad)
            dAd = lsa_hip.CsrMatrix.from_scipy(ctx, sharding.diagonal_block(Ap, part, rank))
            dMd = None if Mp is None else lsa_hip.CsrMatrix.from_scipy(ctx, sharding.diagonal_block(Mp, part, rank))
            if world > 1:
                _dist_comm_init(ctx, world, rank)
        else:
            dA = lsa_hip.CsrMatrix.from_scipy(ctx, Ap)
            dM = None if Mp is None else lsa_hip.CsrMatrix.from_scipy(ctx, Mp)
        if pc_code == 2 and K is not None and forest is None:
            # pattern-only phase of the nested-dissection LU (elimination forest, index tables, memory plan, buffers) for the
            # scalar type the library will give C = A - sigma M (lsa_op_create): complex only for a complex shift or complex
            # operators (K above is complex whenever the target was stored as a Python complex)
            cplx_factors = A.dtype.kind == "c" or (M is not None and M.dtype.kind == "c") or (sinvert and complex(sigma).imag != 0.0)
            fac = dAd if dAd is not None else dA
            try:
                if nd_tree is not None and dAd is None:
                    fac.prepare_lu_tree(cplx_factors, nd_tree["first"], nd_tree["size"], nd_tree["parent"])
                else:  # a rank's diagonal block (block-Jacobi layout), or no ordering asked for: the library dissects by itself
                    zero_diag = None
                    if part is None and fac.nnz > 60 * fac.shape[0]:
                        zd = sp.csr_matrix(K).diagonal()[perm] == 0
                        zero_diag = zd if zd.any() else None
                    fac.prepare_lu(cplx_factors, constraint=zero_diag)
            except lsa_hip.LsaError as exc:
                # The buffers of the exact LU do not fit the device: not an error of this phase.  The operator build meets the
                # same condition and answers it as PETSc's users would have to by hand: ILU(k) + GMRES, said in the statistics
                # (stats["pc_fallback"]) and on stderr.  Anything else is an error here as there.
                if exc.status != lsa_hip.LSA_ERR_OOM:
                    raise
                logger.warning("The exact LU does not fit the device memory (%s); the solve will fall back to ILU(k) + GMRES.", exc)
        self._prepared = {"sig": self._signature(), "ctx": ctx, "dA": dA, "dM": dM, "dAd": dAd, "dMd": dMd, "part": part, "perm": perm,
                          "forest": forest,
                          "n": n, "sinvert": sinvert, "cayley": self._st_type is iSTType.CAYLEY, "sigma": sigma, "pc_code": pc_code,
                          "levels": levels}

    def release(self) -> None:
        """Free the device copies made by :meth:`prepare`."""
        prep = getattr(self, "_prepared", None)
        self._prepared = None
        self._comm_seen = {"allgather_calls": 0, "allgather_bytes_received": 0}  # counters live in the context
        if prep is not None:
            ctx = prep.pop("ctx")
            prep.clear()
            import gc

            gc.collect()  # matrix handles hold device memory; they must go before the context
            ctx.close()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def solve(self) -> None:
        """Run the eigensolver on the GPU (reference: ``self._eps.solve()``, ``Solver/utils.py:268-270``):
        build and factorise ``A - sigma M`` on the device, then Krylov-Schur with device-resident Arnoldi."""
        import lsa_hip
        from lsa_hip.krylov_schur import krylov_schur

        if self._which is iEpsWhich.ALL:
            self._solve_interval()
            return
        self.prepare()
        prep = self._prepared
        ctx, n, sinvert, sigma, perm = prep["ctx"], prep["n"], prep["sinvert"], prep["sigma"], prep["perm"]
        ncv = min(self._ncv if self._ncv is not None else max(2 * self._nev, self._nev + 15), n)
        nev = min(self._nev, ncv)
        cayley = prep["cayley"]
        nu = complex(sigma if self._antishift is None else self._antishift)  # STCayleySetAntishift defaults to the shift
        which = self._which or (iEpsWhich.TARGET_MAGNITUDE if sinvert else iEpsWhich.LARGEST_MAGNITUDE)
        # (like SLEPc, an interval set without iEpsWhich.ALL has no effect; ALL is handled by _solve_interval)
        lam_key = _lambda_rank_key(which, self._target)
        if self._adjoint and (not sinvert or cayley or prep["pc_code"] != 2 or (prep["part"] is not None and prep["forest"] is None)):
            raise NotImplementedError("adjoint=True needs shift-invert with the exact LU (PreconditionerType.LU, lu='nd'), on one GPU or in the "
                                      "subtree-parallel sharded layout")
        ksp_rtol = self._ksp_rtol if self._ksp_rtol is not None else float(np.clip(self._tol * 1e-2, 1e-13, 1e-8))
        op = basis = None
        try:
            op = lsa_hip.ShiftInvertOperator(
                ctx, prep["dA"], prep["dM"], np.conj(sigma) if self._adjoint else sigma, mode=2 if cayley else 0 if sinvert else 1, antishift=nu,
                ilu_levels=prep["levels"],
                ilu_shift=self._ilu_shift,
                ksp_rtol=ksp_rtol, ksp_restart=_restart_length(self._restart_len, n), ksp_maxit=self._ksp_max_it, pc_type=prep["pc_code"],
                A_diag=prep["dAd"], M_diag=prep["dMd"], forest=prep["forest"],
                rows=None if prep["forest"] is None else (_dist_rank_world()[0] * prep["part"].b_pad,
                                                          _dist_rank_world()[0] * prep["part"].b_pad + int(np.diff(prep["part"].starts)[_dist_rank_world()[0]])),
            )
            if self._adjoint:
                op.set_adjoint(True)
            part = prep["part"]
            keep = None
            if self._project_out is not None:
                if self._project_out.size and (self._project_out[0] < 0 or self._project_out[-1] >= n):
                    raise ValueError("project_out holds dof indices outside [0, n)")
                keep = np.ones(n)
                keep[self._project_out] = 0.0
                keep = keep[perm]  # the iteration runs in permuted numbering
            mask = keep if part is None else part.pad_vector(np.ones(n) if keep is None else keep)
            if keep is not None:
                op.set_projection(mask)
            basis = lsa_hip.KrylovBasis(ctx, op, ncv, mask)
            if part is None:
                basis.set_row_permutation(perm)  # Ritz vectors leave the device in the caller's numbering
            tiny = np.finfo(float).tiny
            if cayley:  # theta = (lambda + nu) / (lambda - sigma)
                back = lambda th: (sigma * th + nu) / np.where(th == 1.0, 1.0 + 1e-300, th - 1.0)  # noqa: E731
            elif sinvert:  # theta = 1 / (lambda - sigma)
                back = lambda th: sigma + 1.0 / np.where(th == 0, tiny, th)  # noqa: E731
            else:
                back = lambda th: th + sigma  # noqa: E731
            theta_key = lambda th: lam_key(back(np.asarray(th, dtype=np.complex128)))  # noqa: E731
            import os

            if os.environ.get("LSA_KS_DRIVER", "native") == "python":
                # the same outer iteration in Python over LAPACK (lsa_hip/krylov_schur.py): test double of the library's loop
                res = krylov_schur(basis, nev, self._tol, self._max_it, theta_key, rng_seed=self._seed)
            else:
                # one library call: Krylov-Schur with the library's own dense algebra (lsa_krylov_solve).  The start vector is
                # drawn here so that both drivers begin from the same one.
                cached = getattr(self, "_v0_cache", None)
                if cached is None or cached[0] != (basis.n, self._seed):  # (0.7 ms at 30 k unknowns: a shift sweep draws it once)
                    rng = np.random.default_rng(self._seed)
                    cached = ((basis.n, self._seed), rng.standard_normal(basis.n) + 1j * rng.standard_normal(basis.n))
                    self._v0_cache = cached
                v0 = cached[1]
                res = basis.solve(nev, self._tol, self._max_it, which.value, 2 if cayley else 0 if sinvert else 1, sigma, antishift=nu,
                                  target=self._target, v0=v0, seed=self._seed)
            imag_norms = getattr(basis, "imag_norms", None)  # set when the device already put the vectors into canonical phase
            theta = res.theta
            lam = back(np.asarray(theta, dtype=np.complex128))
            if part is None:
                X = res.vectors
            else:
                vecs = part.unpad_vector(res.vectors)
                X = np.empty_like(vecs)
                X[perm, :] = vecs
            self._stats = op.stats()
            self._stats["krylov_restarts"] = res.restarts
            if res.history and "seconds_dense" in res.history[-1]:  # the library's loop times its phases
                self._stats.update({k: res.history[-1][k] for k in ("seconds_expand", "seconds_dense", "seconds_restart")})
            if part is not None:
                now = ctx.comm_stats()
                last = getattr(self, "_comm_seen", {"allgather_calls": 0, "allgather_bytes_received": 0})
                self._stats.update({k: now[k] - last[k] for k in now})  # of this solve
                self._comm_seen = now
                self._stats["ranks"] = part.nranks
            if self._stats.get("pc_fallback"):
                logger.warning("The exact LU did not fit the device memory: the inner solves ran ILU(%d)-preconditioned GMRES instead.",
                               prep["levels"])
            if self._stats.get("backward_accepted"):
                logger.info("%d inner solves were accepted on their backward error (the shift lies next to eigenvalues: A - sigma M is "
                            "ill-conditioned); worst true relative residual %.2e. A direct solver does no better; check residuals().",
                            self._stats["backward_accepted"], self._stats.get("max_rel_res", 0.0))
            if self._stats.get("stagnated_solves") or (self._stats.get("max_rel_res", 0.0) > 10.0 * ksp_rtol and not self._stats.get("backward_accepted")):
                logger.warning("Inner solves stagnated above the requested tolerance: worst true relative residual %.2e (ksp_rtol %.1e, %d "
                               "solves accepted at the rounding floor). Eigenpairs are those of an inexactly applied operator; check "
                               "residuals().", self._stats.get("max_rel_res", 0.0), ksp_rtol, self._stats.get("stagnated_solves", 0))
        finally:
            basis = None
            op = None
        order = np.argsort(lam_key(lam), kind="stable")
        self._eigenvalues = lam[order]
        # column access (one eigenvector) must be contiguous; the Krylov-Schur driver already returns the wanted pairs first
        self._eigenvectors = X if (np.array_equal(order, np.arange(len(order))) and X.flags.f_contiguous) else np.asfortranarray(X[:, order])
        self._imag_norms = imag_norms[order] if (imag_norms is not None and len(imag_norms) == len(order)) else None
        self._residual_estimates = res.residuals[order]
        self._restarts = res.restarts

    def _solve_interval(self) -> None:
        """``iEpsWhich.ALL`` + ``set_interval(a, b)`` (reference: ``Solver/utils.py:248-254``, SLEPc's spectrum slicing):
        every eigenvalue of a Hermitian problem in ``[a, b]``, by a sweep of shift-invert solves.

        The sweep: shift-invert Krylov-Schur converges on a Hermitian problem nearest the shift first, so a solve that returned
        the m nearest eigenvalues has found everything within the distance of the m-th.  Shifts advance from a to b so that
        those covered intervals overlap; a gap is closed by a shift inside it.  Pairs found from two shifts are merged when
        their eigenvalues agree and their vectors are parallel (a repeated eigenvalue keeps its independent vectors).

        The proof: SLEPc backs its slicing with inertia counts of a symmetric-indefinite factorisation.  For a REAL SYMMETRIC
        pair the exact LU gives the same counts (``_count_below``: the multifrontal elimination is a block congruence, the
        inertia is read off the pivot blocks): the sweep is complete iff it found ``count(b) - count(a)`` pairs; a deficit --
        a multiple eigenvalue whose copies never showed up in a single-vector Krylov space -- is localised by bisection on
        the counts and closed by solves inside it (``stats["interval_expected"]``, ``stats["interval_complete"]``).  Complex
        Hermitian problems, the sharded layout and the ILU path have no such count: there the sweep says that its
        completeness is heuristic."""
        if self._interval is None:
            raise ValueError("iEpsWhich.ALL needs an interval: call set_interval(a, b) first")
        if self._problem_type not in _HERMITIAN:
            raise ValueError("iEpsWhich.ALL (all eigenvalues in an interval) is defined for Hermitian problem types (HEP, GHEP) only, as in SLEPc")
        a, b = self._interval[0], self._interval[1]
        if not (np.isfinite(a) and np.isfinite(b) and a < b):
            raise ValueError(f"bad interval [{a}, {b}]")
        saved = (self._which, self._st_type, self._target, self._nev, self._ncv)
        n = self._A.shape[0]
        ncv = min(self._ncv if self._ncv is not None else 32, n)
        per_shift = max(1, min(ncv // 2, 16))
        found_lam: list[float] = []
        found_vec: list[np.ndarray] = []
        stats_total: dict = {}
        restarts = 0

        Mh = None if self._M is None else self._M.as_scipy_array()
        added = [0]

        def merge(lam, vecs):
            # A pair is new unless its vector lies in the span of the vectors already stored for (numerically) the same
            # eigenvalue: two shifts return differently rotated bases of a multiple eigenvalue's eigenspace, and a test
            # against single stored vectors would count the same eigenspace twice.  M-inner product for generalised problems.
            for lv, v in zip(lam, vecs.T):
                if not (a <= lv <= b):
                    continue
                same = [v0 for l0, v0 in zip(found_lam, found_vec) if abs(lv - l0) <= 1e-7 * max(1.0, abs(lv))]
                if same:
                    B = np.column_stack(same)
                    MB = B if Mh is None else Mh @ B
                    G = B.conj().T @ MB  # Gram matrix of the stored eigenspace basis
                    c = np.linalg.lstsq(G, MB.conj().T @ v, rcond=None)[0]
                    r = v - B @ c
                    nv = np.sqrt(abs(np.vdot(v, v if Mh is None else Mh @ v)))
                    nr = np.sqrt(abs(np.vdot(r, r if Mh is None else Mh @ r)))
                    if nr <= 0.1 * nv:
                        continue
                found_lam.append(float(lv))
                found_vec.append(v.copy())
                added[0] += 1

        try:
            self._which, self._st_type = iEpsWhich.TARGET_MAGNITUDE, iSTType.SINVERT
            self._nev, self._ncv = per_shift, ncv
            covered = a  # everything in [a, covered] has been found
            span = b - a
            sigma = a + 1e-3 * span * (1.0 + 1e-2)  # not exactly on the end point (an eigenvalue could sit there)
            pending: list[float] = []
            for _ in range(200):
                self._target = complex(sigma)
                if getattr(self, "_prepared", None) is not None:  # the uploaded matrices do not depend on the shift
                    self._prepared["sig"] = self._signature()
                    self._prepared["sigma"] = self._target
                # A Krylov space grown from ONE start vector holds one direction of a multiple eigenvalue's eigenspace: the solve at
                # a shift is repeated with fresh start vectors until a repetition adds no pair (two solves per shift for a simple
                # spectrum, m + 1 where an eigenvalue of multiplicity m lies in reach).
                seed0 = self._seed
                try:
                    for rep in range(8):
                        self._seed = seed0 + 7919 * rep
                        self.solve()
                        restarts += self._restarts
                        for k, v in self._stats.items():
                            if isinstance(v, (int, float)) and k not in ("last_rel_res", "max_rel_res"):
                                stats_total[k] = stats_total.get(k, 0) + v
                        if rep == 0:
                            lam = np.real(self._eigenvalues)  # the cover is judged on the first solve's set
                        added[0] = 0
                        merge(np.real(self._eigenvalues), self._eigenvectors)
                        if rep > 0 and added[0] == 0:
                            break
                finally:
                    self._seed = seed0
                # the m nearest eigenvalues were returned: the ball of the m-th is complete; with fewer than asked for, the
                # spectrum is exhausted on this side of the cover
                dist = np.sort(np.abs(lam - sigma))
                radius = dist[-1] if len(dist) >= per_shift else max(span, dist[-1] if len(dist) else span)
                left, right = sigma - radius, sigma + radius
                if left > covered + 1e-12 * span:  # a gap between the cover and this ball: put a shift into it first
                    pending.append(sigma)
                    sigma = 0.5 * (covered + left)
                    continue
                covered = max(covered, right)
                while pending and pending[-1] <= covered:  # postponed shifts that the cover has reached in the meantime
                    pending.pop()
                if covered >= b:
                    break
                sigma = covered + max(0.5 * radius, 1e-6 * span)
            else:
                raise RuntimeError("interval sweep did not cover [a, b] in 200 shifts")
            # ---- the proof SLEPc's spectrum slicing gives: inertia counts.  For a real symmetric pair the exact LU is a block
            # congruence, so the number of eigenvalues below a shift is the number of negative eigenvalues of A - shift M, read
            # off the pivot blocks of its factorisation (lsa_ndlu_inertia).  The sweep above is complete iff it found
            # count(b) - count(a) pairs; a deficit is localised by bisection on the counts and closed by solves inside it.
            proof = None
            if self._can_count_eigenvalues():
                def in_range(lo, hi):
                    return sum(1 for lv in found_lam if lo < lv <= hi)

                def close(lo, hi, n_lo, n_hi, depth):
                    if in_range(lo, hi) >= n_hi - n_lo or depth > 14 or hi - lo <= 1e-10 * max(1.0, abs(hi)):
                        return
                    mid = 0.5 * (lo + hi)
                    n_mid, mid = self._count_below(mid, 1e-7 * (hi - lo))
                    seed0 = self._seed
                    try:  # a solve in the middle of the deficient interval, fresh start vectors
                        self._target = complex(mid)
                        self._prepared["sig"], self._prepared["sigma"] = self._signature(), self._target
                        for rep in range(4):
                            self._seed = seed0 + 104729 * (depth + 1) + 7919 * rep
                            self.solve()
                            added[0] = 0
                            merge(np.real(self._eigenvalues), self._eigenvectors)
                            if in_range(lo, hi) >= n_hi - n_lo:
                                break
                    finally:
                        self._seed = seed0
                    close(lo, mid, n_lo, n_mid, depth + 1)
                    close(mid, hi, n_mid, n_hi, depth + 1)

                n_a, a_used = self._count_below(a, 1e-9 * span)
                n_b, b_used = self._count_below(b, 1e-9 * span)
                close(a_used, b_used, n_a, n_b, 0)
                proof = {"expected": int(n_b - n_a), "found": in_range(a_used, b_used)}
        finally:
            self._which, self._st_type, self._target, self._nev, self._ncv = saved
            if getattr(self, "_prepared", None) is not None:
                self._prepared["sig"] = None  # the next solve() prepares for its own settings
        if proof is not None and proof["found"] == proof["expected"]:
            logger.info("iEpsWhich.ALL on [%g, %g]: %d eigenvalues, complete: the inertia of A - sigma M at the end points counts %d.", a, b,
                        len(found_lam), proof["expected"])
        elif proof is not None:
            logger.warning("iEpsWhich.ALL on [%g, %g]: %d eigenvalues found, but the inertia counts at the end points say %d: the set is "
                           "%s.", a, b, proof["found"], proof["expected"], "incomplete" if proof["found"] < proof["expected"] else "over-counted")
        else:
            logger.warning("iEpsWhich.ALL on [%g, %g]: %d eigenvalues found by a sweep of shift-invert solves; completeness is heuristic "
                           "(inertia counts need a real symmetric pair and the exact LU on one GPU) -- a multiple eigenvalue can be "
                           "under-counted.", a, b, len(found_lam))
        stats_total["interval_expected"] = -1 if proof is None else proof["expected"]
        stats_total["interval_complete"] = int(proof is not None and proof["found"] == proof["expected"])
        order = np.argsort(found_lam)
        self._eigenvalues = np.array(found_lam, dtype=np.complex128)[order]
        self._eigenvectors = np.asfortranarray(np.column_stack([found_vec[i] for i in order])) if found_lam else np.zeros((n, 0), dtype=np.complex128)
        self._imag_norms = None  # (the merged vectors are not the last solve's: phases are fixed on the host when they are handed out)
        self._residual_estimates = np.zeros(len(found_lam))
        self._restarts = restarts
        self._stats = stats_total

    def _can_count_eigenvalues(self) -> bool:
        """Inertia counts are available for a real symmetric pair factored by the exact LU on one GPU."""
        prep = getattr(self, "_prepared", None)
        if prep is None or prep["part"] is not None or prep["pc_code"] != 2:
            return False
        A = self._A.as_scipy_array()
        M = None if self._M is None else self._M.as_scipy_array()
        return A.dtype.kind != "c" and (M is None or M.dtype.kind != "c")

    def _count_below(self, shift: float, nudge: float) -> tuple[int, float]:
        """Number of eigenvalues of the (definite, real symmetric) pair below ``shift``: the negative eigenvalues of
        ``A - shift M``, from the pivot blocks of its exact LU (``lsa_ndlu_inertia``; the reference reaches the same count
        through SLEPc's spectrum slicing, ``Solver/utils.py:248-254``).  A shift that sits on an eigenvalue (zero pivots) is
        moved by ``nudge`` and counted again; returns the count and the shift used."""
        import lsa_hip

        prep = self._prepared
        ctx, perm = prep["ctx"], prep["perm"]
        A = _permute(sp.csr_matrix(self._A.as_scipy_array()), perm)
        M = sp.identity(A.shape[0], format="csr") if self._M is None else _permute(sp.csr_matrix(self._M.as_scipy_array()), perm)
        for attempt in range(6):
            K = sp.csr_matrix(A - shift * M)
            K.sort_indices()
            dK = lsa_hip.CsrMatrix.from_scipy(ctx, K.astype(np.float64))
            f = lsa_hip.NdLu(ctx, dK, 0)
            neg, zero, _pos = f.inertia()
            del f, dK
            if zero == 0:
                return int(neg), float(shift)
            shift += nudge * (attempt + 1)
        raise RuntimeError(f"inertia count: A - sigma M stays singular around sigma = {shift:g}")

    def residuals(self) -> np.ndarray:
        """Relative residuals ``||A v - lam M v|| / (||A v|| + |lam| ||M v||)`` of the converged pairs, evaluated on the
        device (formula of ``Solver/eigen2.py:48-56``)."""
        import lsa_hip

        self.prepare()
        prep = self._prepared
        if prep["part"] is not None or self._adjoint:  # sharded layout: no rank holds the whole matrix on its device; adjoint: A^H, M^H
            A = self._A.as_scipy_array()
            M = None if self._M is None else self._M.as_scipy_array()
            if self._adjoint:
                A, M = A.conj().T, (None if M is None else M.conj().T)
            lam, V = self._eigenvalues, self._eigenvectors
            Av, Mv = A @ V, (M @ V if M is not None else V)
            num = np.linalg.norm(Av - Mv * lam[np.newaxis, :], axis=0)
            return num / (np.linalg.norm(Av, axis=0) + np.abs(lam) * np.linalg.norm(Mv, axis=0) + 1e-16)
        Xp = self._eigenvectors[prep["perm"], :]
        return lsa_hip.eig_residuals(prep["ctx"], prep["dA"], prep["dM"], self._eigenvalues, Xp)

    # ---- results (same names as the reference) ---------------------------------------------------------------------------
    def get_num_converged(self) -> int:
        return int(self._eigenvalues.shape[0])

    def get_eigenvalue(self, idx: int) -> float | complex:
        lam = complex(self._eigenvalues[idx])
        if self._problem_type in _HERMITIAN:
            return float(lam.real)
        return lam

    def get_eigenvector(self, idx: int) -> iComplexPETScVector:
        """Eigenvector ``idx`` with unit 2-norm; the imaginary part is dropped when its norm is <= 1e-6, as the
        reference's real build does (``Solver/utils.py:280-291``)."""
        v = self._eigenvectors[:, idx]
        imag_norms = getattr(self, "_imag_norms", None)
        if imag_norms is not None:  # unit norm and canonical phase were applied on the device (lsa_krylov_ritz_vectors)
            if imag_norms[idx] <= 1e-6:
                vr = v.real
                return iComplexPETScVector(iPETScVector._adopt(vr / np.linalg.norm(vr)))
            v = v.copy()
            return iComplexPETScVector(iPETScVector._adopt(v.real), iPETScVector._adopt(v.imag))
        # fix the arbitrary complex phase so that a real eigenvector comes out real
        k = int(np.argmax(np.abs(v)))
        v = v * (np.abs(v[k]) / v[k]) if v[k] != 0 else v.copy()
        if np.linalg.norm(v.imag) <= 1e-6:
            return iComplexPETScVector(iPETScVector._adopt(v.real / np.linalg.norm(v.real)))
        # v is this call's own array: its real and imaginary parts are handed over as views (a strided copy of each costs
        # 4 ms per pair at 500 k unknowns)
        return iComplexPETScVector(iPETScVector._adopt(v.real), iPETScVector._adopt(v.imag))

    def get_eigenpair(self, idx: int) -> tuple[float | complex, iComplexPETScVector]:
        return self.get_eigenvalue(idx), self.get_eigenvector(idx)

    def get_all_eigenpairs_up_to(self, num: int) -> Iterator[tuple[float | complex, iComplexPETScVector]]:
        # The norms and scalings of the vectors run on ONE host BLAS thread, for the duration of this call only: a threaded
        # BLAS call leaves its worker pool spinning, and the launch-bound factorisation of the caller's NEXT solve then
        # takes 50 ms longer (tools/micro/after_eigensolve.py: 74 -> 125 ms per solve at 30 k unknowns on a 256-core host).
        from lsa_hip.krylov_schur import _single_threaded_blas

        count = min(self.get_num_converged(), num)
        imag_norms = getattr(self, "_imag_norms", None)
        if imag_norms is not None and not np.any(imag_norms[:count] <= 1e-6):
            # complex vectors that left the device normalised and in canonical phase: handing them out calls no BLAS at all
            # (entering the thread-pool guard alone costs 0.6 ms: it walks the loaded libraries)
            pairs = [self.get_eigenpair(i) for i in range(count)]
        else:
            with _single_threaded_blas():
                pairs = [self.get_eigenpair(i) for i in range(count)]
        yield from pairs

    def get_eigenvector_array(self, idx: int) -> np.ndarray:
        """Complex ndarray of eigenvector ``idx`` (convenience; not in the reference)."""
        return self._eigenvectors[:, idx].copy()

    @property
    def stats(self) -> dict:
        """Counters of the last solve (outer applies, inner iterations, kernel launches, factor/solve seconds)."""
        return dict(self._stats)


def _restart_length(requested: int, n: int, budget_bytes: float = 48e9) -> int:
    """GMRES restart length: as long as requested, but the complex basis n x (restart + 1) must fit the HBM budget.
    Long recurrences are cheap in 288 GB and matter: a shift close to an eigenvalue leaves one tiny eigenvalue in the
    preconditioned operator, which full GMRES resolves in a few extra steps and restarted GMRES never does."""
    return int(max(1, min(requested, max(n, 1), budget_bytes // (16 * max(n, 1)) - 1)))


def _dist_rank_world() -> tuple[int, int]:
    """(rank, world size) of the torch.distributed job this process belongs to; (0, 1) outside one."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


_host_group = None


def _dist_comm_init(ctx, world: int, rank: int) -> None:
    """Bootstrap the all-gather of the sharded layout from the torch.distributed job: RCCL over xGMI when the job runs
    the nccl backend (one GPU per rank), the host-staged transport over gloo otherwise (CPU-only process groups, several
    ranks sharing one GPU in tests and rehearsals; ``LSA_COMM_TRANSPORT=host`` forces it)."""
    import os

    import torch.distributed as dist

    import torch

    if dist.get_backend() == "nccl" and os.environ.get("LSA_COMM_TRANSPORT", "rccl") != "host":
        # RCCL is opened by the library itself (dlopen); if that fails on ANY rank, every rank falls back to the host
        # transport together (a collective decision: ranks on different transports would dead-lock)
        # (1) every rank checks that it can open RCCL at all -- asking for a unique id does that -- and the ranks agree on
        # the answer BEFORE anyone enters ncclCommInitRank, which would wait for ever for a rank that never calls it
        ok = 1
        try:
            uid = ctx.unique_id()
        except Exception as exc:  # noqa: BLE001
            logger.warning("RCCL unavailable on rank %d (%s)", rank, exc)
            uid, ok = None, 0
        probe = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(probe, op=dist.ReduceOp.MIN)
        ok = int(probe.item())
        uid = _dist_broadcast_bytes(uid if (rank == 0 and ok) else None)  # (2) rank 0's id for everybody
        if uid is not None and ok:
            try:
                ctx.comm_init(world, rank, uid)
            except Exception as exc:  # noqa: BLE001
                logger.warning("RCCL communicator initialisation failed on rank %d (%s)", rank, exc)
                ok = 0
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return
        logger.warning("Falling back to the host-staged all-gather (gloo) on every rank.")

    global _host_group
    group = None
    if dist.get_backend() != "gloo":
        if _host_group is None:
            _host_group = dist.new_group(backend="gloo")
        group = _host_group

    def exchange(blocks: np.ndarray) -> None:  # (world, bytes) uint8 view of the library's staging buffer
        t = torch.from_numpy(blocks)
        dist.all_gather([t[r] for r in range(world)], t[rank].clone(), group=group)

    ctx.comm_init_host(world, rank, exchange)


def _dist_broadcast_bytes(payload: bytes | None) -> bytes:
    """Broadcast rank 0's bytes (the 128-byte RCCL unique id) to every rank of the torch.distributed job."""
    import torch.distributed as dist

    box = [payload]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _onto_pattern(A: sp.csr_matrix, pattern: sp.csr_matrix) -> sp.csr_matrix:
    """A's values scattered onto ``pattern`` (a superset of A's pattern); missing entries become explicit zeros."""
    n = A.shape[0]
    key_new = np.repeat(np.arange(n, dtype=np.int64), np.diff(pattern.indptr)) * A.shape[1] + pattern.indices
    key_old = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr)) * A.shape[1] + A.indices
    data = np.zeros(pattern.nnz, dtype=A.dtype)
    data[np.searchsorted(key_new, key_old)] = A.data
    return sp.csr_matrix((data, pattern.indices.copy(), pattern.indptr.copy()), shape=A.shape)


def _combine(A: sp.csr_matrix, M: sp.csr_matrix, sigma: complex) -> sp.csr_matrix:
    """A - sigma M on the shared pattern (host copy, used only to look for zero diagonals)."""
    return sp.csr_matrix((A.data - sigma * M.data, A.indices, A.indptr), shape=A.shape)


def _permute(A: sp.csr_matrix, perm: np.ndarray) -> sp.csr_matrix:
    if np.array_equal(perm, np.arange(A.shape[0])):
        out = sp.csr_matrix(A)
    else:
        out = A[perm][:, perm].tocsr()
    out.sort_indices()
    return out
