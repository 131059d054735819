"""CPU suite, part 3: host-side logic of the product (no GPU): Krylov-Schur driver, ordering, drop-in API surface,
boundary shims and MatrixMarket I/O.  The Krylov-Schur driver runs here on a numpy backend (tests/helpers.py) with the
oracle's SuperLU shift-invert as the operator; on the GPU the same driver runs on ``lsa_hip.KrylovBasis``.
"""

import logging

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import helpers
from oracle import shift_invert
from synthetic import fem


# ---- Krylov-Schur ----------------------------------------------------------------------------------------------------------


@pytest.fixture(scope="module")
def s2k():
    return fem.cylinder_case("S2k")


def test_krylov_schur_matches_arpack_oracle(s2k):
    from lsa_hip.krylov_schur import krylov_schur

    sigma = fem.SIGMA_RE50
    lu = spla.splu((s2k.A - sigma * s2k.M).tocsc())
    be = helpers.NumpyKrylovBackend(lambda x: lu.solve(s2k.M @ x), s2k.n, 40)
    res = krylov_schur(be, 8, 1e-11, 200, lambda th: -np.abs(th))
    assert res.nconv >= 8 and res.restarts >= 1  # restarted at least once: exercises the truncation
    lam = sigma + 1.0 / res.theta
    ref, _, _ = shift_invert.solve(s2k.A, s2k.M, sigma, k=8, tol=1e-13, ncv=40)
    assert helpers.match_nearest(lam, ref).max() < 1e-9
    assert np.all(np.diff(np.abs(lam - sigma)) > -1e-9)  # wanted first
    assert np.allclose(np.linalg.norm(res.vectors, axis=0), 1.0, atol=1e-12)
    r = shift_invert.compute_residuals(s2k.A, s2k.M, lam, res.vectors)
    assert r.max() < 1e-9
    assert res.op_applies == be.applies


def test_krylov_schur_small_dense_cases():
    """n <= ncv: exact breakdown; repeated eigenvalues need the fresh-direction continuation (test_eigen.py:284-304)."""
    from lsa_hip.krylov_schur import krylov_schur

    D = np.diag([2.0, 2.0, 3.0]).astype(complex)
    be = helpers.NumpyKrylovBackend(lambda x: D @ x, 3, 3)
    res = krylov_schur(be, 3, 1e-8, 50, lambda th: -np.abs(th))
    assert sorted(np.round(res.theta.real, 8)) == [2.0, 2.0, 3.0]
    assert np.linalg.matrix_rank(res.vectors) == 3
    J = np.array([[1, 1], [0, 1]], dtype=complex)
    be = helpers.NumpyKrylovBackend(lambda x: J @ x, 2, 2)
    res = krylov_schur(be, 2, 1e-6, 50, lambda th: -np.abs(th))
    assert res.theta.real == pytest.approx([1.0, 1.0], abs=1e-6)


def test_krylov_schur_reports_unconverged():
    from lsa_hip.krylov_schur import krylov_schur

    rng = np.random.default_rng(0)
    A = rng.standard_normal((200, 200))
    be = helpers.NumpyKrylovBackend(lambda x: A @ x, 200, 12)
    res = krylov_schur(be, 6, 1e-14, 1, lambda th: np.abs(th))  # smallest magnitude without inversion: hopeless in 1 restart
    assert res.nconv < 6 and len(res.theta) == res.nconv


# ---- ordering ----------------------------------------------------------------------------------------------------------------


def test_pivot_safe_rcm(s2k):
    from oracle import kernels
    from Solver.utils import pivot_safe_rcm

    C = sp.csr_matrix((s2k.A.data - fem.SIGMA_RE50 * s2k.M.data, s2k.A.indices, s2k.A.indptr), shape=s2k.A.shape)
    perm = pivot_safe_rcm(C)
    assert sorted(perm) == list(range(s2k.n))
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    rows = np.repeat(np.arange(s2k.n), np.diff(Cp.indptr))
    bw = np.max(np.abs(rows - Cp.indices))
    rows0 = np.repeat(np.arange(s2k.n), np.diff(C.indptr))
    # the structured assembly order is already banded; RCM must stay in the same class (and beats a random order by far)
    assert bw <= 1.5 * np.max(np.abs(rows0 - C.indices))
    # every zero-diagonal row has a coupled row before it, so ILU(0) has no zero pivot without any shift
    ilu = kernels.ILU0(Cp, 0.0)
    assert ilu.nshift == 0 and np.all(np.abs(ilu.v[ilu.diag]) > 1e-8)
    # ... which natural (assembly) order does not give
    with pytest.raises(ZeroDivisionError):
        kernels.ILU0(C, 0.0)


# ---- drop-in API surface (no solve) ----------------------------------------------------------------------------------------------


def test_enum_surface_matches_reference():
    from Solver.utils import KSPType, PreconditionerType, iEpsProblemType, iEpsWhich, iSTType

    assert [m.name for m in PreconditionerType] == ["NONE", "JACOBI", "SOR", "ASM", "ILU", "ICC", "LU", "CHOLESKY", "GAMG", "HYPRE", "REDUNDANT", "SHELL"]
    assert PreconditionerType.ILU == "ilu" and KSPType.FGMRES.to_petsc() == "fgmres"
    assert iEpsWhich.SMALLEST_MAGNITUDE is iEpsWhich.LARGEST_REAL  # the alias of Solver/utils.py:157
    assert iEpsWhich.LARGEST_REAL.to_arpack() == "LR" and iEpsWhich.LARGEST_MAGNITUDE.to_arpack() == "LM_abs"
    with pytest.raises(ValueError):
        iEpsWhich.TARGET_REAL.to_arpack()
    assert iEpsProblemType.from_string("gnhep") is iEpsProblemType.GNHEP
    with pytest.raises(ValueError):
        iEpsProblemType.from_string("nope")
    assert {t.name for t in iSTType} == {"SHELL", "SHIFT", "SINVERT", "CAYLEY", "PRECOND", "FILTER"}


def test_config_propagates_into_raw_getters():
    """tests/unit/Solver/test_eigen.py:87-104."""
    from FEM.utils import iPETScMatrix
    from Solver.eigen import EigenSolver, EigensolverConfig, iEpsProblemType
    from Solver.utils import iEpsSolver

    A = iPETScMatrix.from_matrix(np.diag([1.0, 1.5, -42.0]))
    cfg = EigensolverConfig(num_eig=3, problem_type=iEpsProblemType.GHEP, atol=1e-3, max_it=100)
    es = EigenSolver(cfg, A=A)  # legacy order
    assert es.config is cfg and isinstance(es.solver, iEpsSolver)
    assert es.solver.raw.getTolerances() == (cfg.atol, cfg.max_it)
    assert es.solver.raw.getDimensions()[0] == cfg.num_eig
    assert es.solver.raw.getProblemType() == cfg.problem_type.to_slepc()
    es2 = EigenSolver(A, None, cfg)  # current order
    assert es2.config is cfg
    assert EigensolverConfig() == EigensolverConfig(num_eig=5, problem_type=iEpsProblemType.GNHEP, atol=1e-6, max_it=500, ncv=80)


def test_constructor_errors():
    """Solver/eigen.py:78-87 and tests/unit/Solver/test_eigen.py:81-84."""
    from FEM.utils import iPETScMatrix
    from Solver.eigen import EigenSolver
    from Solver.utils import iEpsSolver

    with pytest.raises(ValueError):
        iEpsSolver(M=iPETScMatrix.from_matrix(np.eye(3)))
    with pytest.raises(ValueError, match="must be square"):
        EigenSolver(iPETScMatrix(sp.csr_matrix(np.ones((2, 3)))))
    with pytest.raises(ValueError, match="does not match"):
        EigenSolver(iPETScMatrix.from_matrix(np.eye(3)), iPETScMatrix.from_matrix(np.eye(4)))


@pytest.mark.parametrize("pc_name", ["NONE", "JACOBI", "SOR", "ASM", "ILU", "ICC", "LU", "CHOLESKY", "GAMG", "HYPRE", "REDUNDANT", "SHELL"])
def test_set_st_pc_type_round_trip(pc_name):
    """tests/unit/Solver/test_eigen.py:307-322."""
    from FEM.utils import iPETScMatrix
    from Solver.utils import PreconditionerType, iEpsProblemType, iEpsSolver, iSTType

    solver = iEpsSolver(A=iPETScMatrix.from_matrix(np.diag([1.0, 1.5, -42.0])))
    solver.set_problem_type(iEpsProblemType.HEP)
    solver.set_dimensions(number_eigenpairs=3)
    solver.set_tolerances(atol=1e-8, max_it=50)
    solver.set_st_type(iSTType.SINVERT)
    solver.set_target(2.0)
    solver.set_st_pc_type(PreconditionerType[pc_name])
    assert solver.raw.getST().getKSP().getPC().getType() == pc_name.lower()


def test_hermitian_warning(caplog):
    """tests/unit/Solver/test_eigen.py:188-200."""
    from FEM.utils import iPETScMatrix
    from Solver.eigen import EigenSolver, EigensolverConfig, iEpsProblemType

    A = iPETScMatrix.from_matrix(np.diag([1.0, 1.5, -42.0]))
    A[0, 1] = 0.1
    A.assemble()
    caplog.set_level(logging.WARNING)
    EigenSolver(EigensolverConfig(num_eig=3, problem_type=iEpsProblemType.GHEP), A=A)
    assert any("assumes Hermitian A" in r.getMessage() for r in caplog.records)
    caplog.clear()
    EigenSolver(A, None, EigensolverConfig(num_eig=3, problem_type=iEpsProblemType.GHEP), check_hermitian=False)
    assert not caplog.records


# ---- boundary shims ------------------------------------------------------------------------------------------------------------------


def test_matrix_shim_and_matrix_market_round_trip(tmp_path, s2k):
    from FEM.utils import iPETScMatrix

    A = iPETScMatrix(s2k.A)
    assert A.shape == s2k.A.shape and A.nonzero_entries == s2k.A.nnz
    assert A.norm == pytest.approx(np.linalg.norm(s2k.A.data))
    path = tmp_path / "A.mtx"
    A.export(path)
    B = iPETScMatrix.from_path(path)
    a, b = A.as_scipy_array(), B.as_scipy_array()
    assert b.nnz == a.nnz  # explicit zeros survive the stage boundary
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
    # agrees with scipy's reader, including a symmetric file
    import scipy.io

    assert abs(scipy.io.mmread(str(path)).tocsr() - b).max() == 0
    S = sp.csr_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 3.0, 4.0], [0.0, 4.0, 5.0]]))
    scipy.io.mmwrite(str(tmp_path / "S.mtx"), S, symmetry="symmetric")
    assert abs(iPETScMatrix.from_path(tmp_path / "S.mtx").as_scipy_array() - S).max() == 0
    assert iPETScMatrix(S).is_numerically_hermitian() and not A.is_numerically_hermitian()
    assert abs(A.T.as_scipy_array() - s2k.A.T).max() == 0
    Z = (s2k.A.astype(complex) * (1 + 2j)).tocsr()
    iPETScMatrix(Z).export(tmp_path / "Z.mtx")
    assert abs(iPETScMatrix.from_path(tmp_path / "Z.mtx").as_scipy_array() - Z).max() < 1e-15
    # the native reader and the numpy reader agree entry for entry, hermitian / skew / pattern files included
    import lsa_hip
    from FEM import mmio

    H = sp.csr_matrix(np.array([[2.0, 1 + 1j, 0], [1 - 1j, 3.0, 4j], [0, -4j, 5.0]]))
    scipy.io.mmwrite(str(tmp_path / "H.mtx"), H, symmetry="hermitian")
    K = sp.csr_matrix(np.array([[0.0, 2.0, -1.0], [-2.0, 0.0, 3.0], [1.0, -3.0, 0.0]]))
    scipy.io.mmwrite(str(tmp_path / "K.mtx"), K, symmetry="skew-symmetric")
    (tmp_path / "P.mtx").write_text("%%MatrixMarket matrix coordinate pattern general\n% comment\n3 3 3\n1 1\n2 3\n3 1\n")
    for name, want in (("A", a), ("S", S), ("Z", Z), ("H", H), ("K", K), ("P", sp.csr_matrix(([1.0, 1.0, 1.0], ([0, 1, 2], [0, 2, 0])), shape=(3, 3)))):
        nat = lsa_hip.read_matrix_market(tmp_path / f"{name}.mtx")
        ref = mmio._read_matrix_market_numpy(tmp_path / f"{name}.mtx")
        assert nat.shape == ref.shape and np.array_equal(nat.indptr, ref.indptr) and np.array_equal(nat.indices, ref.indices)
        assert np.array_equal(nat.data, ref.data) and abs(nat - want).max() < 1e-15
    (tmp_path / "bad.mtx").write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(ValueError):
        lsa_hip.read_matrix_market(tmp_path / "bad.mtx")


def test_complex_vector_shim():
    from FEM.utils import iComplexPETScVector, iPETScVector

    a = iComplexPETScVector.from_array(np.array([1 + 1j, 2.0, -1j]))
    b = iComplexPETScVector(iPETScVector(np.array([1.0, 0.0, 2.0])))
    assert b.imag is None and a.imag is not None
    assert a.norm() == pytest.approx(np.sqrt(2 + 4 + 1))
    assert a.dot(b) == pytest.approx(np.vdot([1 + 1j, 2.0, -1j], [1.0, 0.0, 2.0]))  # conjugates self (FEM/utils.py:1194-1212)
    a.scale(1j)
    assert np.allclose(a.as_array(), 1j * np.array([1 + 1j, 2.0, -1j]))
    c = a.copy()
    c.scale(0.0)
    assert a.norm() > 0


def test_native_matrix_market_parser_on_awkward_and_malformed_files(tmp_path):
    """The native reader (host code of liblsa_hip.so, the file boundary of the path): duplicates are summed, CRLF and
    blank/comment lines are accepted, an empty matrix is fine; truncated, out-of-range or non-numeric data is a
    ``ValueError``, not a crash or a silently short matrix."""
    import lsa_hip

    def parse(text, name="m.mtx"):
        p = tmp_path / name
        p.write_bytes(text.encode())
        return lsa_hip.read_matrix_market(p)

    m = parse("%%MatrixMarket matrix coordinate real general\r\n% made on another OS\r\n\r\n3 3 4\r\n1 1 1.5\r\n3 2 -2\r\n1 1 0.5\r\n2 2 0\r\n")
    assert m.shape == (3, 3) and m.nnz == 3  # (1,1) twice -> summed; the explicit zero at (2,2) is kept
    assert m[0, 0] == 2.0 and m[2, 1] == -2.0 and m.indptr.tolist() == [0, 1, 2, 3]
    e = parse("%%MatrixMarket matrix coordinate complex general\n4 5 0\n")
    assert e.shape == (4, 5) and e.nnz == 0 and np.iscomplexobj(e.data)
    i = parse("%%MatrixMarket matrix coordinate integer symmetric\n2 2 2\n2 1 7\n2 2 3\n")
    assert i.toarray().tolist() == [[0.0, 7.0], [7.0, 3.0]]
    for bad in (
        "%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.0\n2 2 2.0\n",  # fewer entries than announced
        "%%MatrixMarket matrix coordinate real general\n3 3 1\n4 1 1.0\n",  # row index out of range
        "%%MatrixMarket matrix coordinate real general\n3 3 1\n1 0 1.0\n",  # column index 0 (1-based format)
        "%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 abc\n",  # not a number
        "%%MatrixMarket matrix coordinate real general\n3 x 1\n1 1 1.0\n",  # bad size line
        "%%MatrixMarket matrix coordinate real unknown-symmetry\n1 1 1\n1 1 1.0\n",
        "%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 1.0\n",  # bad banner
        "",
    ):
        with pytest.raises(ValueError):
            parse(bad, "bad.mtx")
    with pytest.raises(ValueError):
        lsa_hip.read_matrix_market(tmp_path / "does_not_exist.mtx")
