"""GPU tests of the CROSS-CHECK library (tests/xcheck: round 1's exact block-tridiagonal LU, an independent direct solver on the
device) against SuperLU, of the product's exact LU against it, and of PreconditionerType.LU through the drop-in surface against
the eigen oracle."""

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402,F401  (sys.path)
import xcheck  # noqa: E402  (tests/xcheck: liblsa_xcheck.so)


def _ordered(case, sigma):
    from synthetic import fem
    from Solver.utils import pivot_safe_rcm

    es = fem.cylinder_case(case)
    C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    perm = pivot_safe_rcm(C)
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    return es, Cp


# block sizes 1536, 2560 and 3584 exercise the other panel-kernel instances (4 rows per thread; 12 and 16 rows per thread
# with the split row-tiled update on a side stream); the default and 256 / 512 use the thread-per-row instance
@pytest.mark.parametrize("case,sigma,block", [("S2k", 0.018 + 0.7379601143282424j, 256), ("S5k", 0.018 + 0.7379601143282424j, 0),
                                              ("S5k", 0.05, 512), ("S5k", 0.018 + 0.7379601143282424j, 1536),
                                              ("S5k", 0.018 + 0.7379601143282424j, 2560), ("S5k", 0.018 + 0.7379601143282424j, 3584)])
def test_block_lu_is_a_direct_solver(hip_ctx, case, sigma, block):
    import lsa_hip

    es, Cp = _ordered(case, sigma)
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp)
    f = xcheck.BlockLu(hip_ctx, dC, block)
    info = f.info()
    assert info["block_size"] > info["bandwidth"] and info["block_size"] % 256 == 0
    assert info["nblocks"] == -(-es.n // info["block_size"])
    rng = np.random.default_rng(3)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    dx = lsa_hip.DeviceVector(hip_ctx, es.n, np.complex128)
    f.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
    x = dx.numpy()
    assert np.linalg.norm(Cp @ x - b) <= 1e-12 * np.linalg.norm(b)
    xref = spla.splu(sp.csc_matrix(Cp.astype(np.complex128))).solve(b)
    assert np.linalg.norm(x - xref) <= 1e-11 * np.linalg.norm(xref)
    f.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)  # graph replay gives the same answer
    assert np.array_equal(dx.numpy(), x)


@pytest.mark.parametrize("block", [256, 1536, 2560, 3584])
def test_block_lu_real_matrix(hip_ctx, block):
    """float64 instantiation: a real, non-symmetric, banded matrix that needs row interchanges (weak diagonal)."""
    import lsa_hip

    n, bw = 3000, 40
    rng = np.random.default_rng(11)
    offs = [-bw, -7, -1, 0, 1, 5, bw]
    A = sp.diags([rng.standard_normal(n - abs(o)) for o in offs], offs, format="csr")
    A = A + sp.diags(0.05 * np.ones(n), 0)
    A = sp.csr_matrix(A)
    A.sort_indices()
    f = xcheck.BlockLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), block)
    b = rng.standard_normal(n)
    dx = lsa_hip.DeviceVector(hip_ctx, n, np.float64)
    f.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
    xref = spla.splu(sp.csc_matrix(A)).solve(b)
    assert np.linalg.norm(dx.numpy() - xref) <= 1e-9 * np.linalg.norm(xref)


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 17, 255, 256, 257, 300, 519])
def test_block_lu_ragged_sizes(hip_ctx, n):
    """Blocks whose size is not a multiple of the panel width, a last block of a few rows, a single row."""
    import lsa_hip

    rng = np.random.default_rng(100 + n)
    offs = [o for o in (-3, -1, 0, 1, 2) if abs(o) < n]
    A = sp.csr_matrix(sp.diags([rng.standard_normal(n - abs(o)) + 1j * rng.standard_normal(n - abs(o)) for o in offs], offs, format="csr"))
    A.sort_indices()
    f = xcheck.BlockLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), 256)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    dx = lsa_hip.DeviceVector(hip_ctx, n, np.complex128)
    f.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
    xref = np.linalg.solve(A.toarray(), b)
    assert np.linalg.norm(dx.numpy() - xref) <= 1e-9 * np.linalg.norm(xref)


@pytest.mark.parametrize("block", [512, 1536])
def test_block_lu_absorbed_and_sparse_sweeps_agree(hip_ctx, monkeypatch, block):
    """The solve on the absorbed couplings (one dense launch per step; default for blocks of <= 1024 rows) and the one on
    Sinv + the sparse rows of C (LSA_BLU_ABSORB=0) are the same direct solve."""
    import lsa_hip

    es, Cp = _ordered("S5k", 0.018 + 0.7379601143282424j)
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp)
    rng = np.random.default_rng(9)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    xs, launches = [], []
    for absorb in ("1", "0"):
        monkeypatch.setenv("LSA_BLU_ABSORB", absorb)
        f = xcheck.BlockLu(hip_ctx, dC, block)
        dx = lsa_hip.DeviceVector(hip_ctx, es.n, np.complex128)
        f.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
        xs.append(dx.numpy())
        launches.append(f.info()["apply_launches"])
        del f
    assert np.linalg.norm(Cp @ xs[0] - b) <= 1e-12 * np.linalg.norm(b)
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-11 * np.linalg.norm(xs[1])
    assert launches[1] == 2 * launches[0]


def test_block_lu_unblocked_elimination_agrees(hip_ctx, monkeypatch):
    """LSA_GJ_PANEL=1 selects the two-launches-per-pivot Gauss-Jordan (what blocks of more than 4096 rows get)."""
    import lsa_hip

    es, Cp = _ordered("S2k", 0.018 + 0.7379601143282424j)
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp)
    b = np.random.default_rng(5).standard_normal(es.n) + 0j
    xs = []
    for panel in ("8", "1"):
        monkeypatch.setenv("LSA_GJ_PANEL", panel)
        dx = lsa_hip.DeviceVector(hip_ctx, es.n, np.complex128)
        xcheck.BlockLu(hip_ctx, dC, 512).solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
        xs.append(dx.numpy())
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-11 * np.linalg.norm(xs[1])


def test_block_lu_reports_singular_block(hip_ctx):
    import lsa_hip

    n = 300
    A = sp.diags([np.ones(n - 1), 2.0 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    A.data[(rows == 10) | (A.indices == 10)] = 0.0  # row and column 10 vanish; the entries stay in the pattern
    with pytest.raises(lsa_hip.LsaError) as ei:
        xcheck.BlockLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), 256)
    assert ei.value.status == -3


def test_eigensolver_with_lu_preconditioner_matches_oracle():
    """PreconditionerType.LU (the reference's own setting, .examples/eigenvalues.py:100): exact inner solves, one
    preconditioner apply per Arnoldi step, eigenvalues within 1e-8 of the oracle."""
    from oracle import shift_invert
    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    es = fem.cylinder_case("S5k")
    sigma = fem.SIGMA_RE50
    ref, _, _ = shift_invert.solve(es.A, es.M, sigma, k=20, tol=1e-13, ncv=80)
    solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80), check_hermitian=False)
    solver.solver.set_st_type(iSTType.SINVERT)
    solver.solver.set_target(sigma)
    solver.solver.set_st_pc_type(PreconditionerType.LU)
    pairs = solver.solve()
    lam = np.array([p[0] for p in pairs])
    assert len(lam) == 20
    for r in ref:
        assert np.min(np.abs(lam - r)) <= 1e-8 * abs(r)
    st = solver.solver.stats
    assert st["gmres_iters"] == 0  # every inner solve was direct (verified against b - C x)
    assert st["max_rel_res"] <= 1e-12
    assert solver.solver.residuals().max() <= 1e-8


def test_lu_request_falls_back_to_ilu_when_the_factors_do_not_fit(monkeypatch):
    """PreconditionerType.LU when the exact factors do not fit the device memory (forced here: LSA_ND_TEST_OOM makes the set-up of the
    nested-dissection LU report LSA_ERR_OOM -- the only failure that is answered by a leaner method): the operator is built on
    ILU(2) + GMRES instead, says so (``stats["pc_fallback"]``, a warning) and the eigenvalues still match the oracle."""
    from oracle import shift_invert
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    n = 200_000
    main = 2.0 + np.linspace(0.0, 1.0, n)
    A = sp.diags([-np.ones(n - 1), main, -np.ones(n - 1)], [-1, 0, 1], format="lil")
    A[0, n - 1] = A[n - 1, 0] = -1.0
    A = sp.csr_matrix(A)
    sigma = 1.7
    ref, _, _ = shift_invert.solve(A, None, sigma, k=3, tol=1e-12, ncv=30)
    for forced in (True, False):
        if forced:
            monkeypatch.setenv("LSA_ND_TEST_OOM", "1")
        else:
            monkeypatch.delenv("LSA_ND_TEST_OOM")
        solver = EigenSolver(A, None, EigensolverConfig(num_eig=3, atol=1e-10, ncv=30), check_hermitian=False, ordering="natural")
        solver.solver.set_st_type(iSTType.SINVERT)
        solver.solver.set_target(sigma)
        solver.solver.set_st_pc_type(PreconditionerType.LU)
        lam = np.array([p[0] for p in solver.solve()])
        for r in ref:
            assert np.min(np.abs(lam - r)) <= 1e-8 * abs(r)
        st = solver.solver.stats
        if forced:  # inner solves were iterative: the exact LU was not available
            assert st["pc_fallback"] == 1 and st["gmres_iters"] > 0
        else:
            assert st["pc_fallback"] == 0 and st["gmres_iters"] == 0
        solver.solver.release()
