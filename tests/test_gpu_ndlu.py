"""GPU parity tests of the nested-dissection multifrontal LU (PreconditionerType.LU on the device) against SuperLU and
against the numpy walk of the same analysis tables (tests/nd_emulation.py)."""

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu

SIGMA = 0.018 + 0.7379601143282424j


def _shifted(case, sigma):
    from synthetic import fem

    es = fem.cylinder_case(case)
    return es, sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)


def _solve(hip_ctx, f, b):
    import lsa_hip

    dx = lsa_hip.DeviceVector(hip_ctx, len(b), b.dtype)
    f.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
    return dx.numpy()


@pytest.mark.parametrize("case,sigma,leaf", [("S2k", SIGMA, 32), ("S2k", SIGMA, 0), ("S5k", SIGMA, 0), ("S5k", 0.05, 64), ("S30k", SIGMA, 0),
                                             ("S30k", SIGMA, 300)])
def test_ndlu_is_a_direct_solver(hip_ctx, case, sigma, leaf):
    """Saddle-point matrices in their assembly order (the dissection is internal); complex and real shifts; leaf sizes
    that exercise the 64-, 128-, 256- and 512-thread panel instances."""
    import lsa_hip

    es, C = _shifted(case, sigma)
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, C)
    f = lsa_hip.NdLu(hip_ctx, dC, leaf)
    info = f.info()
    assert info["tree_nodes"] >= 1 and info["apply_launches"] <= 2 * info["levels"]
    assert info["factor_entries"] < 0.25 * es.n * es.n  # sparse: far from the n^2 of a dense inverse
    rng = np.random.default_rng(3)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    x = _solve(hip_ctx, f, b)
    assert np.linalg.norm(C @ x - b) <= 1e-12 * np.linalg.norm(b)
    xref = spla.splu(sp.csc_matrix(C.astype(np.complex128))).solve(b)
    assert np.linalg.norm(x - xref) <= 1e-10 * np.linalg.norm(xref)
    assert np.array_equal(_solve(hip_ctx, f, b), x)  # fixed summation order: bitwise repeatable


def test_ndlu_matches_the_walk_of_its_tables(hip_ctx):
    """Same analysis, same data flow, LAPACK pivot blocks: the device result agrees to rounding."""
    import lsa_hip
    from nd_emulation import Emulated

    es, C = _shifted("S5k", SIGMA)
    em = Emulated(lsa_hip.NdAnalysis(C, 0).export_tables(), C.data)
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), 0)
    rng = np.random.default_rng(5)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    x = _solve(hip_ctx, f, b)
    xe = em.solve(b)
    assert np.linalg.norm(x - xe) <= 1e-11 * np.linalg.norm(xe)


def test_ndlu_real_factors_real_and_complex_vectors(hip_ctx):
    """float64 instantiation; a complex right-hand side against real factors (the M-solve of iSTType.SHIFT)."""
    import lsa_hip

    es, C = _shifted("S5k", 0.05)
    C = sp.csr_matrix(C.real)
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), 0)
    lu = spla.splu(sp.csc_matrix(C))
    rng = np.random.default_rng(7)
    br = rng.standard_normal(es.n)
    xr = _solve(hip_ctx, f, br)
    assert xr.dtype == np.float64 and np.linalg.norm(xr - lu.solve(br)) <= 1e-10 * np.linalg.norm(xr)
    bc = br + 1j * rng.standard_normal(es.n)
    xc = _solve(hip_ctx, f, bc)
    assert np.linalg.norm(xc - (lu.solve(bc.real) + 1j * lu.solve(bc.imag))) <= 1e-10 * np.linalg.norm(xc)


@pytest.mark.parametrize("case,sigma,leaf", [("S5k", SIGMA, 0), ("S5k", 0.05, 48), ("S30k", SIGMA, 0)])
def test_ndlu_transposed_sweeps_solve_the_adjoint_systems(hip_ctx, case, sigma, leaf):
    """C^-T b and C^-H b from the factors of C (no second factorisation, no transposed matrix)."""
    import lsa_hip

    es, C = _shifted(case, sigma)
    if np.isrealobj(sigma):
        C = sp.csr_matrix(C.real)
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), leaf)
    rng = np.random.default_rng(13)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    db = lsa_hip.DeviceVector.from_numpy(hip_ctx, b)
    dx = lsa_hip.DeviceVector(hip_ctx, es.n, np.complex128)
    for conj in (True, False):
        f.solve_adjoint(db, dx, conj=conj)
        x = dx.numpy()
        Ct = C.conj().T if conj else C.T
        assert np.linalg.norm(Ct @ x - b) <= 1e-12 * np.linalg.norm(b)
    f.solve(db, dx)  # the plain solve is untouched by the adjoint sweeps
    assert np.linalg.norm(C @ dx.numpy() - b) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("n", [1, 2, 3, 9, 65, 130, 519])
def test_ndlu_ragged_sizes_and_general_patterns(hip_ctx, n):
    """Tiny and odd sizes, structurally unsymmetric random patterns, weak diagonals (row pivoting inside the blocks)."""
    import lsa_hip

    rng = np.random.default_rng(n)
    A = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="csr") + sp.diags(0.05 + rng.random(n), 0)
    A = sp.csr_matrix(A)
    A.sort_indices()
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), 16)
    b = rng.standard_normal(n)
    x = _solve(hip_ctx, f, b)
    assert np.linalg.norm(A @ x - b) <= 1e-10 * max(np.linalg.norm(b), np.linalg.norm(A @ x))


def test_ndlu_zero_diagonal_needs_pivoting(hip_ctx):
    """All-zero diagonal, strong coupling inside pairs of unknowns with one shared adjacency (the situation of a mesh
    node's velocity / pressure unknowns): solvable only through row pivoting inside the pivot blocks.  Pivots are not
    moved across fronts, so the pairs must stay together -- they do, having the same neighbours."""
    import lsa_hip

    npair = 300
    rng = np.random.default_rng(2)
    S = sp.random(npair, npair, density=0.01, random_state=rng) + sp.diags([np.ones(npair - 1), np.ones(npair - 1)], [-1, 1])
    S = sp.csr_matrix(S + S.T)
    S.setdiag(0.0)
    S.eliminate_zeros()
    weak = sp.kron(S, np.ones((2, 2))).tocsr()
    weak.data[:] = 0.1 * rng.standard_normal(weak.nnz)
    A = sp.csr_matrix(weak + sp.kron(sp.identity(npair), np.array([[0.0, 2.0], [2.0, 0.0]])))
    A = sp.csr_matrix(A + sp.kron(sp.identity(npair), 1e-300 * np.eye(2)))  # diagonal stored, numerically zero
    A.sort_indices()
    assert np.abs(A.diagonal()).max() < 1e-200
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), 64)
    b = rng.standard_normal(2 * npair)
    x = _solve(hip_ctx, f, b)
    assert np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b)


def test_ndlu_singular_matrix_is_an_error(hip_ctx):
    import lsa_hip

    n = 300
    A = sp.csr_matrix(sp.diags([np.ones(n - 1), 2.0 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]))
    A.data[A.indptr[17]:A.indptr[18]] = 0.0  # a row of stored zeros
    with pytest.raises(lsa_hip.LsaError) as exc:
        lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), 0)
    assert exc.value.status == -3  # LSA_ERR_ZERO_PIVOT


def test_ndlu_refactor_and_cached_analysis(hip_ctx):
    """A shift sweep: new values on the same pattern (explicit refactor, and create-after-destroy through the context's
    cache) give the factorisation of the new matrix."""
    import lsa_hip

    es, C0 = _shifted("S5k", SIGMA)
    _, C1 = _shifted("S5k", 0.05 + 0.744243299635422j)
    rng = np.random.default_rng(9)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    dC0 = lsa_hip.CsrMatrix.from_scipy(hip_ctx, C0)
    dC1 = lsa_hip.CsrMatrix.from_scipy(hip_ctx, C1)
    f = lsa_hip.NdLu(hip_ctx, dC0, 0)
    first = f.info()
    x0 = _solve(hip_ctx, f, b)
    f.refactor(dC1)
    x1 = _solve(hip_ctx, f, b)
    assert np.linalg.norm(C0 @ x0 - b) <= 1e-12 * np.linalg.norm(b) and np.linalg.norm(C1 @ x1 - b) <= 1e-12 * np.linalg.norm(b)
    del f  # parks the factorisation in the context
    g = lsa_hip.NdLu(hip_ctx, dC0, 0)
    assert g.info()["seconds_analyse"] == 0.0 and first["seconds_analyse"] > 0.0  # analysis reused
    assert np.array_equal(_solve(hip_ctx, g, b), x0)


def test_product_lu_against_the_cross_check_library(hip_ctx):
    """The product's direct solver (nested-dissection multifrontal LU) against an independent one on the same device (round 1's
    block-tridiagonal LU of the RCM order, tests/xcheck): the same solution to 1e-10."""
    import helpers  # noqa: F401
    import lsa_hip
    import xcheck
    from synthetic import fem
    from Solver.utils import pivot_safe_rcm

    es = fem.cylinder_case("S5k")
    C = sp.csr_matrix((es.A.data - SIGMA * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    perm = pivot_safe_rcm(C)
    C = C[perm][:, perm].tocsr()
    C.sort_indices()
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, C)
    rng = np.random.default_rng(11)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    db = lsa_hip.DeviceVector.from_numpy(hip_ctx, b)
    xs = []
    for f in (lsa_hip.NdLu(hip_ctx, dC, 0), xcheck.BlockLu(hip_ctx, dC)):
        dx = lsa_hip.DeviceVector(hip_ctx, es.n, np.complex128)
        f.solve(db, dx)
        xs.append(dx.numpy())
        del f
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-10 * np.linalg.norm(xs[1])
    assert np.linalg.norm(C @ xs[0] - b) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("case,sigma,leaf,tp_min", [("S5k", SIGMA, 0, 32), ("S5k", 0.05, 200, 32), ("S30k", SIGMA, 0, 64), ("S30k", SIGMA, 600, 100),
                                                    ("C9k", -5.0, 0, 32), ("C20k", -5.0 + 0.5j, 0, 200)])
def test_ndlu_tournament_pivoting(hip_ctx, monkeypatch, case, sigma, leaf, tp_min):
    """Tall pivot blocks choose the 32 pivot rows of a column block by a tournament (local eliminations over 256 / 512
    rows, 8-way merges, the winners' tile inverted in LDS; ``LSA_ND_TP_MIN`` = the block height from which a level does so,
    512 by default).  Forced onto small and mid-size fronts here -- one to three tournament rounds, partial last blocks,
    real and complex factors, 2D and 3D patterns, the transposed sweeps -- the answers agree with SuperLU like those of the
    panel elimination, and zero diagonals (pressure rows) are pivoted around."""
    import lsa_hip
    from synthetic import fem

    monkeypatch.setenv("LSA_ND_TP_MIN", str(tp_min))
    monkeypatch.setenv("LSA_ND_NO_CACHE", "1")
    if case.startswith("C"):
        es = fem.cube_case(case)
        C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    else:
        es, C = _shifted(case, sigma)
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), leaf)
    rng = np.random.default_rng(11)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    x = _solve(hip_ctx, f, b)
    assert np.linalg.norm(C @ x - b) <= 1e-12 * np.linalg.norm(b)
    xref = spla.splu(sp.csc_matrix(C.astype(np.complex128))).solve(b)
    assert np.linalg.norm(x - xref) <= 1e-10 * np.linalg.norm(xref)
    assert np.array_equal(_solve(hip_ctx, f, b), x)
    dx = lsa_hip.DeviceVector(hip_ctx, es.n, np.complex128)
    f.solve_adjoint(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx)
    xa = dx.numpy()
    assert np.linalg.norm(C.conj().T @ xa - b) <= 1e-12 * np.linalg.norm(b)
    f.refactor(lsa_hip.CsrMatrix.from_scipy(hip_ctx, C))  # same values: the second elimination reproduces the first bit for bit
    assert np.array_equal(_solve(hip_ctx, f, b), x)


@pytest.mark.parametrize("case,sigma,work_mb", [("S5k", 0.018 + 0.7379601143282424j, 1), ("C9k", -5.0, 2), ("S5k", 0.05, 0)])
def test_ndlu_packed_factors_chunked_fronts_and_elimination_order(hip_ctx, monkeypatch, case, sigma, work_mb):
    """The memory plan of round 3: factors packed to m^2 + 2 m b scalars per node, the working fronts of a level factored in
    chunks that share one arena (forced here onto small cases by ``LSA_ND_WORK_MB``: several chunks per level, update matrices
    recycled by the first-fit arena), and the caller's matrix in the elimination order of ``lsa_nd_order`` with the forest
    handed back (sweeps without index lists).  Same answers as SuperLU and as the library's own dissection of the
    unpermuted matrix; the device buffers are smaller than the sum of the fronts."""
    import lsa_hip
    from synthetic import fem

    monkeypatch.setenv("LSA_ND_NO_CACHE", "1")
    if work_mb:
        monkeypatch.setenv("LSA_ND_WORK_MB", str(work_mb))
    if case.startswith("C"):
        es = fem.cube_case(case)
        C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    else:
        es, C = _shifted(case, sigma)
    zd = C.diagonal() == 0
    o = lsa_hip.nd_order(C, 64, constraint=zd if case.startswith("C") else None)
    perm = o["perm"]
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    rng = np.random.default_rng(5)
    b = rng.standard_normal(es.n) + (1j * rng.standard_normal(es.n) if np.iscomplexobj(C.data) else 0.0)
    f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp), tree={"first": o["first"], "size": o["size"], "parent": o["parent"]})
    xp = _solve(hip_ctx, f, b[perm])
    x = np.empty_like(xp)
    x[perm] = xp
    assert np.linalg.norm(C @ x - b) <= 1e-12 * np.linalg.norm(b)
    xref = spla.splu(sp.csc_matrix(C)).solve(b)
    assert np.linalg.norm(x - xref) <= 1e-10 * np.linalg.norm(xref)
    assert np.array_equal(_solve(hip_ctx, f, b[perm]), xp)  # bitwise repeatable
    info = f.info()
    an = lsa_hip.NdAnalysis(Cp, tree={"first": o["first"], "size": o["size"], "parent": o["parent"]})
    assert info["factor_entries"] == an.factor_entries
    if work_mb:  # packed factors + one chunk + live update matrices: less than every front kept whole
        assert info["front_entries"] < an.front_entries
    g = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), 64)  # the library's own dissection, vectors through index lists
    x2 = _solve(hip_ctx, g, b)
    assert np.linalg.norm(x2 - xref) <= 1e-10 * np.linalg.norm(xref)
    dxa = lsa_hip.DeviceVector(hip_ctx, es.n, b.dtype)
    f.solve_adjoint(lsa_hip.DeviceVector.from_numpy(hip_ctx, b[perm]), dxa)  # transposed sweeps on the packed blocks
    xa = dxa.numpy()
    assert np.linalg.norm(Cp.conj().T @ xa - b[perm]) <= 1e-11 * np.linalg.norm(b)


def test_ndlu_lookahead_two_stream_path(hip_ctx, monkeypatch):
    """The tournament of the next column block on a second stream under the current block's rank-32 product (default: pivot
    blocks of 1024 rows and more), forced onto a small case: same factors as the one-stream order."""
    import lsa_hip
    from synthetic import fem

    es = fem.cube_case("C9k")
    C = sp.csr_matrix((es.A.data - fem.SIGMA_CUBE * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    b = np.random.default_rng(8).standard_normal(es.n)
    monkeypatch.setenv("LSA_ND_NO_CACHE", "1")
    monkeypatch.setenv("LSA_ND_TP_MIN", "64")
    xs = []
    for ahead in ("100000", "65"):
        monkeypatch.setenv("LSA_ND_LOOKAHEAD_MIN", ahead)
        f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), 128)
        xs.append(_solve(hip_ctx, f, b))
        assert np.linalg.norm(C @ xs[-1] - b) <= 1e-11 * np.linalg.norm(b)
    assert np.array_equal(xs[0], xs[1])


@pytest.mark.parametrize("case,sigma,leaf", [("S30k", SIGMA, 600), ("C20k", -5.0 + 0.5j, 0), ("C9k", -5.0, 128)])
def test_ndlu_super_blocks_of_128_pivot_columns(hip_ctx, monkeypatch, case, sigma, leaf):
    """Large pivot blocks (1024 rows and more by default, ``LSA_ND_SB_MIN``) are inverted in super-blocks of 128 columns: rank-32
    updates stay inside the super-block, everything outside it is updated once per 128 pivots by a product on the matrix
    cores (``nd_gj_update_kernel``).  Forced onto small fronts here (partial last super-blocks, real and complex factors, with
    and without the second-stream look-ahead): SuperLU's answer, and the look-ahead changes no bit."""
    import lsa_hip
    from synthetic import fem

    monkeypatch.setenv("LSA_ND_TP_MIN", "64")
    monkeypatch.setenv("LSA_ND_SB_MIN", "130")
    monkeypatch.setenv("LSA_ND_NO_CACHE", "1")
    if case.startswith("C"):
        es = fem.cube_case(case)
        C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    else:
        es, C = _shifted(case, sigma)
    rng = np.random.default_rng(12)
    b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    xref = spla.splu(sp.csc_matrix(C.astype(np.complex128))).solve(b)
    xs = []
    for ahead in ("100000", "131"):
        monkeypatch.setenv("LSA_ND_LOOKAHEAD_MIN", ahead)
        f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, C), leaf)
        assert f.info()["max_front"] > 260  # (several super-blocks in the largest pivot blocks)
        xs.append(_solve(hip_ctx, f, b))
        assert np.linalg.norm(C @ xs[-1] - b) <= 1e-12 * np.linalg.norm(b)
        assert np.linalg.norm(xs[-1] - xref) <= 1e-10 * np.linalg.norm(xref)
    assert np.array_equal(xs[0], xs[1])


def test_inertia_counts_the_eigenvalues_below_a_shift(hip_ctx):
    """``lsa_ndlu_inertia``: for a real symmetric C the multifrontal elimination is a block congruence, so the negative
    eigenvalues of ``A - sigma M`` -- the eigenvalues of the definite pencil below sigma -- are counted on the pivot blocks of
    its factors.  The membrane pair of the reference's benchmark (tests/benchmark/vibrating_membrane.md) against the dense
    generalised eigenvalues; a shift on an eigenvalue shows up as a zero count."""
    import lsa_hip
    import scipy.linalg as sla
    from synthetic import fem

    A, M, _ = fem.assemble_membrane(12, 12, 2.0, 4.0)
    A, M = sp.csr_matrix(A), sp.csr_matrix(M)
    w = np.sort(sla.eigh(A.toarray(), M.toarray(), eigvals_only=True))
    assert np.sum(np.isclose(w, 1.0)) == 96  # the identity Dirichlet rows of A and M: eigenvalue 1, 96 times
    for sigma in (0.5, 3.5, 9.0, 40.0, float(0.5 * (w[130] + w[131]))):
        K = sp.csr_matrix(A - sigma * M)
        K.sort_indices()
        f = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, K), 32)
        neg, zero, pos = f.inertia()
        assert zero == 0 and neg == int(np.sum(w < sigma)) and neg + pos == A.shape[0]
        del f
    with pytest.raises(lsa_hip.LsaError):  # a shift ON an eigenvalue (the 96-fold lambda = 1): singular, reported as such
        K1 = sp.csr_matrix(A - 1.0 * M)
        K1.sort_indices()
        lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, K1), 32)
    Kc = sp.csr_matrix(A - (0.3 + 0.1j) * M)
    Kc.sort_indices()
    fc = lsa_hip.NdLu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, Kc), 32)
    with pytest.raises(ValueError):  # complex factors: no inertia
        fc.inertia()
