"""CPU suite, part 1: the oracle pinned against the reference's own known answers and golden vectors.

Everything here is test infrastructure checking test infrastructure: the numpy/scipy/C restatement in ``oracle/``
against (a) data taken from the reference's tests and published benchmark (``tests/golden/reference_known_answers.json``),
(b) dense QZ as an algorithm-independent cross-check, (c) the committed golden vectors of the synthetic cylinder pair,
(d) the structural properties the reference's FEM tests assert on (A, M).
"""

import json
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import kernels, shift_invert
from synthetic import fem

GOLDEN = Path(__file__).resolve().parent / "golden"
KNOWN = json.loads((GOLDEN / "reference_known_answers.json").read_text())


# ---- (a) reference known answers ---------------------------------------------------------------------------------------


def test_membrane_published_eigenvalues():
    """tests/benchmark/vibrating_membrane.md:102-110: P2, 32x32, [0,2]x[0,4]; published to 7 digits."""
    ref = KNOWN["membrane_32x32_p2"]
    A, M, bnd = fem.assemble_membrane(32, 32, *ref["domain"])
    lam, _, res = shift_invert.solve(A, M, 3.0, k=24, tol=1e-12, ncv=80)
    lam = np.sort(lam.real)
    lam = lam[np.abs(lam - 1.0) > 1e-8]  # spurious lambda = 1 of the identity Dirichlet rows (vibrating_membrane.py:169-173)
    for got, pub in zip(lam[:3], ref["published"]):
        assert abs(got - pub) < 5e-7  # every printed digit
    ana = fem.membrane_analytic(15)
    assert np.mean(np.abs(lam[:15] - ana) / ana) == pytest.approx(ref["avg_rel_error_first_15"], abs=5e-8)
    # the spurious eigenvalue has the multiplicity of the pinned dofs
    w = shift_invert.dense_generalized(A[:200, :200], M[:200, :200])  # small dense probe just for finiteness
    assert np.all(np.isfinite(w))


@pytest.mark.parametrize("name", ["diagonal_3x3", "jordan_2x2"])
def test_small_known_answers_dense(name):
    k = KNOWN[name]
    w = np.sort(shift_invert.dense_generalized(np.array(k["A"], dtype=float)).real)
    assert w == pytest.approx(sorted(k["eigenvalues"]), abs=k["atol"])


def test_complex_pair_and_vector_ratio():
    k = KNOWN["complex_pair_2x2"]
    A = sp.csr_matrix(np.array(k["A"], dtype=float))
    lam, V, res = shift_invert.solve(sp.block_diag([A, sp.identity(8) * 50.0]).tocsr(), None, 3.1, k=2, tol=1e-12, ncv=8)
    lam = lam[np.argsort(lam.imag)]
    want = [complex(*z) for z in k["eigenvalues"]]
    assert lam == pytest.approx(want, abs=k["atol"])
    w, vec = np.linalg.eig(np.array(k["A"], dtype=float))
    order = np.argsort(w.imag)
    ratios = [vec[0, i] / vec[1, i] for i in order]
    assert ratios == pytest.approx([complex(*z) for z in k["vector_ratio_v0_over_v1"]], abs=k["atol"])


def test_shift_invert_epsilon_case():
    k = KNOWN["shift_invert_epsilon"]
    A = sp.diags(k["diag"]).tocsr()
    big = sp.block_diag([A, sp.identity(10) * 7.0]).tocsr()  # ARPACK needs k < n - 1
    lam, _, _ = shift_invert.solve(big, None, k["sigma"], k=3, tol=1e-14, ncv=12)
    assert np.sort(lam.real) == pytest.approx(k["expected"], rel=k["rtol"])


# ---- (b) dense QZ cross-check of the shift-invert route ------------------------------------------------------------------


def test_oracle_matches_dense_qz_on_saddle_point_pair():
    es = fem.assemble_linearized_ns(fem.channel_mesh(8, 4, grading=0.3), 50.0)
    assert es.n < 700
    w = shift_invert.dense_generalized(es.A, es.M)
    sigma = fem.SIGMA_RE50
    w = w[np.argsort(np.abs(w - sigma))][:6]
    lam, V, res = shift_invert.solve(es.A, es.M, sigma, k=6, tol=1e-13, ncv=40)
    assert np.max(np.abs(np.sort_complex(lam) - np.sort_complex(w)) / np.abs(np.sort_complex(w))) < 1e-10
    assert res.max() < 1e-11
    assert np.allclose(np.linalg.norm(V, axis=0), 1.0, atol=1e-12)  # unit 2-norm (test_eigen.py:231-239)


# ---- (c) golden vectors of the synthetic cylinder pair ---------------------------------------------------------------------


def test_cylinder_golden_vectors():
    g = json.loads((GOLDEN / "cylinder_s2k.json").read_text())
    es = fem.cylinder_case("S2k")
    assert (es.n, es.A.nnz) == (g["n"], g["nnz"])
    assert np.linalg.norm(es.A.data) == pytest.approx(g["frobenius"][0], rel=1e-12)
    assert np.linalg.norm(es.M.data) == pytest.approx(g["frobenius"][1], rel=1e-12)
    hist = {str(k): int(v) for k, v in zip(*np.unique(np.diff(es.A.indptr), return_counts=True))}
    assert hist == g["row_degree_histogram"]
    lam, _, res = shift_invert.solve(es.A, es.M, complex(*g["sigma"]), k=10, tol=1e-13, ncv=60)
    want = np.array([complex(*z) for z in g["eigenvalues"]])
    for r in want:
        assert np.min(np.abs(lam - r)) <= 1e-10 * abs(r)
    assert res.max() < 1e-10


# ---- (d) structure of (A, M): what tests/unit/FEM/test_operators.py asserts ---------------------------------------------------


@pytest.fixture(scope="module")
def small_pair():
    return fem.assemble_linearized_ns(fem.channel_mesh(10, 6, grading=0.3), 50.0)


def test_shared_pattern_and_row_degrees(small_pair):
    A, M = small_pair.A, small_pair.M
    assert np.array_equal(A.indptr, M.indptr) and np.array_equal(A.indices, M.indices)
    deg = np.diff(A.indptr)
    assert deg.max() == 45  # 2*19 velocity + 7 pressure neighbours of an interior vertex
    assert 22 in deg  # edge-node velocity rows


def test_mass_and_operator_blocks(small_pair):
    """test_operators.py:150-210: M_vv SPD, other M blocks zero; A_pp zero, G and D non-empty and D = G^T."""
    es = small_pair
    u, p = es.dofs_u, es.dofs_p
    free_u = np.setdiff1d(u, es.dirichlet)
    Mvv = es.M[free_u][:, free_u]
    assert abs(Mvv - Mvv.T).max() < 1e-13
    x = np.random.default_rng(0).standard_normal(len(free_u))
    assert x @ (Mvv @ x) > 0
    assert abs(es.M[p][:, p]).max() == 0 and abs(es.M[u][:, p]).max() == 0 and abs(es.M[p][:, u]).max() == 0
    assert es.M[p][:, p].nnz > 0  # stored explicit zeros
    assert abs(es.A[p][:, p]).max() == 0
    G, D = es.A[free_u][:, p], es.A[p][:, free_u]
    assert abs(G).max() > 0 and abs(D - G.T).max() < 1e-13


def test_dirichlet_rows_are_identity_in_both(small_pair):
    """FEM/operators.py:483-485,504-506: identity rows AND columns in A and M -> spurious lambda = 1."""
    es = small_pair
    for K in (es.A, es.M):
        rows = K[es.dirichlet]
        assert np.allclose(rows.diagonal(k=0)[:0], [])  # (shape guard)
        d = K.diagonal()[es.dirichlet]
        assert np.all(d == 1.0)
        off = rows.copy()
        off = off.tolil()
        for i, r in enumerate(es.dirichlet):
            off[i, r] = 0
        assert abs(off.tocsr()).max() == 0
        cols = K[:, es.dirichlet].tolil()
        for i, r in enumerate(es.dirichlet):
            cols[r, i] = 0
        assert abs(cols.tocsr()).max() == 0


# ---- C kernels of the oracle vs scipy -------------------------------------------------------------------------------------


def test_c_kernels_against_scipy(small_pair):
    es = small_pair
    rng = np.random.default_rng(1)
    x = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    assert np.allclose(kernels.spmv(es.A, x), es.A @ x, rtol=1e-13, atol=1e-13)
    assert np.allclose(kernels.spmv(es.A, x.real), es.A @ x.real, rtol=1e-13, atol=1e-13)
    c = kernels.axpby_same_pattern(es.A, es.M, 1.0, -fem.SIGMA_RE50)
    assert np.allclose(c, es.A.data - fem.SIGMA_RE50 * es.M.data, rtol=1e-15, atol=0)


def test_ilu_with_full_fill_is_lu(small_pair):
    """ILU(k) with k large enough is the exact LU: solves to machine precision (pins orc_ilu0 + orc_iluk_symbolic)."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    es = small_pair
    C = sp.csr_matrix((kernels.axpby_same_pattern(es.A, es.M, 1.0, -fem.SIGMA_RE50), es.A.indices, es.A.indptr), shape=es.A.shape)
    # velocity first, pressure last: every pressure pivot receives fill
    perm = np.concatenate([es.dofs_u, es.dofs_p])
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    full = kernels.ILU0(kernels.iluk_pattern(Cp, 10_000), 0.0)
    b = np.random.default_rng(2).standard_normal(es.n) + 0j
    x = full.solve(b)
    assert np.linalg.norm(Cp @ x - b) <= 1e-9 * np.linalg.norm(b)
    # split solves compose to the full solve
    assert np.allclose(full.upper(full.lower(b)), x, rtol=1e-13, atol=1e-13)
    # ILU(0) keeps the pattern, ILU(1) adds fill monotonically
    p0, p1 = kernels.iluk_pattern(Cp, 0), kernels.iluk_pattern(Cp, 1)
    assert p0.nnz == Cp.nnz and p1.nnz > p0.nnz
    with pytest.raises(ZeroDivisionError):
        kernels.ILU0(sp.csr_matrix((np.array([0.0, 1.0, 1.0, 0.0]), np.array([0, 1, 0, 1]), np.array([0, 2, 4])), shape=(2, 2)), 0.0)


def test_projected_operator_of_eigen2_keeps_the_spectrum():
    """``Solver/eigen2.py:164-201`` zeroes the pressure dofs around every apply; M has no pressure columns, so the
    non-zero spectrum is that of the full problem and the vectors are the velocity parts."""
    from oracle import shift_invert
    from synthetic import fem

    es = fem.cylinder_case("S2k")
    sigma = fem.SIGMA_RE50
    full, Vf, _ = shift_invert.solve(es.A, es.M, sigma, k=4, tol=1e-13, ncv=40)
    proj, Vp, _ = shift_invert.solve(es.A, es.M, sigma, k=4, tol=1e-13, ncv=40, project_out=es.dofs_p)
    for r in full:
        assert np.min(np.abs(proj - r)) <= 1e-9 * abs(r)
    assert np.abs(Vp[es.dofs_p, :]).max() <= 1e-14
    j = int(np.argmin(np.abs(proj - full[0])))
    vu = Vf[:, 0].copy()
    vu[es.dofs_p] = 0.0
    assert abs(abs(np.vdot(vu, Vp[:, j])) / np.linalg.norm(vu) - 1.0) <= 1e-8
