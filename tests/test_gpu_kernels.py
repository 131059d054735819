"""GPU parity tests, kernel level: every HIP kernel on the path against the C/numpy oracle on the same seeded inputs.

All calls go through the C-ABI (``lsa_hip`` is a ctypes layer).  Tolerances: SpMV / SpTRSV / ILU are compared to the
scalar C restatement up to floating-point reassociation (different summation order), rtol 1e-12 on well-scaled
data; see each test.
"""

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case5k():
    from synthetic import fem

    return fem.cylinder_case("S5k")


@pytest.fixture(scope="module")
def shifted5k(case5k):
    """(C, perm): complex C = A - sigma M of S5k in pivot-safe RCM ordering, host copy."""
    from oracle import kernels
    from synthetic import fem
    from Solver.utils import pivot_safe_rcm

    A, M = case5k.A, case5k.M
    C = sp.csr_matrix((kernels.axpby_same_pattern(A, M, 1.0, -fem.SIGMA_RE50), A.indices, A.indptr), shape=A.shape)
    perm = pivot_safe_rcm(C)
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    return Cp, perm


def _rng_vec(n, seed, cplx=True):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n)
    return v + 1j * rng.standard_normal(n) if cplx else v


@pytest.mark.parametrize("mat_c,vec_c", [(False, False), (False, True), (True, True)])
def test_spmv_matches_oracle(hip_ctx, case5k, mat_c, vec_c):
    import lsa_hip
    from oracle import kernels

    A = case5k.A.astype(np.complex128) * (1.0 + 0.5j) if mat_c else case5k.A
    x = _rng_vec(A.shape[0], 1, vec_c)
    ref = kernels.spmv(A, x)
    dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, A)
    dx = lsa_hip.DeviceVector.from_numpy(hip_ctx, x)
    dy = lsa_hip.DeviceVector(hip_ctx, A.shape[0], ref.dtype)
    dA.matvec(dx, dy)
    got = dy.numpy()
    assert np.linalg.norm(got - ref) <= 1e-13 * np.linalg.norm(ref)


def test_spmv_edge_rows(hip_ctx):
    """Empty rows, a dense row longer than a wavefront, explicit zeros, 1x1."""
    import lsa_hip
    from oracle import kernels

    rng = np.random.default_rng(3)
    n = 300
    A = sp.random(n, n, density=0.02, random_state=7, format="lil")
    A[5, :] = rng.standard_normal(n)  # long row
    A[7, :] = 0  # empty row
    A = sp.csr_matrix(A)
    A.data[::7] = 0.0  # explicit zeros stay in the pattern
    x = _rng_vec(n, 4, True)
    dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, A)
    dy = lsa_hip.DeviceVector(hip_ctx, n, np.complex128)
    dA.matvec(lsa_hip.DeviceVector.from_numpy(hip_ctx, x), dy)
    ref = kernels.spmv(A, x)
    assert np.linalg.norm(dy.numpy() - ref) <= 1e-13 * np.linalg.norm(ref)
    one = lsa_hip.CsrMatrix.from_scipy(hip_ctx, sp.csr_matrix(np.array([[2.5]])))
    dy1 = lsa_hip.DeviceVector(hip_ctx, 1, np.float64)
    one.matvec(lsa_hip.DeviceVector.from_numpy(hip_ctx, np.array([4.0])), dy1)
    assert dy1.numpy()[0] == 10.0


def test_spmv_compressed_indices(hip_ctx, case5k, monkeypatch):
    """16-bit column indices relative to the row's first column (the default for large complex matrices; forced here
    with the variant word): same product as the oracle, unsorted rows, empty rows and all three scalar pairings included;
    a row that spans 65 536 columns or more silently keeps the 32-bit indices."""
    import lsa_hip
    from oracle import kernels

    monkeypatch.setenv("LSA_SPMV_VARIANT", str(16 | 0x800))
    for mat_c, vec_c in ((False, False), (False, True), (True, True)):
        A = case5k.A.astype(np.complex128) * (1.0 + 0.5j) if mat_c else case5k.A
        x = _rng_vec(A.shape[0], 11, vec_c)
        dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, A)
        dy = lsa_hip.DeviceVector(hip_ctx, A.shape[0], np.complex128 if vec_c else np.float64)
        dA.matvec(lsa_hip.DeviceVector.from_numpy(hip_ctx, x), dy)
        ref = kernels.spmv(A, x)
        assert np.linalg.norm(dy.numpy() - ref) <= 1e-13 * np.linalg.norm(ref)
    n = 70_000
    rows = np.array([0, 0, 3, 3, 3, n - 1])
    cols = np.array([n - 1, 2, 7, 5, 66_000, 0])  # row 0 spans 69 997 columns; row 3 is stored unsorted
    W = sp.csr_matrix((np.arange(1.0, 7.0) * (1 + 1j), (rows, cols)), shape=(n, n))
    x = _rng_vec(n, 12, True)
    dW = lsa_hip.CsrMatrix.from_scipy(hip_ctx, W)
    dy = lsa_hip.DeviceVector(hip_ctx, n, np.complex128)
    dW.matvec(lsa_hip.DeviceVector.from_numpy(hip_ctx, x), dy)
    assert np.linalg.norm(dy.numpy() - W @ x) <= 1e-13 * np.linalg.norm(W @ x)
    N = sp.csr_matrix((np.array([1.0, 2.0, 3.0]) * (1 - 1j), (np.array([4, 4, 9]), np.array([60_000, 100, 9]))), shape=(n, n))  # fits 16 bits
    dN = lsa_hip.CsrMatrix.from_scipy(hip_ctx, N)
    dN.matvec(lsa_hip.DeviceVector.from_numpy(hip_ctx, x), dy)
    assert np.linalg.norm(dy.numpy() - N @ x) <= 1e-13 * np.linalg.norm(N @ x)


def test_spmv_transpose(hip_ctx, case5k):
    import lsa_hip

    A = case5k.A.astype(np.complex128) * (1.0 - 0.25j)
    x = _rng_vec(A.shape[0], 11, True)
    dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, A)
    dx = lsa_hip.DeviceVector.from_numpy(hip_ctx, x)
    dy = lsa_hip.DeviceVector(hip_ctx, A.shape[0], np.complex128)
    dA.rmatvec(dx, dy, conj=True)
    ref = A.conj().T @ x
    assert np.linalg.norm(dy.numpy() - ref) <= 1e-12 * np.linalg.norm(ref)
    dA.rmatvec(dx, dy, conj=False)
    ref = A.T @ x
    assert np.linalg.norm(dy.numpy() - ref) <= 1e-12 * np.linalg.norm(ref)


def test_axpby_same_pattern(hip_ctx, case5k):
    import lsa_hip
    from oracle import kernels
    from synthetic import fem

    dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, case5k.A)
    dM = lsa_hip.CsrMatrix.from_scipy(hip_ctx, case5k.M)
    dC = dA.axpby(dM, 1.0, -fem.SIGMA_RE50)
    ref = kernels.axpby_same_pattern(case5k.A, case5k.M, 1.0, -fem.SIGMA_RE50)
    got = dC.values()
    assert got.dtype == np.complex128
    # one fused multiply-add per part on the device vs separate multiply and add on the host: last-bit differences
    assert np.max(np.abs(got - ref)) <= 4 * np.finfo(float).eps * np.max(np.abs(ref))
    dR = dA.axpby(dM, 2.0, -0.5)
    assert dR.dtype == np.float64
    assert np.allclose(dR.values(), 2.0 * case5k.A.data - 0.5 * case5k.M.data, rtol=1e-15, atol=0)
    other = lsa_hip.CsrMatrix.from_scipy(hip_ctx, sp.identity(case5k.n, format="csr"))
    with pytest.raises(ValueError):
        dA.axpby(other, 1.0, 1.0)


@pytest.mark.parametrize("levels", [0, 2])
def test_ilu_factor_matches_oracle(hip_ctx, shifted5k, levels):
    import lsa_hip
    from oracle import kernels

    Cp, _ = shifted5k
    ref_pat = kernels.iluk_pattern(Cp, levels)
    ref = kernels.ILU0(ref_pat, 0.0)
    pc = lsa_hip.Ilu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp), levels=levels)
    rp, ci, val = pc.factors()
    assert np.array_equal(rp, ref.rp) and np.array_equal(ci, ref.ci)  # symbolic ILU(k): identical pattern
    # numeric: same elimination order per row, products fused differently -> compare relative to the row scale
    scale = np.max(np.abs(ref.v))
    assert np.max(np.abs(val - ref.v)) <= 1e-11 * scale
    info = pc.info()
    assert info["nnz"] == ref_pat.nnz and info["nshift"] == 0
    assert info["levels_lower"] > 1 and info["levels_upper"] > 1


@pytest.mark.parametrize("algo,block", [(0, 0), (1, 0), (2, 256), (2, 1024)])
@pytest.mark.parametrize("which", [0, 1, 2])
def test_sptrsv_matches_oracle(hip_ctx, shifted5k, which, algo, block):
    import lsa_hip
    from oracle import kernels

    Cp, _ = shifted5k
    ref = kernels.ILU0(kernels.iluk_pattern(Cp, 1), 0.0)
    pc = lsa_hip.Ilu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp), levels=1)
    pc.set_algorithm(algo, block)
    b = _rng_vec(Cp.shape[0], 21, True)
    db = lsa_hip.DeviceVector.from_numpy(hip_ctx, b)
    dx = lsa_hip.DeviceVector(hip_ctx, Cp.shape[0], np.complex128)
    pc.solve(db, dx, which)
    if which == 2:  # a second apply replays the captured graph (blocked form)
        pc.solve(db, dx, which)
    want = [ref.lower, ref.upper, ref.solve][which](b)
    got = dx.numpy()
    assert np.all(np.isfinite(got))
    assert np.linalg.norm(got - want) <= 1e-10 * np.linalg.norm(want)


def test_sptrsv_real_factors_complex_vectors(hip_ctx, case5k):
    """Real sigma: real factors applied to complex Krylov vectors."""
    import lsa_hip
    from oracle import kernels
    from Solver.utils import pivot_safe_rcm

    C = sp.csr_matrix((case5k.A.data - 0.05 * case5k.M.data, case5k.A.indices, case5k.A.indptr), shape=case5k.A.shape)
    perm = pivot_safe_rcm(C)
    Cp = C[perm][:, perm].tocsr()
    Cp.sort_indices()
    ref = kernels.ILU0(kernels.iluk_pattern(Cp, 1), 0.0)
    pc = lsa_hip.Ilu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp), levels=1)
    b = _rng_vec(Cp.shape[0], 5, True)
    dx = lsa_hip.DeviceVector(hip_ctx, Cp.shape[0], np.complex128)
    pc.solve(lsa_hip.DeviceVector.from_numpy(hip_ctx, b), dx, 2)
    want = ref.solve(b.real) + 1j * ref.solve(b.imag)
    assert np.linalg.norm(dx.numpy() - want) <= 1e-10 * np.linalg.norm(want)


def test_ilu_zero_pivot_is_an_error(hip_ctx):
    import lsa_hip

    A = sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))  # structurally present, numerically zero pivot
    dA = lsa_hip.CsrMatrix.from_scipy(hip_ctx, sp.csr_matrix((np.array([0.0, 1.0, 1.0, 0.0]), np.array([0, 1, 0, 1]), np.array([0, 2, 4])), shape=(2, 2)))
    with pytest.raises(lsa_hip.LsaError) as ei:
        lsa_hip.Ilu(hip_ctx, dA, levels=0, shift_tol=0.0)
    assert ei.value.status == -3
    pc = lsa_hip.Ilu(hip_ctx, dA, levels=0, shift_tol=1e-8)  # with a shift the factorisation goes through
    assert pc.info()["nshift"] >= 1
    with pytest.raises(lsa_hip.LsaError):  # missing diagonal entry
        lsa_hip.Ilu(hip_ctx, lsa_hip.CsrMatrix.from_scipy(hip_ctx, A), levels=0)


def test_gmres_ilu_solves_shifted_system(hip_ctx, shifted5k, case5k):
    import lsa_hip

    Cp, perm = shifted5k
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp)
    pc = lsa_hip.Ilu(hip_ctx, dC, levels=2)
    b = (case5k.M @ _rng_vec(Cp.shape[0], 9, True))[perm]
    db = lsa_hip.DeviceVector.from_numpy(hip_ctx, b)
    dx = lsa_hip.DeviceVector(hip_ctx, Cp.shape[0], np.complex128)
    its, rr = lsa_hip.gmres(hip_ctx, dC, pc, db, dx, rtol=1e-11, restart=200, maxit=600)
    x = dx.numpy()
    true_rr = np.linalg.norm(Cp @ x - b) / np.linalg.norm(b)
    assert rr <= 1e-11 and true_rr <= 1e-10, (its, rr, true_rr)
    assert its < 200
    # against the direct solve of the oracle
    import scipy.sparse.linalg as spla

    xref = spla.splu(Cp.tocsc()).solve(b)
    assert np.linalg.norm(x - xref) <= 1e-7 * np.linalg.norm(xref)


def test_gmres_reports_divergence(hip_ctx, shifted5k):
    import lsa_hip

    Cp, _ = shifted5k
    dC = lsa_hip.CsrMatrix.from_scipy(hip_ctx, Cp)
    b = lsa_hip.DeviceVector.from_numpy(hip_ctx, _rng_vec(Cp.shape[0], 2, True))
    x = lsa_hip.DeviceVector(hip_ctx, Cp.shape[0], np.complex128)
    with pytest.raises(lsa_hip.LsaError) as ei:
        lsa_hip.gmres(hip_ctx, dC, None, b, x, rtol=1e-12, restart=10, maxit=20)
    assert ei.value.status == -4
