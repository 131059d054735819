"""CPU suite: the library's own small dense eigen-algebra (``csrc/dense.hip``: complex Schur form by Hessenberg reduction + the
shifted QR algorithm, reordering of the Schur form, eigenvectors of the triangular factor) -- what the one-call eigen-solve
``lsa_eigs_sinvert`` uses on the projected matrix instead of LAPACK.  Checked against scipy (LAPACK) on the shapes the
Krylov-Schur iteration produces: Hessenberg (first expansion), triangle + spike row + Hessenberg (after a restart), general."""

import ctypes

import numpy as np
import pytest
import scipy.linalg as sla

import helpers  # noqa: F401  (sys.path)
import lsa_hip


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _schur(A):
    lib = lsa_hip.load_library()
    n = A.shape[0]
    T = np.asfortranarray(A.astype(np.complex128))
    Q = np.zeros((n, n), dtype=np.complex128, order="F")
    assert lib.lsa_dense_schur(n, _p(T), max(n, 1), _p(Q), max(n, 1)) == 0
    return T, Q


def _shapes(rng, n, kind):
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    if kind == "hessenberg":
        A = np.triu(A, -1)
    elif kind == "after_restart" and n > 4:  # leading triangle, spike row, Hessenberg rest (Krylov-Schur after a truncation)
        k = n // 2
        A = np.triu(A, -1)
        A[:k, :k] = np.triu(A[:k, :k])
        A[k, :k] = rng.standard_normal(k) * 10.0 ** (-rng.integers(0, 12, k))  # converging pairs: tiny couplings
    elif kind == "real":
        A = A.real.astype(np.complex128)
    return A


@pytest.mark.parametrize("kind", ["hessenberg", "after_restart", "general", "real"])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 80, 120])
def test_schur_form(kind, n):
    rng = np.random.default_rng(n + len(kind))
    A = _shapes(rng, n, kind)
    T, Q = _schur(A)
    if n == 0:
        return
    scale = max(np.linalg.norm(A), 1e-300)
    assert np.linalg.norm(Q @ T @ Q.conj().T - A) <= 1e-13 * n * scale
    assert np.linalg.norm(Q.conj().T @ Q - np.eye(n)) <= 1e-13 * n
    assert np.count_nonzero(np.tril(T, -1)) == 0
    w, w0 = np.diag(T), sla.eigvals(A)
    assert max(np.min(np.abs(w - z)) for z in w0) <= 1e-9 * max(np.abs(w0).max(), 1e-300)  # (random matrices: well-conditioned spectra)


def test_schur_of_defective_and_repeated_spectra():
    J = np.array([[1, 1, 0, 0], [0, 1, 1, 0], [0, 0, 1, 0], [0, 0, 0, 2]], dtype=np.complex128)  # Jordan block + a simple eigenvalue
    T, Q = _schur(J)
    assert np.linalg.norm(Q @ T @ Q.conj().T - J) <= 1e-14 and np.allclose(np.sort(np.diag(T).real), [1, 1, 1, 2], atol=1e-5)
    D = np.diag([2.0, 2.0, 3.0, 3.0, 3.0]).astype(np.complex128)
    rng = np.random.default_rng(0)
    U = sla.qr(rng.standard_normal((5, 5)) + 1j * rng.standard_normal((5, 5)))[0]
    A = U @ D @ U.conj().T
    T, Q = _schur(A)
    assert np.linalg.norm(Q @ T @ Q.conj().T - A) <= 1e-13 and np.allclose(np.sort(np.diag(T).real), [2, 2, 3, 3, 3], atol=1e-12)
    Z = np.zeros((6, 6), dtype=np.complex128)
    T, Q = _schur(Z)
    assert np.all(T == 0) and np.allclose(Q, np.eye(6))


def test_reordering_and_triangular_eigenvectors():
    lib = lsa_hip.load_library()
    rng = np.random.default_rng(5)
    n = 60
    A = np.triu(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)), -1)
    T, Q = _schur(A)
    d = np.diag(T).copy()
    select = (np.abs(d) > np.median(np.abs(d))).astype(np.int32)
    ns = ctypes.c_int32(0)
    assert lib.lsa_dense_schur_reorder(n, _p(T), n, _p(Q), n, _p(select), ctypes.byref(ns)) == 0
    k = ns.value
    assert k == select.sum()
    assert np.linalg.norm(Q @ T @ Q.conj().T - A) <= 1e-12 * np.linalg.norm(A) and np.count_nonzero(np.tril(T, -1)) == 0
    assert np.allclose(np.diag(T)[:k], d[select == 1], rtol=1e-10) and np.allclose(np.diag(T)[k:], d[select == 0], rtol=1e-10)  # orders kept
    # the leading k Schur vectors span the invariant subspace of the selected eigenvalues
    assert np.linalg.norm(A @ Q[:, :k] - Q[:, :k] @ T[:k, :k]) <= 1e-12 * np.linalg.norm(A)
    S = np.zeros((n, n), dtype=np.complex128, order="F")
    assert lib.lsa_dense_tri_eigenvectors(n, _p(T), n, _p(S), n) == 0
    assert np.abs(T @ S - S * np.diag(T)[None, :]).max() <= 1e-12 * np.abs(T).max()
    assert np.allclose(np.linalg.norm(S, axis=0), 1.0) and np.count_nonzero(np.tril(S, -1)) == 0
    X = Q @ S  # eigenvectors of A
    assert np.abs(A @ X - X * np.diag(T)[None, :]).max() <= 1e-11 * np.abs(A).max()
    assert lib.lsa_dense_schur(-1, None, 1, None, 1) != 0  # bad arguments are refused, not dereferenced


@pytest.mark.parametrize("n,kind", [(1, "pos"), (2, "indef"), (7, "random"), (40, "random"), (150, "random"), (60, "zero-diagonal"), (33, "singular")])
def test_symmetric_inertia_matches_the_eigenvalue_signs(n, kind):
    """``lsa_dense_sym_inertia`` (Bunch-Kaufman diagonal pivoting on the host: what ``lsa_ndlu_inertia`` applies to the pivot
    blocks of the forest) against the signs of numpy's eigenvalues: definite, indefinite, a saddle-point block with a zero diagonal
    (2 x 2 pivots), a singular matrix; both triangles are read and symmetrised (a computed inverse is unsymmetric in the last bits)."""
    import ctypes

    lib = lsa_hip.load_library()
    rng = np.random.default_rng(n)
    if kind == "pos":
        A = np.array([[3.0]])
    elif kind == "indef":
        A = np.array([[0.0, 2.0], [2.0, 0.0]])
    elif kind == "zero-diagonal":
        B = rng.standard_normal((n // 3, n - n // 3))
        K = rng.standard_normal((n - n // 3, n - n // 3))
        A = np.block([[K @ K.T + np.eye(n - n // 3), B.T], [B, np.zeros((n // 3, n // 3))]])
    elif kind == "singular":
        X = rng.standard_normal((n, n - 3))
        A = X @ np.diag(rng.standard_normal(n - 3)) @ X.T
    else:
        X = rng.standard_normal((n, n))
        A = X + X.T
    w = np.linalg.eigvalsh(A)
    scale = np.abs(w).max()
    want = (int(np.sum(w < -1e-10 * scale)), int(np.sum(np.abs(w) <= 1e-10 * scale)), int(np.sum(w > 1e-10 * scale)))
    noisy = np.asfortranarray(A + 1e-16 * scale * rng.standard_normal(A.shape))  # (not exactly symmetric)
    ng, ze, ps = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.lsa_dense_sym_inertia(n, noisy.ctypes.data_as(ctypes.c_void_p), n, 1e-12, ctypes.byref(ng), ctypes.byref(ze), ctypes.byref(ps)) == 0
    assert (ng.value, ze.value, ps.value) == want
    assert lib.lsa_dense_sym_inertia(-1, None, 1, 0.0, ctypes.byref(ng), ctypes.byref(ze), ctypes.byref(ps)) != 0
