"""ORACLE (test infrastructure, CPU only) -- shift-invert eigen-solve restated with scipy.

Only ``tests/``, ``bench.py`` (``cpu_baseline`` leg) and ``__graft_entry__.smoke()`` may import this file.

The reference's hot path is ``SLEPc.EPS.solve()`` (third-party, un-vendored, nominal SLEPc/PETSc 3.22.0:
``README.md:86-87``).  The reference itself states the algorithm SLEPc-free in ``Solver/eigen2.py``; this file
follows that statement line by line with the same libraries where they exist here (ARPACK through scipy) and
SuperLU in place of PETSc's KSP(preonly)+LU:

* ``C = A - sigma M``                              ``Solver/eigen2.py:109-111``
* factorise ``C`` once (LU)                         ``Solver/eigen2.py:121-151``
* ``OP x = C^-1 (M x)``                             ``Solver/eigen2.py:164-201`` (``project_out=`` adds its pressure projection)
* ``eigs(OP, k, which='LM', tol, maxiter, ncv)``    ``Solver/eigen2.py:225-234``
* ``lambda = sigma + 1/mu``                         ``Solver/eigen2.py:209-211,239``
* relative residuals                                ``Solver/eigen2.py:48-56``
* ordering: nearest to the target first (EPS_TARGET_MAGNITUDE, what SLEPc uses with ST = sinvert)

Pinned against: the published vibrating-membrane eigenvalues (``tests/benchmark/vibrating_membrane.md:102-110``),
the known answers of ``tests/unit/Solver/test_eigen.py`` and dense QZ (``scipy.linalg.eig``) on small cases; see
``tests/test_oracle.py``.  The cylinder eigenvalues themselves have no fixture in the reference ("parity unpinned"
for those numbers).
"""

from __future__ import annotations

import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def compute_residuals(A, M, lam: np.ndarray, V: np.ndarray) -> np.ndarray:
    """``||A v - lam M v|| / (||A v|| + |lam| ||M v|| + 1e-16)`` column-wise (``Solver/eigen2.py:48-56``)."""
    Av = A @ V
    Mv = M @ V if M is not None else V
    R = Av - Mv * lam[np.newaxis, :]
    num = np.linalg.norm(R, axis=0)
    den = np.linalg.norm(Av, axis=0) + np.abs(lam) * np.linalg.norm(Mv, axis=0) + 1e-16
    return num / den


def solve(A, M, sigma: complex, k: int = 20, *, tol: float = 1e-12, ncv: int | None = None, maxiter: int = 500,
          v0: np.ndarray | None = None, return_info: bool = False, project_out: np.ndarray | None = None):
    """k eigenpairs of A x = lam M x nearest sigma.  Returns (lam, V, residuals), nearest first, unit 2-norm vectors.

    ``project_out``: dof indices zeroed before and after every operator apply -- the velocity-subspace operator of
    ``Solver/eigen2.py:164-201`` (``x[dofs_p] = 0``; ``M x``; solve; ``y[dofs_p] = 0``)."""
    n = A.shape[0]
    A = sp.csr_matrix(A).astype(np.complex128)
    Mc = None if M is None else sp.csr_matrix(M).astype(np.complex128)
    C = (A - sigma * (Mc if Mc is not None else sp.identity(n, dtype=np.complex128, format="csr"))).tocsc()
    t0 = time.perf_counter()
    lu = spla.splu(C)
    t_factor = time.perf_counter() - t0
    applies = [0]

    def op(x):
        applies[0] += 1
        if project_out is not None:
            x = x.copy()
            x[project_out] = 0.0
        y = lu.solve(Mc @ x if Mc is not None else x)
        if project_out is not None:
            y[project_out] = 0.0
        return y

    lop = spla.LinearOperator((n, n), matvec=op, dtype=np.complex128)
    ncv = ncv if ncv is not None else max(4 * k, 40)
    ncv = min(ncv, n - 1)
    t0 = time.perf_counter()
    mu, W = spla.eigs(lop, k=k, which="LM", tol=tol, maxiter=maxiter, ncv=ncv, v0=v0)
    t_eigs = time.perf_counter() - t0
    lam = sigma + 1.0 / mu
    order = np.argsort(np.abs(lam - sigma), kind="stable")
    lam, V = lam[order], W[:, order]
    V = V / np.linalg.norm(V, axis=0)
    res = compute_residuals(A, Mc, lam, V)
    if return_info:
        return lam, V, res, {"seconds_factor": t_factor, "seconds_eigs": t_eigs, "op_applies": applies[0]}
    return lam, V, res


def dense_generalized(A, M=None) -> np.ndarray:
    """All finite eigenvalues by dense QZ: the algorithm-independent cross-check for n <= a few thousand."""
    Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
    if M is None:
        return sla.eigvals(Ad)
    Md = M.toarray() if sp.issparse(M) else np.asarray(M)
    w = sla.eigvals(Ad, Md)
    return w[np.isfinite(w)]


def solve_two_sided(A, M, sigma: complex, k: int = 20, *, tol: float = 0.0, ncv: int | None = None, maxiter: int = 500):
    """Right AND left eigenpairs nearest ``sigma`` on ONE factorisation, and what they say about each eigenvalue.

    Direct problem as :func:`solve`.  Adjoint problem as the reference states it (``Sensitivity/__init__.py:247-274``):
    eigenpairs of ``(A^H, M^H)`` nearest ``conj(sigma)`` -- here through ``OP_adj x = C^-H (M^H x)`` on the same SuperLU
    factors (``trans='H'``) instead of a second factorisation of the transposed matrices.  Pairing by eigenvalue
    (``lambda_adj = conj(lambda)``), normalisation ``a^H M v = 1`` (``Sensitivity/__init__.py:281-287``).

    Returns a dict: ``lam`` (ARPACK's values, nearest first), ``V``, ``A_left`` (paired, scaled), ``res`` / ``res_left``
    (relative residuals, ``Solver/eigen2.py:48-56``), ``kappa[i] = ||a_i|| ||M v_i|| / |a_i^H M v_i|`` -- first-order,
    ``|delta lambda_i| <= ||a_i|| ||r_i|| / |a_i^H M v_i|`` with ``||r_i|| = res_i (||A v_i|| + |lambda_i| ||M v_i||)``,
    so ``|delta lambda_i| / |lambda_i| <~ 2 kappa_i res_i`` -- and ``lam_rq``, the two-sided Rayleigh quotients
    ``a^H A v / a^H M v``, whose error is of second order (``~ kappa res res_left``): the refined eigenvalue."""
    n = A.shape[0]
    Ac = sp.csr_matrix(A).astype(np.complex128)
    Mc = sp.identity(n, dtype=np.complex128, format="csr") if M is None else sp.csr_matrix(M).astype(np.complex128)
    lu = spla.splu((Ac - sigma * Mc).tocsc())
    MH, AH = Mc.conj().T.tocsr(), Ac.conj().T.tocsr()
    ncv = min(ncv if ncv is not None else max(4 * k, 40), n - 1)
    op = spla.LinearOperator((n, n), matvec=lambda x: lu.solve(Mc @ x), dtype=np.complex128)
    op_adj = spla.LinearOperator((n, n), matvec=lambda x: lu.solve(MH @ x, trans="H"), dtype=np.complex128)
    mu, W = spla.eigs(op, k=k, which="LM", tol=tol, maxiter=maxiter, ncv=ncv)
    lam = sigma + 1.0 / mu
    order = np.argsort(np.abs(lam - sigma), kind="stable")
    lam, V = lam[order], W[:, order] / np.linalg.norm(W[:, order], axis=0)
    mu2, W2 = spla.eigs(op_adj, k=k, which="LM", tol=tol, maxiter=maxiter, ncv=ncv)
    lam_adj = np.conj(sigma) + 1.0 / mu2
    pick = np.array([int(np.argmin(np.abs(np.conj(lam_adj) - z))) for z in lam])
    pair_gap = np.abs(np.conj(lam_adj[pick]) - lam) / np.abs(lam)
    L = W2[:, pick] / np.linalg.norm(W2[:, pick], axis=0)
    MV, AV = Mc @ V, Ac @ V
    prod = np.einsum("ij,ij->j", L.conj(), MV)  # a^H M v
    kappa = np.linalg.norm(L, axis=0) * np.linalg.norm(MV, axis=0) / np.abs(prod)
    lam_rq = np.einsum("ij,ij->j", L.conj(), AV) / prod
    return {"lam": lam, "V": V, "A_left": L / np.conj(prod)[np.newaxis, :], "res": compute_residuals(Ac, Mc, lam, V),
            "res_left": compute_residuals(AH, MH, np.conj(lam), L), "kappa": kappa, "lam_rq": lam_rq, "pair_gap": pair_gap,
            "distinct_left": len(set(pick.tolist())) == len(pick)}
