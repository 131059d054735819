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
