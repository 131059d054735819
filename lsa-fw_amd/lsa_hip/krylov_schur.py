"""Krylov-Schur outer iteration (host logic of ``SLEPc.EPS.solve()``, reached from ``Solver/utils.py:270``).

SLEPc is not vendored in the reference; this restates its default EPS (Krylov-Schur, Stewart 2001; SLEPc
technical report STR-7) at the level the reference depends on:

* Arnoldi decomposition ``OP V_m = V_m H_m + v_{m+1} b^H`` grown to ``m = ncv`` vectors;
* Schur form of the small ``H_m`` with the *wanted* Ritz values moved to the leading block;
* residual estimate of a Ritz pair ``|b^H y|`` and the relative convergence test ``<= tol * |theta|``
  (``EPS_CONV_REL``, the default the reference never overrides; tol = ``EigensolverConfig.atol``);
* truncation to ``k = nconv + (m - nconv) // 2`` vectors and restart, at most ``max_it`` times.

The heavy parts (operator applies, orthogonalisation, ``V <- V Q``) run behind a *backend*; the product backend
is :class:`lsa_hip.KrylovBasis` (HIP).  The backend protocol is ``n``, ``ncv``, ``inject(j, v)``,
``extend(j0, j1, H) -> breakdown``, ``restart(m, Q)``, ``ritz_vectors(m, Y, normalise)``.  Only the O(ncv^3)
dense algebra on ``H`` happens here, with LAPACK through scipy.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable

import contextlib

import numpy as np
import scipy.linalg as sla
from scipy.linalg import lapack as _lapack

try:  # 80 x 80 LAPACK calls are slower on 64 BLAS threads than on one (ztrsen 0.9 -> 0.2 ms, zgeev 13 -> 0.2 ms on the GPU box)
    from threadpoolctl import threadpool_limits as _threadpool_limits
except ImportError:  # pragma: no cover
    _threadpool_limits = None


def _single_threaded_blas():
    return _threadpool_limits(limits=1, user_api="blas") if _threadpool_limits is not None else contextlib.nullcontext()


@dataclass
class KrylovSchurResult:
    theta: np.ndarray  # converged Ritz values of OP, wanted first
    vectors: np.ndarray  # (n, len(theta)) Ritz vectors, unit 2-norm
    residuals: np.ndarray  # relative residual estimates |b^H y| / |theta|
    nconv: int
    restarts: int
    op_applies: int
    history: list = field(default_factory=list)


def _schur_wanted_first(T: np.ndarray, Q: np.ndarray, rank_key, k: int):
    """Reorder the complex Schur form H = Q T Q^H so that (at least) the k best-ranked eigenvalues lead.

    One LAPACK ``ztrsen`` on the factors the convergence test already computed (a second ``zgees`` with a Python sort
    callback costs as much as the first).  Ties at the selection threshold are all selected, like a sort callback would;
    returns (T, Q, sdim) with sdim the size of the leading block actually selected."""
    m = T.shape[0]
    if k <= 0 or k >= m:
        return T, Q, (m if k >= m else 0)
    key_diag = rank_key(np.diag(T))
    keys = np.sort(key_diag)
    thr = 0.5 * (keys[k - 1] + keys[k]) if keys[k] > keys[k - 1] else keys[k - 1]
    select = (key_diag <= thr).astype(np.int32)
    Ts, Qs, _w, sdim, _s, _sep, info = _lapack.ztrsen(select, T, Q, job="N", wantq=1)
    if info != 0:  # reordering failed (ill-conditioned swap): fall back to a sorted factorisation of Q T Q^H
        Ts, Qs, sdim = sla.schur(Q @ T @ Q.conj().T, output="complex", sort=lambda z: bool(rank_key(np.array([z]))[0] <= thr))
    return Ts, Qs, int(sdim)


def krylov_schur(
    backend,
    nev: int,
    tol: float,
    max_restarts: int,
    rank_key: Callable[[np.ndarray], np.ndarray],
    *,
    v0: np.ndarray | None = None,
    rng_seed: int = 0,
    keep_fraction: float = 0.5,
    on_restart: Callable[[dict], None] | None = None,
) -> KrylovSchurResult:
    """Compute ``nev`` eigenpairs of the backend's operator; wanted = smallest ``rank_key(theta)``.

    ``rank_key`` maps an array of Ritz values of OP to real sort keys (small = wanted), e.g. ``-abs(theta)``
    for shift-invert towards the target.  Returns every converged pair (possibly more than ``nev``, like
    ``EPS.getConverged()``), wanted first.
    """
    with _single_threaded_blas():
        return _krylov_schur(backend, nev, tol, max_restarts, rank_key, v0, rng_seed, keep_fraction, on_restart)


def _krylov_schur(backend, nev, tol, max_restarts, rank_key, v0, rng_seed, keep_fraction, on_restart) -> KrylovSchurResult:
    n, m = backend.n, backend.ncv
    if m > n:
        raise ValueError(f"ncv = {m} exceeds the problem size {n}")
    nev = min(nev, m)
    rng = np.random.default_rng(rng_seed)

    def random_vector():
        return rng.standard_normal(n) + 1j * rng.standard_normal(n)

    backend.inject(0, random_vector() if v0 is None else v0)

    H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    k = 0  # basis vectors carried over from the previous restart
    restarts = 0
    applies = 0
    history = []
    while True:
        # ---- expand to m vectors; continue past exact breakdowns (invariant subspace) with a fresh direction ----
        j, m_eff = k, m
        while j < m:
            bd = backend.extend(j, m, H)
            if bd < 0:
                applies += m - j
                break
            applies += bd - j + 1
            H[bd + 1, bd] = 0.0
            if bd + 1 >= m:  # broke down on the last step: the m vectors span an invariant subspace
                break
            backend.inject(bd + 1, random_vector())
            j = bd + 1
        Hm = H[:m_eff, :m_eff]
        b = H[m_eff, :m_eff].copy()  # b^H, the row under the square part

        # ---- Ritz pairs and residual estimates -------------------------------------------------------------------
        T, Q = sla.schur(Hm, output="complex")
        w, S = sla.eig(T)
        rank = np.argsort(rank_key(w), kind="stable")
        w, S = w[rank], S[:, rank]
        S = S / np.linalg.norm(S, axis=0)
        est = np.abs((b @ Q) @ S)
        rel = est / np.maximum(np.abs(w), np.finfo(float).tiny)
        nconv = 0
        while nconv < m_eff and rel[nconv] <= tol:
            nconv += 1
        info = {"restart": restarts, "nconv": nconv, "m": m_eff, "applies": applies,
                "next_unconverged": float(rel[nconv]) if nconv < m_eff else 0.0}
        history.append(info)
        if on_restart is not None:
            on_restart(info)

        if nconv >= nev or nconv >= m_eff or restarts >= max_restarts:
            Y = Q @ S[:, :nconv]
            X = backend.ritz_vectors(m_eff, Y, True) if nconv > 0 else np.zeros((n, 0), dtype=np.complex128)
            return KrylovSchurResult(w[:nconv], X, rel[:nconv], nconv, restarts, applies, history)

        # ---- truncate to the wanted part of the Schur form and restart -----------------------------------------------
        knew = nconv + int((m_eff - nconv) * keep_fraction)
        knew = max(min(knew, m_eff - 1), 1)
        T, Q, sdim = _schur_wanted_first(T, Q, rank_key, knew)
        knew = max(min(sdim, m_eff - 1), 1)
        bt = b @ Q
        backend.restart(m_eff, Q[:, :knew])
        H[:, :] = 0.0
        H[:knew, :knew] = T[:knew, :knew]
        H[knew, :knew] = bt[:knew]
        k = knew
        restarts += 1
