"""GPU parity tests of the SLEPc-free front end ``Solver/eigen2.py`` (``ArpackEigenSolver``: velocity-subspace
shift-invert, reference ``Solver/eigen2.py:71-265``) against the oracle's restatement of the same operator."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_projected_shift_invert_matches_oracle():
    from oracle import shift_invert
    from synthetic import fem
    from Solver.eigen2 import ArpackEigenSolver, ShiftInvertConfig, _compute_residuals
    from Solver.utils import iEpsWhich

    es = fem.cylinder_case("S2k")
    sigma = fem.SIGMA_RE50
    k = 6
    ref, Vref, _ = shift_invert.solve(es.A, es.M, sigma, k=k, tol=1e-13, ncv=40, project_out=es.dofs_p)
    cfg = ShiftInvertConfig(sigma=sigma, k=k, tol=1e-10, ncv=40, which_sort=iEpsWhich.LARGEST_REAL)
    lam, V, res = ArpackEigenSolver(cfg, es.A, es.M, dofs_u=es.dofs_u, dofs_p=es.dofs_p).solve()
    assert lam.shape == (k,) and V.shape == (es.n, k) and res.shape == (k,)
    for r in ref:
        assert np.min(np.abs(lam - r)) <= 1e-8 * abs(r)
    assert np.all(np.diff(lam.real) <= 1e-14)  # which_sort = LARGEST_REAL: descending real parts (eigen2.py:240-242)
    assert np.all(V[es.dofs_p, :] == 0.0)  # the iteration never leaves the velocity subspace
    assert np.allclose(np.linalg.norm(V, axis=0), 1.0, atol=1e-12)
    # eigenvectors: the oracle's projected vectors up to phase
    for j in range(k):
        i = int(np.argmin(np.abs(ref - lam[j])))
        assert abs(abs(np.vdot(Vref[:, i], V[:, j])) - 1.0) <= 1e-7
    # the residual report is the reference's formula on the full (A, M)
    assert np.allclose(res, _compute_residuals(es.A.astype(complex), es.M.astype(complex), lam, V), rtol=1e-6, atol=1e-12)


def test_projection_keeps_the_spectrum_of_the_full_problem():
    """M has no pressure columns, so P C^-1 M P and C^-1 M share their non-zero eigenvalues."""
    from oracle import shift_invert
    from synthetic import fem
    from Solver.eigen2 import ArpackEigenSolver, ShiftInvertConfig

    es = fem.cylinder_case("S2k")
    full, _, _ = shift_invert.solve(es.A, es.M, 0.05, k=4, tol=1e-13, ncv=40)
    lam, _, _ = ArpackEigenSolver(ShiftInvertConfig(sigma=0.05, k=4, tol=1e-10), es.A, es.M, dofs_u=es.dofs_u, dofs_p=es.dofs_p).solve()
    for r in full:
        assert np.min(np.abs(lam - r)) <= 1e-8 * max(abs(r), 1e-3)


def test_shape_errors_match_the_reference():
    import scipy.sparse as sp

    from Solver.eigen2 import ArpackEigenSolver, ShiftInvertConfig, _sort_indices

    with pytest.raises(ValueError, match="square and have the same shape"):
        ArpackEigenSolver(ShiftInvertConfig(), sp.identity(4, format="csr"), sp.identity(5, format="csr"), dofs_u=np.arange(4), dofs_p=np.zeros(0, int))
    with pytest.raises(ValueError, match="Unknown which_sort"):
        _sort_indices(np.ones(3), "XX")
    assert list(_sort_indices(np.array([1 + 3j, 2 - 1j, -5 + 0j]), "LM_abs")) == [2, 0, 1]
