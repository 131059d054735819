"""GPU tests of the assembler-free ``LinearSolver.solve`` front end (reference: Solver/linear.py:38-87)."""

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def system():
    from synthetic import fem

    es = fem.cylinder_case("S2k")
    C = sp.csr_matrix((es.A.data - 0.05 * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    b = es.M @ np.random.default_rng(0).standard_normal(es.n)
    return C, b, spla.splu(C.tocsc()).solve(b)


def test_preonly_is_a_direct_solve(system):
    from FEM.utils import iPETScMatrix, iPETScVector
    from Solver.linear import LinearSolver
    from Solver.utils import KSPType

    C, b, xref = system
    x = LinearSolver.solve(iPETScMatrix(C), iPETScVector(b), ksp_type=KSPType.PREONLY)
    assert isinstance(x, iPETScVector) and x.size == C.shape[0]
    assert np.linalg.norm(x.as_array() - xref) <= 1e-10 * np.linalg.norm(xref)
    # complex right-hand side on a real matrix
    xc = LinearSolver.solve(C, b * (1 + 2j), ksp_type=KSPType.PREONLY)
    assert np.linalg.norm(xc.as_array() - xref * (1 + 2j)) <= 1e-10 * np.linalg.norm(xref) * np.sqrt(5)


def test_gmres_with_and_without_preconditioner(system):
    from Solver.linear import LinearSolver
    from Solver.utils import KSPType, PreconditionerType

    C, b, xref = system
    x = LinearSolver.solve(C, b, ksp_type=KSPType.GMRES, rtol=1e-11, pc=PreconditionerType.ILU)
    assert np.linalg.norm(C @ x.as_array() - b) <= 1e-10 * np.linalg.norm(b)
    # the reference's setting (no preconditioner) on an easy system: the mass matrix
    from synthetic import fem

    es = fem.cylinder_case("S2k")
    u = es.dofs_u[:300]  # velocity-velocity mass block: SPD (the pressure rows of M are zero)
    Mvv = es.M[u][:, u].tocsr()
    xm = LinearSolver.solve(Mvv, np.ones(300), ksp_type=KSPType.GMRES, rtol=1e-10, max_it=300)
    assert np.linalg.norm(Mvv @ xm.as_array() - 1.0) <= 1e-9 * np.sqrt(300)


def test_unsupported_ksp_and_shapes(system):
    from Solver.linear import LinearSolver
    from Solver.utils import KSPType

    C, b, _ = system
    with pytest.raises(ValueError, match="not supported"):
        LinearSolver.solve(C, b, ksp_type=KSPType.CG)
    with pytest.raises(ValueError):
        LinearSolver.solve(C, b[:-1], ksp_type=KSPType.PREONLY)
