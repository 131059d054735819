"""GPU tests of the row-sharded layout that one GPU can run: the degenerate one-rank job (padded block layout,
shard upload, sharded operator builder, masked start vectors, block-local ordering) must reproduce the single-GPU
answer, and shards of a two-rank partition must reproduce their rows of the global SpMV.  The RCCL exchange itself
needs >= 2 GPUs and is covered on CPU by the gloo emulation in tests/test_sharding_cpu.py.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_layout_single_rank_matches_oracle():
    from oracle import fem, shift_invert
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    es = fem.cylinder_case("S2k")
    sigma = fem.SIGMA_RE50
    ref, _, _ = shift_invert.solve(es.A, es.M, sigma, k=5, tol=1e-13)
    solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=5, atol=1e-10, ncv=40), check_hermitian=False, ilu_levels=2, layout="sharded")
    solver.solver.set_st_type(iSTType.SINVERT)
    solver.solver.set_target(sigma)
    solver.solver.set_st_pc_type(PreconditionerType.ILU)
    pairs = solver.solve()
    part = solver.solver._prepared["part"]
    assert part is not None and part.nranks == 1 and part.n_pad > es.n  # padding slots exist and were carried along
    lam = np.array([p[0] for p in pairs])
    for r in ref:
        assert np.min(np.abs(lam - r)) <= 1e-8 * abs(r)
    V = np.column_stack([p[1].as_array() for p in pairs])
    assert V.shape[0] == es.n
    assert shift_invert.compute_residuals(es.A, es.M, lam, V).max() <= 1e-8
    assert solver.solver.residuals().max() <= 1e-8


def test_two_rank_shards_reproduce_global_spmv(hip_ctx):
    import lsa_hip
    from lsa_hip import sharding
    from oracle import fem

    es = fem.cylinder_case("S5k")
    part = sharding.partition_rows(es.A.indptr, 2)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
    xp = part.pad_vector(x)
    dx = lsa_hip.DeviceVector.from_numpy(hip_ctx, xp)
    dy = lsa_hip.DeviceVector(hip_ctx, part.n_pad, np.complex128)
    for r in range(2):  # both shards write their own block of one padded global vector
        rows = sharding.shard_rows(es.A, part, r)
        dA = lsa_hip.CsrMatrix.from_scipy_shard(hip_ctx, rows, part.n_pad, r * part.b_pad)
        dA.matvec(dx, dy)
    yp = dy.numpy()
    ref = es.A @ x
    assert np.linalg.norm(part.unpad_vector(yp) - ref) <= 1e-13 * np.linalg.norm(ref)
    assert np.all(yp[part.pad_vector(np.ones(es.n)) == 0] == 0)  # padding untouched
