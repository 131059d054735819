"""GPU tests of the row-sharded layout that one GPU can run: the degenerate one-rank job (padded block layout,
shard upload, sharded operator builder, masked start vectors, block-local ordering) must reproduce the single-GPU
answer, and shards of a two-rank partition must reproduce their rows of the global SpMV.  The RCCL exchange itself
needs >= 2 GPUs; with several ranks on this one GPU the same device path runs over the host-staged transport (gloo).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_layout_single_rank_matches_oracle():
    from oracle import shift_invert
    from synthetic import fem
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
    from synthetic import fem

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


def _rank_main(rank: int, world: int, port: int, out_dir: str, case: str, pc: str, env: str = "") -> None:
    """One rank of a sharded solve; all ranks share GPU 0 and exchange through the host-staged transport (gloo)."""
    import os
    import sys
    from pathlib import Path

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for item in filter(None, env.split(",")):  # e.g. "LSA_ND_DIST_MIN=1,LSA_ND_XSTAGE_KB=64"
        key, _, val = item.partition("=")
        os.environ[key] = val
    if case.startswith("C"):
        os.environ["LSA_DIST_SPMV"] = "shard"  # the products on this rank's rows + exchange (the default of large 3D patterns)
    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root), str(root / "lsa-fw_amd"), str(root / "tests")]
    import torch.distributed as dist

    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    dist.init_process_group("gloo", rank=rank, world_size=world)
    es = fem.cube_case(case) if case.startswith("C") else fem.cylinder_case(case)
    sigma = fem.SIGMA_CUBE if case.startswith("C") else fem.SIGMA_RE50
    kw = {"ilu_levels": 2} if pc == "ilu" else {}
    solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=5, atol=1e-10, ncv=40), check_hermitian=False, layout="sharded", **kw)
    solver.solver.set_st_type(iSTType.SINVERT)
    solver.solver.set_target(sigma)
    solver.solver.set_st_pc_type(PreconditionerType.ILU if pc == "ilu" else PreconditionerType.LU)
    pairs = solver.solve()
    lam = np.array([p[0] for p in pairs[:5]])
    V = np.column_stack([p[1].as_array() for p in pairs[:5]])
    st = solver.solver.stats
    np.savez(Path(out_dir) / f"rank{rank}.npz", lam=lam, V=V, res=solver.solver.residuals()[:5], gmres=st["gmres_iters"], applies=st["op_applies"],
             gathers=st["allgather_calls"], nbytes=st["allgather_bytes_received"], ranks=st["ranks"])
    solver.solver.release()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pc,case", [(2, "lu", "S2k"), (3, "lu", "S2k"), (4, "lu", "S2k"), (2, "ilu", "S2k"), (4, "lu", "C9k")])
def test_sharded_solve_with_several_ranks_on_one_gpu(tmp_path, world, pc, case):
    """The whole device path of the sharded layouts with more than one rank.  LU: the subtree-parallel exact
    factorisation (every rank factors its subtrees of the nested-dissection forest, one all-gather of the subtree roots'
    fronts, the replicated top; per operator apply two all-gathers on the 2D pattern, four on the 3D one, and no inner iteration).  ILU: row shards in the
    padded block layout, block-Jacobi ILU(2), GMRES over the replicated basis, one all-gather after every SpMV and every
    preconditioner apply.  The ranks share this box's single GPU, so the exchange runs through the host-staged transport
    (gloo) instead of RCCL: same call sites, same layout."""
    import socket

    import torch.multiprocessing as mp

    from oracle import shift_invert
    from synthetic import fem

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), case, pc), nprocs=world, join=True)
    out = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in out[1:]:  # replicated bases, fixed-order reductions: bit-identical on every rank, no all-reduce anywhere
        assert np.array_equal(o["lam"], out[0]["lam"]) and np.array_equal(o["V"], out[0]["V"])
    es = fem.cube_case(case) if case.startswith("C") else fem.cylinder_case(case)  # C9k: the 3D pattern (BASELINE config 4's layout, small)
    ref, _, _ = shift_invert.solve(es.A, es.M, fem.SIGMA_CUBE if case.startswith("C") else fem.SIGMA_RE50, k=5, tol=1e-13)
    for r in ref:
        assert np.min(np.abs(out[0]["lam"] - r)) <= 1e-8 * abs(r)
    assert out[0]["res"].max() <= 1e-8
    assert shift_invert.compute_residuals(es.A, es.M, out[0]["lam"], out[0]["V"]).max() <= 1e-8
    assert int(out[0]["ranks"]) == world
    if pc == "lu":
        # exact, no iteration.  Exchanges per apply: the update vectors of the subtree roots and the solution; on the 3D
        # pattern (~100 entries per row) the two products M x and C x run on this rank's rows and are exchanged as well,
        # on the 2D pattern every rank multiplies the whole matrices (cheaper than the exchange, see solver.hip)
        per_apply = 4 if case.startswith("C") else 2  # (the small 3D case is forced onto the sharded products, see _rank_main)
        assert int(out[0]["gmres"]) == 0 and per_apply * int(out[0]["applies"]) <= int(out[0]["gathers"]) <= per_apply * int(out[0]["applies"]) + 8
    else:  # block-Jacobi over > 1 rank: the inner solves iterate
        assert int(out[0]["gmres"]) > int(out[0]["applies"]) and int(out[0]["gathers"]) > 2 * int(out[0]["applies"])


def test_sharded_3d_case_with_four_ranks_matches_the_single_gpu_solve(tmp_path):
    """BASELINE config 4's discretisation (3D Taylor-Hood on Kuhn tetrahedra) at 83 k unknowns, the elimination forest cut over
    FOUR ranks (subtree-parallel exact LU, products on the ranks' rows, four all-gathers per apply): the same eigenvalues as
    the single-GPU solve to 1e-8, residuals below 1e-8, ranks bit-identical, no inner iteration."""
    import socket

    import torch.multiprocessing as mp

    from oracle import shift_invert
    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(4, port, str(tmp_path), "C80k", "lu"), nprocs=4, join=True)
    out = [np.load(tmp_path / f"rank{r}.npz") for r in range(4)]
    for o in out[1:]:
        assert np.array_equal(o["lam"], out[0]["lam"]) and np.array_equal(o["V"], out[0]["V"])
    es = fem.cube_case("C80k")
    one = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=5, atol=1e-10, ncv=40), check_hermitian=False)
    one.solver.set_st_type(iSTType.SINVERT)
    one.solver.set_target(fem.SIGMA_CUBE)
    one.solver.set_st_pc_type(PreconditionerType.LU)
    ref = np.array([p[0] for p in one.solve()[:5]])
    one.solver.release()
    assert helpers_match(out[0]["lam"], ref).max() <= 1e-8
    assert out[0]["res"].max() <= 1e-8 and int(out[0]["gmres"]) == 0 and int(out[0]["ranks"]) == 4
    assert shift_invert.compute_residuals(es.A, es.M, out[0]["lam"], out[0]["V"]).max() <= 1e-8


@pytest.mark.parametrize("world,case,env", [(2, "S2k", "LSA_ND_DIST_MIN=1"), (4, "S2k", "LSA_ND_DIST_MIN=1,LSA_ND_XSTAGE_KB=8"),
                                            (3, "C9k", "LSA_ND_DIST_MIN=1,LSA_ND_XSTAGE_KB=64"),
                                            (4, "C40k", "LSA_ND_DIST_MIN=2500,LSA_ND_WORK_MB=64"), (4, "C160k", "LSA_ND_DIST_MIN=3000")])
def test_distributed_top_fronts_with_several_ranks_on_one_gpu(tmp_path, world, case, env):
    """The top of the forest DISTRIBUTED over the ranks (what BASELINE config 4's 5 M unknowns on eight GPUs need: replicated, the
    top fronts alone exceed a GPU): every rank keeps the whole pivot block of a top node and its slice of the boundary rows; the
    children's update matrices travel in row chunks through the staging buffer (forced small here: many steps), the sweeps
    exchange slices level by level.  Forced onto every top node (LSA_ND_DIST_MIN=1) or onto the large ones with replicated small
    ones below (the mixed form), several chunks per level (LSA_ND_WORK_MB).  Same eigenvalues as the oracle / the single-GPU
    solve, ranks bit-identical, no inner iteration."""
    import socket

    import torch.multiprocessing as mp

    from oracle import shift_invert
    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), case, "lu", env), nprocs=world, join=True)
    out = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in out[1:]:
        assert np.array_equal(o["lam"], out[0]["lam"]) and np.array_equal(o["V"], out[0]["V"])
    cube = case.startswith("C")
    es = fem.cube_case(case) if cube else fem.cylinder_case(case)
    sigma = fem.SIGMA_CUBE if cube else fem.SIGMA_RE50
    if es.n <= 12000:
        ref, _, _ = shift_invert.solve(es.A, es.M, sigma, k=5, tol=1e-13)
    else:
        one = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=5, atol=1e-10, ncv=40), check_hermitian=False)
        one.solver.set_st_type(iSTType.SINVERT)
        one.solver.set_target(sigma)
        one.solver.set_st_pc_type(PreconditionerType.LU)
        ref = np.array([p[0] for p in one.solve()[:5]])
        one.solver.release()
    assert helpers_match(out[0]["lam"], ref).max() <= 1e-8
    assert out[0]["res"].max() <= 1e-8 and int(out[0]["gmres"]) == 0 and int(out[0]["ranks"]) == world
    assert shift_invert.compute_residuals(es.A, es.M, out[0]["lam"], out[0]["V"]).max() <= 1e-8
    # exchanges per apply: the 2 / 4 of the layout, plus two per level of distributed nodes that have a boundary (a root has none)
    per_apply = 4 if cube else 2
    assert int(out[0]["gathers"]) >= per_apply * int(out[0]["applies"])
    print(case, world, env, "exchanges per apply:", int(out[0]["gathers"]) / int(out[0]["applies"]))


def helpers_match(found, ref):
    return np.array([np.min(np.abs(found - r)) / abs(r) for r in ref])


def _rank_adjoint(rank: int, world: int, port: int, out_dir: str, case: str = "S2k", env: str = "") -> None:
    import os
    import sys
    from pathlib import Path

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for item in filter(None, env.split(",")):
        key, _, val = item.partition("=")
        os.environ[key] = val
    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root), str(root / "lsa-fw_amd"), str(root / "tests")]
    import torch.distributed as dist

    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    dist.init_process_group("gloo", rank=rank, world_size=world)
    cube = case.startswith("C")
    es = fem.cube_case(case) if cube else fem.cylinder_case(case)
    solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=4, atol=1e-10, ncv=40), check_hermitian=False, layout="sharded", adjoint=True)
    solver.solver.set_st_type(iSTType.SINVERT)
    solver.solver.set_target(np.conj(-5.0 + 0.5j if cube else fem.SIGMA_RE50))
    solver.solver.set_st_pc_type(PreconditionerType.LU)
    pairs = solver.solve()
    forest = solver.solver._prepared["forest"]
    np.savez(Path(out_dir) / f"adj{rank}.npz", lam=np.array([p[0] for p in pairs[:4]]), V=np.column_stack([p[1].as_array() for p in pairs[:4]]),
             res=solver.solver.residuals()[:4], gmres=solver.solver.stats["gmres_iters"], ndist=int((forest.owner == -2).sum()))
    solver.solver.release()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_adjoint_solve(tmp_path, world):
    """The adjoint eigenproblem (left eigenvectors: ``Sensitivity/__init__.py:247-274``) in the subtree-parallel layout: the
    transposed sweeps on each rank's subtrees with the same two exchanges as the forward solve, transposed products with the
    whole matrices.  Eigenvalues = the conjugates of the direct ones, vectors = eigenvectors of (A^H, M^H), ranks bit-identical."""
    import socket

    import torch.multiprocessing as mp

    from oracle import shift_invert
    from synthetic import fem

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_adjoint, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    out = [np.load(tmp_path / f"adj{r}.npz") for r in range(world)]
    for o in out[1:]:
        assert np.array_equal(o["lam"], out[0]["lam"]) and np.array_equal(o["V"], out[0]["V"])
    es = fem.cylinder_case("S2k")
    ref, _, _ = shift_invert.solve(es.A, es.M, fem.SIGMA_RE50, k=4, tol=1e-13)
    for r in ref:
        assert np.min(np.abs(out[0]["lam"] - np.conj(r))) <= 1e-8 * abs(r)
    AH, MH = es.A.conj().T.tocsr(), es.M.conj().T.tocsr()
    assert shift_invert.compute_residuals(AH, MH, out[0]["lam"], out[0]["V"]).max() <= 1e-8
    assert out[0]["res"].max() <= 1e-8 and int(out[0]["gmres"]) == 0


@pytest.mark.parametrize("world,case,env", [(2, "S2k", "LSA_ND_DIST_MIN=1"), (3, "C9k", "LSA_ND_DIST_MIN=600"), (4, "C9k", "LSA_ND_DIST_MIN=1")])
def test_sharded_adjoint_solve_with_distributed_top_fronts(tmp_path, world, case, env):
    """The adjoint eigenproblem with the top fronts of the forest distributed over the ranks (round 4, second half): a distributed
    node's transposed sweeps sum over the rows a rank holds, the partial results are exchanged and added in rank order
    (``nd_sweepT_kernel`` with row ranges, ``nd_distT_finish_kernel``).  Eigenvalues = the conjugates of the direct problem's,
    vectors = eigenvectors of (A^H, M^H) to 1e-8, no inner iteration, ranks bit-identical."""
    import socket

    import torch.multiprocessing as mp

    from oracle import shift_invert
    from synthetic import fem

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_adjoint, args=(world, port, str(tmp_path), case, env), nprocs=world, join=True)
    out = [np.load(tmp_path / f"adj{r}.npz") for r in range(world)]
    assert int(out[0]["ndist"]) > 0  # the forest has distributed nodes
    for o in out[1:]:
        assert np.array_equal(o["lam"], out[0]["lam"]) and np.array_equal(o["V"], out[0]["V"])
    cube = case.startswith("C")
    es = fem.cube_case(case) if cube else fem.cylinder_case(case)
    sigma = -5.0 + 0.5j if cube else fem.SIGMA_RE50
    ref, _, _ = shift_invert.solve(es.A, es.M, sigma, k=4, tol=1e-13)
    for r in ref:
        assert np.min(np.abs(out[0]["lam"] - np.conj(r))) <= 1e-8 * abs(r)
    AH, MH = es.A.conj().T.tocsr(), es.M.conj().T.tocsr()
    assert shift_invert.compute_residuals(AH, MH, out[0]["lam"], out[0]["V"]).max() <= 1e-8
    assert out[0]["res"].max() <= 1e-8 and int(out[0]["gmres"]) == 0


def test_rccl_plumbing_with_one_rank(hip_ctx):
    """What a one-GPU box can run of the RCCL path: dlopen, ncclGetUniqueId, ncclCommInitRank (one rank) on the library's
    device, one in-place ncclAllGather on the library's stream, destroy -- the call sequence and signatures the sharded
    layouts use with more ranks (`lsa_comm_selftest`)."""
    hip_ctx.comm_selftest(1 << 20)
    hip_ctx.comm_selftest(4097)
    uid = hip_ctx.unique_id()
    assert len(uid) == 128 and any(uid)


def _nccl_world_of_one(_index: int, port: int, out: str) -> None:
    import os
    import sys
    from pathlib import Path

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root), str(root / "lsa-fw_amd")]
    import torch
    import torch.distributed as dist

    import lsa_hip
    from Solver.utils import _dist_comm_init

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)  # PyTorch's own RCCL is loaded and has a communicator on this device
    ctx = lsa_hip.Context(0)
    ctx.comm_selftest(1 << 16)  # the library's RCCL (the copy already in the process) beside it
    _dist_comm_init(ctx, 1, 0)  # the bootstrap the sharded layouts run: probe, agreement, id broadcast, init
    Path(out).write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_beside_the_nccl_backend_of_torch(tmp_path):
    """bench.py --gpus N runs under torch.distributed's nccl backend: the library's communicator then lives beside
    PyTorch's in one process.  One rank, in a child process (a process group per test process would outlive the test)."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "done"
    mp.spawn(_nccl_world_of_one, args=(port, str(out)), nprocs=1, join=True)
    assert out.read_text() == "ok"
