"""GPU tests that need at least TWO devices: the sharded layouts over RCCL (`comm=PETSc.COMM_WORLD` of the reference,
Solver/utils.py:196-203, becomes one process per GPU and in-place ncclAllGather over xGMI).  On a one-GPU box every test
here is skipped; the same device path runs there with several ranks on one GPU over the host-staged transport
(tests/test_gpu_sharded.py).  The parent process never touches a GPU: `torch.cuda.device_count()` does not initialise
the runtime, and every rank is a freshly spawned child."""

import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_count() -> int:
    try:
        import torch

        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


needs_two = pytest.mark.skipif(_device_count() < 2, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_rccl(rank: int, world: int, port: int, out_dir: str, case: str, batch: str, dist_min: str) -> None:
    """One rank on device `rank`: nccl process group, the library's own RCCL communicator, one sharded solve."""
    import os
    import sys
    from pathlib import Path

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if batch:
        os.environ["LSA_KRYLOV_BATCH"] = batch
    if dist_min:
        os.environ["LSA_ND_DIST_MIN"] = dist_min
    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root), str(root / "lsa-fw_amd"), str(root / "tests")]
    import torch
    import torch.distributed as dist

    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    es = fem.cube_case(case) if case.startswith("C") else fem.cylinder_case(case)
    sigma = fem.SIGMA_CUBE if case.startswith("C") else fem.SIGMA_RE50
    k = 10 if case.startswith("C") else 20
    solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=k, atol=1e-10, ncv=4 * k), check_hermitian=False, layout="sharded", device=rank)
    solver.solver.set_st_type(iSTType.SINVERT)
    solver.solver.set_target(sigma)
    solver.solver.set_st_pc_type(PreconditionerType.LU)
    pairs = solver.solve()
    st = solver.solver.stats
    np.savez(Path(out_dir) / f"rank{rank}_{batch or 'b'}.npz", lam=np.array([p[0] for p in pairs[:k]]), V=np.column_stack([p[1].as_array() for p in pairs[:k]]),
             res=solver.solver.residuals()[:k], gmres=st["gmres_iters"], applies=st["op_applies"], gathers=st["allgather_calls"], ranks=st["ranks"],
             transport=str(st.get("transport", "")))
    solver.solver.release()
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, world, case, batch="", dist_min=""):
    import torch.multiprocessing as mp

    mp.spawn(_rank_rccl, args=(world, _free_port(), str(tmp_path), case, batch, dist_min), nprocs=world, join=True)
    return [np.load(tmp_path / f"rank{r}_{batch or 'b'}.npz") for r in range(world)]


@needs_two
@pytest.mark.parametrize("case", ["S30k", "S120k"])
def test_sharded_solve_over_rccl_two_devices(tmp_path, case):
    """Two ranks on two devices, the exact subtree-parallel LU: eigenvalues to 1e-8 of the golden fixture (S30k) / residual
    test (S120k), ranks bit-identical, batched Arnoldi steps (stream-ordered collectives between kernels) identical to one
    step per read-back."""
    import json
    from pathlib import Path

    batched = _run(tmp_path, 2, case)
    stepwise = _run(tmp_path, 2, case, batch="1")
    for out in (batched, stepwise):
        assert np.array_equal(out[0]["lam"], out[1]["lam"]) and np.array_equal(out[0]["V"], out[1]["V"])
        assert out[0]["res"].max() <= 1e-8 and int(out[0]["gmres"]) == 0 and int(out[0]["ranks"]) == 2
    gap = np.max(np.abs(batched[0]["lam"] - stepwise[0]["lam"]) / np.abs(stepwise[0]["lam"]))
    assert gap <= 1e-9, gap
    if case == "S30k":
        gold = json.loads((Path(__file__).resolve().parent / "golden" / "cylinder_s30k_k20.json").read_text())
        ref = np.array([complex(a, b) for a, b in gold["eigenvalues"]])
        for r in ref:
            assert np.min(np.abs(batched[0]["lam"] - r)) <= 1e-8 * abs(r)


@needs_two
def test_distributed_top_fronts_over_rccl(tmp_path):
    """The 3D discretisation with the top fronts of the forest forced into the row-distributed form (LSA_ND_DIST_MIN=1), over as
    many devices as the box has (at most four): the chunked update-matrix exchange and the per-level exchanges of the sweeps on
    RCCL.  Same eigenvalues as the replicated-top form."""
    world = min(_device_count(), 4)
    dist_on = _run(tmp_path, world, "C40k", dist_min="1")
    replicated = _run(tmp_path, world, "C40k", batch="1")
    for o in dist_on[1:]:
        assert np.array_equal(o["lam"], dist_on[0]["lam"]) and np.array_equal(o["V"], dist_on[0]["V"])
    assert dist_on[0]["res"].max() <= 1e-8 and int(dist_on[0]["gmres"]) == 0
    for r in replicated[0]["lam"]:
        assert np.min(np.abs(dist_on[0]["lam"] - r)) <= 1e-8 * abs(r)
