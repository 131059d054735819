"""One sharded eigen-solve with several ranks on ONE GPU (host-staged exchange over gloo): rehearsal of the subtree-parallel layout at
sizes beyond the test suite's, e.g. the default thresholds of the distributed top on the 3D case.  Says nothing about xGMI scaling.

    python tools/run_dist_one_gpu.py --case C300k --ranks 4 [--env LSA_ND_DIST_MIN=3000]
"""
import argparse
import json
import os
import socket
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd"), str(ROOT / "tests")]


def rank_main(rank, world, port, out_dir, case, k, env):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for item in filter(None, env.split(",")):
        key, _, val = item.partition("=")
        os.environ[key] = val
    if case.startswith("C"):
        os.environ.setdefault("LSA_DIST_SPMV", "shard")
    import numpy as np
    import torch.distributed as dist

    from synthetic import fem
    from Solver.eigen import EigenSolver, EigensolverConfig
    from Solver.utils import PreconditionerType, iSTType

    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:  # a line a minute: a long rehearsal must not look hung
        import threading

        phase = ["assembling"]

        def beat():
            t_start = time.time()
            while phase[0] != "done":
                time.sleep(45)
                print(f"[run_dist_one_gpu] {time.time() - t_start:.0f} s: {phase[0]}", file=sys.stderr, flush=True)

        threading.Thread(target=beat, daemon=True).start()
    else:
        phase = [""]
    t0 = time.time()
    es = fem.cube_case(case) if case.startswith("C") else fem.cylinder_case(case)
    sigma = fem.SIGMA_CUBE if case.startswith("C") else fem.SIGMA_RE50
    t_asm = time.time() - t0
    s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=k, atol=1e-10, ncv=4 * k), check_hermitian=False, layout="sharded")
    s.solver.set_st_type(iSTType.SINVERT)
    s.solver.set_target(sigma)
    s.solver.set_st_pc_type(PreconditionerType.LU)
    t0 = time.time()
    phase[0] = "analysis"
    s.solver.prepare()
    t_prep = time.time() - t0
    forest = s.solver._prepared["forest"]
    t0 = time.time()
    phase[0] = "factorisation and solve"
    pairs = s.solve()
    t_solve = time.time() - t0
    st = s.solver.stats
    res = s.solver.residuals()[:k]
    rec = {"rank": rank, "n": es.n, "assemble_s": t_asm, "prepare_s": t_prep, "solve_s": t_solve, "pairs": len(pairs), "max_residual": float(res.max()),
           "top_nodes_distributed": int((forest.owner == -2).sum()), "top_nodes_replicated": int((forest.owner == -1).sum()),
           "stats": {kk: (float(v) if isinstance(v, (int, float)) else str(v)) for kk, v in st.items()},
           "lambda": [[float(p[0].real), float(p[0].imag)] for p in pairs[:k]]}
    (Path(out_dir) / f"rank{rank}.json").write_text(json.dumps(rec))
    phase[0] = "done"
    s.solver.release()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="C160k")
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--env", default="")
    args = ap.parse_args()
    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(rank_main, args=(args.ranks, port, tmp, args.case, args.k, args.env), nprocs=args.ranks, join=True)
        recs = [json.loads((Path(tmp) / f"rank{r}.json").read_text()) for r in range(args.ranks)]
    same = all(r["lambda"] == recs[0]["lambda"] for r in recs)
    out = {"case": args.case, "ranks": args.ranks, "env": args.env, "ranks_bit_identical": same, "rank0": recs[0],
           "solve_s_per_rank": [r["solve_s"] for r in recs], "factor_s_per_rank": [r["stats"].get("seconds_factor") for r in recs],
           # what must be the same on every rank: a difference says that the ranks took different turns (collectives out of step)
           "per_rank": [{k: r["stats"].get(k) for k in ("op_applies", "refined_solves", "allgather_calls", "max_rel_res", "last_rel_res", "krylov_restarts")}
                        | {"max_residual": r["max_residual"], "lambda0": r["lambda"][0] if r["lambda"] else None} for r in recs]}
    print(json.dumps(out))
