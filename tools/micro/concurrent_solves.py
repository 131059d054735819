"""Throughput of several eigensolves in flight on ONE GPU (one Python thread + one HIP context/stream each)."""
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402

import bench  # noqa: E402
from synthetic import fem  # noqa: E402


class Args:
    k, atol, ncv, restart, ilu_levels = 20, 1e-10, 80, 1000, 2


es = fem.cylinder_case("S30k")
for jobs in (1, 2, 3, 4):
    solvers = []
    for j in range(jobs):
        s = bench.build_solver(es, bench.SWEEP_SIGMAS[2], Args, 0, "lu")
        s.solver.prepare()
        s.solve()
        solvers.append(s)
    reps = 6
    done = [0] * jobs

    def work(j):
        for _ in range(reps):
            done[j] += len(solvers[j].solve())

    threads = [threading.Thread(target=work, args=(j,)) for j in range(jobs)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    print(f"{jobs} solves in flight: {sum(done) / dt:7.1f} eigenpairs/s  ({dt / reps * 1e3:.0f} ms per round of {jobs})", flush=True)
    for s in solvers:
        s.solver.release()
