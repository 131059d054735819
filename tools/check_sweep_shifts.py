"""Every shift of bench.py's replica layout (one per rank at N = 8) on one GPU: converged pairs, residuals, time."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402

import bench  # noqa: E402
from synthetic import fem  # noqa: E402


class Args:
    k, atol, ncv, restart, ilu_levels = 20, 1e-10, 80, 1000, 2


es = fem.cylinder_case("S30k")
for rank in range(8):
    sigma = bench.SWEEP_SIGMAS[(2 + rank) % len(bench.SWEEP_SIGMAS)]
    solver = bench.build_solver(es, sigma, Args, 0, "lu")
    solver.solver.prepare()
    solver.solve()
    t0 = time.perf_counter()
    solver.solve()
    dt = time.perf_counter() - t0
    res = solver.solver.residuals()
    st = solver.solver.stats
    print(f"rank {rank} sigma {sigma}: {int(np.sum(res[:20] <= 1e-8))} pairs pass, max residual {res[:20].max():.1e}, {dt * 1e3:.0f} ms, "
          f"{st['op_applies']} applies, inner GMRES iterations {st['gmres_iters']}", flush=True)
    solver.solver.release()
