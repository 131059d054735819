"""Time one shift-invert eigensolve on the GPU and print the solver counters (development aid)."""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]

import numpy as np  # noqa: E402

from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="S5k")
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--levels", type=int, default=2)
ap.add_argument("--atol", type=float, default=1e-10)
ap.add_argument("--restart", type=int, default=1000)
ap.add_argument("--sigma-real", action="store_true")
args = ap.parse_args()

t0 = time.time()
es = fem.cylinder_case(args.case)
print(f"assembled {args.case}: n={es.n} nnz={es.A.nnz} in {time.time()-t0:.1f}s", flush=True)
sigma = 0.05 if args.sigma_real else fem.SIGMA_RE50
solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=args.k, atol=args.atol, ncv=80), check_hermitian=False,
                     ilu_levels=args.levels, restart=args.restart)
solver.solver.set_st_type(iSTType.SINVERT)
solver.solver.set_target(sigma)
solver.solver.set_st_pc_type(PreconditionerType.ILU)
t0 = time.time()
pairs = solver.solve()
dt = time.time() - t0
st = solver.solver.stats
print(f"solve {dt:.2f}s pairs={len(pairs)} stats={st}", flush=True)
its = max(st["gmres_iters"], 1)
print(f"inner its/apply {its/max(st['op_applies'],1):.1f}; solve-seconds per inner iteration {st['seconds_solve']/its*1e6:.1f} us; "
      f"factor {st['seconds_factor']:.3f}s", flush=True)
