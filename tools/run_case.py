"""One eigensolve of a named synthetic case with either inner solver; prints timings, counters and true residuals."""
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
ap.add_argument("--case", default="S120k")
ap.add_argument("--pc", default="lu")
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--atol", type=float, default=1e-10)
ap.add_argument("--levels", type=int, default=2)
ap.add_argument("--ncv", type=int, default=80)
args = ap.parse_args()
import threading  # noqa: E402


def _heartbeat():  # (a 2 M-unknown 3D assembly is minutes of silent numpy: the GPU box ends commands that say nothing for 7 minutes)
    while True:
        time.sleep(60)
        print(f"... {time.time() - T_START:.0f}s", flush=True)


T_START = time.time()
threading.Thread(target=_heartbeat, daemon=True).start()
t0 = time.time()
es = fem.cube_case(args.case) if args.case.startswith("C") else fem.cylinder_case(args.case)
sigma = fem.SIGMA_CUBE if args.case.startswith("C") else fem.SIGMA_RE50
print(f"{args.case}: n={es.n} nnz={es.A.nnz} assembled in {time.time() - t0:.1f}s", flush=True)
kw = {"ilu_levels": args.levels} if args.pc == "ilu" else {}
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=args.k, atol=args.atol, ncv=args.ncv), check_hermitian=False, **kw)
s.solver.set_st_type(iSTType.SINVERT)
s.solver.set_target(sigma)
s.solver.set_st_pc_type(PreconditionerType.LU if args.pc == "lu" else PreconditionerType.ILU)
t0 = time.time()
s.solver.prepare()
print(f"prepare (ordering + upload) {time.time() - t0:.2f}s", flush=True)
t0 = time.time()
pairs = s.solve()
dt = time.time() - t0
res = s.solver.residuals()
print(f"solve {dt:.2f}s -> {len(pairs) / dt:.2f} eigenpairs/s; converged {len(pairs)}; max residual {res.max():.2e}", flush=True)
print("stats", s.solver.stats, flush=True)
print("lambda[:5]", [complex(p[0]) for p in pairs[:5]], flush=True)
t0 = time.time()
s.solve()
print(f"second solve {time.time() - t0:.2f}s; stats {s.solver.stats}", flush=True)
