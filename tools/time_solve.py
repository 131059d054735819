"""Phase timing of repeated eigen-solves of one case through the drop-in API (development aid)."""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
os.environ.setdefault("LSA_HOST_BLAS_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np  # noqa: E402

from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="S30k")
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--ncv", type=int, default=80)
ap.add_argument("--reps", type=int, default=8)
args = ap.parse_args()
es = fem.cube_case(args.case) if args.case.startswith("C") else fem.cylinder_case(args.case)
sigma = fem.SIGMA_CUBE if args.case.startswith("C") else fem.SIGMA_RE50
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=args.k, atol=1e-10, ncv=args.ncv), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT)
s.solver.set_target(sigma)
s.solver.set_st_pc_type(PreconditionerType.LU)
t0 = time.perf_counter()
s.solver.prepare()
print(f"prepare {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for r in range(args.reps):
    t0 = time.perf_counter()
    pairs = s.solve()
    dt = time.perf_counter() - t0
    st = s.solver.stats
    print(f"solve {1e3 * dt:7.2f} ms  pairs {len(pairs)}  applies {st['op_applies']}  factor {1e3 * st['seconds_factor']:.2f}  expand {1e3 * st.get('seconds_expand', 0):.2f}  "
          f"dense {1e3 * st.get('seconds_dense', 0):.2f}  restart+ritz {1e3 * st.get('seconds_restart', 0):.2f}  restarts {st['krylov_restarts']}", flush=True)
