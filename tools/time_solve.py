"""cProfile of one warm EigenSolver.solve() on the bench workload (development aid)."""
import argparse
import cProfile
import pstats
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
ap.add_argument("--case", default="S30k")
ap.add_argument("--pc", default="lu")
ap.add_argument("--plain", action="store_true", help="no cProfile pass: the last thing the process does is one plain warm solve")
ap.add_argument("--torch", action="store_true", help="initialise PyTorch's HIP context first, as bench.py does")
args = ap.parse_args()
if args.torch:
    import torch

    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch initialised; threads", torch.get_num_threads())
es = fem.cylinder_case(args.case)
cfg = EigensolverConfig(num_eig=20, atol=1e-10, ncv=80, max_it=500)
pc = PreconditionerType.LU if args.pc == "lu" else PreconditionerType.ILU
solver = EigenSolver(es.A, es.M, cfg, check_hermitian=False)
inner = solver.solver
inner.set_st_type(iSTType.SINVERT)
inner.set_target(fem.SIGMA_RE50)
inner.set_st_pc_type(pc)
inner.prepare()
inner.solve()
for _ in range(3):
    t0 = time.time()
    inner.solve()
    print(f"warm solve {time.time() - t0:.3f} s  factor {inner.stats['seconds_factor']:.3f} solve {inner.stats['seconds_solve']:.3f}")
t0 = time.time()
solver.solve()
print(f"warm EigenSolver.solve() {time.time() - t0:.3f} s")
if args.plain:
    t0 = time.time()
    inner.solve()
    print(f"last solve {1e3 * (time.time() - t0):.1f} ms; stats {inner.stats}")
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
solver.solve()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
