"""Wall-clock phases of one LU-path eigensolve (development aid)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import cProfile  # noqa: E402
import pstats  # noqa: E402

from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

es = fem.cylinder_case(sys.argv[1] if len(sys.argv) > 1 else "S30k")
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT)
s.solver.set_target(fem.SIGMA_RE50)
s.solver.set_st_pc_type(PreconditionerType.LU)
s.solver.prepare()
s.solve()
pr = cProfile.Profile()
t = time.time()
pr.enable()
s.solve()
pr.disable()
print("solve wall", time.time() - t, s.solver.stats)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
