import cProfile
import os
import pstats
import sys

sys.path[:0] = ["/root/repo", "/root/repo/lsa-fw_amd"]
os.environ.setdefault("LSA_HOST_BLAS_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
from synthetic import fem
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iSTType

case = sys.argv[1] if len(sys.argv) > 1 else "S30k"
es = fem.cylinder_case(case)


def make():
    s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80), check_hermitian=False)
    s.solver.set_st_type(iSTType.SINVERT)
    s.solver.set_target(fem.SIGMA_RE50)
    s.solver.set_st_pc_type(PreconditionerType.LU)
    return s


s = make()
s.solver.prepare()
s.solve()
s.solver.release()
s = make()
pr = cProfile.Profile()
pr.enable()
s.solver.prepare()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
