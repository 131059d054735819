import sys, os, cProfile, pstats, time
sys.path[:0]=['/root/repo','/root/repo/lsa-fw_amd']
os.environ.setdefault("LSA_HOST_BLAS_THREADS","1"); os.environ.setdefault("OPENBLAS_NUM_THREADS","1")
from synthetic import fem
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iSTType
es=fem.cylinder_case("S30k")
s=EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT); s.solver.set_target(fem.SIGMA_RE50); s.solver.set_st_pc_type(PreconditionerType.LU)
for _ in range(3): s.solve()
pr=cProfile.Profile(); pr.enable()
for _ in range(5): s.solve()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
