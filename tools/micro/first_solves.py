"""Timing of the first few solves after prepare(): which of them still pay one-time costs?"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import os  # noqa: E402

os.environ.setdefault("LSA_HOST_BLAS_THREADS", "1")
if "torch" in sys.argv:  # PyTorch first, as in bench.py (imported after the library it finds no GPU: two ROCm trees in one process)
    import torch

    torch.cuda.set_device(0)
    torch.cuda.synchronize()
from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

es = fem.cylinder_case(sys.argv[1] if len(sys.argv) > 1 else "S30k")
solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80, max_it=500), check_hermitian=False)
inner = solver.solver
inner.set_st_type(iSTType.SINVERT)
inner.set_target(fem.SIGMA_RE50)
inner.set_st_pc_type(PreconditionerType.LU)
t0 = time.perf_counter()
inner.prepare()
print(f"prepare {1e3 * (time.perf_counter() - t0):.1f} ms")
import cProfile  # noqa: E402
import pstats  # noqa: E402

for i in range(5):
    pr = cProfile.Profile() if (i == 1 and "profile1" in sys.argv) else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    solver.solve()
    if pr:
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
    dt = time.perf_counter() - t0
    st = inner.stats
    print(f"solve {i}: {1e3 * dt:7.1f} ms  factor {1e3 * st['seconds_factor']:6.1f}  arnoldi {1e3 * st['seconds_solve']:6.1f}  other {1e3 * (dt - st['seconds_factor'] - st['seconds_solve']):6.1f}  analysis_reused {st['analysis_reused']}")
if "torch" in sys.argv:
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.solve()
        print(f"after torch.cuda.synchronize(): solve {1e3 * (time.perf_counter() - t0):7.1f} ms")
    import gc

    gc.collect()
    t0 = time.perf_counter()
    solver.solve()
    print(f"after gc.collect(): solve {1e3 * (time.perf_counter() - t0):7.1f} ms")
if "profile" in sys.argv:
    import cProfile
    import pstats

    solver2 = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80, max_it=500), check_hermitian=False)
    i2 = solver2.solver
    i2.set_st_type(iSTType.SINVERT)
    i2.set_target(fem.SIGMA_RE50)
    i2.set_st_pc_type(PreconditionerType.LU)
    i2.prepare()
    solver2.solve()
    pr = cProfile.Profile()
    pr.enable()
    solver2.solve()  # the second solve of a fresh solver: the slow one under PyTorch's runtime
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
