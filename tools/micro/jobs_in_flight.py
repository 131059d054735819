"""How many independent S30k solves per second does ONE GPU deliver with J solver threads (own context + streams each)?"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "LSA_HOST_BLAS_THREADS"):
    os.environ.setdefault(v, "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[2] if len(sys.argv) > 2 else "8")
import threading  # noqa: E402

import numpy as np  # noqa: E402

from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

es = fem.cylinder_case("S30k")
J = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def make():
    s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80), check_hermitian=False)
    s.solver.set_st_type(iSTType.SINVERT)
    s.solver.set_target(fem.SIGMA_RE50)
    s.solver.set_st_pc_type(PreconditionerType.LU)
    s.solver.prepare()
    s.solve()
    s.solve()
    return s


solvers = [make() for _ in range(J)]
rounds = 6
done = [0] * J


def work(j):
    for _ in range(rounds):
        done[j] += len(solvers[j].solve())


th = [threading.Thread(target=work, args=(j,)) for j in range(J)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
print(f"J={J} queues={os.environ['GPU_MAX_HW_QUEUES']}: {sum(done) / dt:.1f} eigenpairs/s, {1e3 * dt / rounds:.1f} ms per round of {J} solves", flush=True)
