"""Which host-side activity between two solves slows the next factorisation?  Development aid."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402

from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

es = fem.cylinder_case("S30k")
solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80, max_it=500), check_hermitian=False)
inner = solver.solver
inner.set_st_type(iSTType.SINVERT)
inner.set_target(fem.SIGMA_RE50)
inner.set_st_pc_type(PreconditionerType.LU)
inner.prepare()
inner.solve()
big = np.random.default_rng(0).standard_normal((30876, 20)) + 0j


def run(label, between):
    for _ in range(3):
        between()
        t0 = time.perf_counter()
        inner.solve()
        print(f"{label:34s}: solve {1e3 * (time.perf_counter() - t0):6.1f} ms, factor {1e3 * inner.stats['seconds_factor']:5.1f} ms", flush=True)


run("nothing", lambda: None)
run("get_all_eigenpairs_up_to(20)", lambda: list(inner.get_all_eigenpairs_up_to(20)))
run("nothing again", lambda: None)
run("20 x numpy norm of 30k complex", lambda: [np.linalg.norm(big[:, j]) for j in range(20)])
run("nothing again", lambda: None)
run("20 x vdot (BLAS zdotc)", lambda: [np.vdot(big[:, j], big[:, j]) for j in range(20)])
run("nothing again", lambda: None)
run("matmul 2000x2000 (threaded BLAS)", lambda: np.ones((2000, 2000)) @ np.ones((2000, 2000)))
run("nothing again", lambda: None)
