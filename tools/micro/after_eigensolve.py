"""Why is the solve that follows an EigenSolver.solve() slow?  Variants: result kept / dropped / dropped + gc / sleep."""
import gc
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
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
inner.solve()


def timed(label, fn):
    t0 = time.perf_counter()
    out = fn()
    print(f"{label:46s} {1e3 * (time.perf_counter() - t0):7.1f} ms  (solve phase {1e3 * inner.stats['seconds_solve']:.1f}, analysis reused {inner.stats['analysis_reused']})", flush=True)
    return out


timed("inner.solve()", inner.solve)
kept = timed("EigenSolver.solve(), result kept", solver.solve)
timed("inner.solve() after it", inner.solve)
timed("EigenSolver.solve(), result dropped", lambda: (solver.solve(), None)[1])
timed("inner.solve() after it", inner.solve)
timed("inner.solve() again", inner.solve)
timed("EigenSolver.solve(), dropped, then gc", lambda: (solver.solve(), gc.collect())[1])
timed("inner.solve() after it", inner.solve)
kept2 = timed("EigenSolver.solve() kept #2", solver.solve)
kept3 = timed("EigenSolver.solve() kept #3", solver.solve)
kept3 = timed("EigenSolver.solve() rebinding #4", solver.solve)
kept3 = timed("EigenSolver.solve() rebinding #5", solver.solve)
