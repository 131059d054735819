"""A few S30k eigen-solves in a row, nothing else: the workload for rocprofv3 A/B runs of the kernels of an Arnoldi step
(`rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/micro/solve_loop.py`)."""
import os, sys, time
from pathlib import Path
root = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(root), str(root / "lsa-fw_amd")]
from synthetic import fem
from Solver.eigen import EigenSolver, EigensolverConfig
from Solver.utils import PreconditionerType, iSTType

case = sys.argv[1] if len(sys.argv) > 1 else "S30k"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
es = fem.cylinder_case(case)
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT)
s.solver.set_target(fem.SIGMA_RE50)
s.solver.set_st_pc_type(PreconditionerType.LU)
for _ in range(2):
    s.solve()
t = time.perf_counter()
for _ in range(reps):
    pairs = s.solve()
dt = (time.perf_counter() - t) / reps
st = s.solver.stats
print(f"{case}: {1e3 * dt:.2f} ms per solve, expand {1e3 * st.get('seconds_expand', 0):.2f} dense {1e3 * st.get('seconds_dense', 0):.2f} "
      f"applies {st.get('op_applies')} restart {1e3 * st.get('seconds_restart', 0):.2f} factor {1e3 * st.get('seconds_factor', 0):.2f}; lambda0 {pairs[0][0]!r}")
