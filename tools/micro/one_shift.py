"""One eigen-solve at a given shift, eigenvalues and true residuals printed (development aid): one_shift.py CASE RE IM [K NCV]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402
case, sigma = sys.argv[1], complex(float(sys.argv[2]), float(sys.argv[3]))
k, ncv = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (10, 40)
es = fem.cube_case(case) if case.startswith("C") else fem.cylinder_case(case)
s = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=k, atol=1e-10, ncv=ncv, max_it=500), check_hermitian=False)
s.solver.set_st_type(iSTType.SINVERT); s.solver.set_st_pc_type(PreconditionerType.LU); s.solver.set_target(sigma)
pairs = s.solve()
res = s.solver.residuals()[: len(pairs)]
est = s.solver._residual_estimates[: len(pairs)] if hasattr(s.solver, "_residual_estimates") else [float("nan")] * len(pairs)
for (lam, _), r, e in zip(pairs, res, est):
    print(f"  lambda {complex(lam):.10f}  |lambda - sigma| {abs(complex(lam) - sigma):.4f}  true residual {r:.2e}  estimate {e:.2e}")
print({k_: v for k_, v in s.solver.stats.items() if k_ in ("op_applies", "max_rel_res", "backward_accepted", "krylov_restarts", "stagnated_solves")})
