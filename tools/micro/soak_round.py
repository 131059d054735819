"""Re-run ONE round of tools/soak_nd.py (same random stream) and look at it closely."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402

from synthetic import fem  # noqa: E402
from Solver.eigen import EigenSolver, EigensolverConfig  # noqa: E402
from Solver.utils import PreconditionerType, iSTType  # noqa: E402

want = int(sys.argv[1])
es = fem.cylinder_case("S30k")
rng = np.random.default_rng(2026)
for r in range(want + 1):
    sigma = fem.SIGMA_RE50 + 0.05 * (rng.standard_normal() + 1j * rng.standard_normal())
solver = EigenSolver(es.A, es.M, EigensolverConfig(num_eig=20, atol=1e-10, ncv=80, max_it=500), check_hermitian=False)
inner = solver.solver
inner.set_st_type(iSTType.SINVERT)
inner.set_st_pc_type(PreconditionerType.LU)
inner.set_target(complex(sigma))
pairs = solver.solve()
res = inner.residuals()[: len(pairs)]
print("sigma", complex(sigma), "pairs", len(pairs), "stats", inner.stats)
print("residuals", np.array2string(res, precision=2))
lam = np.array([p[0] for p in pairs])
print("lambda", np.array2string(lam[:6], precision=6))
print("|lambda - sigma|", np.array2string(np.abs(lam - sigma)[:8], precision=4))
