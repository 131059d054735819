"""Factor + solve timing of the exact block LU against the direct (SuperLU) answer (development aid)."""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as spla  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402
from Solver.utils import pivot_safe_rcm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="S5k")
ap.add_argument("--block", type=int, default=0)
args = ap.parse_args()
es = fem.cylinder_case(args.case)
C = sp.csr_matrix((es.A.data - fem.SIGMA_RE50 * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
perm = pivot_safe_rcm(C)
Cp = C[perm][:, perm].tocsr()
Cp.sort_indices()
ctx = lsa_hip.Context(0)
dC = lsa_hip.CsrMatrix.from_scipy(ctx, Cp)
t0 = time.time()
f = lsa_hip.BlockLu(ctx, dC, args.block)
print(f"{args.case}: n={es.n} factor {time.time() - t0:.3f}s info={f.info()}", flush=True)
b = np.random.default_rng(0).standard_normal(es.n) + 1j * np.random.default_rng(1).standard_normal(es.n)
db = lsa_hip.DeviceVector.from_numpy(ctx, b)
dx = lsa_hip.DeviceVector(ctx, es.n, np.complex128)
f.solve(db, dx)
x = dx.numpy()
print("relative residual", np.linalg.norm(Cp @ x - b) / np.linalg.norm(b), flush=True)
if es.n < 40000:
    xr = spla.splu(Cp.tocsc()).solve(b)
    print("vs SuperLU", np.linalg.norm(x - xr) / np.linalg.norm(xr), flush=True)
print(f"solve {f.time_solve(db, dx, 100) * 1e3:.1f} us per apply", flush=True)
