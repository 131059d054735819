"""Time the preconditioner apply (L then U solve) for each SpTRSV algorithm / block size (development aid)."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402
from Solver.utils import pivot_safe_rcm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="S30k")
ap.add_argument("--levels", type=int, default=2)
ap.add_argument("--blocks", default="512,1024,2048,4096")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--real", action="store_true")
args = ap.parse_args()
es = fem.cylinder_case(args.case)
sigma = 0.05 if args.real else fem.SIGMA_RE50
C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
perm = pivot_safe_rcm(C)
Cp = C[perm][:, perm].tocsr()
Cp.sort_indices()
ctx = lsa_hip.Context(0)
dC = lsa_hip.CsrMatrix.from_scipy(ctx, Cp)
pc = lsa_hip.Ilu(ctx, dC, levels=args.levels)
info = pc.info()
n = es.n
b = lsa_hip.DeviceVector.from_numpy(ctx, np.random.default_rng(0).standard_normal(n) + 0j)
x = lsa_hip.DeviceVector(ctx, n, np.complex128)
esz = 16 if Cp.dtype.kind == "c" else 8
print(f"{args.case}: n={n} factor nnz={info['nnz']} levels={info['levels_lower']}/{info['levels_upper']}", flush=True)
for B in [int(t) for t in args.blocks.split(",")]:
    pc.set_algorithm(2, B)
    ms = pc.time_solve(b, x, args.iters)
    dense_bytes = 2 * n * (B / 2) * esz
    print(f"blocked B={B:5d}: {ms * 1e3:8.1f} us per apply; dense bytes {dense_bytes / 1e6:7.1f} MB -> {dense_bytes / ms / 1e6:7.1f} GB/s; "
          f"{2 * 2 * ((n + B - 1) // B)} launches", flush=True)
pc.set_algorithm(1)
ms = pc.time_solve(b, x, 5)
print(f"sync-free : {ms * 1e3:8.1f} us per apply", flush=True)
dx = lsa_hip.DeviceVector.from_numpy(ctx, np.random.default_rng(1).standard_normal(n) + 0j)
dy = lsa_hip.DeviceVector(ctx, n, np.complex128)
print(f"spmv C    : {dC.time_matvec(dx, dy, 500) * 1e3:8.1f} us", flush=True)
