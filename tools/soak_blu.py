"""Soak test: many block LU factorisations + solves in one process (different shifts, cache reuse, both sweep forms), every
result checked -- to shake out rare races in the fused Gauss-Jordan / twisted sweeps."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402
from Solver.utils import pivot_safe_rcm  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "S30k"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
es = fem.cylinder_case(case)
C0 = sp.csr_matrix((es.A.data - fem.SIGMA_RE50 * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
perm = pivot_safe_rcm(C0)
A = es.A[perm][:, perm].tocsr(); A.sort_indices()
M = es.M[perm][:, perm].tocsr(); M.sort_indices()
ctx = lsa_hip.Context(0)
rng = np.random.default_rng(0)
worst = 0.0
t0 = time.time()
for it in range(rounds):
    sigma = fem.SIGMA_RE50 + 0.02 * (rng.standard_normal() + 1j * rng.standard_normal())
    C = sp.csr_matrix((A.data - sigma * M.data, A.indices, A.indptr), shape=A.shape)
    os.environ["LSA_BLU_ABSORB"] = "1" if it % 3 else "0"
    dC = lsa_hip.CsrMatrix.from_scipy(ctx, C)
    f = lsa_hip.BlockLu(ctx, dC)
    for _ in range(3):
        b = rng.standard_normal(es.n) + 1j * rng.standard_normal(es.n)
        dx = lsa_hip.DeviceVector(ctx, es.n, np.complex128)
        f.solve(lsa_hip.DeviceVector.from_numpy(ctx, b), dx)
        r = np.linalg.norm(C @ dx.numpy() - b) / np.linalg.norm(b)
        worst = max(worst, r)
        if not (r <= 1e-11):
            print(f"round {it}: relative residual {r:.3e} at sigma {sigma}", flush=True)
            sys.exit(1)
    del f, dC
    if it % 10 == 9:
        print(f"{it + 1} factorisations, worst relative residual so far {worst:.2e}, {time.time() - t0:.1f} s", flush=True)
print("soak ok", worst)
