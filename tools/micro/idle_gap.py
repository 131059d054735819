"""Does an idle gap before a factorisation slow it down (clock / power state)?  Development aid."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402
from Solver.utils import pivot_safe_rcm  # noqa: E402

es = fem.cylinder_case("S30k")
C = sp.csr_matrix((es.A.data - fem.SIGMA_RE50 * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
perm = pivot_safe_rcm(C)
Cp = C[perm][:, perm].tocsr()
Cp.sort_indices()
ctx = lsa_hip.Context(0)
dC = lsa_hip.CsrMatrix.from_scipy(ctx, Cp)
for gap_ms in (0, 0, 1, 5, 10, 20, 50, 200, 0, 0):
    time.sleep(gap_ms * 1e-3)
    t0 = time.perf_counter()
    f = lsa_hip.NdLu(ctx, dC, 0)
    dt = time.perf_counter() - t0
    del f
    print(f"idle {gap_ms:4d} ms before -> factorisation {dt * 1e3:.1f} ms", flush=True)
