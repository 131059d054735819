"""Golden vector of BASELINE config 3's workload: the oracle's 20 eigenvalues nearest the Re = 50 target on the 500 k-unknown
cylinder pair (scipy ARPACK + SuperLU, about ten minutes and 10 GB on one core; kept apart from make_golden.py for that
reason).  Writes tests/golden/cylinder_s500k_k20.json."""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import fem, shift_invert  # noqa: E402

t0 = time.time()
es = fem.cylinder_case("S500k")
print(f"assembled n={es.n} nnz={es.A.nnz} in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
lam, V, res = shift_invert.solve(es.A, es.M, fem.SIGMA_RE50, k=20, tol=1e-12, ncv=80)
print(f"oracle solve {time.time() - t0:.1f} s, max residual {res.max():.2e}", flush=True)
h = hashlib.sha256()
for arr in (es.A.indptr, es.A.indices, np.round(es.A.data, 10), np.round(es.M.data, 10)):
    h.update(np.ascontiguousarray(arr).tobytes())
(Path(__file__).resolve().parent / "cylinder_s500k_k20.json").write_text(json.dumps({
    "case": "S500k", "re": 50.0, "n": es.n, "nnz": int(es.A.nnz), "sigma": [fem.SIGMA_RE50.real, fem.SIGMA_RE50.imag], "k": 20, "ncv": 80,
    "matrix_sha256_rounded_1e-10": h.hexdigest(), "eigenvalues": [[float(z.real), float(z.imag)] for z in lam], "max_residual": float(res.max()),
}, indent=1))
print("written", flush=True)
