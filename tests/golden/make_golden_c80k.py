"""Regenerates tests/golden/cube_c80k.json: the oracle's 10 eigenvalues nearest the 3D target on the C80k unit-cube pair
(BASELINE config 4's discretisation, .examples/cube.py:37, at 76.5 k unknowns: SuperLU needs minutes and several GB here, which
is why this fixture has a script of its own).  Run in the build container; needs only numpy / scipy."""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import shift_invert  # noqa: E402
from synthetic import fem  # noqa: E402

HERE = Path(__file__).resolve().parent
t0 = time.time()
es = fem.cube_case("C80k")
print(f"C80k: n={es.n} nnz={es.A.nnz} assembled in {time.time() - t0:.0f}s", flush=True)
t0 = time.time()
lam, V, res = shift_invert.solve(es.A, es.M, fem.SIGMA_CUBE, k=10, tol=1e-13, ncv=60)
print(f"oracle solve {time.time() - t0:.0f}s, max residual {res.max():.2e}", flush=True)
h = hashlib.sha256()
for arr in (es.A.indptr, es.A.indices, np.round(es.A.data, 10), np.round(es.M.data, 10)):
    h.update(np.ascontiguousarray(arr).tobytes())
(HERE / "cube_c80k.json").write_text(json.dumps({
    "case": "C80k", "re": 10.0, "n": es.n, "nnz": int(es.A.nnz), "sigma": [float(np.real(fem.SIGMA_CUBE)), float(np.imag(fem.SIGMA_CUBE))], "k": 10, "ncv": 60,
    "matrix_sha256_rounded_1e-10": h.hexdigest(), "eigenvalues": [[float(z.real), float(z.imag)] for z in lam], "max_residual": float(res.max()),
    "oracle": "oracle/shift_invert.py (scipy ARPACK + SuperLU), tol 1e-13",
}, indent=1))
print("written", flush=True)
