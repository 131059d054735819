"""Golden vector of BASELINE config 3's workload: the oracle's 20 eigenvalues nearest the Re = 50 target on the 500 k-unknown
cylinder pair, WITH the evidence a tolerance on them has to rest on: ARPACK run to machine precision (tol = 0), the true
residuals of the right and the left eigenpairs, the condition number of every eigenvalue (left eigenvectors from the adjoint
problem on the same SuperLU factors) and the two-sided Rayleigh quotients (second-order accurate eigenvalues).
scipy ARPACK + SuperLU, about twenty minutes and 10 GB on one core; kept apart from make_golden.py for that reason.
Writes tests/golden/cylinder_s500k_k20.json."""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import shift_invert  # noqa: E402
from synthetic import fem

case = sys.argv[1] if len(sys.argv) > 1 else "S500k"
t0 = time.time()
es = fem.cylinder_case(case)
print(f"assembled n={es.n} nnz={es.A.nnz} in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
r = shift_invert.solve_two_sided(es.A, es.M, fem.SIGMA_RE50, k=20, tol=0.0, ncv=80)
print(f"oracle solves {time.time() - t0:.1f} s, max residual {r['res'].max():.2e} (left {r['res_left'].max():.2e}), "
      f"kappa {r['kappa'].min():.2e} .. {r['kappa'].max():.2e}, |lam - lam_rq|/|lam| max {np.max(np.abs(r['lam'] - r['lam_rq']) / np.abs(r['lam'])):.2e}", flush=True)
print("left/right pairing gaps |conj(lam_adj) - lam| / |lam|:", np.array2string(r["pair_gap"], precision=1), "distinct:", r["distinct_left"], flush=True)
h = hashlib.sha256()
for arr in (es.A.indptr, es.A.indices, np.round(es.A.data, 10), np.round(es.M.data, 10)):
    h.update(np.ascontiguousarray(arr).tobytes())
pairs = lambda z: [[float(c.real), float(c.imag)] for c in z]  # noqa: E731
(Path(__file__).resolve().parent / f"cylinder_{case.lower()}_k20.json").write_text(json.dumps({
    "case": case, "re": 50.0, "n": es.n, "nnz": int(es.A.nnz), "sigma": [fem.SIGMA_RE50.real, fem.SIGMA_RE50.imag], "k": 20, "ncv": 80,
    "arpack_tol": 0.0, "matrix_sha256_rounded_1e-10": h.hexdigest(), "eigenvalues": pairs(r["lam"]), "eigenvalues_two_sided_rq": pairs(r["lam_rq"]),
    "residuals": [float(x) for x in r["res"]], "residuals_left": [float(x) for x in r["res_left"]], "kappa": [float(x) for x in r["kappa"]],
    "left_right_pairing_gap": [float(x) for x in r["pair_gap"]], "left_vectors_distinct": bool(r["distinct_left"]), "max_residual": float(r["res"].max()),
}, indent=1))
print("written", flush=True)
