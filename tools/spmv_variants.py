"""A/B the SpMV kernel variants on SROOF in one process (development aid; LSA_SPMV_VARIANT is read at every launch)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402
from Solver.utils import pivot_safe_rcm  # noqa: E402

case, reps = (sys.argv[1] if len(sys.argv) > 1 else "S500k"), int(sys.argv[2]) if len(sys.argv) > 2 else 10
es = fem.cylinder_case(case)
C = sp.csr_matrix((es.A.data - fem.SIGMA_RE50 * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
perm = pivot_safe_rcm(C)
C = C[perm][:, perm].tocsr()
C.sort_indices()
n1, nnz1 = C.shape[0], C.nnz
rp = np.concatenate([[0], (C.indptr[1:][None, :] + (np.arange(reps) * nnz1)[:, None]).ravel()]).astype(np.int32)
ci = (C.indices[None, :] + (np.arange(reps, dtype=np.int64) * n1)[:, None]).ravel().astype(np.int32)
big = sp.csr_matrix((np.tile(C.data, reps), ci, rp), shape=(n1 * reps, n1 * reps))
n, nnz = big.shape[0], big.nnz
ctx = lsa_hip.Context(0)
x = np.random.default_rng(0).standard_normal(n) + 1j * np.random.default_rng(1).standard_normal(n)
dC = lsa_hip.CsrMatrix.from_scipy(ctx, big)
dx = lsa_hip.DeviceVector.from_numpy(ctx, x)
dy = lsa_hip.DeviceVector(ctx, n, np.complex128)
bytes_c = 20.0 * nnz + 36.0 * n
variants = {
    "default (groups, 32-bit, 256 wg/cu)": 0,
    "groups, 32-bit, wg/cu64": 16 | 0x2000 | 0x1000 | (64 << 16),
    "groups, xcd chunks, wg/cu256": 16 | 0x2000 | 0x1000 | 0x8000 | (256 << 16),
    "groups, xcd chunks, wg/cu64": 16 | 0x2000 | 0x1000 | 0x8000 | (64 << 16),
    "groups, xcd chunks, wg/cu1024": 16 | 0x2000 | 0x1000 | 0x8000 | (1024 << 16),
    "groups, 32-bit, wg/cu1024": 16 | 0x2000 | 0x1000 | (1024 << 16),
    "groups, 32-bit, wg/cu4096": 16 | 0x2000 | 0x1000 | (4096 << 16),
    "groups, 32-bit, lpr8, wg/cu1024": 8 | 0x2000 | 0x1000 | (1024 << 16),
    "groups + ci16, wg/cu256": 16 | 0x2000 | 0x800 | (256 << 16),
    "ci16, no groups (round 1 default)": 16 | 0x800 | 0x4000,
    "plain 32-bit, no groups": 0x1000 | 0x4000,
}
best = {}
for rnd in range(3):
    for name, v in variants.items():
        os.environ["LSA_SPMV_VARIANT"] = str(v)
        dC.time_matvec(dx, dy, 3)
        ms = dC.time_matvec(dx, dy, 20)
        best.setdefault(name, []).append(ms)
xr = big @ x
for name, v in variants.items():  # every variant computes the same product
    os.environ["LSA_SPMV_VARIANT"] = str(v)
    dC.matvec(dx, dy)
    err = np.linalg.norm(dy.numpy() - xr) / np.linalg.norm(xr)
    assert err < 1e-13, (name, err)
for name, ms in sorted(best.items(), key=lambda kv: np.median(kv[1])):
    print(f"{name:36s} median {np.median(ms) * 1e3:7.1f} us  min {min(ms) * 1e3:7.1f} us  -> {bytes_c / np.median(ms) / 1e6:7.1f} GB/s", flush=True)
