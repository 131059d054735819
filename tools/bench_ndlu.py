"""Analysis, factor and solve timing of the nested-dissection LU against the direct (SuperLU) answer (development aid)."""
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

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="S5k")
ap.add_argument("--leaf", type=int, default=0)
ap.add_argument("--refactors", type=int, default=3)
ap.add_argument("--unordered", action="store_true", help="the library's own dissection of the matrix as it comes (vectors addressed through index lists)")
ap.add_argument("--complex", action="store_true", help="3D cases: complex shift (default: the real shift of the 3D tests, float64 factors)")
args = ap.parse_args()
es = fem.cube_case(args.case) if args.case.startswith("C") else fem.cylinder_case(args.case)
sigma = fem.SIGMA_RE50 if not args.case.startswith("C") else (fem.SIGMA_CUBE + 0.5j if args.complex else fem.SIGMA_CUBE)
C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
ctx = lsa_hip.Context(0)
tree = None
if not args.unordered:  # what Solver/utils.py does: the matrix in the elimination order, the forest handed back
    t0 = time.time()
    zd = C.diagonal() == 0
    o = lsa_hip.nd_order(C, args.leaf, constraint=zd if (zd.any() and C.nnz > 60 * es.n) else None)
    C = C[o["perm"]][:, o["perm"]].tocsr()
    C.sort_indices()
    tree = {"first": o["first"], "size": o["size"], "parent": o["parent"]}
    print(f"ordering + permutation {time.time() - t0:.3f}s", flush=True)
dC = lsa_hip.CsrMatrix.from_scipy(ctx, C)
t0 = time.time()
f = lsa_hip.NdLu(ctx, dC, args.leaf, tree=tree)
print(f"{args.case}: n={es.n} nnz={C.nnz} create {time.time() - t0:.3f}s info={f.info()}", flush=True)
for _ in range(args.refactors):
    t0 = time.time()
    f.refactor(dC)
    print(f"  refactor {1e3 * (time.time() - t0):.2f} ms", flush=True)
b = np.random.default_rng(0).standard_normal(es.n) + (1j * np.random.default_rng(1).standard_normal(es.n) if np.iscomplexobj(C.data) else 0.0)
db = lsa_hip.DeviceVector.from_numpy(ctx, b)
dx = lsa_hip.DeviceVector(ctx, es.n, b.dtype)
f.solve(db, dx)
x = dx.numpy()
print("relative residual", np.linalg.norm(C @ x - b) / np.linalg.norm(b), flush=True)
if es.n < 40000:
    xr = spla.splu(C.tocsc()).solve(b)
    print("vs SuperLU", np.linalg.norm(x - xr) / np.linalg.norm(xr), flush=True)
f.time_solve(db, dx, 10)
ms = f.time_solve(db, dx, 100)
info = f.info()
print(f"solve {ms * 1e3:.1f} us per apply; {info['apply_bytes'] / 1e6:.1f} MB -> {info['apply_bytes'] / ms / 1e6:.0f} GB/s; {info['apply_launches']} launches", flush=True)
