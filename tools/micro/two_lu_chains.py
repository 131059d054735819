"""Do two independent chains of LU sweeps overlap on one GPU?  Two factorisations of the same case in two contexts (own streams),
timed alone and together: the aggregate rate of the pair against one chain's tells what streams over subtrees could buy."""
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "S120k"
es = fem.cylinder_case(case)
C = sp.csr_matrix((es.A.data - fem.SIGMA_RE50 * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
o = lsa_hip.nd_order(C)
Cp = C[o["perm"]][:, o["perm"]].tocsr()
Cp.sort_indices()
tree = {"first": o["first"], "size": o["size"], "parent": o["parent"]}
objs = []
for _ in range(2):
    ctx = lsa_hip.Context(0)
    dC = lsa_hip.CsrMatrix.from_scipy(ctx, Cp)
    f = lsa_hip.NdLu(ctx, dC, tree=tree)
    b = lsa_hip.DeviceVector.from_numpy(ctx, np.ones(es.n, dtype=np.complex128))
    x = lsa_hip.DeviceVector(ctx, es.n, np.complex128)
    f.time_solve(b, x, 10)
    objs.append((ctx, dC, f, b, x))
alone = objs[0][2].time_solve(objs[0][3], objs[0][4], 100)
res = [0.0, 0.0]


def run(i):
    res[i] = objs[i][2].time_solve(objs[i][3], objs[i][4], 100)


th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
wall = time.perf_counter() - t0
print(f"{case}: one chain {alone * 1e3:.1f} us per apply; two chains together {res[0] * 1e3:.1f} / {res[1] * 1e3:.1f} us per apply each, "
      f"{200 / wall / (1 / (alone * 1e-3)):.2f} x the rate of one chain")
