"""Per-rank device memory of the exact nested-dissection LU for a 3D case, from the PATTERN alone (no GPU, no matrix values):
analysis -> forest cut over the ranks (lsa_hip.sharding.partition_forest) -> per-rank localised analysis ->
lsa_nd_sym_memory, the plan lsa_ndlu_create executes.  Prints one JSON record (DESIGN.md section 8 quotes it).

    python tools/plan_memory.py --case C5M --ranks 8 --scalar-bytes 8
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from lsa_hip import sharding  # noqa: E402
from synthetic import fem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="C1M")
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--scalar-bytes", type=int, default=8, help="8: real shift (the 3D tests), 16: complex factors")
ap.add_argument("--work-gb", type=float, default=16.0, help="working-front budget per rank")
ap.add_argument("--ncv", type=int, default=40)
args = ap.parse_args()
GB = 1e9
t0 = time.time()
nc = fem.CUBE_CASES[args.case]
P = fem.cube_pattern(nc)
n, nnz = P.shape[0], P.nnz
mesh = fem.CubeMesh(nc)
isv = mesh.is_vertex()
node_offset = 3 * np.arange(mesh.n_nodes, dtype=np.int64) + np.concatenate([[0], np.cumsum(isv)[:-1]])
flags = np.zeros(n, dtype=np.int8)
flags[node_offset[isv] + 3] = 1  # pressure unknowns: zero diagonal, eliminated after their neighbours
print(f"{args.case}: n={n} nnz={nnz} pattern in {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
t0 = time.time()
an = lsa_hip.NdAnalysis(P, 0, constraint=flags)
ex = an.export()
print(f"analysis {time.time() - t0:.0f}s: {an.ntree} nodes, {an.nlevels} levels, largest front {an.max_front}, factor entries {an.factor_entries:.3e}", file=sys.stderr, flush=True)
one = an.memory(args.scalar_bytes, int(args.work_gb * GB))
rec = {"case": args.case, "n": n, "nnz": int(nnz), "scalar_bytes": args.scalar_bytes, "tree_nodes": an.ntree, "levels": an.nlevels, "largest_front": an.max_front,
       "factor_entries": int(an.factor_entries), "flops": an.flops,
       "matrices_GB": (3 * nnz * args.scalar_bytes + 4 * nnz + 4 * n) / GB, "krylov_GB": 2 * (args.ncv + 1) * n * 16 / GB,
       "one_gpu_GB": {k: v / GB for k, v in one.items() if k != "chunks"}, "one_gpu_chunks": one["chunks"]}
if args.ranks > 1:
    t0 = time.time()
    fp = sharding.partition_forest(ex["perm"], ex["node_start"], ex["parent"], ex["front_size"], args.ranks)
    del an
    Pp = P[fp.order][:, fp.order].tocsr()
    del P
    Pp.sort_indices()
    Ppad = sharding.pad_square(Pp, fp.rows)
    del Pp
    tree = {"first": fp.first, "size": fp.size, "parent": fp.parent, "owner": fp.owner}
    print(f"forest cut over {args.ranks} ranks + permutation {time.time() - t0:.0f}s: top holds {int((fp.owner < 0).sum())} nodes", file=sys.stderr, flush=True)
    rec["ranks"] = []
    for r in range(args.ranks):
        t0 = time.time()
        ar = lsa_hip.NdAnalysis(Ppad, tree=tree, rank=r, nranks=args.ranks)
        mem = ar.memory(args.scalar_bytes, int(args.work_gb * GB))
        row = {k: v / GB for k, v in mem.items() if k != "chunks"}
        row["chunks"] = mem["chunks"]
        row["rank"] = r
        row["with_matrices_and_basis_GB"] = row["total"] + rec["matrices_GB"] + rec["krylov_GB"]
        rec["ranks"].append(row)
        print(f"rank {r}: {row} ({time.time() - t0:.0f}s)", file=sys.stderr, flush=True)
        del ar
    rec["max_rank_GB"] = max(x["with_matrices_and_basis_GB"] for x in rec["ranks"])
print(json.dumps(rec))
