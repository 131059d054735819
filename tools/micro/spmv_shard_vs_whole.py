"""What the sparse products of one operator apply cost replicated (every rank multiplies the whole matrix) against sharded (a
rank multiplies its row block, the result is all-gathered) -- the kernel side, measured on one GPU; the exchange is a model
(n / P x 16 B per rank over the xGMI mesh).  Patterns of the sharded configurations: S500k (BASELINE config 3, 2D) and the 3D
row pattern (C300k), matrices in the solver's order, values random.

    python tools/micro/spmv_shard_vs_whole.py      # one JSON record
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402


def case(ctx, name, P):
    n = P.shape[0]
    perm = lsa_hip.nd_order(sp.csr_matrix((np.ones(P.nnz), P.indices, P.indptr), shape=P.shape), 0)["perm"]
    P = P[perm][:, perm].tocsr()
    P.sort_indices()
    rng = np.random.default_rng(0)
    C = sp.csr_matrix((rng.standard_normal(P.nnz) + 1j * rng.standard_normal(P.nnz), P.indices, P.indptr), shape=P.shape)
    M = sp.csr_matrix((rng.standard_normal(P.nnz), P.indices, P.indptr), shape=P.shape)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    dx, dy = lsa_hip.DeviceVector.from_numpy(ctx, x), lsa_hip.DeviceVector(ctx, n, np.complex128)
    out = {"n": n, "nnz": int(P.nnz), "entries_per_row": P.nnz / n, "ranks": {}}
    for ranks in (1, 2, 4, 8):
        r1 = n // ranks  # the first rank's block (blocks are balanced by construction of the forest cut; this is the kernel side only)
        rec = {}
        for tag, A in (("C_c128", C), ("M_f64", M)):
            dA = lsa_hip.CsrMatrix.from_scipy(ctx, A) if ranks == 1 else lsa_hip.CsrMatrix.from_scipy_shard(ctx, A[:r1], n, 0)
            dyl = dy if ranks == 1 else lsa_hip.DeviceVector(ctx, r1, np.complex128)  # (the timing entry point writes the shard's own rows)
            dA.time_matvec(dx, dyl, 10)
            rec[tag + "_us"] = 1e3 * dA.time_matvec(dx, dyl, 200)
            del dA, dyl
        rec["both_products_us"] = rec["C_c128_us"] + rec["M_f64_us"]
        rec["allgather_payload_per_rank_MB"] = 0.0 if ranks == 1 else n / ranks * 16 / 1e6
        out["ranks"][str(ranks)] = rec
    return out


def main():
    ctx = lsa_hip.Context(0)
    rec = {"S500k_2D": case(ctx, "S500k", fem.channel_pattern(*fem.CASES["S500k"])), "C300k_3D": case(ctx, "C300k", fem.cube_pattern(fem.CUBE_CASES["C300k"]))}
    ctx.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
