"""SROOF as ONE mesh: the SpMV roofline matrix of SURVEY 8(d) (2D Taylor-Hood row pattern, n ~ 5 M, nnz ~ 1.5e8) built as a single
1056 x 528 channel mesh in RCM order, against the form bench.py times (10 block-diagonal replicas of the S500k matrix in RCM
order), in one process on one GPU.  The replica form is what the default bench run can afford to build (its 5 M-row assembly
would take minutes); this tool measures once whether the gather locality of the two forms differs.

    python tools/sroof_single_mesh.py            # prints one JSON record
"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "lsa-fw_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
from scipy.sparse.csgraph import reverse_cuthill_mckee  # noqa: E402

import lsa_hip  # noqa: E402
from synthetic import fem  # noqa: E402


def timed(ctx, M, iters=50):
    n = M.shape[0]
    rng = np.random.default_rng(0)
    out = {}
    for name, dt in (("c128", np.complex128), ("f64", np.float64)):
        A = sp.csr_matrix((M.data.astype(dt) if dt is np.complex128 else np.ascontiguousarray(M.data.real), M.indices, M.indptr), shape=M.shape)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt) if dt is np.complex128 else rng.standard_normal(n)
        dA = lsa_hip.CsrMatrix.from_scipy(ctx, A)
        dx, dy = lsa_hip.DeviceVector.from_numpy(ctx, x), lsa_hip.DeviceVector(ctx, n, dt)
        dA.time_matvec(dx, dy, 5)
        ms = dA.time_matvec(dx, dy, iters)
        nbytes = (20.0 if dt is np.complex128 else 12.0) * A.nnz + (36.0 if dt is np.complex128 else 20.0) * n
        out[name] = {"ms": ms, "algorithmic_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000.0, "kernel": dA.matvec_info(dt)["kernel"]}
        del dA, dx, dy
    return out


def main():
    rec = {}
    ctx = lsa_hip.Context(0)
    rng = np.random.default_rng(1)
    t0 = time.time()
    P = fem.channel_pattern(1056, 528)
    perm = reverse_cuthill_mckee(sp.csr_matrix((np.ones(P.nnz, np.int8), P.indices, P.indptr), shape=P.shape), symmetric_mode=True)
    P = P[perm][:, perm].tocsr()
    P.sort_indices()
    one = sp.csr_matrix((rng.standard_normal(P.nnz) + 1j * rng.standard_normal(P.nnz), P.indices, P.indptr), shape=P.shape)
    rec["single_mesh"] = {"n": one.shape[0], "nnz": one.nnz, "build_s": time.time() - t0, **timed(ctx, one)}
    del one, P
    t0 = time.time()
    nx, ny = fem.CASES["S500k"]
    Q = fem.channel_pattern(nx, ny)
    perm = reverse_cuthill_mckee(sp.csr_matrix((np.ones(Q.nnz, np.int8), Q.indices, Q.indptr), shape=Q.shape), symmetric_mode=True)
    Q = Q[perm][:, perm].tocsr()
    Q.sort_indices()
    n1, nnz1, reps = Q.shape[0], Q.nnz, 10
    rp = np.concatenate([[0], (Q.indptr[1:][None, :] + (np.arange(reps) * nnz1)[:, None]).ravel()]).astype(np.int32)
    ci = (Q.indices[None, :] + (np.arange(reps, dtype=np.int64) * n1)[:, None]).ravel().astype(np.int32)
    big = sp.csr_matrix((rng.standard_normal(nnz1 * reps) + 1j * rng.standard_normal(nnz1 * reps), ci, rp), shape=(n1 * reps, n1 * reps))
    rec["ten_replicas_of_s500k"] = {"n": big.shape[0], "nnz": big.nnz, "build_s": time.time() - t0, **timed(ctx, big)}
    ctx.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
