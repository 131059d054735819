"""CPU suite, part 4: the N > 1 layout with ``gloo`` and world_size 2 (no GPU).

The product's host-side sharding logic (``lsa_hip.sharding``: nnz-balanced row blocks, padded block layout, column
renumbering, diagonal blocks) is exercised by a numpy emulation of the sharded operator apply -- local rows times a
replicated vector, in-place all-gather of equal padded blocks, block-Jacobi ILU(k) from the oracle's C kernels,
right-preconditioned GMRES -- driven by the product's Krylov-Schur.  Checks: every rank gets bit-identical Ritz values
(replicated basis, no all-reduce), and they match the single-process oracle to 1e-8.
"""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

import helpers  # noqa: F401  (sys.path)
from lsa_hip import sharding
from synthetic import fem


def test_partition_and_padded_layout():
    es = fem.cylinder_case("S2k")
    for P in (1, 2, 3, 8):
        part = sharding.partition_rows(es.A.indptr, P)
        assert part.starts[0] == 0 and part.starts[-1] == es.n and np.all(np.diff(part.starts) > 0)
        assert part.b_pad % 64 == 0 and part.b_pad >= np.max(np.diff(part.starts)) and part.n_pad == P * part.b_pad
        nnz_blocks = np.diff(es.A.indptr[part.starts])
        assert nnz_blocks.max() <= 1.25 * es.A.nnz / P + 64  # balanced by stored entries
        v = np.random.default_rng(0).standard_normal(es.n)
        vp = part.pad_vector(v)
        assert vp.shape == (part.n_pad,) and np.array_equal(part.unpad_vector(vp), v)
        assert np.count_nonzero(vp) == np.count_nonzero(v)
        # shards times the padded vector reproduce A v block by block; padding is never referenced
        y = np.zeros(part.n_pad)
        for r in range(P):
            rows = sharding.shard_rows(es.A, part, r)
            assert rows.shape == (part.rows(r)[1] - part.rows(r)[0], part.n_pad) and rows.has_sorted_indices
            y[r * part.b_pad : r * part.b_pad + rows.shape[0]] = rows @ vp
            d = sharding.diagonal_block(es.A, part, r)
            r0, r1 = part.rows(r)
            assert abs(d - es.A[r0:r1][:, r0:r1]).max() == 0
        assert np.allclose(part.unpad_vector(y), es.A @ v, rtol=1e-14, atol=1e-14)
    with pytest.raises(ValueError):
        sharding.partition_rows(es.A.indptr, 0)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = Path(__file__).resolve().parents[1]
    sys.path[:0] = [str(root), str(root / "lsa-fw_amd"), str(root / "tests")]
    import torch
    import torch.distributed as dist

    from lsa_hip import sharding as sh
    from lsa_hip.krylov_schur import krylov_schur
    from oracle import kernels
    from synthetic import fem as ofem
    from Solver.utils import _dist_rank_world, delay_zero_diagonal_rows, pivot_safe_rcm
    import helpers as hp

    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert _dist_rank_world() == (rank, world)
    es = ofem.assemble_linearized_ns(ofem.channel_mesh(14, 7, grading=0.3), 50.0)  # n = 1006
    sigma = ofem.SIGMA_RE50
    C = sp.csr_matrix((es.A.data - sigma * es.M.data, es.A.indices, es.A.indptr), shape=es.A.shape)
    perm = pivot_safe_rcm(C)
    Cp = C[perm][:, perm].tocsr()
    Mp = es.M[perm][:, perm].tocsr()
    part = sh.partition_rows(Cp.indptr, world)
    q = delay_zero_diagonal_rows(Cp, part.starts)  # what Solver.utils.iEpsSolver.prepare() does for the sharded layout
    perm = perm[q]
    Cp, Mp = Cp[q][:, q].tocsr(), Mp[q][:, q].tocsr()
    Cp.sort_indices()
    Mp.sort_indices()
    C_rows, M_rows = sh.shard_rows(Cp, part, rank), sh.shard_rows(Mp, part, rank)
    ilu = kernels.ILU0(kernels.iluk_pattern(sh.diagonal_block(Cp, part, rank), 2), 0.0)
    r0 = rank * part.b_pad
    nloc = C_rows.shape[0]

    def allgather(v):
        t = torch.from_numpy(np.ascontiguousarray(v).view(np.float64).copy())
        chunks = [torch.empty(2 * part.b_pad, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(chunks, t[2 * r0 : 2 * (r0 + part.b_pad)].contiguous())
        return torch.cat(chunks).numpy().view(np.complex128)

    def spmv_global(rows, x):
        y = np.zeros(part.n_pad, dtype=np.complex128)
        y[r0 : r0 + nloc] = rows @ x
        return allgather(y)

    def pc_global(b):
        z = np.zeros(part.n_pad, dtype=np.complex128)
        z[r0 : r0 + nloc] = ilu.solve(b[r0 : r0 + nloc])
        return allgather(z)

    def gmres(b, rtol=1e-12, m=300):
        """right-preconditioned GMRES with CGS2 and Givens rotations, same recurrences as csrc/solver.hip"""
        bn = np.linalg.norm(b)
        V = np.zeros((part.n_pad, m + 1), dtype=complex)
        H = np.zeros((m + 1, m), dtype=complex)
        cs, sn, g = np.zeros(m, dtype=complex), np.zeros(m, dtype=complex), np.zeros(m + 1, dtype=complex)
        g[0] = bn
        V[:, 0] = b / bn
        for j in range(m):
            w = spmv_global(C_rows, pc_global(V[:, j]))
            for _ in range(2):
                h = V[:, : j + 1].conj().T @ w
                w = w - V[:, : j + 1] @ h
                H[: j + 1, j] += h
            H[j + 1, j] = np.linalg.norm(w)
            V[:, j + 1] = w / H[j + 1, j]
            for i in range(j):
                t = np.conj(cs[i]) * H[i, j] + np.conj(sn[i]) * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            rr = np.hypot(abs(H[j, j]), abs(H[j + 1, j]))
            cs[j], sn[j] = H[j, j] / rr, H[j + 1, j] / rr
            H[j, j] = np.conj(cs[j]) * H[j, j] + np.conj(sn[j]) * H[j + 1, j]
            H[j + 1, j] = 0
            g[j + 1] = -sn[j] * g[j]
            g[j] = np.conj(cs[j]) * g[j]
            if abs(g[j + 1]) / bn <= rtol:
                y = np.linalg.solve(np.triu(H[: j + 1, : j + 1]), g[: j + 1])
                return pc_global(V[:, : j + 1] @ y), j + 1
        raise RuntimeError("sharded GMRES did not converge")

    its = []

    def op(x):
        y, k = gmres(spmv_global(M_rows, x))
        its.append(k)
        return y

    class Backend(hp.NumpyKrylovBackend):
        def inject(self, j, v):  # padding slots stay zero
            super().inject(j, v * part.pad_vector(np.ones(es.n)))

    be = Backend(op, part.n_pad, 16)
    res = krylov_schur(be, 3, 1e-10, 100, lambda th: -np.abs(th))
    lam = sigma + 1.0 / res.theta[:3]
    X = np.empty((es.n, 3), dtype=complex)
    X[perm] = part.unpad_vector(res.vectors[:, :3])
    np.savez(Path(out_dir) / f"rank{rank}.npz", lam=lam, X=X, its=np.array(its), pad_max=np.abs(res.vectors[part.pad_vector(np.ones(es.n)) == 0]).max())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_operator_world_size_2(tmp_path):
    import torch.multiprocessing as mp

    from oracle import shift_invert

    port = _free_port()
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # two ranks share this box's cores
        os.environ.setdefault(var, "2")
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # replicated basis + fixed-order reductions: both ranks hold bit-identical results without any all-reduce
    assert np.array_equal(r0["lam"], r1["lam"]) and np.array_equal(r0["X"], r1["X"])
    assert r0["pad_max"] == 0.0  # padding slots of the block layout stay exactly zero
    es = fem.assemble_linearized_ns(fem.channel_mesh(14, 7, grading=0.3), 50.0)
    ref, _, _ = shift_invert.solve(es.A, es.M, fem.SIGMA_RE50, k=3, tol=1e-13)
    assert helpers.match_nearest(r0["lam"], ref).max() < 1e-8
    res = shift_invert.compute_residuals(es.A, es.M, r0["lam"], r0["X"])
    assert res.max() < 1e-8
    # block-Jacobi ILU(2) over 2 ranks still converges (single-rank ILU(2) needs ~20 iterations on this case)
    assert r0["its"].max() < 300
