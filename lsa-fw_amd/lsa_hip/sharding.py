"""Row-block sharding of the eigen path across GPUs (one process per GPU).

The reference inherits PETSc's MPI row-block distribution (``PETSc.COMM_WORLD`` at ``Solver/utils.py:200``).  Here the
rows of ``A``, ``M`` (hence ``C = A - sigma M``) are cut into ``P`` contiguous blocks balanced by stored entries, and
the index space is *padded* to ``P`` equal blocks of ``B_pad`` slots so that the all-gather after every SpMV /
preconditioner apply moves equal counts:

    global (permuted) row i in block r   ->   padded index  r * B_pad + (i - starts[r])

Padding slots are zero in every vector and no column index points at them.  The Krylov bases are replicated; each
rank factors only its diagonal block (block-Jacobi ILU(k), PETSc's parallel default ``bjacobi`` + ``ilu``).

This module is pure host logic (numpy/scipy) and is exercised on CPU with ``gloo`` in ``tests/test_sharding_cpu.py``.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass(frozen=True)
class RowPartition:
    starts: np.ndarray  # (P+1,) first row of each block in the unpadded numbering
    b_pad: int  # slots per block in the padded layout

    @property
    def nranks(self) -> int:
        return len(self.starts) - 1

    @property
    def n(self) -> int:
        return int(self.starts[-1])

    @property
    def n_pad(self) -> int:
        return self.nranks * self.b_pad

    def rows(self, rank: int) -> tuple[int, int]:
        return int(self.starts[rank]), int(self.starts[rank + 1])

    def to_padded(self, idx: np.ndarray) -> np.ndarray:
        """Padded index of unpadded indices."""
        idx = np.asarray(idx)
        owner = np.searchsorted(self.starts, idx, side="right") - 1
        return owner * self.b_pad + (idx - self.starts[owner])

    def pad_vector(self, v: np.ndarray) -> np.ndarray:
        out = np.zeros((self.n_pad,) + v.shape[1:], dtype=v.dtype)
        out[self.to_padded(np.arange(self.n))] = v
        return out

    def unpad_vector(self, vp: np.ndarray) -> np.ndarray:
        return vp[self.to_padded(np.arange(self.n))]


def partition_rows(indptr: np.ndarray, nranks: int, align: int = 64) -> RowPartition:
    """Contiguous row blocks with (nearly) equal numbers of stored entries; B_pad rounded up to ``align``."""
    n = len(indptr) - 1
    if nranks < 1 or nranks > max(n, 1):
        raise ValueError(f"cannot cut {n} rows into {nranks} blocks")
    nnz = int(indptr[-1])
    targets = (np.arange(1, nranks) * nnz) // nranks
    cuts = np.searchsorted(indptr, targets, side="left")
    starts = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    # no empty block: push cuts apart if entries are very uneven
    for r in range(1, nranks + 1):
        starts[r] = max(starts[r], starts[r - 1] + 1)
    for r in range(nranks - 1, -1, -1):
        starts[r] = min(starts[r], starts[r + 1] - 1)
    starts[0], starts[-1] = 0, n
    b = int(np.max(np.diff(starts)))
    b_pad = ((b + align - 1) // align) * align
    return RowPartition(starts, b_pad)


def shard_rows(A: sp.csr_matrix, part: RowPartition, rank: int) -> sp.csr_matrix:
    """Rows of block ``rank`` with columns renumbered into the padded layout: shape (rows_r, n_pad)."""
    r0, r1 = part.rows(rank)
    rows = sp.csr_matrix(A[r0:r1])
    cols = part.to_padded(rows.indices).astype(np.int32)  # monotone map: columns stay sorted inside each row
    return sp.csr_matrix((rows.data, cols, rows.indptr), shape=(r1 - r0, part.n_pad))


def diagonal_block(A: sp.csr_matrix, part: RowPartition, rank: int) -> sp.csr_matrix:
    """Square diagonal block of block ``rank`` in local numbering (input of the block-Jacobi ILU)."""
    r0, r1 = part.rows(rank)
    blk = sp.csr_matrix(A[r0:r1][:, r0:r1])
    blk.sort_indices()
    return blk


# ---- subtree-parallel exact LU: the elimination forest cut over the ranks ------------------------------------------------


@dataclass(frozen=True)
class ForestPartition:
    """The nested-dissection forest of ``lsa_hip.NdAnalysis`` cut over ``P`` ranks.

    Every rank factors whole subtrees; the top of the forest (``owner == -1``) is replicated.  ``order`` lists the
    unknowns (original numbering) rank by rank -- each rank's subtrees in elimination order, the top last, at the end of
    the last rank's block -- so that a rank's unknowns are one contiguous row block and the order is still a post-order
    of the same forest.  ``first`` / ``size`` / ``parent`` / ``owner`` describe the forest in the padded block layout of
    ``rows`` (what ``lsa_op_create_dist`` takes)."""

    order: np.ndarray
    rows: RowPartition
    first: np.ndarray
    size: np.ndarray
    parent: np.ndarray
    owner: np.ndarray
    subtree_work: np.ndarray  # per rank: factor scalars of its subtrees
    top_work: int  # factor scalars of the replicated top


def dist_front_min() -> int:
    """Front size from which a top node of the cut forest is DISTRIBUTED over the ranks instead of replicated
    (``LSA_ND_DIST_MIN``; 0 = never).  Replicating the top is the cheaper form while its fronts are small (2D: a few thousand
    rows, every exchange of the sweeps costs more than the redundant work); in 3D the top fronts ARE the factorisation (tens of
    thousands of rows each) and replicating them is what kept a 5 M-unknown problem from fitting eight GPUs."""
    import os

    raw = os.environ.get("LSA_ND_DIST_MIN", "").strip()
    return int(raw) if raw else 6144


def dist_split_min() -> int:
    """Front size from which a subtree root is moved into the distributed top whatever the load balance says
    (``LSA_ND_DIST_SPLIT``; 0 = never): keeps the largest front and update matrix a single rank must hold below ~ 2 x 2.6 GB."""
    import os

    raw = os.environ.get("LSA_ND_DIST_SPLIT", "").strip()
    return int(raw) if raw else 18000


def partition_forest(perm: np.ndarray, node_start: np.ndarray, parent: np.ndarray, front_size: np.ndarray, nranks: int, align: int = 64,
                     imbalance: float = 1.25, dist_min: int | None = None) -> ForestPartition:
    """Cut the forest (arrays of ``NdAnalysis.export()``: nodes in post-order) over ``nranks`` ranks.

    Top nodes whose front has at least ``dist_min`` rows (default: :func:`dist_front_min`), and every top node above one of
    them, get owner -2: distributed over the ranks (rows of the boundary part of their fronts, see ``include/lsa_hip.h``);
    the other top nodes owner -1: replicated.

    The frontier starts at the roots; its heaviest subtree is split (its root joins the replicated top, its children the
    frontier) until there are at least ``nranks`` subtrees and the heaviest one is within ``imbalance`` of the mean load,
    or nothing is left to split.  Subtrees go to ranks heaviest first, each to the least loaded rank."""
    nt = len(parent)
    m = np.diff(node_start).astype(np.int64)
    b = front_size.astype(np.int64) - m
    work = m * m + 2 * m * b
    nodes_in = np.ones(nt, dtype=np.int64)
    sub = work.copy()
    children = [[] for _ in range(nt)]
    for t in range(nt):  # post-order: children come first
        p = int(parent[t])
        if p >= 0:
            sub[p] += sub[t]
            nodes_in[p] += nodes_in[t]
            children[p].append(t)
    frontier = [t for t in range(nt) if parent[t] < 0]
    top: list[int] = []
    while True:
        splittable = [t for t in frontier if children[t]]
        if not splittable:
            break
        heavy = max(frontier, key=lambda t: sub[t])
        mean = sum(sub[t] for t in frontier) / nranks
        if len(frontier) >= nranks and (sub[heavy] <= imbalance * mean or len(frontier) >= 8 * nranks):
            break
        t = heavy if children[heavy] else max(splittable, key=lambda t: sub[t])
        frontier.remove(t)
        top.append(t)
        frontier.extend(children[t])
    # A subtree root with a very large front does not belong to ONE rank whatever the balance says: its working front and the
    # update matrix it hands up (b^2 scalars, alive until its parent is factored) would sit on that rank alone -- at 5 M
    # unknowns in 3D a 66 k-row root front is 35 GB plus a 51 GB update matrix.  Such roots join the (distributed) top.
    thr = dist_front_min() if dist_min is None else int(dist_min)
    big = dist_split_min() if thr > 0 else 0
    while big > 0 and nranks > 1:
        huge = [t for t in frontier if children[t] and front_size[t] >= big]
        if not huge:
            break
        for t in huge:
            frontier.remove(t)
            top.append(t)
            frontier.extend(children[t])
    load = np.zeros(nranks, dtype=np.int64)
    rank_of_root = {}
    for t in sorted(frontier, key=lambda t: (-sub[t], t)):
        r = int(np.argmin(load))
        rank_of_root[t] = r
        load[r] += sub[t]
    owner = np.full(nt, -1, dtype=np.int32)
    for t, r in rank_of_root.items():
        owner[t - nodes_in[t] + 1 : t + 1] = r  # a subtree is a contiguous run of post-order ids ending at its root
    if thr > 0 and nranks > 1:
        for t in range(nt):  # post-order: a node's children are decided before it
            if owner[t] == -1 and (front_size[t] >= thr or any(owner[c] == -2 for c in children[t])):
                owner[t] = -2
    # new node order: rank by rank, then the top
    new_nodes = np.concatenate([np.flatnonzero(owner == r) for r in range(nranks)] + [np.flatnonzero(owner < 0)])
    new_id = np.empty(nt, dtype=np.int64)
    new_id[new_nodes] = np.arange(nt)
    order = np.concatenate([perm[node_start[t] : node_start[t + 1]] for t in new_nodes]) if nt else np.zeros(0, dtype=np.int64)
    sizes = m[new_nodes]
    first_unpadded = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    rows_per_rank = np.array([m[owner == r].sum() for r in range(nranks)], dtype=np.int64)
    rows_per_rank[-1] += m[owner < 0].sum()
    if np.any(rows_per_rank == 0):
        raise ValueError(f"the elimination forest is too small to be cut over {nranks} ranks")
    starts = np.concatenate([[0], np.cumsum(rows_per_rank)])
    b_pad = ((int(rows_per_rank.max()) + align - 1) // align) * align
    rows = RowPartition(starts, b_pad)
    par_new = np.where(parent[new_nodes] >= 0, new_id[np.maximum(parent[new_nodes], 0)], -1).astype(np.int32)
    return ForestPartition(order=np.asarray(order, dtype=np.int64), rows=rows, first=rows.to_padded(first_unpadded).astype(np.int32),
                           size=sizes.astype(np.int32), parent=par_new, owner=owner[new_nodes].astype(np.int32),
                           subtree_work=load, top_work=int(work[owner < 0].sum()))


def pad_square(A: sp.csr_matrix, part: RowPartition) -> sp.csr_matrix:
    """The whole (already permuted) matrix embedded in the padded block layout: n_pad x n_pad with empty padding rows."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    slots = part.to_padded(np.arange(n))
    counts = np.zeros(part.n_pad, dtype=np.int64)
    counts[slots] = np.diff(A.indptr)
    indptr = np.concatenate([[0], np.cumsum(counts)])
    out = sp.csr_matrix((A.data, part.to_padded(A.indices).astype(np.int32), indptr), shape=(part.n_pad, part.n_pad))
    return out
