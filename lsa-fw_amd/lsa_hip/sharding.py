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
