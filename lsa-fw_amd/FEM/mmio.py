"""MatrixMarket coordinate I/O for the (A, M) stage boundary.

The reference writes ``A.mtx`` / ``M.mtx`` with ``scipy.io.mmwrite`` (``FEM/utils.py:634-635``) and reads them back
with ``scipy.io.mmread`` followed by a per-entry PETSc ``setValue`` loop (``FEM/utils.py:143-147,208-215``).  This
module reads the same files straight into CSR in bulk (no per-entry Python work), keeps explicit zeros and expands
``symmetric`` / ``hermitian`` / ``skew-symmetric`` storage.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import scipy.sparse as sp


def read_matrix_market(path: Path) -> sp.csr_matrix:
    """CSR of a MatrixMarket coordinate file.  Uses the native one-pass reader of liblsa_hip.so (``lsa_mm_open``, host
    code, no GPU needed); the pure-numpy reader below is the same algorithm for machines where the library has not been
    built yet (file I/O is not part of the eigen path's arithmetic)."""
    try:
        import lsa_hip

        return lsa_hip.read_matrix_market(path)
    except (ImportError, RuntimeError, OSError):
        return _read_matrix_market_numpy(path)


def _read_matrix_market_numpy(path: Path) -> sp.csr_matrix:
    with open(path, "rb") as fh:
        header = fh.readline().decode().strip().lower().split()
        if len(header) < 5 or header[0] != "%%matrixmarket" or header[1] != "matrix":
            raise ValueError(f"{path}: not a MatrixMarket matrix file")
        fmt, field, symmetry = header[2], header[3], header[4]
        if fmt != "coordinate":
            raise ValueError(f"{path}: only coordinate format is supported (got {fmt})")
        line = fh.readline()
        while line.startswith(b"%") or not line.strip():
            line = fh.readline()
        nrows, ncols, nnz = (int(t) for t in line.split())
        body = fh.read()
    per = {"real": 3, "integer": 3, "double": 3, "complex": 4, "pattern": 2}.get(field)
    if per is None:
        raise ValueError(f"{path}: unsupported field {field}")
    flat = np.array(body.split(), dtype=np.float64) if nnz else np.zeros(0)
    if flat.size != nnz * per:
        raise ValueError(f"{path}: expected {nnz} entries of {per} numbers, found {flat.size} numbers")
    flat = flat.reshape(nnz, per)
    rows = flat[:, 0].astype(np.int64) - 1
    cols = flat[:, 1].astype(np.int64) - 1
    if field == "complex":
        vals = flat[:, 2] + 1j * flat[:, 3]
    elif field == "pattern":
        vals = np.ones(nnz)
    else:
        vals = flat[:, 2].copy()
    if symmetry != "general":
        off = rows != cols
        mirror = {"symmetric": vals[off], "hermitian": np.conj(vals[off]), "skew-symmetric": -vals[off]}.get(symmetry)
        if mirror is None:
            raise ValueError(f"{path}: unsupported symmetry {symmetry}")
        rows, cols, vals = np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]), np.concatenate([vals, mirror])
    # CSR without dropping explicit zeros; duplicate coordinates are summed like scipy does
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    if len(rows) > 1:
        dup = (rows[1:] == rows[:-1]) & (cols[1:] == cols[:-1])
        if dup.any():
            first = np.concatenate([[True], ~dup])
            seg = np.cumsum(first) - 1
            vals = np.bincount(seg, weights=vals.real) + (1j * np.bincount(seg, weights=vals.imag) if np.iscomplexobj(vals) else 0)
            rows, cols = rows[first], cols[first]
    indptr = np.zeros(nrows + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr)
    return sp.csr_matrix((vals, cols.astype(np.int32), indptr.astype(np.int32)), shape=(nrows, ncols))


def write_matrix_market(path: Path, A: sp.spmatrix) -> None:
    A = sp.csr_matrix(A)
    coo_rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr)) + 1
    cplx = np.iscomplexobj(A.data)
    with open(path, "w") as fh:
        fh.write(f"%%MatrixMarket matrix coordinate {'complex' if cplx else 'real'} general\n%\n")
        fh.write(f"{A.shape[0]} {A.shape[1]} {A.nnz}\n")
        if cplx:
            cols = np.column_stack([coo_rows, A.indices + 1, A.data.real, A.data.imag])
            np.savetxt(fh, cols, fmt="%d %d %.17e %.17e")
        else:
            cols = np.column_stack([coo_rows, A.indices + 1, A.data])
            np.savetxt(fh, cols, fmt="%d %d %.17e")
