"""PETSc-free stand-ins for the boundary types of the eigen path.

Mirrors the members of ``/root/reference/FEM/utils.py`` that the eigen path and its callers touch
(SURVEY.md section 8, rows A9 / A10), backed by scipy CSR / numpy instead of ``petsc4py``:

* ``iPETScMatrix`` (``FEM/utils.py:104``): ``from_path`` ``:143-147``, ``zeros`` ``:149``, ``from_matrix`` ``:183-220``,
  ``shape`` ``:373``, ``raw`` ``:354``, ``nonzero_entries`` ``:377``, ``norm`` ``:401``, ``T``/``H`` ``:363-371``,
  ``is_numerically_hermitian`` ``:436-448``, ``assemble`` ``:450``, ``as_scipy_array`` ``:585-588``,
  ``export`` (MatrixMarket) ``:616-636``;
* ``iPETScVector`` (``:662``) and ``iComplexPETScVector`` (``:911``) with the *real-build* layout the callers rely
  on: a real part, an optional imaginary part, a conjugating ``dot`` (``:1194-1212``).

The matrix shim keeps explicit zeros (they are part of the mixed-space sparsity pattern) and loads MatrixMarket
files without the reference's per-entry ``setValue`` loop (``FEM/utils.py:208-215``).
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import scipy.sparse as sp


class iPETScVector:
    """Dense vector (numpy-backed) with the reference wrapper's surface."""

    def __init__(self, data: np.ndarray):
        self._a = np.array(data, copy=True).ravel()

    @classmethod
    def _adopt(cls, data: np.ndarray) -> "iPETScVector":
        """Wrap an array the caller hands over (no copy; may be a strided view of memory nobody else writes)."""
        out = cls.__new__(cls)
        out._a = data.reshape(-1)
        return out

    @classmethod
    def from_array(cls, data: np.ndarray) -> "iPETScVector":
        return cls(np.asarray(data))

    @classmethod
    def zeros(cls, n: int) -> "iPETScVector":
        return cls(np.zeros(n))

    @property
    def raw(self) -> np.ndarray:
        return self._a

    @property
    def size(self) -> int:
        return int(self._a.shape[0])

    @property
    def norm(self) -> float:
        return float(np.linalg.norm(self._a))

    def as_array(self) -> np.ndarray:
        return self._a.copy()

    def get_value(self, index: int):
        return self._a[index]

    def __getitem__(self, i):
        return self._a[i]

    def __setitem__(self, i, v) -> None:
        self._a[i] = v

    def assemble(self) -> None:
        pass

    def dot(self, other: "iPETScVector"):
        # PETSc VecDot(x, y) = y^H x; the reference calls self._vec.dot(other) -> conj(other) . self
        return np.vdot(other._a, self._a) if np.iscomplexobj(other._a) else np.dot(self._a, other._a)

    def scale(self, alpha) -> None:
        self._a = self._a * alpha

    def copy(self) -> "iPETScVector":
        return iPETScVector(self._a)

    def __mul__(self, alpha) -> "iPETScVector":
        return iPETScVector(self._a * alpha)

    __rmul__ = __mul__

    def __add__(self, other: "iPETScVector") -> "iPETScVector":
        return iPETScVector(self._a + other._a)

    def __sub__(self, other: "iPETScVector") -> "iPETScVector":
        return iPETScVector(self._a - other._a)

    def export(self, path: Path) -> None:
        np.save(str(path), self._a)


class iComplexPETScVector:
    """Real-build complex vector: real part + optional imaginary part (``FEM/utils.py:911-1240``)."""

    def __init__(self, real, imag=None):
        self._real = real if isinstance(real, iPETScVector) else iPETScVector(real)
        self._imag = None if imag is None else imag if isinstance(imag, iPETScVector) else iPETScVector(imag)

    @classmethod
    def from_array(cls, data: np.ndarray) -> "iComplexPETScVector":
        arr = np.asarray(data).ravel()
        if np.iscomplexobj(arr):
            return cls(iPETScVector(arr.real), iPETScVector(arr.imag))
        return cls(iPETScVector(arr))

    @property
    def real(self) -> iPETScVector:
        return self._real

    @property
    def imag(self) -> iPETScVector | None:
        return self._imag

    @property
    def is_complex(self) -> bool:
        return self._imag is not None

    @property
    def size(self) -> int:
        return self._real.size

    def as_array(self) -> np.ndarray:
        if self._imag is None:
            return self._real.as_array()
        return self._real.as_array() + 1j * self._imag.as_array()

    def get_value(self, index: int):
        if self._imag is None:
            return self._real.get_value(index)
        return self._real.get_value(index) + 1j * self._imag.get_value(index)

    def assemble(self) -> None:
        pass

    def norm(self) -> float:
        if self._imag is None:
            return self._real.norm
        return float(np.sqrt(self._real.norm**2 + self._imag.norm**2))

    def dot(self, other) -> complex:
        """Hermitian inner product, conjugating *self* (``FEM/utils.py:1194-1212``)."""
        if isinstance(other, iPETScVector):
            other = iComplexPETScVector(other)
        if not isinstance(other, iComplexPETScVector):
            raise TypeError("Dot product requires a iComplexPETScVector.")
        return complex(np.vdot(self.as_array(), other.as_array()))

    def scale(self, scalar) -> None:
        z = self.as_array() * scalar
        if np.iscomplexobj(z) and np.any(z.imag != 0.0):
            self._real, self._imag = iPETScVector(z.real), iPETScVector(z.imag)
        else:
            self._real, self._imag = iPETScVector(np.real(z)), None

    def copy(self) -> "iComplexPETScVector":
        return iComplexPETScVector(self._real.copy(), None if self._imag is None else self._imag.copy())


class iPETScMatrix:
    """Sparse matrix (scipy CSR, explicit zeros kept) with the reference wrapper's surface."""

    def __init__(self, mat):
        if isinstance(mat, iPETScMatrix):
            mat = mat._mat
        if isinstance(mat, np.ndarray):
            dense = np.asarray(mat)
            dt = np.complex128 if np.iscomplexobj(dense) else np.float64
            # a dense input keeps its full pattern, like PETSc's MATDENSE in the reference's from_matrix
            rows, cols = np.indices(dense.shape)
            mat = sp.csr_matrix((dense.astype(dt).ravel(), (rows.ravel(), cols.ravel())), shape=dense.shape)
        self._mat = sp.csr_matrix(mat)
        self._mat.sort_indices()

    # ---- constructors ---------------------------------------------------------------------------------------
    @classmethod
    def from_path(cls, path: Path) -> "iPETScMatrix":
        """Load a MatrixMarket file (``general`` or ``symmetric``; explicit zeros preserved)."""
        from .mmio import read_matrix_market

        return cls(read_matrix_market(Path(path)))

    @classmethod
    def zeros(cls, shape: tuple[int, int], nnz=None) -> "iPETScMatrix":
        return cls(sp.csr_matrix(shape, dtype=np.float64))

    @classmethod
    def from_matrix(cls, matrix) -> "iPETScMatrix":
        if isinstance(matrix, iPETScMatrix):
            return matrix
        if isinstance(matrix, np.ndarray) or sp.issparse(matrix):
            return cls(matrix)
        raise TypeError(f"Cannot construct iPETScMatrix from object of type {type(matrix)}")

    # ---- properties -------------------------------------------------------------------------------------------
    @property
    def raw(self) -> sp.csr_matrix:
        return self._mat

    @property
    def shape(self) -> tuple[int, int]:
        return tuple(self._mat.shape)

    @property
    def nonzero_entries(self) -> int:
        return int(self._mat.nnz)

    @property
    def norm(self) -> float:
        return float(np.sqrt(np.sum(np.abs(self._mat.data) ** 2)))

    @property
    def T(self) -> "iPETScMatrix":
        return iPETScMatrix(self._mat.T.tocsr())

    @property
    def H(self) -> "iPETScMatrix":
        return iPETScMatrix(self._mat.conj().T.tocsr())

    def __str__(self) -> str:
        return f"iPETScMatrix(shape={self.shape}, nnz={self.nonzero_entries})"

    # ---- element access (tests build small matrices entry by entry) ------------------------------------------------
    def __getitem__(self, idx):
        return self._mat[idx]

    def __setitem__(self, idx, value) -> None:
        lil = self._mat.tolil()
        lil[idx] = value
        self._mat = lil.tocsr()
        self._mat.sort_indices()

    def zero_all_entries(self) -> None:
        self._mat.data[:] = 0

    def assemble(self) -> None:
        self._mat.sort_indices()

    # ---- algebra used on the path ----------------------------------------------------------------------------------
    def as_scipy_array(self) -> sp.csr_matrix:
        """CSR triple ``ia, ja, aa`` as a scipy matrix (``FEM/utils.py:585-588``)."""
        return self._mat

    def is_numerically_hermitian(self, tol: float = 1e-4) -> bool:
        d = self._mat - self._mat.conj().T
        return bool(np.sqrt(np.sum(np.abs(d.data) ** 2)) < tol) if d.nnz else True

    def is_numerically_symmetric(self, tol: float = 1e-4) -> bool:
        d = self._mat - self._mat.T
        return bool(np.sqrt(np.sum(np.abs(d.data) ** 2)) < tol) if d.nnz else True

    def mult(self, x: iPETScVector, y: iPETScVector | None = None) -> iPETScVector:
        out = iPETScVector(self._mat @ x.raw)
        if y is not None:
            y._a = out._a
            return y
        return out

    def __matmul__(self, x):
        if isinstance(x, iPETScVector):
            return self.mult(x)
        return self._mat @ x

    def export(self, path: Path) -> None:
        """Write MatrixMarket ``coordinate general`` (``FEM/utils.py:616-636``), explicit zeros included."""
        from .mmio import write_matrix_market

        write_matrix_market(Path(path), self._mat)
