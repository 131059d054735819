"""ORACLE (test infrastructure, CPU only) -- ctypes front end of ``oracle/kernels.c``.

Only ``tests/``, ``bench.py`` (``cpu_baseline`` leg) and ``__graft_entry__.smoke()`` may import this.
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liblsa_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """Compile kernels.c with gcc (a few hundred ms)."""
    src = [_HERE / "kernels.c", _HERE / "kernels_impl.h"]
    if not force and _LIB_PATH.exists() and all(_LIB_PATH.stat().st_mtime >= s.stat().st_mtime for s in src):
        return _LIB_PATH
    _LIB_PATH.parent.mkdir(exist_ok=True)
    subprocess.check_call(
        ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", str(_LIB_PATH), str(src[0]), "-lm"], cwd=str(_HERE)
    )
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(build()))
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype) -> str:
    return "c128" if np.dtype(dtype).kind == "c" else "f64"


def _csr(A):
    rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(A.indices, dtype=np.int32)
    return rp, ci, np.ascontiguousarray(A.data)


def spmv(A, x: np.ndarray) -> np.ndarray:
    """y = A x; A real or complex CSR, x real or complex (real A with complex x uses the mixed kernel)."""
    rp, ci, v = _csr(A)
    n = A.shape[0]
    if v.dtype.kind != "c" and x.dtype.kind == "c":
        x = np.ascontiguousarray(x, dtype=np.complex128)
        y = np.empty(n, dtype=np.complex128)
        lib().orc_spmv_rc(n, _p(rp), _p(ci), _p(v), _p(x), _p(y))
        return y
    dt = np.result_type(v.dtype, x.dtype)
    v = np.ascontiguousarray(v, dtype=dt)
    x = np.ascontiguousarray(x, dtype=dt)
    y = np.empty(n, dtype=dt)
    getattr(lib(), f"orc_spmv_{_sfx(dt)}")(n, _p(rp), _p(ci), _p(v), _p(x), _p(y))
    return y


def axpby_same_pattern(A, M, alpha: complex, beta: complex) -> np.ndarray:
    """values of alpha*A + beta*M (A, M real, one shared pattern) as complex128."""
    a = np.ascontiguousarray(A.data, dtype=np.float64)
    m = np.ascontiguousarray(M.data, dtype=np.float64)
    c = np.empty(a.shape[0], dtype=np.complex128)
    f = lib().orc_axpby_rc
    f.argtypes = [ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_double] * 4 + [ctypes.c_void_p]
    alpha, beta = complex(alpha), complex(beta)
    f(a.shape[0], _p(a), _p(m), alpha.real, alpha.imag, beta.real, beta.imag, _p(c))
    return c


class ILU0:
    """ILU(0) of a CSR matrix with sorted columns (PETSc PCILU, levels = 0, natural ordering)."""

    def __init__(self, C, shift_tol: float = 0.0):
        self.rp, self.ci, v = _csr(C)
        self.n = C.shape[0]
        self.v = v.copy()
        self.dt = self.v.dtype
        self.diag = np.empty(self.n, dtype=np.int32)
        ns = ctypes.c_int(0)
        rc = getattr(lib(), f"orc_ilu0_{_sfx(self.dt)}")(
            self.n, _p(self.rp), _p(self.ci), _p(self.v), _p(self.diag), ctypes.c_double(shift_tol), ctypes.byref(ns)
        )
        if rc != 0:
            raise ZeroDivisionError(f"ILU(0): zero pivot / missing diagonal at row {-rc - 1}")
        self.nshift = ns.value

    def solve(self, b: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(b, dtype=self.dt)
        x = np.empty_like(b)
        getattr(lib(), f"orc_ilu_solve_{_sfx(self.dt)}")(self.n, _p(self.rp), _p(self.ci), _p(self.v), _p(self.diag), _p(b), _p(x))
        return x

    def lower(self, b):
        b = np.ascontiguousarray(b, dtype=self.dt)
        x = np.empty_like(b)
        getattr(lib(), f"orc_sptrsv_lower_unit_{_sfx(self.dt)}")(self.n, _p(self.rp), _p(self.ci), _p(self.v), _p(self.diag), _p(b), _p(x))
        return x

    def upper(self, b):
        b = np.ascontiguousarray(b, dtype=self.dt)
        x = np.empty_like(b)
        getattr(lib(), f"orc_sptrsv_upper_{_sfx(self.dt)}")(self.n, _p(self.rp), _p(self.ci), _p(self.v), _p(self.diag), _p(b), _p(x))
        return x


def iluk_pattern(C, levels: int):
    """CSR matrix with C's values scattered into the ILU(levels) pattern (fill entries are explicit zeros)."""
    import scipy.sparse as sp

    rp, ci, v = _csr(C)
    n = C.shape[0]
    f = lib().orc_iluk_symbolic
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    cap = int(C.nnz * (1 + 2 * levels)) + n
    while True:
        orp = np.empty(n + 1, dtype=np.int32)
        oci = np.empty(cap, dtype=np.int32)
        need = f(n, _p(rp), _p(ci), levels, cap, _p(orp), _p(oci))
        if need <= cap:
            break
        cap = int(need)
    oci = oci[:need].copy()
    # scatter values: positions of original entries inside the new rows
    P = sp.csr_matrix((np.ones(need, dtype=np.int8), oci, orp), shape=C.shape)
    out = sp.csr_matrix((np.zeros(need, dtype=v.dtype), oci, orp), shape=C.shape)
    key_new = np.repeat(np.arange(n, dtype=np.int64), np.diff(orp)) * n + oci
    key_old = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp)) * n + ci
    pos = np.searchsorted(key_new, key_old)
    out.data[pos] = v
    return out
