"""ctypes binding of ``liblsa_hip.so`` (declared in ``include/lsa_hip.h``).

This is the thin host layer the SLEPc-shaped API in ``Solver/eigen.py`` sits on.  There is no CPU fallback:
importing works anywhere (so the symbol table can be checked without a GPU), but creating a
:class:`Context` raises ``RuntimeError`` when the library or a GPU is missing.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

LSA_F64, LSA_C128 = 0, 1

_STATUS_NAMES = {
    0: "LSA_OK",
    -1: "LSA_ERR_ARG",
    -2: "LSA_ERR_HIP",
    -3: "LSA_ERR_ZERO_PIVOT",
    -4: "LSA_ERR_DIVERGED",
    -5: "LSA_ERR_NONFINITE",
    -6: "LSA_ERR_TIMEOUT",
    -7: "LSA_ERR_COMM",
    -8: "LSA_ERR_OOM",
}

# the status codes of include/lsa_hip.h by name (LSA_OK, LSA_ERR_ARG, ... LSA_ERR_OOM)
globals().update({name: code for code, name in _STATUS_NAMES.items()})

LIB_PATH = Path(os.environ.get("LSA_HIP_LIB", Path(__file__).resolve().parent / "liblsa_hip.so"))


class LsaError(RuntimeError):
    """A device-side or numerical failure reported by the library (status < -1)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{_STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class lsa_stats(ctypes.Structure):
    _fields_ = [
        ("op_applies", ctypes.c_int64),
        ("gmres_iters", ctypes.c_int64),
        ("spmv_calls", ctypes.c_int64),
        ("sptrsv_calls", ctypes.c_int64),
        ("last_rel_res", ctypes.c_double),
        ("max_rel_res", ctypes.c_double),
        ("seconds_factor", ctypes.c_double),
        ("seconds_solve", ctypes.c_double),
        ("stagnated_solves", ctypes.c_int32),
        ("pc_fallback", ctypes.c_int32),
        ("backward_accepted", ctypes.c_int32),
        ("analysis_reused", ctypes.c_int32),
        ("refined_solves", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
    ]


class lsa_op_options(ctypes.Structure):
    _fields_ = [
        ("ilu_levels", ctypes.c_int32),
        ("ilu_shift", ctypes.c_double),
        ("ksp_rtol", ctypes.c_double),
        ("ksp_restart", ctypes.c_int32),
        ("ksp_maxit", ctypes.c_int32),
        ("pc_type", ctypes.c_int32),
        ("antishift", ctypes.c_double * 2),
    ]


class lsa_ks_options(ctypes.Structure):
    _fields_ = [
        ("nev", ctypes.c_int32),
        ("max_restarts", ctypes.c_int32),
        ("tol", ctypes.c_double),
        ("which", ctypes.c_int32),
        ("transform", ctypes.c_int32),
        ("sigma", ctypes.c_double * 2),
        ("antishift", ctypes.c_double * 2),
        ("target", ctypes.c_double * 2),
        ("seed", ctypes.c_uint64),
        ("keep_fraction", ctypes.c_double),
    ]


class lsa_ks_result(ctypes.Structure):
    _fields_ = [
        ("nconv", ctypes.c_int32),
        ("nout", ctypes.c_int32),
        ("restarts", ctypes.c_int32),
        ("pad", ctypes.c_int32),
        ("op_applies", ctypes.c_int64),
        ("next_unconverged", ctypes.c_double),
        ("seconds_expand", ctypes.c_double),
        ("seconds_dense", ctypes.c_double),
        ("seconds_restart", ctypes.c_double),
    ]


_P = ctypes.c_void_p
_I32, _I64, _DBL = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
_PP = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); kept in step with include/lsa_hip.h (tests/test_abi.py checks both directions)
SIGNATURES = {
    "lsa_ctx_create": (ctypes.c_int, [ctypes.c_int, _PP]),
    "lsa_ctx_destroy": (None, [_P]),
    "lsa_last_error": (ctypes.c_char_p, [_P]),
    "lsa_ctx_synchronize": (ctypes.c_int, [_P]),
    "lsa_ctx_arch": (ctypes.c_char_p, [_P]),
    "lsa_vec_create": (ctypes.c_int, [_P, _I64, ctypes.c_int, _PP]),
    "lsa_vec_destroy": (None, [_P]),
    "lsa_vec_upload": (ctypes.c_int, [_P, _P, _P]),
    "lsa_vec_download": (ctypes.c_int, [_P, _P, _P]),
    "lsa_csr_upload": (ctypes.c_int, [_P, _I32, _I64, _P, _P, _P, ctypes.c_int, _PP]),
    "lsa_mat_destroy": (None, [_P]),
    "lsa_mat_download_values": (ctypes.c_int, [_P, _P, _P]),
    "lsa_csr_axpby": (ctypes.c_int, [_P, _P, _P, _DBL * 2, _DBL * 2, ctypes.c_int, _PP]),
    "lsa_spmv": (ctypes.c_int, [_P, _P, _P, _P]),
    "lsa_spmv_transpose": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P]),
    "lsa_spmv_info": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_char_p, _I32, ctypes.POINTER(_I64)]),
    "lsa_spmv_time": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.POINTER(_DBL)]),
    "lsa_ilu_create": (ctypes.c_int, [_P, _P, ctypes.c_int, _DBL, _PP]),
    "lsa_ilu_destroy": (None, [_P]),
    "lsa_ilu_set_algorithm": (ctypes.c_int, [_P, _P, ctypes.c_int, _I32]),
    "lsa_ilu_solve": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P]),
    "lsa_ilu_solve_time": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.POINTER(_DBL)]),
    "lsa_ilu_info": (ctypes.c_int, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32)]),
    "lsa_ilu_download": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "lsa_nd_analyse": (ctypes.c_int, [_I32, _P, _P, _I32, _P, _PP]),
    "lsa_nd_order": (ctypes.c_int, [_I32, _P, _P, _I32, _P, _PP]),
    "lsa_nd_analyse_tree": (ctypes.c_int, [_I32, _P, _P, _I32, _P, _P, _P, _P, _I32, _I32, _PP]),
    "lsa_nd_sym_export_dist": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "lsa_nd_sym_export_top": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "lsa_ndlu_inertia": (ctypes.c_int, [_P, _P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "lsa_dense_sym_inertia": (ctypes.c_int, [_I32, _P, _I32, _DBL, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "lsa_nd_sym_error": (ctypes.c_char_p, [_P]),
    "lsa_nd_sym_destroy": (None, [_P]),
    "lsa_nd_sym_info": (ctypes.c_int, [_P, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I64),
                                       ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_DBL)]),
    "lsa_nd_sym_export": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "lsa_nd_sym_memory": (ctypes.c_int, [_P, _I32, _I64, _P, _P, _P, _P]),
    "lsa_nd_sym_export_tables": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "lsa_ndlu_create": (ctypes.c_int, [_P, _P, _I32, _PP]),
    "lsa_ndlu_prepare": (ctypes.c_int, [_P, _P, ctypes.c_int, _I32, _P]),
    "lsa_ndlu_prepare_tree": (ctypes.c_int, [_P, _P, ctypes.c_int, _I32, _P, _P, _P]),
    "lsa_ndlu_create_tree": (ctypes.c_int, [_P, _P, _I32, _P, _P, _P, _P, _PP]),
    "lsa_ndlu_refactor": (ctypes.c_int, [_P, _P, _P]),
    "lsa_ndlu_destroy": (None, [_P]),
    "lsa_ndlu_solve": (ctypes.c_int, [_P, _P, _P, _P]),
    "lsa_ndlu_solve_adjoint": (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P]),
    "lsa_ndlu_solve_time": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.POINTER(_DBL)]),
    "lsa_ndlu_info": (ctypes.c_int, [_P, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I64),
                                     ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I32), ctypes.POINTER(_DBL), ctypes.POINTER(_DBL)]),
    "lsa_gmres": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, _DBL, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_I32), ctypes.POINTER(_DBL)]),
    "lsa_op_create": (ctypes.c_int, [_P, _P, _P, _DBL * 2, ctypes.c_int, ctypes.POINTER(lsa_op_options), _PP]),
    "lsa_op_create_sharded": (ctypes.c_int, [_P, _P, _P, _P, _P, _DBL * 2, ctypes.c_int, ctypes.POINTER(lsa_op_options), _PP]),
    "lsa_op_create_dist": (ctypes.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _DBL * 2, ctypes.c_int, ctypes.POINTER(lsa_op_options), _PP]),
    "lsa_op_destroy": (None, [_P]),
    "lsa_op_apply": (ctypes.c_int, [_P, _P, _P, _P]),
    "lsa_op_stats": (ctypes.c_int, [_P, ctypes.POINTER(lsa_stats)]),
    "lsa_op_set_adjoint": (ctypes.c_int, [_P, _P, ctypes.c_int]),
    "lsa_op_set_projection": (ctypes.c_int, [_P, _P, _P]),
    "lsa_krylov_create": (ctypes.c_int, [_P, _P, _I32, _PP]),
    "lsa_krylov_destroy": (None, [_P]),
    "lsa_krylov_set_row_permutation": (ctypes.c_int, [_P, _P, _P]),
    "lsa_krylov_set_start": (ctypes.c_int, [_P, _P, _P]),
    "lsa_krylov_inject": (ctypes.c_int, [_P, _P, _I32, _P]),
    "lsa_krylov_extend": (ctypes.c_int, [_P, _P, _I32, _I32, _P, _I32, ctypes.POINTER(_I32)]),
    "lsa_krylov_restart": (ctypes.c_int, [_P, _P, _I32, _I32, _P, _I32]),
    "lsa_krylov_ritz_vectors": (ctypes.c_int, [_P, _P, _I32, _I32, _P, _I32, ctypes.c_int, _P]),
    "lsa_krylov_imag_norms": (ctypes.c_int, [_P, _I32, _P]),
    "lsa_krylov_shape": (ctypes.c_int, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I32)]),
    "lsa_mat_rows": (_I64, [_P]),
    "lsa_krylov_solve": (ctypes.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "lsa_eigs_sinvert": (ctypes.c_int, [_P, _P, _P, _DBL * 2, _I32, _I32, _DBL, _I32, ctypes.POINTER(lsa_op_options), _P, _P, _I32, _P, _P, _P, _P, _P]),
    "lsa_dense_schur": (ctypes.c_int, [_I32, _P, _I32, _P, _I32]),
    "lsa_dense_schur_reorder": (ctypes.c_int, [_I32, _P, _I32, _P, _I32, _P, ctypes.POINTER(_I32)]),
    "lsa_dense_tri_eigenvectors": (ctypes.c_int, [_I32, _P, _I32, _P, _I32]),
    "lsa_eig_residuals": (ctypes.c_int, [_P, _P, _P, _I32, _P, _P, _P]),
    "lsa_mm_open": (ctypes.c_int, [ctypes.c_char_p, _PP, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I64), ctypes.POINTER(ctypes.c_int)]),
    "lsa_mm_read_csr": (ctypes.c_int, [_P, _P, _P, _P]),
    "lsa_mm_error": (ctypes.c_char_p, [_P]),
    "lsa_mm_close": (None, [_P]),
    "lsa_comm_unique_id": (ctypes.c_int, [_P]),
    "lsa_comm_init": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P]),
    "lsa_comm_init_host": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, _P]),
    "lsa_comm_stats": (ctypes.c_int, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "lsa_comm_selftest": (ctypes.c_int, [_P, _I64]),
    "lsa_csr_upload_shard": (ctypes.c_int, [_P, _I32, _I32, _I32, _I64, _P, _P, _P, ctypes.c_int, _PP]),
}

_HOST_GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen liblsa_hip.so and attach the signatures; raises RuntimeError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C lsa-fw_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the eigen path."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float64:
        return LSA_F64
    if dt == np.complex128:
        return LSA_C128
    raise ValueError(f"unsupported dtype {dt}: use float64 or complex128")


def _np_dtype(code: int):
    return np.complex128 if code == LSA_C128 else np.float64


_host_blas_limit = None


def _calm_host_blas() -> None:
    """Opt-in (``LSA_HOST_BLAS_THREADS=<n>``): pin the host BLAS pools to ``n`` threads for the life of the process.

    The GPU path is a chain of dependent launches and leans on the HIP runtime's helper thread.  numpy's OpenBLAS workers
    spin for ~0.1 s after any threaded call; on a 16-core share of a GPU node 64 of them starve that thread and the next
    factorisation is 1.3-2x slower (``tools/micro/after_solve.py``).  A library must not change its caller's thread pools
    behind its back, so nothing happens unless the variable is set (``bench.py`` and the examples set it to 1); the dense
    80 x 80 algebra of the Krylov-Schur driver limits the pools only for the duration of its own LAPACK calls."""
    global _host_blas_limit
    want = os.environ.get("LSA_HOST_BLAS_THREADS", "keep")
    if _host_blas_limit is not None or want == "keep":
        return
    try:
        from threadpoolctl import threadpool_limits

        _host_blas_limit = threadpool_limits(limits=max(1, int(want)), user_api="blas")
    except Exception:  # threadpoolctl missing or an unknown BLAS: the path still works, only slower under oversubscription
        _host_blas_limit = False


class Context:
    """One GPU + one HIP stream.  Not thread-safe (mirrors the single blocking ``eps.solve()`` of the reference)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        _calm_host_blas()
        h = ctypes.c_void_p()
        rc = self._lib.lsa_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(
                f"lsa_ctx_create(device={device}) failed with {_STATUS_NAMES.get(rc, rc)}: no usable AMD GPU. "
                "The eigen path has no CPU fallback."
            )
        self.handle = h
        self.device = device

    @property
    def arch(self) -> str:
        return self._lib.lsa_ctx_arch(self.handle).decode()

    def check(self, rc: int) -> None:
        if rc == 0:
            return
        msg = self._lib.lsa_last_error(self.handle).decode(errors="replace")
        if rc == -1:
            raise ValueError(msg)
        raise LsaError(rc, msg)

    def synchronize(self) -> None:
        self.check(self._lib.lsa_ctx_synchronize(self.handle))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.lsa_ctx_destroy(self.handle)
            self.handle = None

    # multi-GPU bootstrap -------------------------------------------------------------------------------
    def unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(128)
        rc = self._lib.lsa_comm_unique_id(buf)
        if rc != 0:
            raise LsaError(rc, "RCCL unavailable")
        return buf.raw

    def comm_init(self, nranks: int, rank: int, uid: bytes) -> None:
        buf = ctypes.create_string_buffer(uid, 128)
        self.check(self._lib.lsa_comm_init(self.handle, nranks, rank, buf))

    def comm_init_host(self, nranks: int, rank: int, exchange) -> None:
        """Host-staged all-gather: ``exchange(buffer)`` receives a writable uint8 numpy view of ``nranks`` equal blocks with
        this rank's block filled in and must fill the others (``lsa_comm_init_host``)."""

        def _cb(host_buf, bytes_per_rank, _user):
            try:
                view = np.ctypeslib.as_array((ctypes.c_uint8 * (int(bytes_per_rank) * nranks)).from_address(host_buf))
                exchange(view.reshape(nranks, int(bytes_per_rank)))
                return 0
            except Exception:  # noqa: BLE001  (must not propagate through the C frame)
                import traceback

                traceback.print_exc()
                return 1

        self._host_cb = _HOST_GATHER_FN(_cb)  # keep the trampoline alive as long as the context
        self.check(self._lib.lsa_comm_init_host(self.handle, nranks, rank, ctypes.cast(self._host_cb, ctypes.c_void_p), None))

    def comm_selftest(self, nbytes: int = 1 << 20) -> None:
        """RCCL with one rank on this device: open, unique id, communicator, one in-place all-gather, destroy (raises on failure)."""
        self.check(self._lib.lsa_comm_selftest(self.handle, int(nbytes)))

    def comm_stats(self) -> dict:
        calls, nbytes = _I64(0), _I64(0)
        self._lib.lsa_comm_stats(self.handle, ctypes.byref(calls), ctypes.byref(nbytes))
        return {"allgather_calls": calls.value, "allgather_bytes_received": nbytes.value}


class DeviceVector:
    def __init__(self, ctx: Context, n: int, dtype=np.complex128):
        self.ctx = ctx
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.lsa_vec_create(ctx.handle, self.n, _dtype_code(dtype), ctypes.byref(h)))
        self.handle = h

    @classmethod
    def from_numpy(cls, ctx: Context, a: np.ndarray) -> "DeviceVector":
        a = np.ascontiguousarray(a)
        if a.dtype not in (np.float64, np.complex128):
            a = a.astype(np.complex128 if a.dtype.kind == "c" else np.float64)
        v = cls(ctx, a.shape[0], a.dtype)
        v.upload(a)
        return v

    def upload(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != (self.n,):
            raise ValueError(f"expected shape ({self.n},), got {a.shape}")
        self.ctx.check(self.ctx._lib.lsa_vec_upload(self.ctx.handle, self.handle, _ptr(a)))

    def numpy(self) -> np.ndarray:
        out = np.empty(self.n, dtype=self.dtype)
        self.ctx.check(self.ctx._lib.lsa_vec_download(self.ctx.handle, self.handle, _ptr(out)))
        return out

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.lsa_vec_destroy(self.handle)
            self.handle = None


class CsrMatrix:
    """CSR matrix resident in HBM (int32 indices, float64 or complex128 values, sorted columns)."""

    def __init__(self, ctx: Context, handle, shape, nnz: int, dtype, keep=()):
        self.ctx, self.handle, self.shape, self.nnz, self.dtype = ctx, handle, shape, int(nnz), np.dtype(dtype)
        self._keep = keep  # matrices whose index arrays this one shares

    @classmethod
    def from_scipy(cls, ctx: Context, A) -> "CsrMatrix":
        import scipy.sparse as sp

        A = sp.csr_matrix(A)
        if A.shape[0] != A.shape[1]:
            raise ValueError("lsa_csr_upload handles square matrices; use from_scipy_shard for row blocks")
        if not A.has_sorted_indices:
            A = A.copy()
            A.sort_indices()
        dt = np.complex128 if A.dtype.kind == "c" else np.float64
        rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ci = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data, dtype=dt)
        h = ctypes.c_void_p()
        ctx.check(
            ctx._lib.lsa_csr_upload(ctx.handle, A.shape[0], A.nnz, _ptr(rp), _ptr(ci), _ptr(val), _dtype_code(dt), ctypes.byref(h))
        )
        return cls(ctx, h, A.shape, A.nnz, dt)

    @classmethod
    def from_scipy_shard(cls, ctx: Context, A_rows, n_global: int, row0: int) -> "CsrMatrix":
        import scipy.sparse as sp

        A = sp.csr_matrix(A_rows)
        A.sort_indices()
        dt = np.complex128 if A.dtype.kind == "c" else np.float64
        rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ci = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data, dtype=dt)
        h = ctypes.c_void_p()
        ctx.check(
            ctx._lib.lsa_csr_upload_shard(
                ctx.handle, n_global, row0, row0 + A.shape[0], A.nnz, _ptr(rp), _ptr(ci), _ptr(val), _dtype_code(dt), ctypes.byref(h)
            )
        )
        return cls(ctx, h, (A.shape[0], n_global), A.nnz, dt)

    def prepare_lu(self, complex_factors: bool, leaf_size: int = 0, constraint=None) -> None:
        """Pattern-only phase of the nested-dissection LU for matrices with this pattern (``lsa_ndlu_prepare``);
        ``constraint``: optional mask of the zero-diagonal unknowns to eliminate after their neighbours."""
        flags = None if constraint is None else np.ascontiguousarray(constraint, dtype=np.int8)
        self.ctx.check(self.ctx._lib.lsa_ndlu_prepare(self.ctx.handle, self.handle, LSA_C128 if complex_factors else LSA_F64, int(leaf_size),
                                                      None if flags is None else _ptr(flags)))

    def prepare_lu_tree(self, complex_factors: bool, first, size, parent) -> None:
        """The same phase with the caller's forest (``lsa_ndlu_prepare_tree``): for a matrix already permuted into the
        elimination order of :func:`nd_order`, whose forest this is."""
        first, size, parent = (np.ascontiguousarray(a, dtype=np.int32) for a in (first, size, parent))
        self.ctx.check(self.ctx._lib.lsa_ndlu_prepare_tree(self.ctx.handle, self.handle, LSA_C128 if complex_factors else LSA_F64, len(parent),
                                                           _ptr(first), _ptr(size), _ptr(parent)))

    def axpby(self, other: "CsrMatrix", alpha: complex, beta: complex, dtype=None) -> "CsrMatrix":
        """alpha*self + beta*other on the shared pattern (MatAXPY, Solver/eigen2.py:110-111)."""
        alpha, beta = complex(alpha), complex(beta)
        if dtype is None:
            cplx = alpha.imag != 0 or beta.imag != 0 or self.dtype.kind == "c" or other.dtype.kind == "c"
            dtype = np.complex128 if cplx else np.float64
        h = ctypes.c_void_p()
        self.ctx.check(
            self.ctx._lib.lsa_csr_axpby(
                self.ctx.handle, self.handle, other.handle, (_DBL * 2)(alpha.real, alpha.imag), (_DBL * 2)(beta.real, beta.imag),
                _dtype_code(dtype), ctypes.byref(h),
            )
        )
        return CsrMatrix(self.ctx, h, self.shape, self.nnz, dtype, keep=(self, other))

    def values(self) -> np.ndarray:
        out = np.empty(self.nnz, dtype=self.dtype)
        self.ctx.check(self.ctx._lib.lsa_mat_download_values(self.ctx.handle, self.handle, _ptr(out)))
        return out

    def matvec(self, x: DeviceVector, y: DeviceVector) -> None:
        self.ctx.check(self.ctx._lib.lsa_spmv(self.ctx.handle, self.handle, x.handle, y.handle))

    def rmatvec(self, x: DeviceVector, y: DeviceVector, conj: bool = True) -> None:
        self.ctx.check(self.ctx._lib.lsa_spmv_transpose(self.ctx.handle, self.handle, int(conj), x.handle, y.handle))

    def matvec_info(self, vector_dtype=np.complex128) -> dict:
        """Kernel ``lsa_spmv`` launches for this matrix / vector type and the bytes it moves per launch."""
        name = ctypes.create_string_buffer(128)
        nbytes = _I64(0)
        self.ctx.check(self.ctx._lib.lsa_spmv_info(self.ctx.handle, self.handle, _dtype_code(vector_dtype), name, 128, ctypes.byref(nbytes)))
        return {"kernel": name.value.decode(), "bytes_moved": nbytes.value}

    def time_matvec(self, x: DeviceVector, y: DeviceVector, iters: int) -> float:
        """Mean milliseconds per SpMV launch over ``iters`` back-to-back launches (HIP events on the library's stream)."""
        ms = _DBL(0.0)
        self.ctx.check(self.ctx._lib.lsa_spmv_time(self.ctx.handle, self.handle, x.handle, y.handle, int(iters), ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.lsa_mat_destroy(self.handle)
            self.handle = None


class Ilu:
    """ILU(k) factors + triangular-solve schedule on the device."""

    def __init__(self, ctx: Context, C: CsrMatrix, levels: int = 0, shift_tol: float = 0.0):
        self.ctx, self._C = ctx, C
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.lsa_ilu_create(ctx.handle, C.handle, int(levels), float(shift_tol), ctypes.byref(h)))
        self.handle = h
        self.n = C.shape[0]
        self.dtype = C.dtype

    def info(self) -> dict:
        nnz, ll, lu, ns = _I64(0), _I32(0), _I32(0), _I32(0)
        self.ctx._lib.lsa_ilu_info(self.handle, ctypes.byref(nnz), ctypes.byref(ll), ctypes.byref(lu), ctypes.byref(ns))
        return {"nnz": nnz.value, "levels_lower": ll.value, "levels_upper": lu.value, "nshift": ns.value}

    def set_algorithm(self, algo: int, block_size: int = 0) -> None:
        """0 = level launches, 1 = sync-free, 2 = blocked with inverted diagonal blocks (see include/lsa_hip.h)."""
        self.ctx.check(self.ctx._lib.lsa_ilu_set_algorithm(self.ctx.handle, self.handle, int(algo), int(block_size)))

    def solve(self, b: DeviceVector, x: DeviceVector, which: int = 2) -> None:
        self.ctx.check(self.ctx._lib.lsa_ilu_solve(self.ctx.handle, self.handle, int(which), b.handle, x.handle))

    def time_solve(self, b: DeviceVector, x: DeviceVector, iters: int, which: int = 2) -> float:
        """Mean milliseconds per triangular solve (HIP events on the library's stream)."""
        ms = _DBL(0.0)
        self.ctx.check(self.ctx._lib.lsa_ilu_solve_time(self.ctx.handle, self.handle, int(which), b.handle, x.handle, int(iters), ctypes.byref(ms)))
        return ms.value

    def factors(self):
        """(rowptr, col, val) of the combined L\\U factor."""
        nnz = self.info()["nnz"]
        rp = np.empty(self.n + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=self.dtype)
        self.ctx.check(self.ctx._lib.lsa_ilu_download(self.ctx.handle, self.handle, _ptr(rp), _ptr(ci), _ptr(val)))
        return rp, ci, val

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.lsa_ilu_destroy(self.handle)
            self.handle = None


class NdAnalysis:
    """Host-only analysis of the nested-dissection multifrontal LU (``lsa_nd_analyse``): ordering, elimination forest
    and the index tables of the device kernels.  Needs no GPU; used by the tests and by sizing tools."""

    def __init__(self, A, leaf_size: int = 0, constraint=None, tree=None, rank: int = 0, nranks: int = 1, order_only: bool = False):
        """``order_only``: the elimination order and the forest without the index tables (``lsa_nd_order``).
        ``constraint``: optional boolean mask of the unknowns with a numerically zero diagonal (eliminated after all
        their neighbours: the pressure rows of a saddle-point matrix).  ``tree``: instead of dissecting, take the forest
        ``{"first", "size", "parent"[, "owner"]}`` (``lsa_nd_analyse_tree``), localised for ``rank`` of ``nranks``."""
        import scipy.sparse as sp

        A = sp.csr_matrix(A)
        if A.shape[0] != A.shape[1]:
            raise ValueError("the analysis needs a square pattern")
        self._lib = load_library()
        self.n, self.nnz = A.shape[0], A.nnz
        rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ci = np.ascontiguousarray(A.indices, dtype=np.int32)
        h = ctypes.c_void_p()
        flags = None if constraint is None else np.ascontiguousarray(constraint, dtype=np.int8)
        if flags is not None and flags.shape != (self.n,):
            raise ValueError("constraint mask must have one entry per row")
        if tree is None and order_only:
            rc = self._lib.lsa_nd_order(self.n, _ptr(rp), _ptr(ci), int(leaf_size), None if flags is None else _ptr(flags), ctypes.byref(h))
        elif tree is None:
            rc = self._lib.lsa_nd_analyse(self.n, _ptr(rp), _ptr(ci), int(leaf_size), None if flags is None else _ptr(flags), ctypes.byref(h))
        else:
            first, size, parent = (np.ascontiguousarray(tree[k], dtype=np.int32) for k in ("first", "size", "parent"))
            owner = None if tree.get("owner") is None else np.ascontiguousarray(tree["owner"], dtype=np.int32)
            rc = self._lib.lsa_nd_analyse_tree(self.n, _ptr(rp), _ptr(ci), len(parent), _ptr(first), _ptr(size), _ptr(parent),
                                               None if owner is None else _ptr(owner), int(rank), int(nranks), ctypes.byref(h))
        self.handle = h
        if rc != 0:
            msg = self._lib.lsa_nd_sym_error(h).decode(errors="replace")
            raise ValueError(msg)
        nt, nl, mf = _I32(0), _I32(0), _I32(0)
        ie, fe, fre, fl = _I64(0), _I64(0), _I64(0), _DBL(0.0)
        self._lib.lsa_nd_sym_info(h, ctypes.byref(nt), ctypes.byref(nl), ctypes.byref(mf), ctypes.byref(ie), ctypes.byref(fe), ctypes.byref(fre),
                                  ctypes.byref(fl))
        self.ntree, self.nlevels, self.max_front = nt.value, nl.value, mf.value
        self.index_entries, self.factor_entries, self.front_entries, self.flops = ie.value, fe.value, fre.value, fl.value

    def export(self) -> dict:
        """perm, node_start, parent, level, front_size, idx (see include/lsa_hip.h)."""
        out = {"perm": np.empty(self.n, np.int32), "node_start": np.empty(self.ntree + 1, np.int32), "parent": np.empty(self.ntree, np.int32),
               "level": np.empty(self.ntree, np.int32), "front_size": np.empty(self.ntree, np.int32), "idx": np.empty(self.index_entries, np.int32)}
        self._lib.lsa_nd_sym_export(self.handle, *[_ptr(out[k]) for k in ("perm", "node_start", "parent", "level", "front_size", "idx")])
        return out

    def memory(self, scalar_bytes: int = 16, work_budget_bytes: int = 0, detail: bool = False) -> dict:
        """Device memory a factorisation of this analysis allocates, in bytes (``lsa_nd_sym_memory``); with ``detail`` also the
        per-node plan (update-arena and working-arena offsets in scalars, chunk of every node)."""
        out = np.zeros(8, np.int64)
        upd = np.zeros(self.ntree, np.int64) if detail else None
        work = np.zeros(self.ntree, np.int64) if detail else None
        chunk = np.full(self.ntree, -1, np.int32) if detail else None
        rc = self._lib.lsa_nd_sym_memory(self.handle, int(scalar_bytes), int(work_budget_bytes), _ptr(out), None if upd is None else _ptr(upd),
                                         None if work is None else _ptr(work), None if chunk is None else _ptr(chunk))
        if rc != 0:
            raise ValueError("lsa_nd_sym_memory: bad argument")
        rec = {"factors": int(out[0]), "working_arena": int(out[1]), "update_arena": int(out[2]), "exchange_region": int(out[3]), "sweep_buffers": int(out[4]),
               "chunks": int(out[5]), "largest_front": int(out[6]), "index_tables": int(out[7])}
        rec["total"] = rec["factors"] + rec["working_arena"] + rec["update_arena"] + rec["sweep_buffers"] + rec["index_tables"]
        if detail:
            rec.update({"upd_off": upd, "work_off": work, "chunk_of": chunk})
        return rec

    def export_tables(self) -> dict:
        """cmap, gptr, gidx, asm_dst, lvl_ptr, lvl_nodes (the tables the device kernels walk)."""
        ex = self.export()
        m = np.diff(ex["node_start"])
        b = ex["front_size"] - m
        scal = np.zeros(6, np.int64)
        self._lib.lsa_nd_sym_export_dist(self.handle, None, None, None, None, None, None, _ptr(scal))
        nlist = int(np.count_nonzero(np.ones(self.ntree)))  # lvl_nodes holds the nodes this rank works on (ghosts excluded)
        out = {"cmap": np.empty(int(b.sum()), np.int32), "gptr": np.empty(int((ex["front_size"] + 1).sum()), np.int32),
               "gidx": np.empty(int(b.sum()), np.int32), "asm_dst": np.empty(int(scal[3]), np.int64), "lvl_ptr": np.empty(self.nlevels + 1, np.int32),
               "lvl_nodes": np.full(nlist, -1, np.int32), "kind": np.empty(self.ntree, np.int32), "front_off": np.empty(self.ntree + 1, np.int64),
               "u_off": np.empty(self.ntree + 1, np.int64), "asm_src": np.empty(int(scal[3]), np.int32), "child_ptr": np.empty(self.ntree + 1, np.int32)}
        self._lib.lsa_nd_sym_export_tables(self.handle, *[_ptr(out[k]) for k in ("cmap", "gptr", "gidx", "asm_dst", "lvl_ptr", "lvl_nodes")])
        out["lvl_nodes"] = out["lvl_nodes"][: int(out["lvl_ptr"][-1])]
        self._lib.lsa_nd_sym_export_dist(self.handle, _ptr(out["kind"]), _ptr(out["front_off"]), _ptr(out["u_off"]), _ptr(out["asm_src"]),
                                         _ptr(out["child_ptr"]), None, None)
        out["child_idx"] = np.empty(int(out["child_ptr"][-1]), np.int32)
        self._lib.lsa_nd_sym_export_dist(self.handle, None, None, None, None, None, _ptr(out["child_idx"]), None)
        out["owner"] = np.zeros(self.ntree, np.int32)
        rows, exch, totals = np.zeros(4 * self.ntree, np.int32), np.zeros(4 * self.ntree, np.int64), np.zeros(2, np.int64)
        self._lib.lsa_nd_sym_export_top(self.handle, _ptr(out["owner"]), _ptr(rows), _ptr(exch), _ptr(totals))
        rows, exch = rows.reshape(-1, 4), exch.reshape(-1, 4)
        out.update({"brow0": rows[:, 0].copy(), "brow": rows[:, 1].copy(), "orow0": rows[:, 2].copy(), "orows": rows[:, 3].copy(),
                    "ux_base": exch[:, 0].copy(), "ux_stride": exch[:, 1].copy(), "xg_base": exch[:, 2].copy(), "xg_stride": exch[:, 3].copy(),
                    "u_entries": int(totals[0]), "xg_entries": int(totals[1])})
        out.update(ex)
        out.update({"front_slot": int(scal[0]), "u_slot": int(scal[1]), "phase_b_level": int(scal[2]), "nranks": int(scal[4]), "rank": int(scal[5])})
        return out

    def __del__(self):
        if getattr(self, "handle", None):
            self._lib.lsa_nd_sym_destroy(self.handle)
            self.handle = None


def nd_order(A, leaf_size: int = 0, constraint=None) -> dict:
    """Elimination order of the nested-dissection LU for the pattern of ``A`` (host only): ``perm`` (new -> old index) and the
    forest ``first`` / ``size`` / ``parent`` of contiguous index ranges of the permuted matrix ``A[perm][:, perm]``."""
    an = NdAnalysis(A, leaf_size, constraint=constraint, order_only=True)
    ex = {"perm": np.empty(an.n, np.int32), "node_start": np.empty(an.ntree + 1, np.int32), "parent": np.empty(an.ntree, np.int32),
          "level": np.empty(an.ntree, np.int32)}
    an._lib.lsa_nd_sym_export(an.handle, _ptr(ex["perm"]), _ptr(ex["node_start"]), _ptr(ex["parent"]), _ptr(ex["level"]), None, None)
    return {"perm": ex["perm"].astype(np.int64), "first": ex["node_start"][:-1].copy(), "size": np.diff(ex["node_start"]).astype(np.int32),
            "parent": ex["parent"], "level": ex["level"]}


class NdLu:
    """Nested-dissection multifrontal LU of a CSR matrix, resident on the device (``lsa_ndlu_*``)."""

    def __init__(self, ctx: Context, C: CsrMatrix, leaf_size: int = 0, tree: dict | None = None):
        """``tree``: ``{"first", "size", "parent"[, "owner"]}`` selects ``lsa_ndlu_create_tree`` (subtree-parallel when
        the context has more than one rank and ``owner`` is given)."""
        self.ctx, self._C = ctx, C
        h = ctypes.c_void_p()
        if tree is None:
            ctx.check(ctx._lib.lsa_ndlu_create(ctx.handle, C.handle, int(leaf_size), ctypes.byref(h)))
        else:
            first, size, parent = (np.ascontiguousarray(tree[k], dtype=np.int32) for k in ("first", "size", "parent"))
            owner = None if tree.get("owner") is None else np.ascontiguousarray(tree["owner"], dtype=np.int32)
            ctx.check(ctx._lib.lsa_ndlu_create_tree(ctx.handle, C.handle, len(parent), _ptr(first), _ptr(size), _ptr(parent),
                                                    None if owner is None else _ptr(owner), ctypes.byref(h)))
        self.handle = h
        self.n = C.shape[0]

    def refactor(self, C: CsrMatrix) -> None:
        self.ctx.check(self.ctx._lib.lsa_ndlu_refactor(self.ctx.handle, self.handle, C.handle))
        self._C = C

    def info(self) -> dict:
        nt, nl, mf, nla = _I32(0), _I32(0), _I32(0), _I32(0)
        fe, fre, ab = _I64(0), _I64(0), _I64(0)
        sa, sn = _DBL(0.0), _DBL(0.0)
        self.ctx._lib.lsa_ndlu_info(self.handle, ctypes.byref(nt), ctypes.byref(nl), ctypes.byref(mf), ctypes.byref(fe), ctypes.byref(fre),
                                    ctypes.byref(ab), ctypes.byref(nla), ctypes.byref(sa), ctypes.byref(sn))
        return {"tree_nodes": nt.value, "levels": nl.value, "max_front": mf.value, "factor_entries": fe.value, "front_entries": fre.value,
                "apply_bytes": ab.value, "apply_launches": nla.value, "seconds_analyse": sa.value, "seconds_numeric": sn.value}

    def solve(self, b: DeviceVector, x: DeviceVector) -> None:
        self.ctx.check(self.ctx._lib.lsa_ndlu_solve(self.ctx.handle, self.handle, b.handle, x.handle))

    def solve_adjoint(self, b: DeviceVector, x: DeviceVector, conj: bool = True) -> None:
        """x = C^-H b (``conj``) or C^-T b on the same factors."""
        self.ctx.check(self.ctx._lib.lsa_ndlu_solve_adjoint(self.ctx.handle, self.handle, int(conj), b.handle, x.handle))

    def time_solve(self, b: DeviceVector, x: DeviceVector, iters: int) -> float:
        ms = _DBL(0.0)
        self.ctx.check(self.ctx._lib.lsa_ndlu_solve_time(self.ctx.handle, self.handle, b.handle, x.handle, int(iters), ctypes.byref(ms)))
        return ms.value

    def inertia(self) -> tuple[int, int, int]:
        """(negative, zero, positive) eigenvalue counts of a REAL SYMMETRIC matrix from its factors (``lsa_ndlu_inertia``): for
        ``C = A - sigma M`` of a definite pencil, ``negative`` is the number of eigenvalues below ``sigma``."""
        ng, ze, ps = _I64(0), _I64(0), _I64(0)
        self.ctx.check(self.ctx._lib.lsa_ndlu_inertia(self.ctx.handle, self.handle, ctypes.byref(ng), ctypes.byref(ze), ctypes.byref(ps)))
        return ng.value, ze.value, ps.value

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.lsa_ndlu_destroy(self.handle)
            self.handle = None


def gmres(ctx: Context, C: CsrMatrix, pc: Ilu | None, b: DeviceVector, x: DeviceVector, *, rtol=1e-10, restart=200, maxit=2000,
          use_x0=False):
    """Right-preconditioned GMRES on the device; returns (iterations, relative residual)."""
    its, rr = _I32(0), _DBL(0.0)
    rc = ctx._lib.lsa_gmres(ctx.handle, C.handle, pc.handle if pc else None, b.handle, x.handle, int(use_x0), float(rtol), int(restart),
                            int(maxit), ctypes.byref(its), ctypes.byref(rr))
    ctx.check(rc)
    return its.value, rr.value


class ShiftInvertOperator:
    """``y = (A - sigma M)^-1 M x`` (mode 0, iSTType.SINVERT), ``y = M^-1 (A - sigma M) x`` (mode 1, iSTType.SHIFT) or
    ``y = (A - sigma M)^-1 (A + nu M) x`` (mode 2, iSTType.CAYLEY with antishift ``nu``)."""

    def __init__(self, ctx: Context, A: CsrMatrix, M: CsrMatrix | None, sigma: complex, *, mode: int = 0, antishift: complex = 0.0,
                 ilu_levels: int = 0,
                 ilu_shift: float = 0.0, ksp_rtol: float = 1e-11, ksp_restart: int = 200, ksp_maxit: int = 2000, pc_type: int = 1,
                 A_diag: CsrMatrix | None = None, M_diag: CsrMatrix | None = None, forest=None, rows: tuple[int, int] | None = None):
        """With ``A_diag`` (and ``M_diag``) given, ``A`` / ``M`` are this rank's row shards in the padded block layout
        (:mod:`lsa_hip.sharding`) and the preconditioner is block-Jacobi over ranks.  With ``forest`` (a
        :class:`lsa_hip.sharding.ForestPartition`) and ``rows`` (this rank's padded row range), ``A`` / ``M`` are the WHOLE
        padded matrices and the inner solve is the subtree-parallel exact LU (``lsa_op_create_dist``)."""
        self.ctx, self._A, self._M, self._Ad, self._Md = ctx, A, M, A_diag, M_diag
        sigma = complex(sigma)
        antishift = complex(antishift)
        opts = lsa_op_options(int(ilu_levels), float(ilu_shift), float(ksp_rtol), int(ksp_restart), int(ksp_maxit), int(pc_type),
                              (_DBL * 2)(antishift.real, antishift.imag))
        h = ctypes.c_void_p()
        sig = (_DBL * 2)(sigma.real, sigma.imag)
        if forest is not None:
            self._forest = tuple(np.ascontiguousarray(getattr(forest, k), dtype=np.int32) for k in ("first", "size", "parent", "owner"))
            f0, f1, f2, f3 = self._forest
            ctx.check(ctx._lib.lsa_op_create_dist(ctx.handle, A.handle, M.handle if M is not None else None, int(rows[0]), int(rows[1]), len(f2),
                                                  _ptr(f0), _ptr(f1), _ptr(f2), _ptr(f3), sig, int(mode), ctypes.byref(opts), ctypes.byref(h)))
        elif A_diag is None:
            ctx.check(ctx._lib.lsa_op_create(ctx.handle, A.handle, M.handle if M is not None else None, sig, int(mode), ctypes.byref(opts), ctypes.byref(h)))
        else:
            ctx.check(
                ctx._lib.lsa_op_create_sharded(ctx.handle, A.handle, M.handle if M is not None else None, A_diag.handle,
                                               M_diag.handle if M_diag is not None else None, sig, int(mode), ctypes.byref(opts), ctypes.byref(h))
            )
        self.handle = h
        self.n = A.shape[1]  # vector length (padded global length when sharded)
        self.sigma = sigma
        self.mode = mode

    def apply(self, x: DeviceVector, y: DeviceVector) -> None:
        self.ctx.check(self.ctx._lib.lsa_op_apply(self.ctx.handle, self.handle, x.handle, y.handle))

    def set_adjoint(self, on: bool = True) -> None:
        """Apply ``(A - sigma M)^-H M^H`` on the same factors (``lsa_op_set_adjoint``)."""
        self.ctx.check(self.ctx._lib.lsa_op_set_adjoint(self.ctx.handle, self.handle, int(on)))

    def set_projection(self, keep: np.ndarray | None) -> None:
        """Projected operator ``y = P Kfac^-1 Kmul x``, ``P = diag(keep)`` with 0/1 entries (``None`` removes it)."""
        if keep is None:
            self.ctx.check(self.ctx._lib.lsa_op_set_projection(self.ctx.handle, self.handle, None))
            return
        keep = np.ascontiguousarray(keep, dtype=np.float64)
        if keep.shape != (self.n,):
            raise ValueError(f"projection mask must have shape ({self.n},)")
        self.ctx.check(self.ctx._lib.lsa_op_set_projection(self.ctx.handle, self.handle, _ptr(keep)))

    def stats(self) -> dict:
        st = lsa_stats()
        self.ctx._lib.lsa_op_stats(self.handle, ctypes.byref(st))
        return {name: getattr(st, name) for name, _ in lsa_stats._fields_}

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.lsa_op_destroy(self.handle)
            self.handle = None


class KrylovBasis:
    """Arnoldi basis in HBM + the recurrences Krylov-Schur needs.  Implements the backend protocol of
    :func:`lsa_hip.krylov_schur.krylov_schur` (``n``, ``ncv``, ``inject``, ``extend``, ``restart``, ``ritz_vectors``)."""

    def __init__(self, ctx: Context, op: ShiftInvertOperator, ncv: int, mask: np.ndarray | None = None):
        """``mask`` (0/1 per slot) zeroes the padding slots of injected vectors in the sharded block layout."""
        self.ctx, self._op, self._mask = ctx, op, mask
        self.n, self.ncv = op.n, int(ncv)
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.lsa_krylov_create(ctx.handle, op.handle, self.ncv, ctypes.byref(h)))
        self.handle = h

    def set_row_permutation(self, perm: np.ndarray | None) -> None:
        """``perm[i]`` = the caller's index of basis row ``i``: Ritz vectors then come back in the caller's numbering."""
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        self.ctx.check(self.ctx._lib.lsa_krylov_set_row_permutation(self.ctx.handle, self.handle, None if p is None else _ptr(p)))

    def inject(self, j: int, v: np.ndarray) -> None:
        v = np.ascontiguousarray(v, dtype=np.complex128)
        if v.shape != (self.n,):
            raise ValueError(f"start vector must have shape ({self.n},)")
        if self._mask is not None:
            v = np.ascontiguousarray(v * self._mask)
        self.ctx.check(self.ctx._lib.lsa_krylov_inject(self.ctx.handle, self.handle, int(j), _ptr(v)))

    def extend(self, j0: int, j1: int, H: np.ndarray) -> int:
        """Arnoldi steps j0..j1-1 writing columns of the Fortran-ordered (ncv+1, ncv) complex H; returns the
        breakdown step or -1."""
        assert H.flags.f_contiguous and H.dtype == np.complex128
        bd = _I32(-1)
        self.ctx.check(self.ctx._lib.lsa_krylov_extend(self.ctx.handle, self.handle, int(j0), int(j1), _ptr(H), H.shape[0], ctypes.byref(bd)))
        return bd.value

    def restart(self, m: int, Q: np.ndarray) -> None:
        Q = np.asfortranarray(Q, dtype=np.complex128)
        self.ctx.check(self.ctx._lib.lsa_krylov_restart(self.ctx.handle, self.handle, int(m), Q.shape[1], _ptr(Q), Q.shape[0]))

    def ritz_vectors(self, m: int, Y: np.ndarray, normalise: bool = True, canonical_phase: bool = True) -> np.ndarray:
        """``X = V[:, :m] Y`` on the host; unit columns when ``normalise``; with ``canonical_phase`` also rotated so that the
        entry of largest magnitude is real and positive, and ``self.imag_norms`` = the 2-norms of the imaginary parts."""
        Y = np.asfortranarray(Y, dtype=np.complex128)
        X = np.empty((self.n, Y.shape[1]), dtype=np.complex128, order="F")
        mode = (1 if normalise else 0) | (2 if (normalise and canonical_phase) else 0)
        self.ctx.check(self.ctx._lib.lsa_krylov_ritz_vectors(self.ctx.handle, self.handle, int(m), Y.shape[1], _ptr(Y), Y.shape[0], mode, _ptr(X)))
        self.imag_norms = None
        if mode & 2:
            out = np.zeros(Y.shape[1], dtype=np.float64)
            if self.ctx._lib.lsa_krylov_imag_norms(self.handle, Y.shape[1], _ptr(out)) == 0:
                self.imag_norms = out
        return X

    def solve(self, nev: int, tol: float, max_restarts: int, which: int, transform: int, sigma: complex, *, antishift: complex = 0.0,
              target: complex | None = None, v0: np.ndarray | None = None, seed: int = 0, max_out: int | None = None, vectors: bool = True):
        """The whole Krylov-Schur iteration inside the library (``lsa_krylov_solve``: the dense algebra on the projected
        matrix is the library's own).  ``which``: EPSWhich code (``iEpsWhich.value``); ``transform``: 0 shift-invert,
        1 shift, 2 Cayley.  Returns a :class:`lsa_hip.krylov_schur.KrylovSchurResult` plus the eigenvalues ``lam``."""
        from .krylov_schur import KrylovSchurResult

        max_out = self.ncv if max_out is None else int(max_out)
        target = sigma if target is None else target
        o = lsa_ks_options(nev=int(nev), max_restarts=int(max_restarts), tol=float(tol), which=int(which), transform=int(transform),
                           sigma=(_DBL * 2)(complex(sigma).real, complex(sigma).imag), antishift=(_DBL * 2)(complex(antishift).real, complex(antishift).imag),
                           target=(_DBL * 2)(complex(target).real, complex(target).imag), seed=int(seed), keep_fraction=0.5)
        theta = np.zeros(max(max_out, 1), dtype=np.complex128)
        lam = np.zeros(max(max_out, 1), dtype=np.complex128)
        est = np.zeros(max(max_out, 1), dtype=np.float64)
        X = np.empty((self.n, max_out), dtype=np.complex128, order="F") if vectors else None
        res = lsa_ks_result()
        v = None
        if v0 is not None:
            v = np.ascontiguousarray(v0, dtype=np.complex128)
            if v.shape != (self.n,):
                raise ValueError(f"start vector must have shape ({self.n},)")
        mask = None if self._mask is None else np.ascontiguousarray(self._mask, dtype=np.float64)
        self.ctx.check(self.ctx._lib.lsa_krylov_solve(self.ctx.handle, self.handle, ctypes.byref(o), None if v is None else _ptr(v),
                                                      None if mask is None else _ptr(mask), max_out, _ptr(theta), _ptr(lam),
                                                      None if X is None else _ptr(X), _ptr(est), ctypes.byref(res)))
        k = res.nout
        self.imag_norms = None
        if X is not None and k > 0:
            out = np.zeros(k, dtype=np.float64)
            if self.ctx._lib.lsa_krylov_imag_norms(self.handle, k, _ptr(out)) == 0:
                self.imag_norms = out
        vec = np.zeros((self.n, 0), dtype=np.complex128) if X is None else (X[:, :k] if k == max_out else np.asfortranarray(X[:, :k]))
        r = KrylovSchurResult(theta[:k].copy(), vec, est[:k].copy(), int(res.nconv), int(res.restarts), int(res.op_applies),
                              [{"nconv": int(res.nconv), "next_unconverged": float(res.next_unconverged), "seconds_expand": res.seconds_expand,
                                "seconds_dense": res.seconds_dense, "seconds_restart": res.seconds_restart}])
        r.lam = lam[:k].copy()
        return r

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.lsa_krylov_destroy(self.handle)
            self.handle = None


def eigs_sinvert(ctx: Context, A: CsrMatrix, M: CsrMatrix | None, sigma: complex, nev: int, *, ncv: int = 0, tol: float = 1e-10, max_restarts: int = 500,
                 pc_type: int = 2, ksp_rtol: float = 1e-12, v0: np.ndarray | None = None, row_perm: np.ndarray | None = None):
    """``lsa_eigs_sinvert``: the eigenpairs of ``A x = lam M x`` nearest ``sigma`` from ONE library call on uploaded matrices
    (factorisation, Krylov-Schur, Ritz vectors).  Returns ``(lam, X, estimates, result dict, stats dict)``."""
    n = A.shape[0]
    nout = int(ncv) if ncv > 0 else max(2 * nev, nev + 15)
    nout = min(nout, n)
    opts = lsa_op_options(ilu_levels=2, ilu_shift=0.0, ksp_rtol=float(ksp_rtol), ksp_restart=min(200, max(n, 1)), ksp_maxit=4000, pc_type=int(pc_type),
                          antishift=(_DBL * 2)(0.0, 0.0))
    lam = np.zeros(max(nout, 1), dtype=np.complex128)
    est = np.zeros(max(nout, 1), dtype=np.float64)
    X = np.empty((n, nout), dtype=np.complex128, order="F")
    res, st = lsa_ks_result(), lsa_stats()
    v = None if v0 is None else np.ascontiguousarray(v0, dtype=np.complex128)
    p = None if row_perm is None else np.ascontiguousarray(row_perm, dtype=np.int32)
    sg = (_DBL * 2)(complex(sigma).real, complex(sigma).imag)
    ctx.check(ctx._lib.lsa_eigs_sinvert(ctx.handle, A.handle, None if M is None else M.handle, sg, int(nev), int(ncv), float(tol), int(max_restarts),
                                        ctypes.byref(opts), None if v is None else _ptr(v), None if p is None else _ptr(p), nout, _ptr(lam), _ptr(X),
                                        _ptr(est), ctypes.byref(res), ctypes.byref(st)))
    k = res.nout
    return (lam[:k].copy(), np.asfortranarray(X[:, :k]), est[:k].copy(),
            {"nconv": res.nconv, "restarts": res.restarts, "op_applies": res.op_applies, "next_unconverged": res.next_unconverged,
             "seconds_expand": res.seconds_expand, "seconds_dense": res.seconds_dense, "seconds_restart": res.seconds_restart},
            {name: getattr(st, name) for name, _ in lsa_stats._fields_})


def read_matrix_market(path) -> "scipy.sparse.csr_matrix":  # noqa: F821
    """Parse a MatrixMarket coordinate file with the native reader (host only, no GPU needed)."""
    import scipy.sparse as sp

    lib = load_library()
    h = ctypes.c_void_p()
    nr, nc, nnz, cx = _I32(0), _I32(0), _I64(0), ctypes.c_int(0)
    rc = lib.lsa_mm_open(str(path).encode(), ctypes.byref(h), ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nnz), ctypes.byref(cx))
    try:
        if rc != 0:
            raise ValueError(f"{path}: {lib.lsa_mm_error(h).decode(errors='replace')}")
        rp = np.empty(nr.value + 1, dtype=np.int32)
        ci = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value, dtype=np.complex128 if cx.value else np.float64)
        lib.lsa_mm_read_csr(h, _ptr(rp), _ptr(ci), _ptr(val))
    finally:
        lib.lsa_mm_close(h)
    return sp.csr_matrix((val, ci, rp), shape=(nr.value, nc.value))


def eig_residuals(ctx: Context, A: CsrMatrix, M: CsrMatrix | None, lam: np.ndarray, X: np.ndarray) -> np.ndarray:
    """Relative residuals of Solver/eigen2.py:48-56, evaluated on the device."""
    lam = np.ascontiguousarray(lam, dtype=np.complex128)
    X = np.asfortranarray(X, dtype=np.complex128)
    res = np.empty(lam.shape[0], dtype=np.float64)
    ctx.check(ctx._lib.lsa_eig_residuals(ctx.handle, A.handle, M.handle if M is not None else None, lam.shape[0], _ptr(lam), _ptr(X), _ptr(res)))
    return res
