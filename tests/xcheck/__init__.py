"""ctypes binding of the cross-check library ``liblsa_xcheck.so`` (tests only): the exact block-tridiagonal LU of round 1
(``tests/xcheck/blocklu.hip``, ``lsa_blu_*``), an independent direct solver on the device.  Never imported by the product."""

from __future__ import annotations

import ctypes
from pathlib import Path

import lsa_hip
from lsa_hip import _DBL, _I32, _I64, _P, _PP, CsrMatrix, DeviceVector

LIB_PATH = Path(__file__).resolve().parent / "liblsa_xcheck.so"
SIGNATURES = {
    "lsa_blu_create": (ctypes.c_int, [_P, _P, _I32, _PP]),
    "lsa_blu_destroy": (None, [_P]),
    "lsa_blu_solve": (ctypes.c_int, [_P, _P, _P, _P]),
    "lsa_blu_solve_time": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.POINTER(_DBL)]),
    "lsa_blu_info": (ctypes.c_int, [_P, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_DBL)]),
    "lsa_blu_apply_bytes": (ctypes.c_int, [_P, ctypes.POINTER(_I64)]),
    "lsa_blu_apply_launches": (ctypes.c_int, [_P, ctypes.POINTER(_I32)]),
}
_lib = None


def load_library():
    """The product library first (its symbols are what this one links against), then the cross-check library."""
    global _lib
    if _lib is None:
        lsa_hip.load_library()
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C tests/xcheck` (or __graft_entry__.build())")
        lib = ctypes.CDLL(str(LIB_PATH), mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class BlockLu:
    """Exact block-tridiagonal LU of a banded CSR matrix, resident on the device (``lsa_blu_*``)."""

    def __init__(self, ctx, C: CsrMatrix, block_size: int = 0):
        self.ctx, self._C, self._lib = ctx, C, load_library()
        h = ctypes.c_void_p()
        ctx.check(self._lib.lsa_blu_create(ctx.handle, C.handle, int(block_size), ctypes.byref(h)))
        self.handle = h

    def info(self) -> dict:
        B, nb, bw, sec = _I32(0), _I32(0), _I32(0), _DBL(0.0)
        self._lib.lsa_blu_info(self.handle, ctypes.byref(B), ctypes.byref(nb), ctypes.byref(bw), ctypes.byref(sec))
        nbytes, nl = _I64(0), _I32(0)
        self._lib.lsa_blu_apply_bytes(self.handle, ctypes.byref(nbytes))
        self._lib.lsa_blu_apply_launches(self.handle, ctypes.byref(nl))
        return {"block_size": B.value, "nblocks": nb.value, "bandwidth": bw.value, "seconds": sec.value, "apply_bytes": nbytes.value,
                "apply_launches": nl.value}

    def solve(self, b: DeviceVector, x: DeviceVector) -> None:
        self.ctx.check(self._lib.lsa_blu_solve(self.ctx.handle, self.handle, b.handle, x.handle))

    def time_solve(self, b: DeviceVector, x: DeviceVector, iters: int = 20) -> float:
        ms = _DBL(0.0)
        self.ctx.check(self._lib.lsa_blu_solve_time(self.ctx.handle, self.handle, b.handle, x.handle, int(iters), ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self._lib.lsa_blu_destroy(self.handle)
            self.handle = None
