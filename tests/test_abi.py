"""CPU suite, part 2: the C-ABI library loads and exports exactly what ``include/lsa_hip.h`` declares.

No compute call is made (there is no GPU here); creating a context must fail loudly rather than fall back.
"""

import ctypes
import re
from pathlib import Path

import pytest

import lsa_hip

HEADER = Path(__file__).resolve().parents[1] / "include" / "lsa_hip.h"


def declared_functions() -> set[str]:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return set(re.findall(r"\b(lsa_[a-z0-9_]+)\s*\(", text))


def test_header_library_and_binding_agree():
    lib = lsa_hip.load_library()
    declared = declared_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/lsa_hip.h but not exported by liblsa_hip.so"
    assert declared == set(lsa_hip.SIGNATURES), (
        f"binding out of step with the header: only in header {sorted(declared - set(lsa_hip.SIGNATURES))}, "
        f"only in binding {sorted(set(lsa_hip.SIGNATURES) - declared)}"
    )


def test_library_is_in_tree_and_built_for_gfx950():
    assert lsa_hip.LIB_PATH.exists() and "lsa-fw_amd" in str(lsa_hip.LIB_PATH)
    blob = lsa_hip.LIB_PATH.read_bytes()
    assert b"gfx950" in blob  # the offload bundle carries the target name


def test_struct_layouts_match_header():
    assert ctypes.sizeof(lsa_hip.lsa_stats) == 4 * 8 + 4 * 8 + 6 * 4
    # int32, (pad), double, double, int32, int32, int32, (pad), double[2] -> 56 bytes with natural alignment
    # (csrc/solver.hip static_asserts the same number on the C side)
    assert ctypes.sizeof(lsa_hip.lsa_op_options) == 56


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lsa_hip.Context(0)
    from Solver.eigen import EigenSolver, EigensolverConfig
    import numpy as np

    es = EigenSolver(np.diag([1.0, 2.0, 3.0]), None, EigensolverConfig(num_eig=1))
    with pytest.raises(RuntimeError):
        es.solve()


def test_cross_check_library_exports_its_header():
    """tests/xcheck/liblsa_xcheck.so (test-only: round 1's banded block LU) loads beside the product library and exports what its
    own header declares; none of it is part of include/lsa_hip.h any more."""
    import helpers  # noqa: F401
    import xcheck

    lib = xcheck.load_library()
    text = re.sub(r"/\*.*?\*/", "", (Path(__file__).resolve().parent / "xcheck" / "lsa_xcheck.h").read_text(), flags=re.S)
    declared = set(re.findall(r"\b(lsa_blu_[a-z0-9_]+)\s*\(", text))
    assert declared == set(xcheck.SIGNATURES) and len(declared) == 7
    for name in declared:
        assert hasattr(lib, name)
    assert not any(name.startswith("lsa_blu") for name in declared_functions())
