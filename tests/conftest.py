"""pytest configuration: registers the ``gpu`` marker and puts the product package on sys.path."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "lsa-fw_amd"), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One device context for the whole GPU session; fails loudly (no CPU fallback) if there is no GPU."""
    import lsa_hip

    ctx = lsa_hip.Context(0)
    yield ctx
    ctx.close()
