"""Shared fixtures.  Tests marked ``gpu`` need a real MI355X (run with ``-m gpu`` on the GPU
box); everything else runs on CPU (``-m "not gpu"``).  ``oracle/`` is imported here and in
tests only — it is the checker, never the thing under test on the product side."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "tests", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def built():
    """Make sure libtbk_hip.so and the oracle are built (cross-compiles without a GPU)."""
    import __graft_entry__ as entry

    entry.build()
    return True


@pytest.fixture(scope="session")
def orc(built):
    import oracle

    return oracle.load()


@pytest.fixture(scope="session")
def gpu(built):
    from trio_binning_amd import _lib

    if _lib.device_count() < 1:
        pytest.fail("gpu-marked test running without a visible HIP device")
    return _lib
