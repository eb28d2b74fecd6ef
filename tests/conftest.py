import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/mik_oracle.c via ctypes) -- the checker, never the thing under test."""
    o = graft.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    """The product package (iterativesolvers.jl_amd); loading it needs libmik.so built."""
    if not os.path.exists(os.path.join(graft.PKG_DIR, "libmik.so")):
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def ctx(pkg):
    """A device context; GPU tests only.  Fails loudly (no CPU fallback) if there is no device."""
    return pkg.default_context()


@pytest.fixture(scope="session")
def dist(pkg):
    """The row-partition layer of the product package."""
    from importlib import import_module
    return import_module(pkg.__name__ + ".dist")


@pytest.fixture(autouse=True)
def _development_knobs_back_to_default(request):
    """a test that fails between setting a development knob (include/mik_dev.h) and resetting it must not leak it into the next one"""
    yield
    if "pkg" in request.fixturenames:
        L = request.getfixturevalue("pkg").lib()
        for k in range(KN.COUNT):
            L.mik_set_tuning(k, 0)


class KN:
    """the development knobs of include/mik_dev.h by name (tests/test_abi.py checks this table against the header's enum)"""
    LAYOUTS, CSR_KERNEL, SDIA_KERNEL, LONG_SEGMENT, UPLOAD, GS, TRANSPORT, NO_LOOKAHEAD, SOLVER_FORM, GS_TIMEOUT, CG_STEP, HOST_WAIT, MACHINE = range(13)
    COUNT = 13
    # bits of LAYOUTS
    CSR_ONLY, NO_SLICE_CONSTANT, NO_SLICE_OFFSETS, NO_JAGGED, JAGGED_ALWAYS, NO_XWIN, XWIN_UNUSED = 1, 2, 4, 8, 16, 32, 64
    # bits of CG_STEP
    X_IN_STEP, PCG_THREE_SWEEPS, HALO_AFTER_SWEEP, SEPARATE_ALPHA = 1, 2, 4, 8


def fromhex(lst):
    return np.array([float.fromhex(s) for s in lst], dtype=np.float64)
