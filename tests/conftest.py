import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TARGETS = {"astc": (0, 16), "bc7": (1, 16), "etc1": (2, 8), "etc2": (3, 16), "rgba": (4, 64)}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Every GPU test gets a deadline (pytest-timeout, method "thread": the whole run is ended -- a test stuck inside a C call never
    returns to the interpreter, so a signal handler would not run).  A deadlock in the library then fails the run in minutes
    instead of holding the GPU box until the job limit."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(900, method="thread"))


def pytest_sessionstart(session):
    """a fresh checkout has no built artefacts (they are git-ignored): build the C-ABI library once, up front"""
    from basisu_rs_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        from basisu_rs_amd import build

        build.build_hip()


def _make(path, target):
    if not os.path.exists(os.path.join(path, target)):
        subprocess.run(["make", "-C", path, target], check=True, capture_output=True)


@pytest.fixture(scope="session")
def golden():
    from basisu_rs_amd import synth

    return synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))


from oracle.pyoracle import Oracle  # noqa: E402  (the checker)


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


class Emul:
    """host build of the device headers (tests/host_emul) -- test-only"""

    def __init__(self):
        _make(os.path.join(ROOT, "tests", "host_emul"), "libbu_emul.so")
        self.lib = ctypes.CDLL(os.path.join(ROOT, "tests", "host_emul", "libbu_emul.so"))
        self.lib.bu_emul_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.bu_emul_batch.restype = None

    def batch(self, target, blocks):
        t, obs = TARGETS[target]
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
        out = np.zeros((blocks.shape[0], obs), dtype=np.uint8)
        st = np.zeros(blocks.shape[0], dtype=np.uint8)
        self.lib.bu_emul_batch(t, blocks.ctypes.data, blocks.shape[0], out.ctypes.data, st.ctypes.data)
        return out, st


@pytest.fixture(scope="session")
def emul():
    return Emul()


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI (gpu tests only)"""
    from basisu_rs_amd import Context

    c = Context(0)
    yield c
    c.close()
