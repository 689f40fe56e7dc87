"""The C-ABI shared library loads without a GPU and exports exactly what include/basisu_hip.h declares."""
import ctypes
import os
import re

import pytest

from basisu_rs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "basisu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from basisu_rs_amd import build

        build.build_hip()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a gfx950 device context creation must fail (BU_ERR_NO_DEVICE); there is no CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from basisu_rs_amd import BasisuError, Context

    with pytest.raises(BasisuError) as e:
        Context(0)
    assert e.value.status in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP)


def test_pure_host_helpers_of_the_abi(oracle):
    """entry points that need no device: block sizes, status strings, status-word decode, selector build"""
    import numpy as np

    lib = _lib.load()
    assert [lib.bu_target_block_bytes(t) for t in range(6)] == [16, 16, 8, 16, 64, 0]
    assert lib.bu_status_string(1) == b"invalid mode index"  # uastc.rs:336
    assert lib.bu_status_string(2) == b"block pattern is not valid"  # uastc.rs:364
    assert lib.bu_status_string(3) == b"data length is not divisible by UASTC block size (16)"  # uastc.rs:56
    bad = ctypes.c_uint64(0)
    assert lib.bu_status_word_decode(_lib.STATUS_WORD_CLEAR, ctypes.byref(bad)) == 0
    assert lib.bu_status_word_decode((1234 << 8) | 2, ctypes.byref(bad)) == 2 and bad.value == 1234
    from basisu_rs_amd import etc1s_selector_from_rows

    rows = np.random.default_rng(0).integers(0, 256, size=(500, 4), dtype=np.uint8)
    assert (etc1s_selector_from_rows(rows) == oracle.selectors_from_rows(rows)).all()
