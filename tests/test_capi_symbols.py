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


def test_rust_facade_declarations_match_the_header():
    """rust/src/ffi.rs (source only, no toolchain here): every extern fn it declares must be a symbol of the header with the
    same number of parameters -- the cheapest check that the uncompiled binding has not drifted from the C ABI"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "basisu_hip.h")).read(), flags=re.S)
    c_arity = {}
    for m in re.finditer(r"\b(bu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        c_arity[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    rs = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    rs = re.sub(r"//.*", "", rs)
    seen = 0
    for m in re.finditer(r"pub fn (bu_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", rs, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if not args else args.count(":")
        assert name in c_arity, name
        assert n == c_arity[name], (name, n, c_arity[name])
        seen += 1
    assert seen >= 35
    # the eleven public functions of the reference crate (lib.rs:20-53) exist in the facade
    lib_rs = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    for fn in ("read_to_rgba", "read_to_etc1", "read_to_etc2", "read_to_uastc", "read_to_astc", "read_to_bc7", "unpack_uastc_block_to_rgba",
               "transcode_uastc_block_to_astc", "transcode_uastc_block_to_bc7", "transcode_uastc_block_to_etc1", "transcode_uastc_block_to_etc2"):
        assert re.search(r"pub fn %s\(" % fn, lib_rs), fn
