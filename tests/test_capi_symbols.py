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


# ---- type-level check of the uncompiled Rust binding -------------------------------------------------------------------------
_C_SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "int": "c_int", "float": "f32",
              "char": "c_char", "void": "c_void", "bu_status": "c_int", "bu_target": "c_int", "bu_read_target": "c_int", "bu_launch_policy": "c_int",
              "bu_context": "bu_context", "bu_comm": "bu_comm", "bu_basis_header": "bu_basis_header", "bu_slice_desc": "bu_slice_desc",
              "bu_image": "bu_image"}


def _c_type_to_rust(decl):
    """one C parameter / return type (name stripped) -> the Rust FFI type it must be bound as"""
    d = decl.strip()
    m = re.match(r"^(const\s+)?(\w+)\s*((?:\*\s*(?:const\s*)?)*)$", d)
    assert m, "cannot parse C type %r" % decl
    const_base, base, stars = bool(m.group(1)), m.group(2), m.group(3)
    assert base in _C_SCALARS, "unknown C type %r" % base
    t = _C_SCALARS[base]
    # pointers, innermost first: `const T*` -> *const T; `T* const*` -> *const *mut T
    levels = re.findall(r"\*\s*(const)?", stars)
    for k, q in enumerate(levels):
        is_const = const_base if k == 0 else bool(levels[k - 1])
        t = ("*const " if is_const else "*mut ") + t
    return t


def _split_c_params(args):
    out = []
    for a in [x.strip() for x in args.split(",")] if args.strip() not in ("", "void") else []:
        arr = re.search(r"\[[^\]]*\]\s*$", a)  # `uint8_t id[BU_COMM_ID_BYTES]` decays to a pointer
        a = re.sub(r"\[[^\]]*\]\s*$", "", a)
        m = re.match(r"^(.*?)(\w+)$", a.strip())
        typ = m.group(1).strip()
        if arr:
            typ += "*"
        out.append(typ)
    return out


def test_rust_facade_parameter_types_match_the_header():
    """position by position: every parameter and return type of every extern fn in rust/src/ffi.rs is the Rust spelling of
    the C type in include/basisu_hip.h (a `*const u32` bound to a `uint64_t*` passes an arity check; not this one), and the
    three #[repr(C)] structs have the header's fields in the header's order with the header's widths."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "basisu_hip.h")).read(), flags=re.S)
    c_fns = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(bu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        ret = m.group(1).strip().split("\n")[-1].strip()
        ret = re.sub(r"^(extern\s+)?", "", ret)
        c_fns[m.group(2)] = (ret, _split_c_params(" ".join(m.group(3).split())))
    rs = re.sub(r"//.*", "", open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read())
    seen = 0
    for m in re.finditer(r"pub fn (bu_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", rs, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), (m.group(3) or "").strip()
        c_ret, c_params = c_fns[name]
        r_params = [a.split(":", 1)[1].strip() for a in args.split(",")] if args else []
        want = [_c_type_to_rust(p) for p in c_params]
        assert r_params == want, (name, r_params, want)
        want_ret = "" if c_ret == "void" else _c_type_to_rust(c_ret)
        assert ret == want_ret, (name, ret, want_ret)
        seen += 1
    assert seen >= 36
    # structs: field names, order and widths
    for sname in ("bu_basis_header", "bu_slice_desc", "bu_image"):
        end = re.search(r"\}\s*%s\s*;" % sname, hdr)
        assert end, sname
        body = hdr[hdr.rindex("{", 0, end.start()) + 1: end.start()]
        c_fields = []
        for line in body.split(";"):
            line = " ".join(line.split())
            if not line:
                continue
            typ, names = line.split(" ", 1)
            for n in names.split(","):
                c_fields.append((n.strip(), _C_SCALARS[typ]))
        rm = re.search(r"#\[repr\(C\)\][^{]*pub struct %s\s*\{(.*?)\}" % sname, rs, flags=re.S)
        assert rm, sname
        r_fields = [(f.split(":")[0].replace("pub", "").strip(), f.split(":")[1].strip()) for f in rm.group(1).split(",") if ":" in f]
        assert r_fields == c_fields, (sname, r_fields, c_fields)
