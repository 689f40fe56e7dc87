"""Python driver over the C ABI, shaped like the reference crate's API for the hot path so that the
parity tests read like the reference's own tests (tests/transcode_uastc_block.rs).

Reference items mirrored (file:line in /root/reference):
  lib.rs:29-53      unpack_uastc_block_to_rgba, transcode_uastc_block_to_{astc,bc7,etc1,etc2}
  uastc.rs:41-47    TargetTextureFormat
  uastc.rs:77-146   Decoder::{read_to_uastc, decode_to_rgba, transcode}
  lib.rs:26-27      Error = String  ->  BasisuError(message)

The host logic itself (argument checks, staging, launch, status decode) is C++ inside
libbasisu_hip.so; this module only marshals buffers.  All work runs on the GPU -- there is no CPU
path here, and importing this module without the built library raises.
"""
import ctypes
import enum

import numpy as np

from . import _lib


class BasisuError(Exception):
    """The reference's `Err(String)`; str(e) is the reference's message for hot-path errors."""

    def __init__(self, status, first_bad_block=None, detail=None):
        self.status = status
        self.first_bad_block = first_bad_block
        msg = _lib.load().bu_status_string(status).decode()
        if detail:
            msg += " (" + detail + ")"
        super().__init__(msg)


class TargetTextureFormat(enum.IntEnum):  # uastc.rs:41-47
    Astc = _lib.ASTC
    Bc7 = _lib.BC7
    Etc1 = _lib.ETC1
    Etc2 = _lib.ETC2


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    a = np.ascontiguousarray(a.reshape(-1).view(np.uint8))
    return a


class Context:
    """Owns one bu_context (device tables, stream, staging buffers)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        st = self._lib.bu_context_create(int(device), ctypes.byref(h))
        if st != _lib.OK:
            raise BasisuError(st)
        self._h = h
        self.device = int(device)
        self._pinned = {}

    def close(self):
        if getattr(self, "_h", None):
            for addr in list(getattr(self, "_pinned", {}).values()):
                self._lib.bu_host_free(self._h, addr)
            self._pinned = {}
            self._lib.bu_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _check(self, st, bad=None):
        if st == _lib.OK:
            return
        detail = None
        if st == _lib.ERR_HIP:
            detail = self._lib.bu_last_error(self._h).decode()
        raise BasisuError(st, bad.value if bad is not None and st in (_lib.ERR_INVALID_MODE, _lib.ERR_INVALID_PATTERN, _lib.ERR_INDEX_RANGE) else None, detail)

    # ---- host-pointer slice API -----------------------------------------------------------------
    def host_alloc(self, nbytes):
        """A page-locked uint8 buffer (bu_host_alloc).  Slices handed to transcode()/decode_to_rgba() in such
        buffers (input and `out=`) stream over PCIe with upload, kernels and download overlapped.  The memory
        lives until host_free(buffer) or close()."""
        p = ctypes.c_void_p(0)
        self._check(self._lib.bu_host_alloc(self._h, int(nbytes), ctypes.byref(p)))
        if not p.value:
            return np.empty(0, dtype=np.uint8)
        arr = np.ctypeslib.as_array((ctypes.c_uint8 * int(nbytes)).from_address(p.value))
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        addr = self._pinned.pop(arr.ctypes.data, None)
        if addr is not None:
            self._check(self._lib.bu_host_free(self._h, addr))

    def transcode(self, fmt, data, out=None):
        """uastc::Decoder::transcode (uastc.rs:112-121): bytes in -> bytes out.  `out` (optional) is a caller
        buffer of at least n_blocks * block_bytes, e.g. from host_alloc()."""
        a = _as_u8(data)
        n = a.size // 16
        if out is None:
            out = np.empty(max(n, 1) * _lib.BLOCK_BYTES[int(fmt)], dtype=np.uint8)
        bad = ctypes.c_uint64(0)
        st = self._lib.bu_uastc_transcode(self._h, int(fmt), a.ctypes.data, a.size, out.ctypes.data, out.size, ctypes.byref(bad))
        self._check(st, bad)
        return out[: n * _lib.BLOCK_BYTES[int(fmt)]]

    def decode_to_rgba(self, data, blocks_per_row, out=None):
        """uastc::Decoder::decode_to_rgba (uastc.rs:89-110): row-major RGBA8 bytes."""
        a = _as_u8(data)
        n = a.size // 16
        if out is None:
            out = np.empty(max(n, 1) * 64, dtype=np.uint8)
        bad = ctypes.c_uint64(0)
        st = self._lib.bu_uastc_decode_to_rgba(self._h, a.ctypes.data, a.size, int(blocks_per_row), out.ctypes.data, out.size, ctypes.byref(bad))
        self._check(st, bad)
        return out[: n * 64]

    def set_launch_policy(self, shared):
        """bu_context_set_launch_policy.  "auto" / None (the default of a new context) = BU_LAUNCH_AUTO: decided per call -- shared when the launch goes
        to one of the context's own streams and another of them has work in flight, exclusive otherwise; False = BU_LAUNCH_EXCLUSIVE: a large launch
        fills the chip by itself; True = BU_LAUNCH_SHARED: it keeps at most half of every CU so that launches queued on different streams run side by
        side (include/basisu_hip.h)."""
        code = 2 if shared is None or shared == "auto" else (1 if shared else 0)
        self._check(self._lib.bu_context_set_launch_policy(self._h, code))

    def stream(self, index):
        """the context's own stream `index` (0..7) as a raw hipStream_t value.  The library checks when it creates them (in groups of four) that each
        has a hardware queue of its own and re-creates the group with CU masks if the runtime's queue pool (GPU_MAX_HW_QUEUES) is too small;
        query_in_flight() tells what the context got"""
        p = ctypes.c_void_p(0)
        self._check(self._lib.bu_context_stream(self._h, int(index), ctypes.byref(p)))
        return p.value

    def probe_streams(self, n_streams=4):
        """the largest number of the context's streams 0..n_streams-1 sharing one hardware queue in this process, measured now (1: each has its own)"""
        k = ctypes.c_int(0)
        self._check(self._lib.bu_context_probe_streams(self._h, int(n_streams), ctypes.byref(k)))
        return k.value

    def query_in_flight(self, n_streams=4):
        """bu_context_query_in_flight: (effective_streams, "pool" | "cu_mask") -- how many launches a pipeline over the context's streams
        0..n_streams-1 really keeps in flight in this process, and which kind of stream the context created"""
        eff, mode = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.bu_context_query_in_flight(self._h, int(n_streams), ctypes.byref(eff), ctypes.byref(mode)))
        return eff.value, ("cu_mask" if mode.value == 1 else "pool")

    def block_api_on_device(self, enable):
        """Per-block API: False (default) = the library's own block code on the calling thread, True = a one-block kernel launch."""
        self._check(self._lib.bu_block_api_on_device(self._h, 1 if enable else 0))

    def _block(self, fn, block, out_bytes):
        a = _as_u8(block)
        if a.size != 16:
            raise ValueError("a UASTC block is 16 bytes")
        out = np.empty(out_bytes, dtype=np.uint8)
        st = fn(self._h, a.ctypes.data, out.ctypes.data)
        self._check(st)
        return out

    # ---- ETC1S back-end -------------------------------------------------------------------------
    def etc1s_transcode_to_etc1(self, idx, endpoints, selectors):
        """basis_lz::Decoder::transcode_to_etc1 back-end (basis_lz/mod.rs:153-186)."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        endpoints = np.ascontiguousarray(endpoints, dtype=np.uint32)
        selectors = np.ascontiguousarray(selectors, dtype=np.uint8).reshape(-1, 8)
        out = np.empty(max(idx.size, 1) * 8, dtype=np.uint8)
        bad = ctypes.c_uint64(0)
        st = self._lib.bu_etc1s_transcode_etc1(self._h, idx.ctypes.data, idx.size, endpoints.ctypes.data, endpoints.size,
                                               selectors.ctypes.data, selectors.shape[0], out.ctypes.data, out.size, ctypes.byref(bad))
        self._check(st, bad)
        return out[: idx.size * 8]

    def etc1s_decode_to_rgba(self, idx, alpha_idx, nbx, nby, endpoints, selectors):
        """basis_lz::Decoder::decode_to_rgba back-end (basis_lz/mod.rs:97-151)."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        aptr = None
        if alpha_idx is not None:
            alpha_idx = np.ascontiguousarray(alpha_idx, dtype=np.uint32)
            aptr = alpha_idx.ctypes.data
        endpoints = np.ascontiguousarray(endpoints, dtype=np.uint32)
        selectors = np.ascontiguousarray(selectors, dtype=np.uint8).reshape(-1, 8)
        n = int(nbx) * int(nby)
        if idx.size != n:
            raise ValueError("idx must hold nbx*nby entries")
        out = np.empty(max(n, 1) * 64, dtype=np.uint8)
        bad = ctypes.c_uint64(0)
        st = self._lib.bu_etc1s_decode_rgba(self._h, idx.ctypes.data, aptr, int(nbx), int(nby), endpoints.ctypes.data, endpoints.size,
                                            selectors.ctypes.data, selectors.shape[0], out.ctypes.data, out.size, ctypes.byref(bad))
        self._check(st, bad)
        return out[: n * 64]

    # ---- device-pointer API (torch tensors or raw pointers) ---------------------------------------
    def transcode_device(self, fmt, d_in, n_blocks, d_out, blocks_per_row=0, block_index_base=0, d_status=None, stream=None):
        st = self._lib.bu_uastc_transcode_device(self._h, int(fmt), _ptr(d_in), int(n_blocks), _ptr(d_out), int(blocks_per_row),
                                                 int(block_index_base), _ptr(d_status), _stream_ptr(stream))
        self._check(st)

    def transcode_device_sync(self, fmt, d_in, n_blocks, d_out, blocks_per_row=0, block_index_base=0):
        """bu_uastc_transcode_device_sync: a contiguous device-resident range, blocking; one exclusive launch, tile tickets on long walks.
        Returns the status word (STATUS_WORD_CLEAR or lowest failing block << 8 | status); never raises for a block error"""
        word = ctypes.c_uint64(0)
        st = self._lib.bu_uastc_transcode_device_sync(self._h, int(fmt), _ptr(d_in), int(n_blocks), _ptr(d_out), int(blocks_per_row), int(block_index_base),
                                                      ctypes.byref(word))
        self._check(st)
        return word.value

    def transcode_batch_in_flight(self, fmt, d_ins, n_blocks, d_outs, blocks_per_row=0, index_base=None, d_status=None, n_streams=4):
        """bu_uastc_transcode_batch_in_flight: independent slices as a pipeline of launches on the context's own streams; only enqueues --
        synchronize() waits.  d_ins / d_outs: device tensors or raw pointers, n_blocks: blocks per slice"""
        n = len(d_ins)
        VP, SZ = ctypes.c_void_p * n, ctypes.c_size_t * n
        ib = (ctypes.c_uint64 * n)(*[int(x) for x in index_base]) if index_base is not None else None
        st = self._lib.bu_uastc_transcode_batch_in_flight(self._h, int(fmt), n, VP(*[_ptr(x) for x in d_ins]), SZ(*[int(x) for x in n_blocks]),
                                                          VP(*[_ptr(x) for x in d_outs]), int(blocks_per_row), ib, _ptr(d_status), int(n_streams))
        self._check(st)

    def synchronize(self):
        """bu_context_synchronize: everything enqueued on the context's own streams has completed"""
        self._check(self._lib.bu_context_synchronize(self._h))

    def status_word_reset(self, d_status, stream=None):
        self._check(self._lib.bu_status_word_reset(self._h, _ptr(d_status), _stream_ptr(stream)))

    def status_word_check(self, word):
        bad = ctypes.c_uint64(0)
        st = self._lib.bu_status_word_decode(int(word) & 0xFFFFFFFFFFFFFFFF, ctypes.byref(bad))
        self._check(st, bad)


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(int(x))


def _stream_ptr(stream):
    if stream is None:
        try:
            import torch

            if torch.cuda.is_available():
                return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return None
    if hasattr(stream, "cuda_stream"):
        return ctypes.c_void_p(stream.cuda_stream)
    return ctypes.c_void_p(int(stream))


def etc1s_selector_from_rows(rows):
    """etc::Selector::set_selector for the 16 texels of one codebook entry (etc.rs:363-393)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 4)
    out = np.empty((rows.shape[0], 8), dtype=np.uint8)
    lib = _lib.load()
    for i in range(rows.shape[0]):
        lib.bu_etc1s_selector_from_rows(rows[i].ctypes.data, out[i].ctypes.data)
    return out


# ---- whole-file API: basis.rs / lib.rs:20-22 -----------------------------------------------------------
class Image:
    """lib.rs:63-68 Image<u8>: w, h = original pixel size, stride in bytes, data = bytes"""

    def __init__(self, w, h, stride, data):
        self.w, self.h, self.stride, self.data = w, h, stride, data

    def __repr__(self):
        return "Image(w=%d, h=%d, stride=%d, %d bytes)" % (self.w, self.h, self.stride, len(self.data))


def _check_host(st):
    if st != _lib.OK:
        raise BasisuError(st)


def read_header(buf):
    """basis::read_header (basis.rs:307-336) -> _lib.BasisHeader"""
    lib = _lib.load()
    a = _as_u8(buf)
    h = _lib.BasisHeader()
    _check_host(lib.bu_basis_read_header(a.ctypes.data, a.size, ctypes.byref(h)))
    return h


def read_slice_descs(buf, header=None):
    """basis::read_slice_descs (basis.rs:343-362)"""
    lib = _lib.load()
    a = _as_u8(buf)
    h = header or read_header(buf)
    n = ctypes.c_size_t(0)
    _check_host(lib.bu_basis_read_slice_descs(a.ctypes.data, a.size, ctypes.byref(h), None, 0, ctypes.byref(n)))
    arr = (_lib.SliceDesc * max(n.value, 1))()
    _check_host(lib.bu_basis_read_slice_descs(a.ctypes.data, a.size, ctypes.byref(h), arr, n.value, ctypes.byref(n)))
    return list(arr[: n.value])


def crc16(data, crc=0):
    a = _as_u8(data)
    return _lib.load().bu_basis_crc16(a.ctypes.data, a.size, crc)


def read_query(target, buf):
    """(number of images, total output bytes) a read_to_* call on this file produces (bu_read_query; host only)"""
    lib = _lib.load()
    a = _as_u8(buf)
    n, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _check_host(lib.bu_read_query(int(target), a.ctypes.data, a.size, ctypes.byref(n), ctypes.byref(nb)))
    return n.value, nb.value


def _read_to(target, buf, ctx=None, out=None):
    """`out`: optional caller buffer of at least read_query()[1] bytes; one from Context.host_alloc() is written by the
    kernels directly over PCIe (no device output buffer, no download)."""
    lib = _lib.load()
    a = _as_u8(buf)
    n, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if out is None:
        _check_host(lib.bu_read_query(target, a.ctypes.data, a.size, ctypes.byref(n), ctypes.byref(nb)))
        out = np.empty(max(nb.value, 1), dtype=np.uint8)
        max_images = n.value
    else:  # the caller sized the buffer (read_query): no second pass over the payload CRC, one image per slice at most
        hq = _lib.BasisHeader()
        _check_host(lib.bu_basis_read_header(a.ctypes.data, a.size, ctypes.byref(hq)))
        max_images = hq.total_slices
    c = ctx or default_context()
    imgs = (_lib.ImageDesc * max(max_images, 1))()
    h = _lib.BasisHeader()
    st = lib.bu_read_to(c.handle, target, a.ctypes.data, a.size, ctypes.byref(h), imgs, max_images, ctypes.byref(n), out.ctypes.data, out.size)
    c._check(st)
    return h, [Image(im.w, im.h, im.stride, out[im.offset:im.offset + im.size]) for im in imgs[: n.value]]


def read_to_rgba(buf, ctx=None, out=None):  # basis.rs:8-90 -> (Header, Vec<Image<u8>>)
    return _read_to(_lib.READ_RGBA, buf, ctx, out)


def read_to_etc1(buf, ctx=None, out=None):  # basis.rs:92-143
    return _read_to(_lib.READ_ETC1, buf, ctx, out)[1]


def read_to_etc2(buf, ctx=None, out=None):  # basis.rs:145-173
    return _read_to(_lib.READ_ETC2, buf, ctx, out)[1]


def read_to_uastc(buf, ctx=None, out=None):  # basis.rs:175-202
    return _read_to(_lib.READ_UASTC, buf, ctx, out)[1]


def read_to_astc(buf, ctx=None, out=None):  # basis.rs:204-231
    return _read_to(_lib.READ_ASTC, buf, ctx, out)[1]


def read_to_bc7(buf, ctx=None, out=None):  # basis.rs:233-260
    return _read_to(_lib.READ_BC7, buf, ctx, out)[1]


def basislz_decode(buf, slice_index=None):
    """host-only BasisLZ decode of an ETC1S file -> (endpoints u32[n], selectors u8[n,8], idx u32[blocks] or None)"""
    lib = _lib.load()
    a = _as_u8(buf)
    h = read_header(buf)
    n = h.total_selectors  # reference quirk: both codebooks are sized by total_selectors (basis.rs:289-291)
    ep = np.zeros(max(n, 1), dtype=np.uint32)
    sel = np.zeros((max(n, 1), 8), dtype=np.uint8)
    idx = None
    iptr = None
    if slice_index is not None:
        sd = read_slice_descs(buf, h)[slice_index]
        idx = np.zeros(max(sd.num_blocks_x * sd.num_blocks_y, 1), dtype=np.uint32)
        iptr = idx.ctypes.data
    _check_host(lib.bu_basislz_decode(a.ctypes.data, a.size, slice_index or 0, ep.ctypes.data, sel.ctypes.data, iptr))
    if idx is not None:
        idx = idx[: sd.num_blocks_x * sd.num_blocks_y]
    return ep[:n], sel[:n], idx


def write_uastc_file(slices, header_flags=0, tex_type=0):
    """slices: list of dict(data=bytes-like, orig_w, orig_h, nbx, nby, image_index=0, level=0, flags=0) -> bytes of a .basis file"""
    lib = _lib.load()
    n = len(slices)
    descs = (_lib.SliceDesc * max(n, 1))()
    keep = [_as_u8(s["data"]) for s in slices]
    ptrs = (ctypes.c_void_p * max(n, 1))(*[k.ctypes.data for k in keep])
    sizes = (ctypes.c_size_t * max(n, 1))(*[k.size for k in keep])
    for i, s in enumerate(slices):
        d = descs[i]
        d.image_index, d.level_index, d.flags = s.get("image_index", 0), s.get("level", 0), s.get("flags", 0)
        d.orig_width, d.orig_height, d.num_blocks_x, d.num_blocks_y = s["orig_w"], s["orig_h"], s["nbx"], s["nby"]
    ln = ctypes.c_size_t(0)
    _check_host(lib.bu_basis_write_uastc(descs, ptrs, sizes, n, header_flags, tex_type, None, 0, ctypes.byref(ln)))
    out = np.empty(ln.value, dtype=np.uint8)
    _check_host(lib.bu_basis_write_uastc(descs, ptrs, sizes, n, header_flags, tex_type, out.ctypes.data, out.size, ctypes.byref(ln)))
    return out.tobytes()


# ---- module-level mirror of the reference's free functions / Decoder --------------------------------
_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Decoder:
    """uastc::Decoder (uastc.rs:77-146)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def read_to_uastc(self, data):  # uastc.rs:85-87
        return bytes(data)

    def decode_to_rgba(self, data, blocks_per_row):  # uastc.rs:89-110
        return self.ctx.decode_to_rgba(data, blocks_per_row)

    def transcode(self, fmt, data):  # uastc.rs:112-121
        return self.ctx.transcode(TargetTextureFormat(fmt), data)


def unpack_uastc_block_to_rgba(data, ctx=None):  # lib.rs:29-31 -> [u32; 16]
    c = ctx or default_context()
    return c._block(c._lib.bu_unpack_uastc_block_to_rgba, data, 64).view("<u4").copy()


def transcode_uastc_block_to_astc(data, ctx=None):  # lib.rs:33-37
    c = ctx or default_context()
    return c._block(c._lib.bu_transcode_uastc_block_to_astc, data, 16)


def transcode_uastc_block_to_bc7(data, ctx=None):  # lib.rs:39-41
    c = ctx or default_context()
    return c._block(c._lib.bu_transcode_uastc_block_to_bc7, data, 16)


def transcode_uastc_block_to_etc1(data, ctx=None):  # lib.rs:43-47
    c = ctx or default_context()
    return c._block(c._lib.bu_transcode_uastc_block_to_etc1, data, 8)


def transcode_uastc_block_to_etc2(data, ctx=None):  # lib.rs:49-53
    c = ctx or default_context()
    return c._block(c._lib.bu_transcode_uastc_block_to_etc2, data, 16)
