"""ctypes view of the CPU oracle (oracle/libbu_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
always as the checker / the reported CPU baseline, never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TARGETS = {"astc": (0, 16), "bc7": (1, 16), "etc1": (2, 8), "etc2": (3, 16), "rgba": (4, 64)}


def _make(path, target):
    if not os.path.exists(os.path.join(path, target)):
        subprocess.run(["make", "-C", path, target], check=True, capture_output=True)


class Oracle:
    """ctypes view of oracle/libbu_oracle.so -- the CHECKER (never the thing under test)."""

    def __init__(self):
        _make(os.path.join(ROOT, "oracle"), "libbu_oracle.so")
        self.lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libbu_oracle.so"))
        L, c = self.lib, ctypes
        L.bu_oracle_batch.argtypes = [c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.bu_oracle_batch.restype = None
        L.bu_oracle_transcode.argtypes = [c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.POINTER(c.c_size_t)]
        L.bu_oracle_decode_to_rgba.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_void_p, c.POINTER(c.c_size_t)]
        L.bu_oracle_transcode_mt.argtypes = [c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.c_int]
        L.bu_oracle_etc1s_to_etc1.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p]
        L.bu_oracle_etc1s_to_etc1.restype = None
        L.bu_oracle_etc1s_to_rgba.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p]
        L.bu_oracle_etc1s_to_rgba.restype = None
        L.bu_oracle_selector_from_rows.argtypes = [c.c_void_p, c.c_void_p]
        L.bu_oracle_selector_from_rows.restype = None
        L.bu_oracle_decode_etc1_block.argtypes = [c.c_void_p, c.c_void_p]
        L.bu_oracle_decode_etc1_block.restype = None
        for n in ("bu_oracle_prove_shared_pbits", "bu_oracle_prove_unique_pbits", "bu_oracle_prove_eac_center"):
            getattr(L, n).restype = c.c_uint64
        L.bu_oracle_prove_unique_pbits.argtypes = [c.c_uint, c.c_uint, c.c_uint64, c.c_uint64]

    # ---- container / BasisLZ half (bu_oracle_basis.c) ----
    READ = {"rgba": 0, "etc1": 1, "etc2": 2, "uastc": 3, "astc": 4, "bc7": 5}

    def crc16(self, data, crc=0):
        self.lib.bu_oracle_crc16.restype = ctypes.c_uint16
        self.lib.bu_oracle_crc16.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint16]
        return self.lib.bu_oracle_crc16(bytes(data), len(data), crc)

    def header_from_bytes(self, data):
        out = (ctypes.c_uint32 * 26)()
        self.lib.bu_oracle_header_from_bytes(bytes(data), out)
        return list(out)

    def read_to(self, which, data):
        """-> (status, header[26], [(w, h, stride, bytes)...])"""
        class Img(ctypes.Structure):
            _fields_ = [("w", ctypes.c_uint32), ("h", ctypes.c_uint32), ("stride", ctypes.c_uint32), ("offset", ctypes.c_uint64), ("size", ctypes.c_uint64)]

        data = bytes(data)
        L = self.lib
        L.bu_oracle_read_to.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        hdr = (ctypes.c_uint32 * 26)()
        n, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        st = L.bu_oracle_read_to(self.READ[which], data, len(data), hdr, None, 0, ctypes.byref(n), None, 0, ctypes.byref(nb))
        if st:
            return st, list(hdr), []
        imgs = (Img * max(n.value, 1))()
        out = np.zeros(max(nb.value, 1), dtype=np.uint8)
        st = L.bu_oracle_read_to(self.READ[which], data, len(data), hdr, imgs, n.value, ctypes.byref(n), out.ctypes.data, out.size, ctypes.byref(nb))
        res = [(im.w, im.h, im.stride, out[im.offset:im.offset + im.size].copy()) for im in imgs[: n.value]]
        return st, list(hdr), res

    def lz_decode(self, ecb, scb, tables, n_ep, n_sel, is_video, slice_bytes, nbx, nby):
        L = self.lib
        L.bu_oracle_lz_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t,
                                          ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        ep = np.zeros(n_ep * 4, dtype=np.uint8)
        sel = np.zeros(n_sel * 8, dtype=np.uint8)
        idx = np.zeros(nbx * nby * 2, dtype=np.uint16)
        st = L.bu_oracle_lz_decode(ecb, len(ecb), scb, len(scb), tables, len(tables), n_ep, n_sel, int(is_video), slice_bytes, len(slice_bytes),
                                   nbx, nby, ep.ctypes.data, sel.ctypes.data, idx.ctypes.data)
        return st, ep.view("<u4"), sel.reshape(-1, 8), idx.reshape(-1, 2)

    def batch(self, target, blocks):
        """blocks [n,16] uint8 -> (out [n,bytes], status [n])"""
        t, obs = TARGETS[target]
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
        out = np.zeros((blocks.shape[0], obs), dtype=np.uint8)
        st = np.zeros(blocks.shape[0], dtype=np.uint8)
        self.lib.bu_oracle_batch(t, blocks.ctypes.data, blocks.shape[0], out.ctypes.data, st.ctypes.data)
        return out, st

    def transcode(self, target, data):
        """slice driver with the reference's first-error-aborts semantics -> (status, first_bad, out)"""
        t, obs = TARGETS[target]
        data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        out = np.zeros((data.size // 16 + 1) * obs, dtype=np.uint8)
        fb = ctypes.c_size_t(0)
        st = self.lib.bu_oracle_transcode(t, data.ctypes.data, data.size, out.ctypes.data, ctypes.byref(fb))
        return st, fb.value, out[: (data.size // 16) * obs]

    def decode_to_rgba(self, data, bpr):
        data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        out = np.zeros((data.size // 16 + 1) * 64, dtype=np.uint8)
        fb = ctypes.c_size_t(0)
        st = self.lib.bu_oracle_decode_to_rgba(data.ctypes.data, data.size, bpr, out.ctypes.data, ctypes.byref(fb))
        return st, fb.value, out[: (data.size // 16) * 64]

    def selectors_from_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 4)
        out = np.zeros((rows.shape[0], 8), dtype=np.uint8)
        for i in range(rows.shape[0]):
            self.lib.bu_oracle_selector_from_rows(rows[i].ctypes.data, out[i].ctypes.data)
        return out

    def etc1s_to_etc1(self, idx, endpoints, selectors):
        idx16 = np.ascontiguousarray(idx, dtype=np.uint32).view(np.uint16)  # {ep, sel} little endian pairs
        ep = np.ascontiguousarray(endpoints, dtype=np.uint32)
        sel = np.ascontiguousarray(selectors, dtype=np.uint8)
        out = np.zeros(idx.size * 8, dtype=np.uint8)
        self.lib.bu_oracle_etc1s_to_etc1(idx16.ctypes.data, idx.size, ep.ctypes.data, sel.ctypes.data, out.ctypes.data)
        return out

    def etc1s_to_rgba(self, idx, alpha_idx, nbx, nby, endpoints, selectors):
        idx16 = np.ascontiguousarray(idx, dtype=np.uint32).view(np.uint16)
        a16 = None if alpha_idx is None else np.ascontiguousarray(alpha_idx, dtype=np.uint32).view(np.uint16)
        ep = np.ascontiguousarray(endpoints, dtype=np.uint32)
        sel = np.ascontiguousarray(selectors, dtype=np.uint8)
        out = np.zeros(nbx * nby * 64, dtype=np.uint8)
        self.lib.bu_oracle_etc1s_to_rgba(idx16.ctypes.data, None if a16 is None else a16.ctypes.data, nbx, nby, ep.ctypes.data,
                                         sel.ctypes.data, out.ctypes.data)
        return out


class Decoders:
    """ctypes view of oracle/libbu_decoders.so: independent ASTC / BC7 / EAC decoders (bu_decoders.c) -- test-only."""

    def __init__(self):
        _make(os.path.join(ROOT, "oracle"), "libbu_decoders.so")
        self.lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libbu_decoders.so"))
        L, c = self.lib, ctypes
        L.bu_dec_astc_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.bu_dec_astc_batch.restype = None
        L.bu_dec_bc7_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.bu_dec_bc7_batch.restype = None
        L.bu_dec_eac_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_void_p]
        L.bu_dec_eac_batch.restype = None
        L.bu_dec_astc_partition_4x4.argtypes = [c.c_int] * 4
        L.bu_dec_bc7_subset.argtypes = [c.c_int] * 3
        L.bu_dec_bc7_anchor.argtypes = [c.c_int] * 3

    def _batch(self, fn, blocks):
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
        out = np.zeros((blocks.shape[0], 64), dtype=np.uint8)
        st = np.zeros(blocks.shape[0], dtype=np.uint8)
        fn(blocks.ctypes.data, blocks.shape[0], out.ctypes.data, st.ctypes.data)
        return out, st

    def astc(self, blocks):
        """[n,16] ASTC 4x4 LDR blocks -> ([n,64] RGBA8 row-major texels, status[n])"""
        return self._batch(self.lib.bu_dec_astc_batch, blocks)

    def bc7(self, blocks):
        return self._batch(self.lib.bu_dec_bc7_batch, blocks)

    def eac_alpha(self, etc2_blocks):
        """[n,16] ETC2 RGBA blocks -> [n,16] alpha of the EAC half, row-major texels"""
        b = np.ascontiguousarray(etc2_blocks, dtype=np.uint8).reshape(-1, 16)
        out = np.zeros((b.shape[0], 16), dtype=np.uint8)
        self.lib.bu_dec_eac_batch(b.ctypes.data, 16, b.shape[0], out.ctypes.data)
        return out
