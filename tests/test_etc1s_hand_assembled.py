"""A hand-assembled ETC1S `.basis` file: every field and every Huffman code is written out below, bit string by bit string, each with
the line of the reference that reads it -- independent of tests/basis_builder.py (the test-only encoder), so a misreading shared by
that encoder and the decoders cannot hide here.  The expected codebooks and block indices are hand-traced from the reference's text
(basis_lz/mod.rs, huffman.rs) and written as literals; the product's host decoder (bu_basislz_decode: fast path AND exact path) and
the oracle must both reproduce them.

The reference pins nothing in this half (tests/corpus_tests.rs:54-73 is #[ignore]d and its corpus is absent), so this is still a
reading of the source, not a vector of the reference's CI -- but it is a second, independent reading, bit by bit.

Conventions of the bit stream (bitreader.rs:3-61): fields are read LSB first -- `U(value, n)` below -- and a Huffman code appears in
the stream most significant bit of its canonical code first (huffman.rs:163-170 stores the codes bit-reversed because
decode_symbol peeks LSB-first, :186-198) -- `C("101")` below writes the bits in stream order."""
import struct

import numpy as np

from basisu_rs_amd import _lib


class Bits:
    def __init__(self):
        self.bits = []

    def U(self, value, n, _why=""):  # an n-bit field, least significant bit first (bitreader.rs:27-47)
        assert 0 <= value < (1 << n)
        self.bits += [(value >> k) & 1 for k in range(n)]
        return self

    def C(self, code, _why=""):  # a Huffman code, in stream order
        self.bits += [int(c) for c in code]
        return self

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(sum(b[8 * i + k] << k for k in range(8)) for i in range(len(b) // 8))


# order in which the code-length code sizes are stored (huffman.rs:52-57)
CL_ORDER = [17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16]


def table_head(w, total_used_syms, cl_sizes):
    """huffman.rs:46-63: 14 bits total_used_syms, 5 bits num_codelength_codes, that many 3-bit sizes in CL_ORDER"""
    n = max(CL_ORDER.index(s) for s in cl_sizes) + 1
    w.U(total_used_syms, 14, "huffman.rs:46").U(n, 5, "huffman.rs:49")
    for s in CL_ORDER[:n]:
        w.U(cl_sizes.get(s, 0), 3, "huffman.rs:59-61")


def one_symbol_table(w):
    """a table whose only symbol is 0, code '0' (1 bit): sizes = [1]; code-length alphabet: only symbol 1, code '0'"""
    table_head(w, 1, {1: 1})
    w.C("0", "size of symbol 0 = 1 (huffman.rs:66-70)")


def crc16(data):  # basis.rs:364-372 with crc = 0
    crc = 0xFFFF
    for b in data:
        q = (b ^ (crc >> 8)) & 0xFFFF
        k = ((q >> 4) ^ q) & 0xFFFF
        crc = (((crc << 8) ^ k) ^ (k << 5) ^ (k << 12)) & 0xFFFF
    return crc ^ 0xFFFF


def endpoint_codebook():
    """mod.rs:461-516.  Four endpoints; start colour (16,16,16), intensity 0 (:472-473).  Colour deltas come from model 0 / 1 / 2 by
    the channel's previous value (0-9 / 10-21 / 22-31, mod.rs:28-37)."""
    w = Bits()
    one_symbol_table(w)  # colour-delta model 0: symbol 0 = '0'
    # colour-delta model 1: symbols 0 ('0'), 5 ('10'), 27 ('11').  28 code sizes: [1, 0,0,0,0, 2, 0 x 21, 2].
    # code-length alphabet: 1 -> '00', 2 -> '01', 17 -> '10', 18 -> '11' (all 2 bits, canonical in symbol order)
    table_head(w, 28, {1: 2, 2: 2, 17: 2, 18: 2})
    w.C("00", "symbol 0: size 1")
    w.C("10", "small zero run (17)").U(1, 3, "3 + 1 = 4 zeros: symbols 1-4 (huffman.rs:71-74)")
    w.C("01", "symbol 5: size 2")
    w.C("11", "big zero run (18)").U(10, 7, "11 + 10 = 21 zeros: symbols 6-26 (huffman.rs:75-78)")
    w.C("01", "symbol 27: size 2")
    one_symbol_table(w)  # colour-delta model 2: symbol 0 = '0'
    # intensity-delta model: symbols 1 ('0') and 6 ('1').  7 code sizes [0,1,0,0,0,0,1] as literals; code-length alphabet 0 -> '0', 1 -> '1'
    table_head(w, 7, {0: 1, 1: 1})
    for size in "0100001":
        w.C(size, "a literal code size")
    w.U(0, 1, "grayscale = false (mod.rs:469)")
    # entry 0: intensity 0 + 1 = 1; R 16 (model 1) + 5 = 21; G 16 + 0 = 16; B 16 + 27 = 43 & 31 = 11
    w.C("0", "inten delta 1").C("10", "R: model 1, delta 5").C("0", "G: model 1, delta 0").C("11", "B: model 1, delta 27")
    # entry 1: intensity 1 + 6 = 7; R 21 (model 1) + 5 = 26; G 16 + 27 = 11; B 11 (model 1) + 27 = 38 & 31 = 6
    w.C("1", "inten delta 6").C("10", "R: model 1, delta 5").C("11", "G: model 1, delta 27").C("11", "B: model 1, delta 27")
    # entry 2: intensity (7 + 1) & 7 = 0; R 26 (model 2) + 0; G 11 (model 1) + 0; B 6 (model 0) + 0
    w.C("0", "inten delta 1").C("0", "R: model 2, delta 0").C("0", "G: model 1, delta 0").C("0", "B: model 0, delta 0")
    # entry 3: intensity 0 + 6 = 6; R 26 (model 2) + 0; G 11 (model 1) + 5 = 16; B 6 (model 0) + 0
    w.C("1", "inten delta 6").C("0", "R: model 2, delta 0").C("10", "G: model 1, delta 5").C("0", "B: model 0, delta 0")
    return w.bytes()


ENDPOINTS = [(21, 16, 11, 1), (26, 11, 6, 7), (26, 11, 6, 0), (26, 16, 6, 6)]  # (r5, g5, b5, inten)


def selector_codebook():
    """mod.rs:524-583, the DPCM form: selector 0 raw (4 row bytes, x = 0 in the low bits), later ones 4 symbols XOR-ed onto the
    previous selector's rows."""
    w = Bits()
    w.U(0, 1, "global = false (mod.rs:527)").U(0, 1, "hybrid = false").U(0, 1, "raw = false")
    # delta table over 256 byte values: 0x00 ('0'), 0x1B ('10'), 0xFF ('11'); sizes [1, 0 x 26, 2, 0 x 227, 2]
    # code-length alphabet: 18 -> '0' (1 bit); 1 -> '10', 2 -> '11'
    table_head(w, 256, {18: 1, 1: 2, 2: 2})
    w.C("10", "symbol 0x00: size 1")
    w.C("0", "big zero run").U(15, 7, "11 + 15 = 26 zeros: symbols 1-26")
    w.C("11", "symbol 0x1B: size 2")
    w.C("0", "big zero run").U(127, 7, "11 + 127 = 138 zeros: symbols 28-165")
    w.C("0", "big zero run").U(78, 7, "11 + 78 = 89 zeros: symbols 166-254")
    w.C("11", "symbol 0xFF: size 2")
    for row in (0xE4, 0x1B, 0x00, 0xFF):
        w.U(row, 8, "selector 0, raw rows (mod.rs:545-553)")
    for delta in ("0", "11", "10", "0"):  # ^00 ^FF ^1B ^00 -> E4 E4 1B FF
        w.C(delta, "selector 1 (mod.rs:556-566)")
    for delta in ("11", "0", "0", "10"):  # ^FF ^00 ^00 ^1B -> 1B E4 1B E4
        w.C(delta, "selector 2")
    for delta in ("10", "10", "11", "11"):  # ^1B ^1B ^FF ^FF -> 00 FF E4 1B
        w.C(delta, "selector 3")
    return w.bytes()


SELECTOR_ROWS = [(0xE4, 0x1B, 0x00, 0xFF), (0xE4, 0xE4, 0x1B, 0xFF), (0x1B, 0xE4, 0x1B, 0xE4), (0x00, 0xFF, 0xE4, 0x1B)]


def slice_tables():
    """mod.rs:77-83: endpoint-predictor, delta-endpoint, selector, selector-history-RLE tables, then 13 bits history size"""
    w = Bits()
    # endpoint predictor: symbols 147 = 0x93 ('0': predictors 3, 0 / 1, 2 for the 2 x 2 group) and 256 ('1': repeat, mod.rs:14-17).
    # 257 sizes: 147 zeros, 1, 108 zeros, 1.  code-length alphabet: 1 -> '0'; 17 -> '10', 18 -> '11'
    table_head(w, 257, {1: 1, 17: 2, 18: 2})
    w.C("11", "big zero run").U(127, 7, "138 zeros: symbols 0-137")
    w.C("10", "small zero run").U(6, 3, "3 + 6 = 9 zeros: symbols 138-146")
    w.C("0", "symbol 147: size 1")
    w.C("11", "big zero run").U(97, 7, "11 + 97 = 108 zeros: symbols 148-255")
    w.C("0", "symbol 256: size 1")
    # delta endpoint over symbols 0-3: sizes [0, 2, 2, 1] -> 3 = '0', 1 = '10', 2 = '11'.  code-length alphabet: 2 -> '0'; 0 -> '10', 1 -> '11'
    table_head(w, 4, {2: 1, 0: 2, 1: 2})
    w.C("10", "symbol 0: size 0").C("0", "symbol 1: size 2").C("0", "symbol 2: size 2").C("11", "symbol 3: size 1")
    # selector over symbols 0-6 (4 codebook entries, 2 history entries, RLE): sizes [0, 3, 0, 3, 2, 2, 2]
    #   -> 4 = '00', 5 = '01', 6 = '10', 1 = '110', 3 = '111'.  code-length alphabet: 2 -> '0'; 0 -> '10', 3 -> '11'
    table_head(w, 7, {2: 1, 0: 2, 3: 2})
    for code in ("10", "11", "10", "11", "0", "0", "0"):
        w.C(code, "sizes 0 3 0 3 2 2 2")
    # selector-history RLE over run symbols 0-63: sizes [1, 0 x 62, 1] -> 0 = '0', 63 = '1'.  code-length alphabet: 1 -> '0', 18 -> '1'
    table_head(w, 64, {1: 1, 18: 1})
    w.C("0", "symbol 0: size 1").C("1", "big zero run").U(51, 7, "11 + 51 = 62 zeros").C("0", "symbol 63: size 1")
    w.U(2, 13, "selector history buffer size (mod.rs:83)")
    return w.bytes()


def slice_payload():
    """mod.rs:188-458 for a 4 x 4-block slice, raster order.  All four 2 x 2 groups use predictor symbol 0x93: top-left block
    delta-coded (3), top-right = left (0), bottom-left = above (1), bottom-right = above-left (2)."""
    w = Bits()
    # row 0
    w.C("0", "(0,0) endpoint predictors 0x93 (mod.rs:262)").C("10", "delta 1: endpoint 0 + 1 = 1 (mod.rs:343-351)").C("111", "selector 3; history [0, 3] (mod.rs:610-620)")
    w.C("01", "(1,0) history entry 1 = 3, swapped to the front: [3, 0] (mod.rs:635-642)")
    w.C("1", "(2,0) predictor symbol 256: repeat (mod.rs:263-271)").U(0, 5, "vlc(4) = 0: 0 + 3 - 1 = 2 more groups repeat 0x93 (mod.rs:585-608)")
    w.C("0", "delta 3: endpoint 1 + 3 = 4 -> wraps to 0").C("110", "selector 1; history [3, 1]")
    w.C("10", "(3,0) selector RLE").C("0", "run symbol 0: 3 blocks of history entry 0 = 3 (mod.rs:380-396)")
    # row 1: (0,1) and (1,1) finish the run; predictors come from the row above (mod.rs:281-298)
    w.C("01", "(2,1) history entry 1 = 1, swapped: [1, 3]")
    w.C("00", "(3,1) history entry 0 = 1")
    # row 2
    w.C("11", "(0,2) repeated predictors; delta 2: endpoint 0 + 2 = 2").C("10", "selector RLE").C("1", "run symbol 63: escape").U(1, 8, "vlc(7) = 1: 3 + 1 = 4 blocks of entry 0 = 1")
    w.C("0", "(2,2) repeated predictors; delta 3: endpoint 2 + 3 = 5 -> wraps to 1")
    # row 3
    w.C("111", "(0,3) selector 3; history [1, 3]")
    w.C("110", "(1,3) selector 1; history [1, 1]")
    w.C("01", "(2,3) history entry 1 = 1")
    w.C("00", "(3,3) history entry 0 = 1")
    return w.bytes()


ENDPOINT_INDEX = [1, 1, 0, 0, 1, 1, 0, 0, 2, 2, 1, 1, 2, 2, 1, 1]
SELECTOR_INDEX = [3, 3, 1, 3, 3, 3, 1, 1, 1, 1, 1, 1, 3, 1, 1, 1]


def the_file():
    ecb, scb, tab, sl = endpoint_codebook(), selector_codebook(), slice_tables(), slice_payload()
    ofs_desc = 77
    ofs_ecb = ofs_desc + 23
    ofs_scb = ofs_ecb + len(ecb)
    ofs_tab = ofs_scb + len(scb)
    ofs_slice = ofs_tab + len(tab)
    # slice descriptor, 23 bytes (basis.rs:554-571)
    desc = (struct.pack("<I", 0)[:3]  # image_index u24
            + struct.pack("<BBHHHHIIH", 0, 0, 16, 16, 4, 4, ofs_slice, len(sl), crc16(sl)))  # level, flags, orig w/h, blocks x/y, ofs, size, crc
    payload = desc + ecb + scb + tab + sl
    h = bytearray(77)  # basis.rs:578-617
    struct.pack_into("<HHH", h, 0, 0x4273, 0x13, 77)  # sig, ver, header_size
    struct.pack_into("<I", h, 8, len(payload))        # data_size
    struct.pack_into("<H", h, 12, crc16(payload))     # data_crc16 = crc16(bytes[77..])
    h[14:17] = struct.pack("<I", 1)[:3]                # total_slices
    h[17:20] = struct.pack("<I", 1)[:3]                # total_images
    h[20] = 0                                          # tex_format ETC1S
    struct.pack_into("<H", h, 21, 1)                   # flags: ETC1S
    struct.pack_into("<H", h, 39, 4)                   # total_endpoints
    struct.pack_into("<I", h, 41, ofs_ecb)
    h[45:48] = struct.pack("<I", len(ecb))[:3]
    struct.pack_into("<H", h, 48, 4)                   # total_selectors (sizes BOTH codebooks: basis.rs:289-291)
    struct.pack_into("<I", h, 50, ofs_scb)
    h[54:57] = struct.pack("<I", len(scb))[:3]
    struct.pack_into("<I", h, 57, ofs_tab)
    struct.pack_into("<I", h, 61, len(tab))
    struct.pack_into("<I", h, 65, ofs_desc)            # slice_desc_file_ofs
    struct.pack_into("<H", h, 6, crc16(bytes(h[8:77])))  # header_crc16 = crc16(bytes[8..77])
    return bytes(h) + payload


# the four sections, byte for byte (what the annotated bit strings above assemble to): a change of the helpers cannot move them silently
ECB_HEX = "01c0040000000000008203981200000000008228ae300098000000000000401c00130002000000008890bc8702"
SCB_HEX = "000826020000000080a03cf6e79c7c03e0df99ea01"
TAB_HEX = "01c194000000000000f23f3b8c009800200000000081e20320028000000020a41b40c044000000000000e28c0000"
SLICE_HEX = "ba81456e804f00"


def test_sections_are_the_bytes_written_out_here():
    assert endpoint_codebook().hex() == ECB_HEX
    assert selector_codebook().hex() == SCB_HEX
    assert slice_tables().hex() == TAB_HEX
    assert slice_payload().hex() == SLICE_HEX


def _expected_idx():
    return np.array([e | (s << 16) for e, s in zip(ENDPOINT_INDEX, SELECTOR_INDEX)], dtype=np.uint32)


def test_product_host_decoder_reproduces_the_hand_trace():
    """bu_basis_read_header / bu_basis_read_slice_descs / bu_basislz_decode (host only: no GPU involved) on the hand-assembled file"""
    import basisu_rs_amd as bu

    f = the_file()
    h = bu.read_header(f)
    assert (h.total_slices, h.tex_format, h.total_endpoints, h.total_selectors) == (1, 0, 4, 4)
    sd = bu.read_slice_descs(f, h)[0]
    assert (sd.num_blocks_x, sd.num_blocks_y, sd.orig_width, sd.orig_height) == (4, 4, 16, 16)
    assert bu.crc16(f[77:]) == h.data_crc16
    ep, sel, idx = bu.basislz_decode(f, 0)
    assert [(int(e) & 31, (int(e) >> 8) & 31, (int(e) >> 16) & 31, int(e) >> 24) for e in ep] == ENDPOINTS
    assert [tuple(int(x) for x in s[:4]) for s in sel] == SELECTOR_ROWS
    assert (sel == bu.etc1s_selector_from_rows(np.array(SELECTOR_ROWS, dtype=np.uint8))).all()  # the ETC1 bit planes of the same rows
    assert (idx == _expected_idx()).all()


def test_exact_and_fast_slice_loops_agree_on_it():
    """the slice loop has a fast form and the exact form it falls back to (csrc/bu_basis.hpp): damage that makes the fast form give
    up (a predictor without a source: the top-left block may not copy from its left) must be reported exactly as the oracle does"""
    import basisu_rs_amd as bu
    from basisu_rs_amd import BasisuError
    from oracle.pyoracle import Oracle

    f = bytearray(the_file())
    h = bu.read_header(bytes(f))
    sd = bu.read_slice_descs(bytes(f), h)[0]
    # flip the first slice bit: predictor symbol '0' (0x93) becomes '1' (256: repeat the "previous" symbol 0 -> predictor 0 at (0,0))
    f[sd.file_ofs] ^= 1
    g = bytearray(f)
    payload = bytes(g[77:])
    struct.pack_into("<H", g, 12, crc16(payload))
    struct.pack_into("<H", g, 6, crc16(bytes(g[8:77])))
    st_o = Oracle().read_to("rgba", bytes(g))[0]
    assert st_o != 0
    try:
        bu.basislz_decode(bytes(g), 0)
        st_p = 0
    except BasisuError as e:
        st_p = e.status
    assert st_p == st_o


def test_oracle_reproduces_the_hand_trace(oracle):
    f = the_file()
    st, hdr, imgs = oracle.read_to("rgba", f)
    assert st == 0 and len(imgs) == 1 and imgs[0][0] == 16 and imgs[0][1] == 16
    import basisu_rs_amd as bu

    hh = bu.read_header(f)
    sd = bu.read_slice_descs(f, hh)[0]
    ecb = f[hh.endpoint_cb_file_ofs: hh.endpoint_cb_file_ofs + hh.endpoint_cb_file_size]
    scb = f[hh.selector_cb_file_ofs: hh.selector_cb_file_ofs + hh.selector_cb_file_size]
    tab = f[hh.tables_file_ofs: hh.tables_file_ofs + hh.tables_file_size]
    sl = f[sd.file_ofs: sd.file_ofs + sd.file_size]
    st, ep, sel, idx = oracle.lz_decode(ecb, scb, tab, 4, 4, False, sl, 4, 4)
    assert st == 0
    assert [(int(e) & 31, (int(e) >> 8) & 31, (int(e) >> 16) & 31, int(e) >> 24) for e in ep] == ENDPOINTS
    assert [tuple(int(x) for x in s[:4]) for s in sel] == SELECTOR_ROWS
    assert [int(x) for x in idx[:, 0]] == ENDPOINT_INDEX and [int(x) for x in idx[:, 1]] == SELECTOR_INDEX
