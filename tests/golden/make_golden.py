#!/usr/bin/env python3
"""Build tests/golden/uastc_kat.bin from the reference's known-answer block vectors.

Source of the vectors (data tables only, no code): the reference's
tests/block_test_cases/uastc_{astc,bc7,etc1,etc2,rgba}.rs, which
tests/transcode_uastc_block.rs:35-78 asserts bit-exactly in the reference CI.
They hold 19 modes x 32 vectors of (uastc[16], expected) for the five targets;
the same 608 UASTC inputs appear in all five files.

File layout (little endian):
  magic  "BUKAT1\\0\\0"            8 bytes
  count  u32 (= 608), rec_size u32 (= 136)
  count records: uastc[16] astc[16] bc7[16] etc1[8] etc2[16] rgba[64]
  rgba = 16 texels, row-major inside the block, bytes R,G,B,A  (src/color.rs:22-24: the
  reference's u32 is little-endian RGBA)
Records are ordered mode-major: record 32*m+k is vector k of UASTC mode m.

Run in the build container only (needs /root/reference); the .bin is committed.
"""
import os
import re
import struct
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

PAIR = re.compile(r"\(\s*\[(\s*0x[^\]]*)\]\s*,\s*\[(\s*0x[^\]]*)\]\s*\)")


def load(name):
    text = open(os.path.join(REF, "tests", "block_test_cases", name)).read()
    out = []
    for a, b in PAIR.findall(text):
        ia = [int(t, 0) for t in a.replace(" ", "").split(",") if t]
        ib = [int(t, 0) for t in b.replace(" ", "").split(",") if t]
        out.append((ia, ib))
    return out


def main():
    astc = load("uastc_astc.rs")
    bc7 = load("uastc_bc7.rs")
    etc1 = load("uastc_etc1.rs")
    etc2 = load("uastc_etc2.rs")
    rgba = load("uastc_rgba.rs")
    n = len(astc)
    assert n == 608 and all(len(x) == n for x in (bc7, etc1, etc2, rgba))
    blob = bytearray()
    blob += b"BUKAT1\0\0" + struct.pack("<II", n, 136)
    for i in range(n):
        u = astc[i][0]
        assert len(u) == 16
        for other in (bc7, etc1, etc2, rgba):
            assert other[i][0] == u, "input mismatch at %d" % i
        assert len(astc[i][1]) == 16 and len(bc7[i][1]) == 16 and len(etc1[i][1]) == 8
        assert len(etc2[i][1]) == 16 and len(rgba[i][1]) == 16
        blob += bytes(u) + bytes(astc[i][1]) + bytes(bc7[i][1]) + bytes(etc1[i][1]) + bytes(etc2[i][1])
        blob += b"".join(struct.pack("<I", v) for v in rgba[i][1])
    assert len(blob) == 16 + n * 136
    path = os.path.join(HERE, "uastc_kat.bin")
    open(path, "wb").write(blob)
    print("wrote", path, len(blob), "bytes")


if __name__ == "__main__":
    main()
