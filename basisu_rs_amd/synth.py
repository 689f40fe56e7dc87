"""Synthetic UASTC / ETC1S inputs of the measurement plan.

  A-gold  block i = G[h(i) mod 608], G = the reference's 608 known-answer UASTC blocks: the expected
          output of every target is the known-answer output, so a 4096x4096 result can be verified
          without any CPU transcode; modes are uniformly mixed (worst case for divergence).
  A-coh   as A-gold but the mode is chosen per 8x8-block tile (texture-like coherence).
  A-rand  128 random bits per block, repaired to be valid (mode code != 69, pattern index in range):
          exercises paths the known-answer vectors do not reach; expected output from the oracle.
  A-err   A-gold with a few blocks replaced by invalid ones (error-reporting parity).
Pure numpy; no dependency on the oracle or the HIP library.
"""
import struct

import numpy as np

GOLD_SEED = 0xBA515

# (code_size, tf_bits, pattern_bits, pattern_count) for the modes that carry a pattern index
# (uastc.rs:528-557, 352-366); all of them are single-plane so the field follows the flags directly
_PATTERN_FIELD = {2: (5, 15, 5, 30), 3: (5, 15, 4, 11), 4: (5, 15, 5, 30), 7: (5, 15, 5, 19), 9: (5, 23, 5, 30), 16: (6, 23, 5, 30)}
_MODE_LUT = None


def load_golden(path):
    """tests/golden/uastc_kat.bin -> dict of uint8 arrays [608, n] (layout: tests/golden/make_golden.py)."""
    d = open(path, "rb").read()
    assert d[:8] == b"BUKAT1\0\0"
    n, rs = struct.unpack_from("<II", d, 8)
    a = np.frombuffer(d, dtype=np.uint8, offset=16).reshape(n, rs)
    return {
        "uastc": a[:, 0:16].copy(), "astc": a[:, 16:32].copy(), "bc7": a[:, 32:48].copy(),
        "etc1": a[:, 48:56].copy(), "etc2": a[:, 56:72].copy(), "rgba": a[:, 72:136].copy(),
    }


def hash32(i, seed):
    """32-bit xorshift-multiply hash of the block index (vectorised)."""
    x = (np.asarray(i, dtype=np.uint64) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def gold_indices(n_blocks, seed=GOLD_SEED):
    return (hash32(np.arange(n_blocks, dtype=np.uint64), seed) % np.uint64(608)).astype(np.int64)


def coh_indices(nbx, nby, seed=GOLD_SEED):
    """mode per 8x8-block tile, vector within the mode per block"""
    by, bx = np.divmod(np.arange(nbx * nby, dtype=np.uint64), np.uint64(nbx))
    tile = (by // np.uint64(8)) * np.uint64((nbx + 7) // 8) + bx // np.uint64(8)
    mode = hash32(tile, seed ^ 0x5A5A) % np.uint64(19)
    vec = hash32(np.arange(nbx * nby, dtype=np.uint64), seed) % np.uint64(32)
    return (mode * np.uint64(32) + vec).astype(np.int64)


def atlas_from_indices(golden_uastc, idx):
    return np.ascontiguousarray(golden_uastc[idx])


def _mode_lut():
    global _MODE_LUT
    if _MODE_LUT is None:
        # prefix codes of the 19 modes (uastc.rs:560-577), rebuilt from (code value, code size)
        codes = {0: (0x1, 4), 1: (0x35, 6), 2: (0x1D, 5), 3: (0x3, 5), 4: (0x13, 5), 5: (0xB, 5), 6: (0x1B, 5), 7: (0x7, 5),
                 8: (0x17, 5), 9: (0xF, 5), 10: (0x2, 3), 11: (0x0, 2), 12: (0x6, 3), 13: (0x1F, 5), 14: (0xD, 5),
                 15: (0x5, 7), 16: (0x15, 6), 17: (0x25, 6), 18: (0x9, 4)}
        lut = np.full(128, 19, dtype=np.uint8)
        for m, (v, n) in codes.items():
            for hi in range(1 << (7 - n)):
                lut[v | (hi << n)] = m
        _MODE_LUT = lut
    return _MODE_LUT


def block_modes(blocks):
    return _mode_lut()[blocks[:, 0] & 127]


def atlas_rand(n_blocks, seed=1):
    """random-valid blocks: every block transcodes without error"""
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = rng.integers(0, 256, size=(n_blocks, 16), dtype=np.uint8)
    while True:  # the only invalid 7-bit code is 69
        bad = np.nonzero((blocks[:, 0] & 127) == 69)[0]
        if bad.size == 0:
            break
        blocks[bad, 0] = rng.integers(0, 256, size=bad.size, dtype=np.uint8)
    modes = block_modes(blocks)
    lo = blocks[:, :8].copy().view("<u8").reshape(-1)
    for m, (code, tf, pb, count) in _PATTERN_FIELD.items():
        sel = np.nonzero(modes == m)[0]
        if sel.size == 0:
            continue
        pos = np.uint64(code + tf)
        mask = np.uint64((1 << pb) - 1)
        v = (lo[sel] >> pos) & mask
        v = v % np.uint64(count)
        lo[sel] = (lo[sel] & ~(mask << pos)) | (v << pos)
    blocks[:, :8] = lo.view(np.uint8).reshape(-1, 8)
    return blocks


def _fix_pattern_fields(blocks):
    modes = block_modes(blocks)
    lo = blocks[:, :8].copy().view("<u8").reshape(-1)
    for m, (code, tf, pb, count) in _PATTERN_FIELD.items():
        sel = np.nonzero(modes == m)[0]
        if sel.size == 0:
            continue
        pos = np.uint64(code + tf)
        mask = np.uint64((1 << pb) - 1)
        v = ((lo[sel] >> pos) & mask) % np.uint64(count)
        lo[sel] = (lo[sel] & ~(mask << pos)) | (v << pos)
    blocks[:, :8] = lo.view(np.uint8).reshape(-1, 8)
    return blocks


def atlas_contrast(n_blocks, seed=1):
    """valid blocks of extreme contrast: the endpoint region is runs of all-zero and all-one bytes (endpoints at or near 0 and
    255), the weight region sparse or dense (most texels at one endpoint, a few at the other).  What uniform random bits almost never
    produce: texels far from their half's average colour -- the clamps of the ETC1 modifier tables, lumas more than 2^15 from the
    thresholds (the saturating i16 lanes of the GPU's selector stage), EAC tables run into 0 and 255."""
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = atlas_rand(n_blocks, seed=seed + 1000)
    ext = np.where(rng.integers(0, 2, size=(n_blocks, 9), dtype=np.uint8) == 1, 0xFF, 0x00).astype(np.uint8)
    keep = rng.integers(0, 4, size=(n_blocks, 9)) == 0  # a quarter of the bytes stay random
    blocks[:, 3:12] = np.where(keep, blocks[:, 3:12], ext)
    sparse = rng.integers(0, 256, size=(n_blocks, 6), dtype=np.uint8) & rng.integers(0, 256, size=(n_blocks, 6), dtype=np.uint8) & rng.integers(0, 256, size=(n_blocks, 6), dtype=np.uint8)
    dense = rng.integers(0, 2, size=(n_blocks, 1), dtype=np.uint8) == 1
    blocks[:, 10:16] = np.where(dense, ~sparse, sparse)
    return _fix_pattern_fields(blocks)


def atlas_err(golden_uastc, n_blocks, bad_at, seed=GOLD_SEED):
    """A-gold with invalid blocks at the given indices: even slots get mode code 69, odd slots an
    out-of-range pattern index (UASTC mode 3 with pattern 15 >= 11)."""
    blocks = atlas_from_indices(golden_uastc, gold_indices(n_blocks, seed))
    for k, i in enumerate(bad_at):
        if k % 2 == 0:
            blocks[i, 0] = (blocks[i, 0] & 0x80) | 69
        else:
            b = np.zeros(16, dtype=np.uint8)
            v = 0x3 | (15 << 20)  # mode 3 code (5 bits) + 15 flag bits, then the 4-bit pattern = 15
            b[:4] = np.frombuffer(struct.pack("<I", v), dtype=np.uint8)
            blocks[i] = b
    return blocks


def etc1s_codebooks(n_endpoints=4096, n_selectors=8192, seed=2):
    """random-valid ETC1S codebooks: endpoints r5|g5<<8|b5<<16|inten<<24, selector rows [n,4]"""
    rng = np.random.Generator(np.random.PCG64(seed))
    c5 = rng.integers(0, 32, size=(n_endpoints, 3), dtype=np.uint32)
    inten = rng.integers(0, 8, size=n_endpoints, dtype=np.uint32)
    endpoints = c5[:, 0] | (c5[:, 1] << 8) | (c5[:, 2] << 16) | (inten << 24)
    rows = rng.integers(0, 256, size=(n_selectors, 4), dtype=np.uint8)
    return endpoints.astype(np.uint32), rows


def etc1s_indices(n_blocks, n_endpoints, n_selectors, seed=2):
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    e = rng.integers(0, n_endpoints, size=n_blocks, dtype=np.uint32)
    s = rng.integers(0, n_selectors, size=n_blocks, dtype=np.uint32)
    return (e | (s << 16)).astype(np.uint32)
