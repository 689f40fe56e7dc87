"""Round trips through INDEPENDENT decoders of the target formats (oracle/bu_decoders.c, written from the Khronos
format specifications -- the reference crate has no ASTC / BC7 / EAC decoders).  SURVEY.md 8c / 8f-4: the reference's
3 040 known-answer vectors cover 32 blocks per UASTC mode; these properties hold for EVERY valid block, so they reach
the partition patterns, anchors, endpoint ranges and p-bit cases the vectors do not.

  ASTC  exact:  decode(transcode_to_astc(b)) == decode_block_to_rgba(b)     (UASTC is an ASTC subset)
  BC7   close:  |decode(transcode_to_bc7(b)) - rgba(b)| <= a small per-mode bound (endpoint / weight requantisation)
  EAC   the alpha selectors are the nearest table values; on encoder-made blocks the result is close to the source
"""
import numpy as np
import pytest

from basisu_rs_amd import synth
from oracle.pyoracle import Decoders


@pytest.fixture(scope="module")
def dec():
    return Decoders()


N_RAND = 200_000

# per-mode bound on |BC7 - UASTC| per channel.  Exact: solid colour.  <= 2: 8-bit endpoints -> 7+p bits with weights
# kept or widened losslessly.  Others requantise endpoints to 5..7 bits and/or weights 3 -> 4, 5 -> 4 bits.
BC7_BOUND = {8: 0, 1: 2, 4: 2, 6: 1, 11: 1, 13: 1, 14: 2, 17: 1}
BC7_BOUND_DEFAULT = 10


def test_decoders_reproduce_the_reference_vectors(golden, dec):
    """pins the decoders themselves: the reference's ASTC vectors decode EXACTLY to the reference's RGBA vectors"""
    out, st = dec.astc(golden["astc"])
    assert (st == 0).all()
    assert (out == golden["rgba"]).all()
    out, st = dec.bc7(golden["bc7"])
    assert (st == 0).all()
    err = np.abs(out.astype(int) - golden["rgba"].astype(int))
    assert err.max() <= 6 and err.mean() < 0.7
    a = dec.eac_alpha(golden["etc2"])
    ea = np.abs(a.astype(int) - golden["rgba"].reshape(-1, 16, 4)[:, :, 3].astype(int))
    assert ea.max() <= 16 and ea.mean() < 0.5


def _check_astc_bc7(dec, blocks, astc, bc7, rgba):
    modes = synth.block_modes(blocks)
    out, st = dec.astc(astc)
    assert (st == 0).all(), "the transcoder emitted something a generic ASTC decoder rejects"
    bad = np.where((out != rgba).any(axis=1))[0]
    assert bad.size == 0, "ASTC round trip differs at blocks %s (modes %s)" % (bad[:5], modes[bad[:5]])
    out, st = dec.bc7(bc7)
    assert (st == 0).all()
    err = np.abs(out.astype(int) - rgba.astype(int)).max(axis=1)
    for m in range(19):
        sel = modes == m
        if sel.any():
            assert err[sel].max() <= BC7_BOUND.get(m, BC7_BOUND_DEFAULT), (m, int(err[sel].max()))


def test_round_trip_of_the_oracle_on_random_valid_blocks(oracle, dec):
    blocks = synth.atlas_rand(N_RAND, seed=7)
    astc, st = oracle.batch("astc", blocks)
    assert (st == 0).all()
    bc7, _ = oracle.batch("bc7", blocks)
    rgba, _ = oracle.batch("rgba", blocks)
    _check_astc_bc7(dec, blocks, astc, bc7, rgba)
    assert set(synth.block_modes(blocks)) == set(range(19))


def test_round_trip_of_the_device_code_host_build(emul, dec):
    """same property on the per-block code the kernels run (compiled for the host, tests/host_emul)"""
    blocks = synth.atlas_rand(60_000, seed=21)
    astc, st = emul.batch("astc", blocks)
    assert (st == 0).all()
    bc7, _ = emul.batch("bc7", blocks)
    rgba, _ = emul.batch("rgba", blocks)
    _check_astc_bc7(dec, blocks, astc, bc7, rgba)


def test_eac_selectors_are_nearest_table_values(oracle, dec):
    """etc.rs:277-341: each texel takes the first of the 8 table values nearest to its alpha.  Checked from the emitted
    block alone: decode it, rebuild the 8 candidates from its header, compare distances."""
    blocks = synth.atlas_rand(50_000, seed=3)
    modes = synth.block_modes(blocks)
    etc2, st = oracle.batch("etc2", blocks)
    assert (st == 0).all()
    rgba, _ = oracle.batch("rgba", blocks)
    alpha = rgba.reshape(-1, 16, 4)[:, :, 3].astype(int)
    got = dec.eac_alpha(etc2).astype(int)
    # candidate values from the block header, with the specification's table (independent of the transcoder's copy)
    from ctypes import c_int8
    tab = np.ctypeslib.as_array((c_int8 * 128).in_dll(dec.lib, "EAC_MOD_EXPORT")).reshape(16, 8).astype(int)
    base = etc2[:, 0].astype(int)
    mult = (etc2[:, 1] >> 4).astype(int)
    table = (etc2[:, 1] & 15).astype(int)
    cand = np.clip(base[:, None] + tab[table] * mult[:, None], 0, 255)  # [n, 8]
    best = np.abs(cand[:, None, :] - alpha[:, :, None]).min(axis=2)  # [n, 16]
    # the one exception the reference makes: a zero table/multiplier hint byte means "opaque", whatever the texels hold
    # (etc.rs:283-286).  The hint sits right after the ETC1 flags (uastc.rs:411-436).
    code_size = np.array([4, 6, 5, 5, 5, 5, 5, 5, 5, 5, 3, 2, 3, 5, 5, 7, 6, 6, 4])
    m1012 = (modes >= 10) & (modes <= 12)
    pos = code_size[modes] + np.where(m1012, 1, 2) + 8 + np.where(m1012, 0, 5)
    word = blocks[:, :8].copy().view("<u8").reshape(-1)
    hint = ((word >> pos.astype(np.uint64)) & np.uint64(0xFF)).astype(int)
    has_alpha = ~np.isin(modes, [0, 1, 2, 3, 4, 5, 6, 7, 8, 18])
    opaque_hint = has_alpha & (hint == 0)
    assert opaque_hint.sum() > 50  # the case is exercised
    assert (got[opaque_hint] == 255).all()
    rest = ~opaque_hint
    assert (np.abs(got - alpha)[rest] == best[rest]).all()
    rgb_modes = np.isin(modes, [0, 1, 2, 3, 4, 5, 6, 7, 18])
    assert (got[rgb_modes] == 255).all()  # opaque formats: the constant-255 block


@pytest.mark.gpu
def test_round_trip_of_the_hip_outputs(ctx, dec):
    from basisu_rs_amd import _lib

    n = 100_000
    blocks = synth.atlas_rand(n, seed=33)
    astc = ctx.transcode(_lib.ASTC, blocks).reshape(n, 16)
    bc7 = ctx.transcode(_lib.BC7, blocks).reshape(n, 16)
    img = ctx.decode_to_rgba(blocks, 1000).reshape(n // 1000, 4, 1000, 16)
    rgba = np.ascontiguousarray(img.transpose(0, 2, 1, 3)).reshape(n, 64)
    _check_astc_bc7(dec, blocks, astc, bc7, rgba)
