"""Parity tests proper: the HIP path, called through the C ABI, against the oracle and the reference's
known-answer vectors.  Bit-exact (integer / byte work).  Run on the GPU box: pytest -m gpu."""
import ctypes

import numpy as np
import pytest

from basisu_rs_amd import _lib, synth

pytestmark = pytest.mark.gpu

FMT = {"astc": _lib.ASTC, "bc7": _lib.BC7, "etc1": _lib.ETC1, "etc2": _lib.ETC2}
ALL = ["astc", "bc7", "etc1", "etc2", "rgba"]


def _gpu(ctx, target, blocks, bpr=None):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    if target == "rgba":  # block-linear view of the row-major image
        n = blocks.shape[0]
        bpr = bpr or n
        img = ctx.decode_to_rgba(blocks, bpr).reshape(n // bpr, 4, bpr, 16)
        return np.ascontiguousarray(img.transpose(0, 2, 1, 3)).reshape(n, 64)
    return ctx.transcode(FMT[target], blocks).reshape(blocks.shape[0], -1)


def test_per_block_api_reproduces_reference_vectors(ctx, golden):
    """lib.rs:29-53 through the drop-in entry points, the shape of tests/transcode_uastc_block.rs"""
    import basisu_rs_amd as bu

    fns = {"astc": bu.transcode_uastc_block_to_astc, "bc7": bu.transcode_uastc_block_to_bc7,
           "etc1": bu.transcode_uastc_block_to_etc1, "etc2": bu.transcode_uastc_block_to_etc2}

    def check(i):
        u = golden["uastc"][i]
        for t, f in fns.items():
            assert (f(u, ctx) == golden[t][i]).all(), (t, i)
        assert (bu.unpack_uastc_block_to_rgba(u, ctx).view(np.uint8) == golden["rgba"][i]).all()

    # default: the library's own block code on the calling thread -- all 608 x 5 vectors (tests/transcode_uastc_block.rs:35-78)
    for i in range(608):
        check(i)
    # the same entry points through the one-block launch (every mode, sampled: a launch per call is slow)
    ctx.block_api_on_device(True)
    try:
        for i in range(0, 608, 7):
            check(i)
    finally:
        ctx.block_api_on_device(False)


def test_per_block_api_errors_and_host_device_agreement(ctx, golden):
    """Error contract of lib.rs:26-53 on both routes of the per-block API (invalid mode code 69, pattern index out of range:
    uastc.rs:336,364), and host route == device route == oracle on random valid and raw random blocks."""
    import basisu_rs_amd as bu
    from basisu_rs_amd import BasisuError as BasisError, synth

    bad_mode = np.zeros(16, dtype=np.uint8)
    bad_mode[0] = 69
    bad_pat = golden["uastc"][2 * 32].copy()  # a mode-2 vector: the 5-bit pattern field sits right behind the 5-bit mode code + 15 flag bits
    bits = int.from_bytes(bad_pat.tobytes(), "little")
    bits = (bits & ~(31 << 20)) | (31 << 20)  # pattern 31 >= 30 patterns
    bad_pat = np.frombuffer(bits.to_bytes(16, "little"), dtype=np.uint8)
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, (300, 16), dtype=np.uint8)
    valid = synth.atlas_rand(300, seed=5)
    for on_device in (False, True):
        ctx.block_api_on_device(on_device)
        try:
            for blk, want in ((bad_mode, 1), (bad_pat, 2)):  # BU_ERR_INVALID_MODE, BU_ERR_INVALID_PATTERN
                for f in (bu.transcode_uastc_block_to_astc, bu.transcode_uastc_block_to_bc7, bu.transcode_uastc_block_to_etc1,
                          bu.transcode_uastc_block_to_etc2, bu.unpack_uastc_block_to_rgba):
                    with pytest.raises(BasisError) as e:
                        f(blk, ctx)
                    assert e.value.status == want
        finally:
            ctx.block_api_on_device(False)
    # host route against the slice path (the kernels) on the same blocks, statuses included
    for blocks in (valid, raw):
        for t, f in (("astc", bu.transcode_uastc_block_to_astc), ("bc7", bu.transcode_uastc_block_to_bc7),
                     ("etc1", bu.transcode_uastc_block_to_etc1), ("etc2", bu.transcode_uastc_block_to_etc2)):
            for u in blocks[:120]:
                try:
                    one = f(u, ctx)
                except BasisError as e1:
                    with pytest.raises(BasisError) as e2:
                        ctx.transcode(FMT[t], u.reshape(1, 16))
                    assert e2.value.status == e1.status
                    continue
                assert (ctx.transcode(FMT[t], u.reshape(1, 16)).reshape(-1) == one).all(), t


@pytest.mark.parametrize("target", ALL)
def test_slice_api_reproduces_all_reference_vectors(ctx, golden, target):
    out = _gpu(ctx, target, golden["uastc"], bpr=32)
    assert (out == golden[target]).all()


@pytest.mark.parametrize("target", ALL)
def test_raw_random_blocks_match_oracle(ctx, oracle, target):
    """per-block results on random 128-bit blocks; invalid ones must report the lowest failing index"""
    rng = np.random.default_rng(21)
    blocks = rng.integers(0, 256, size=(1 << 16, 16), dtype=np.uint8)
    oo, ost = oracle.batch(target, blocks)
    good = ost == 0
    out = _gpu(ctx, target, blocks[good], bpr=1)
    assert (out == oo[good]).all()
    from basisu_rs_amd import BasisuError

    first = int(np.nonzero(~good)[0][0])
    with pytest.raises(BasisuError) as e:
        _gpu(ctx, target, blocks, bpr=256)
    assert e.value.status == int(ost[first]) and e.value.first_bad_block == first


@pytest.mark.parametrize("target", ALL)
def test_random_valid_atlas_matches_oracle(ctx, oracle, target):
    blocks = synth.atlas_rand(1 << 18, seed=4)
    oo, ost = oracle.batch(target, blocks)
    assert (ost == 0).all()
    assert (_gpu(ctx, target, blocks, bpr=512) == oo).all()


@pytest.mark.parametrize("target", ALL)
def test_high_contrast_atlas_matches_oracle(ctx, oracle, target):
    """synth.atlas_contrast: endpoints at 0 / 255, most texels at one of them -- texels far from their ETC half's base colour
    (etc.rs:160-198 with lumas beyond the i16 range of the GPU's packed selector stage), clamped modifier and EAC tables"""
    blocks = synth.atlas_contrast(1 << 19, seed=23)
    oo, ost = oracle.batch(target, blocks)
    assert (ost == 0).all()
    assert (_gpu(ctx, target, blocks, bpr=512) == oo).all()


def test_error_strings_and_first_error_semantics(ctx, golden):
    from basisu_rs_amd import BasisuError, Decoder, TargetTextureFormat

    dec = Decoder(ctx)
    with pytest.raises(BasisuError, match="data length is not divisible by UASTC block size"):
        dec.transcode(TargetTextureFormat.Bc7, bytes(33))
    e = synth.atlas_err(golden["uastc"], 4096, [3000, 77, 2048])
    with pytest.raises(BasisuError, match="block pattern is not valid") as ex:
        dec.transcode(TargetTextureFormat.Astc, e)
    assert ex.value.first_bad_block == 77
    e = synth.atlas_err(golden["uastc"], 4096, [77, 100])
    with pytest.raises(BasisuError, match="invalid mode index") as ex:
        dec.decode_to_rgba(e, 64)
    assert ex.value.first_bad_block == 77  # lowest index wins, like the sequential loop
    assert dec.transcode(TargetTextureFormat.Bc7, b"").size == 0  # empty slice is Ok(empty)


def test_rgba_image_layout_and_ragged_sizes(ctx, golden, oracle):
    for nbx, nby in ((1, 1), (3, 5), (64, 2), (65, 3), (257, 1)):
        idx = synth.gold_indices(nbx * nby, seed=nbx * 131 + nby)
        blocks = golden["uastc"][idx]
        img = ctx.decode_to_rgba(blocks, nbx)
        st, _, ref = oracle.decode_to_rgba(blocks.tobytes(), nbx)
        assert st == 0 and (img == ref).all(), (nbx, nby)
    for n in (1, 63, 64, 65, 255, 257, 1000):  # block-linear targets at ragged sizes
        blocks = golden["uastc"][synth.gold_indices(n, seed=n)]
        for t in ("bc7", "etc1"):
            assert (_gpu(ctx, t, blocks) == golden[t][synth.gold_indices(n, seed=n)]).all()


def test_full_size_atlas_4096_is_self_verifying(ctx, golden):
    """BASELINE config: 4096x4096 px = 1 048 576 blocks.  Block i is golden block h(i) mod 608, so the
    expected output is the known-answer output -- checked in full for BC7 and RGBA32."""
    n = 1 << 20
    idx = synth.gold_indices(n)
    blocks = golden["uastc"][idx]
    out = _gpu(ctx, "bc7", blocks)
    assert (out == golden["bc7"][idx]).all()
    img = ctx.decode_to_rgba(blocks, 1024).reshape(1024, 4, 1024, 16)
    lin = np.ascontiguousarray(img.transpose(0, 2, 1, 3)).reshape(n, 64)
    assert (lin == golden["rgba"][idx]).all()
    # size-independent property: transcoding is per-block, so any permutation commutes with it
    perm = np.random.default_rng(0).permutation(n)
    assert (_gpu(ctx, "bc7", blocks[perm]) == out[perm]).all()


def test_device_pointer_api_with_torch(ctx, golden):
    """bu_uastc_transcode_device on HBM-resident tensors + the status word protocol"""
    import torch

    n = 100_000
    idx = synth.gold_indices(n, seed=99)
    blocks = golden["uastc"][idx].copy()
    d_in = torch.from_numpy(blocks).cuda()
    d_out = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    d_status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(d_status)
    ctx.transcode_device(_lib.BC7, d_in, n, d_out, d_status=d_status)
    torch.cuda.synchronize()
    ctx.status_word_check(int(d_status.item()) & 0xFFFFFFFFFFFFFFFF)
    assert (d_out.cpu().numpy() == golden["bc7"][idx]).all()
    # two launches into one status word with disjoint index bases: the lowest global index is reported
    blocks[70_000, 0] = 69
    d_in = torch.from_numpy(blocks).cuda()
    ctx.status_word_reset(d_status)
    half = n // 2
    ctx.transcode_device(_lib.BC7, d_in[half:], n - half, d_out[half:], block_index_base=half, d_status=d_status)
    ctx.transcode_device(_lib.BC7, d_in[:half], half, d_out[:half], block_index_base=0, d_status=d_status)
    torch.cuda.synchronize()
    from basisu_rs_amd import BasisuError

    with pytest.raises(BasisuError, match="invalid mode index") as e:
        ctx.status_word_check(int(d_status.item()) & 0xFFFFFFFFFFFFFFFF)
    assert e.value.first_bad_block == 70_000


def test_etc1s_backend_matches_oracle(ctx, oracle):
    """config 4 shape: 512x512 blocks, 4096-entry endpoint / 8192-entry selector codebooks"""
    from basisu_rs_amd import BasisuError, etc1s_selector_from_rows

    ep, rows = synth.etc1s_codebooks(4096, 8192, seed=2)
    sel = etc1s_selector_from_rows(rows)
    nbx = nby = 512
    idx = synth.etc1s_indices(nbx * nby, 4096, 8192, seed=2)
    aidx = synth.etc1s_indices(nbx * nby, 4096, 8192, seed=7)
    assert (ctx.etc1s_transcode_to_etc1(idx, ep, sel) == oracle.etc1s_to_etc1(idx, ep, sel)).all()
    assert (ctx.etc1s_decode_to_rgba(idx, None, nbx, nby, ep, sel) == oracle.etc1s_to_rgba(idx, None, nbx, nby, ep, sel)).all()
    assert (ctx.etc1s_decode_to_rgba(idx, aidx, nbx, nby, ep, sel) == oracle.etc1s_to_rgba(idx, aidx, nbx, nby, ep, sel)).all()
    # ragged sizes
    for bx, by in ((1, 1), (5, 3), (65, 2)):
        i2 = idx[: bx * by]
        assert (ctx.etc1s_decode_to_rgba(i2, None, bx, by, ep, sel) == oracle.etc1s_to_rgba(i2, None, bx, by, ep, sel)).all()
        assert (ctx.etc1s_transcode_to_etc1(i2, ep, sel) == oracle.etc1s_to_etc1(i2, ep, sel)).all()
    # out-of-range index -> error with the lowest failing block (reference: assert!, basis_lz/mod.rs:443-445)
    bad = idx.copy()
    bad[1234] = 5000  # endpoint 5000 >= 4096
    with pytest.raises(BasisuError) as e:
        ctx.etc1s_transcode_to_etc1(bad, ep, sel)
    assert e.value.status == _lib.ERR_INDEX_RANGE and e.value.first_bad_block == 1234


def test_virtual_ranks_on_one_device_match_single_rank(ctx, golden):
    """P virtual ranks (P streams, contiguous slice ranges, gather = D2D copies into one buffer)
    must give byte-identical output to P = 1"""
    import torch

    from basisu_rs_amd import sharded

    n_slices, bps = 32, 4096  # 32 slices of 64x64 blocks
    idx = synth.gold_indices(n_slices * bps, seed=77)
    slices = torch.from_numpy(golden["uastc"][idx].reshape(n_slices, bps, 16).copy()).cuda()
    fn = sharded.gpu_transcode_fn(ctx, _lib.BC7)
    whole = torch.empty((n_slices, bps, 16), dtype=torch.uint8, device="cuda")
    assert fn(slices.reshape(-1, 16), whole.view(-1, 16), 0) == sharded._CLEAR
    assert torch.equal(sharded.transcode_array_sharded(slices, fn), whole)  # world size 1: the driver is a plain call
    assert (whole.cpu().numpy() == golden["bc7"][idx].reshape(n_slices, bps, 16)).all()
    for P in (2, 3, 8):
        full = torch.empty_like(whole)
        streams = [torch.cuda.Stream() for _ in range(P)]
        status = torch.empty(P, dtype=torch.int64, device="cuda")
        for r in range(P):
            lo, hi = sharded.partition(n_slices, P, r)
            with torch.cuda.stream(streams[r]):
                ctx.status_word_reset(status[r:r + 1], stream=streams[r])
                ctx.transcode_device(_lib.BC7, slices[lo:hi], (hi - lo) * bps, full[lo:hi], block_index_base=lo * bps,
                                     d_status=status[r:r + 1], stream=streams[r])
        torch.cuda.synchronize()
        for r in range(P):
            ctx.status_word_check(int(status[r].item()) & 0xFFFFFFFFFFFFFFFF)
        assert torch.equal(full, whole), P


# ---- whole-file API (lib.rs:20-22 read_to_*) -----------------------------------------------------------
def _images_equal(got, want):
    assert len(got) == len(want)
    for g, (w, h, stride, data) in zip(got, want):
        assert (g.w, g.h, g.stride) == (w, h, stride)
        assert g.data.tobytes() == data.tobytes()


def test_read_to_all_targets_on_a_uastc_file(ctx, golden, oracle):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb
    import basisu_rs_amd as bu

    dims = [(64, 32), (3, 5), (1, 1), (128, 96)]  # mip-chain-like sizes; the last one takes the sorted kernel
    blocks = [np.concatenate([golden["uastc"][synth.gold_indices(x * y // 2, seed=i)], synth.atlas_rand(x * y - x * y // 2, seed=i)]) for i, (x, y) in enumerate(dims)]
    f = bb.uastc_file(blocks, dims)
    fns = {"rgba": lambda b: bu.read_to_rgba(b, ctx)[1], "etc1": lambda b: bu.read_to_etc1(b, ctx), "etc2": lambda b: bu.read_to_etc2(b, ctx),
           "uastc": lambda b: bu.read_to_uastc(b, ctx), "astc": lambda b: bu.read_to_astc(b, ctx), "bc7": lambda b: bu.read_to_bc7(b, ctx)}
    for name, fn in fns.items():
        st, hdr, want = oracle.read_to(name, f)
        assert st == 0
        _images_equal(fn(f), want)
    h, _ = bu.read_to_rgba(f, ctx)
    assert h.as_list() == oracle.read_to("rgba", f)[1]
    # the same calls into one page-locked output buffer: the kernels store straight into it (no device output, no download)
    pinned = ctx.host_alloc(max(bu.read_query(t, f)[1] for t in range(6)))
    for name, fn in (("rgba", lambda o: bu.read_to_rgba(f, ctx, out=o)[1]), ("etc1", lambda o: bu.read_to_etc1(f, ctx, out=o)),
                     ("bc7", lambda o: bu.read_to_bc7(f, ctx, out=o)), ("astc", lambda o: bu.read_to_astc(f, ctx, out=o))):
        pinned[:] = 0x5A
        got = fn(pinned)
        assert got[0].data.ctypes.data == pinned.ctypes.data
        _images_equal(got, oracle.read_to(name, f)[2])
    ctx.host_free(pinned)
    # a bad block inside slice 1 aborts the whole call with the reference's message
    blocks[1] = blocks[1].copy()
    blocks[1][7, 0] = 69
    with pytest.raises(bu.BasisuError, match="invalid mode index"):
        bu.read_to_bc7(bb.uastc_file(blocks, dims), ctx)
    with pytest.raises(bu.BasisuError, match="Data CRC16 failed"):
        bu.read_to_bc7(bb.uastc_file(blocks, dims, corrupt="data_crc"), ctx)


@pytest.mark.parametrize("kw", [dict(), dict(alpha=True), dict(raw_selectors=False), dict(is_video=True), dict(grayscale=True)])
def test_read_to_on_etc1s_files(ctx, oracle, kw):
    """config 4 end to end: host BasisLZ decode + GPU codebook lookup / repack, against the oracle's whole-file path"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb
    import basisu_rs_amd as bu

    rng = np.random.default_rng(17)
    f, _, _ = bb.etc1s_file(rng, [(64, 64), (33, 17), (1, 1), (100, 80)], n_codebook=1024, **kw)
    st, hdr, want = oracle.read_to("etc1", f)
    assert st == 0
    _images_equal(bu.read_to_etc1(f, ctx), want)
    st, hdr, want = oracle.read_to("rgba", f)
    assert st == 0
    h, got = bu.read_to_rgba(f, ctx)
    _images_equal(got, want)
    assert h.as_list() == hdr
    pinned = ctx.host_alloc(bu.read_query(0, f)[1])
    _images_equal(bu.read_to_rgba(f, ctx, out=pinned)[1], want)  # page-locked output, written by the kernel directly
    ctx.host_free(pinned)
    with pytest.raises(bu.BasisuError, match="not implemented"):
        bu.read_to_bc7(f, ctx)


def test_config5_texture_array_full_size_on_device(ctx, golden):
    """BASELINE config 5 at full size: 512 slices x (1024x1024 px = 65 536 blocks) = 33 554 432 blocks, 512 MiB in,
    512 MiB out, generated and verified on the device (A-gold: expected output = known-answer output)."""
    import torch

    n = 512 * 65536
    gu = torch.from_numpy(golden["uastc"]).cuda()
    gb = torch.from_numpy(golden["bc7"]).cuda()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    idx = torch.randint(0, 608, (n,), device="cuda", generator=gen)
    d_in = gu[idx].contiguous()
    d_out = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    ctx.transcode_device(_lib.BC7, d_in, n, d_out, d_status=status)
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    ok = True
    for lo in range(0, n, 1 << 22):  # compare in pieces to bound temporary memory
        ok = ok and torch.equal(d_out[lo:lo + (1 << 22)], gb[idx[lo:lo + (1 << 22)]])
    assert ok


def test_launch_splitting_above_2_pow_26_blocks(ctx, golden):
    """the kernels index with 32 bits; the host cuts larger inputs into launches of <= 2^26 blocks.  One block past the
    limit, with an invalid block in the second piece to check that reported indices stay global."""
    import torch

    from basisu_rs_amd import BasisuError

    n = (1 << 26) + 4097
    gu = torch.from_numpy(golden["uastc"]).cuda()
    gb = torch.from_numpy(golden["bc7"]).cuda()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(6)
    idx = torch.randint(0, 608, (n,), device="cuda", generator=gen)
    d_in = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    for lo in range(0, n, 1 << 22):  # torch's row gather rejects 2^26-row launches: build the input in pieces
        d_in[lo:lo + (1 << 22)] = gu[idx[lo:lo + (1 << 22)]]
    d_out = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    ctx.transcode_device(_lib.BC7, d_in, n, d_out, d_status=status)
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    for lo in (0, (1 << 26) - (1 << 20), n - (1 << 20)):
        assert torch.equal(d_out[lo:lo + (1 << 20)], gb[idx[lo:lo + (1 << 20)]])
    bad = (1 << 26) + 77
    d_in[bad, 0] = 69
    ctx.status_word_reset(status)
    ctx.transcode_device(_lib.BC7, d_in, n, d_out, d_status=status)
    torch.cuda.synchronize()
    with pytest.raises(BasisuError, match="invalid mode index") as e:
        ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    assert e.value.first_bad_block == bad


def test_page_locked_buffers_take_the_zero_copy_path_with_identical_results(ctx, golden):
    """bu_host_alloc buffers: the kernels read and write host memory directly (no staging), on a small persistent grid.
    Same bytes as the staged path, same lowest-failing-block semantics, RGBA32 row addressing, tiny and ragged sizes."""
    from basisu_rs_amd import BasisuError

    n = (1 << 19) + 4097  # ragged tail
    idx = synth.gold_indices(n, seed=11)
    pin_in = ctx.host_alloc(n * 16)
    pin_in[:] = golden["uastc"][idx].reshape(-1)
    for t in ("bc7", "etc1", "astc", "etc2"):
        bb = golden[t].shape[1]
        pin_out = ctx.host_alloc(n * bb)
        pin_out[:] = 0xAA
        out = ctx.transcode(FMT[t], pin_in, out=pin_out)
        assert out.ctypes.data == pin_out.ctypes.data
        assert (out.reshape(n, bb) == golden[t][idx]).all(), t
        ctx.host_free(pin_out)
    for m in (1, 63, 2047, 2049):  # below / around the sorted-kernel threshold
        pin_out = ctx.host_alloc(m * 16)
        assert (ctx.transcode(FMT["bc7"], pin_in[: m * 16], out=pin_out).reshape(m, 16) == golden["bc7"][idx[:m]]).all(), m
        ctx.host_free(pin_out)
    # RGBA32 with a row pitch that is not a multiple of anything convenient
    nbx, nby = 1000, 525
    m = nbx * nby
    pin_rgba = ctx.host_alloc(m * 64)
    img = ctx.decode_to_rgba(pin_in[: m * 16], nbx, out=pin_rgba).reshape(nby, 4, nbx, 16)
    lin = np.ascontiguousarray(img.transpose(0, 2, 1, 3)).reshape(m, 64)
    assert (lin == golden["rgba"][idx[:m]]).all()
    ctx.host_free(pin_rgba)
    # mixed: only one side page-locked (the other side is staged)
    pageable_out = ctx.transcode(FMT["bc7"], pin_in)
    assert (pageable_out.reshape(n, 16) == golden["bc7"][idx]).all()
    pin_out = ctx.host_alloc(n * 16)
    assert (ctx.transcode(FMT["bc7"], golden["uastc"][idx], out=pin_out).reshape(n, 16) == golden["bc7"][idx]).all()
    ctx.host_free(pin_out)
    with pytest.raises(BasisuError, match="invalid argument"):  # ragged last block row: rejected as on the staged path
        ctx.decode_to_rgba(pin_in[: (1000 * 3 + 17) * 16], 1000)
    # errors far apart (different workgroups): the lowest block index is reported, as the reference's sequential loop would
    bad = synth.atlas_err(golden["uastc"], n, [400000, 140001, 300000])
    pin_in[:] = bad.reshape(-1)
    pin_out = ctx.host_alloc(n * 16)
    with pytest.raises(BasisuError) as ex:
        ctx.transcode(FMT["bc7"], pin_in, out=pin_out)
    assert ex.value.first_bad_block == 140001
    ctx.host_free(pin_out)
    ctx.host_free(pin_in)


def test_read_to_on_a_many_slice_etc1s_file_decodes_slices_concurrently(ctx, oracle):
    """24 576 blocks in 6 slices (+ alpha): enough for bu_read_to to spread the BasisLZ decode over host threads;
    the result must be that of the oracle's sequential whole-file path"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb
    import basisu_rs_amd as bu

    f, _, _ = bb.etc1s_file(np.random.default_rng(23), [(64, 64)] * 6, n_codebook=2048, alpha=True)
    st, hdr, want = oracle.read_to("rgba", f)
    assert st == 0 and len(want) == 6
    h, got = bu.read_to_rgba(f, ctx)
    _images_equal(got, want)
    st, _, want = oracle.read_to("etc1", f)
    assert st == 0
    _images_equal(bu.read_to_etc1(f, ctx), want)
    # a corrupted symbol stream in slice 3 fails the whole call, as the sequential loop would
    sd = bu.read_slice_descs(f)
    g = bytearray(f)
    for k in range(sd[3].file_ofs, sd[3].file_ofs + min(64, sd[3].file_size)):
        g[k] ^= 0xFF
    g = bb.reseal(bytes(g))
    st_o = oracle.read_to("rgba", g)[0]
    assert st_o != 0  # 64 inverted bytes do not survive the symbol decoder
    with pytest.raises(bu.BasisuError):
        bu.read_to_rgba(g, ctx)


def test_large_uastc_file_checks_the_payload_crc_beside_the_upload(ctx, golden, oracle):
    """files >= 1 MiB: bu_read_to verifies the payload CRC on host threads while the slice uploads; results and error
    precedence must be those of the reference's order of checks (CRC before anything behind the header)"""
    import basisu_rs_amd as bu

    nbx, nby = 320, 300  # 96 000 blocks = 1.5 MB
    idx = synth.gold_indices(nbx * nby, seed=5)
    blocks = golden["uastc"][idx]
    f = bu.write_uastc_file([dict(data=blocks, orig_w=4 * nbx, orig_h=4 * nby, nbx=nbx, nby=nby)])
    assert len(f) >= 1 << 20
    imgs = bu.read_to_bc7(f, ctx)
    assert (np.asarray(imgs[0].data).reshape(-1, 16) == golden["bc7"][idx]).all()
    _images_equal(bu.read_to_rgba(f, ctx)[1], oracle.read_to("rgba", f)[2])
    g = bytearray(f)
    g[len(g) // 2] ^= 0x10  # payload damage, CRC not re-sealed
    with pytest.raises(bu.BasisuError, match="Data CRC16 failed"):
        bu.read_to_bc7(bytes(g), ctx)
    # payload damage that also makes a block invalid: the CRC failure is what the reference would report
    g = bytearray(f)
    ofs = bu.read_slice_descs(f)[0].file_ofs
    g[ofs + 16 * 1000] = (g[ofs + 16 * 1000] & 0x80) | 69
    with pytest.raises(bu.BasisuError, match="Data CRC16 failed"):
        bu.read_to_astc(bytes(g), ctx)
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb

    with pytest.raises(bu.BasisuError, match="invalid mode index"):  # CRC re-sealed: now the block error surfaces
        bu.read_to_astc(bb.reseal(bytes(g)), ctx)


@pytest.mark.parametrize("target", ["bc7", "astc", "etc1", "etc2"])
def test_large_configuration_with_a_ragged_last_tile(ctx, golden, target):
    """the two-workgroups-per-CU configuration (>= 2^19 blocks) on a size that is not a multiple of its 2048-block
    tile, with an invalid block in the ragged tail: every block transcoded, lowest failing index reported"""
    import torch

    from basisu_rs_amd import BasisuError

    n = (1 << 19) + 2048 * 3 + 777
    idx = synth.gold_indices(n, seed=77)
    blocks = golden["uastc"][idx].copy()
    d_in = torch.from_numpy(blocks).cuda()
    bb_out = golden[target].shape[1]
    d_out = torch.full((n + 64, bb_out), 0xCD, dtype=torch.uint8, device="cuda")  # 64 guard rows behind the output
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    ctx.transcode_device(FMT[target], d_in, n, d_out, d_status=status)
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    got = d_out.cpu().numpy()
    assert (got[:n] == golden[target][idx]).all()
    assert (got[n:] == 0xCD).all()  # nothing written past the last block
    blocks[n - 5, 0] = 69
    blocks[n - 300, 0] = 69
    d_in = torch.from_numpy(blocks).cuda()
    ctx.status_word_reset(status)
    ctx.transcode_device(FMT[target], d_in, n, d_out, d_status=status)
    torch.cuda.synchronize()
    with pytest.raises(BasisuError, match="invalid mode index") as e:
        ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    assert e.value.first_bad_block == n - 300


def test_read_to_merges_back_to_back_slices_and_keeps_gapped_ones_apart(ctx, golden, oracle):
    """bu_read_to uploads and (for block-linear targets) launches once per run of slices that sit back to back in the
    file.  Mixed layout: runs {0,1,2}, {3,4} behind a 7-byte gap (unaligned file offset), {5} behind a 16-byte gap,
    an empty slice, and a 40-slice array; results and error reporting must be those of the slice-by-slice oracle."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb
    import basisu_rs_amd as bu

    dims = [(40, 30), (20, 15), (10, 8), (64, 64), (32, 32), (5, 3), (0, 0)] + [(32, 32)] * 40
    pads = [0, 0, 0, 7, 0, 16, 0] + [0] * 40
    blocks = [np.concatenate([golden["uastc"][synth.gold_indices(x * y // 2, seed=i)], synth.atlas_rand(x * y - x * y // 2, seed=i)]).reshape(-1, 16)
              for i, (x, y) in enumerate(dims)]
    f = bb.uastc_file(blocks, dims, pads=pads)
    fns = {"rgba": lambda b: bu.read_to_rgba(b, ctx)[1], "etc1": lambda b: bu.read_to_etc1(b, ctx), "etc2": lambda b: bu.read_to_etc2(b, ctx),
           "astc": lambda b: bu.read_to_astc(b, ctx), "bc7": lambda b: bu.read_to_bc7(b, ctx)}
    for name, fn in fns.items():
        st, _, want = oracle.read_to(name, f)
        assert st == 0 and len(want) == len(dims)
        _images_equal(fn(f), want)
    # errors: a bad pattern in slice 4 (second run) and a bad mode in slice 20 (array run): slice 4's error is reported
    blocks[20] = blocks[20].copy()
    blocks[20][5, 0] = 69
    bad4 = blocks[4].copy()
    bad4[100] = synth.atlas_err(golden["uastc"], 2, [0, 1])[1]  # mode 3 with an out-of-range pattern
    with pytest.raises(bu.BasisuError, match="invalid mode index"):
        bu.read_to_bc7(bb.uastc_file(blocks, dims, pads=pads), ctx)
    blocks[4] = bad4
    g = bb.uastc_file(blocks, dims, pads=pads)
    assert oracle.read_to("bc7", g)[0] != 0
    with pytest.raises(bu.BasisuError, match="block pattern is not valid"):
        bu.read_to_bc7(g, ctx)
    with pytest.raises(bu.BasisuError, match="block pattern is not valid"):
        bu.read_to_rgba(g, ctx)


def test_read_to_pieced_two_stream_pipeline_on_a_mapped_output(ctx, golden, monkeypatch):
    """large runs with a page-locked output are uploaded and transcoded in pieces on two streams (16 MiB pieces by
    default; 1 MiB here to exercise it on a small file): same bytes, lowest failing block across pieces"""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb
    import basisu_rs_amd as bu

    monkeypatch.setenv("BU_RUN_PIECE_MIB", "1")
    dims = [(256, 256)] * 3 + [(128, 37)]  # 3 x 1 MiB + 74 KiB: one run of 3.07 MiB -> 4 pieces, the last one ragged
    idx = [synth.gold_indices(x * y, seed=90 + i) for i, (x, y) in enumerate(dims)]
    blocks = [golden["uastc"][ix].copy() for ix in idx]
    f = bb.uastc_file(blocks, dims)
    for name, fn, tgt in (("bc7", bu.read_to_bc7, _lib.READ_BC7), ("etc1", bu.read_to_etc1, _lib.READ_ETC1), ("astc", bu.read_to_astc, _lib.READ_ASTC)):
        pinned = ctx.host_alloc(bu.read_query(tgt, f)[1])
        pinned[:] = 0x77
        imgs = fn(f, ctx, out=pinned)
        for k in range(len(dims)):
            bb_out = golden[name].shape[1]
            assert (np.asarray(imgs[k].data).reshape(-1, bb_out) == golden[name][idx[k]]).all(), (name, k)
        ctx.host_free(pinned)
    blocks[2][40000, 0] = 69  # third piece
    blocks[1][65535, 0] = 69  # end of the second piece: the lower global index
    bad = synth.atlas_err(golden["uastc"], 2, [0, 1])[1]
    blocks[2][50000] = bad
    pinned = ctx.host_alloc(bu.read_query(_lib.READ_BC7, f)[1])
    with pytest.raises(bu.BasisuError, match="invalid mode index"):
        bu.read_to_bc7(bb.uastc_file(blocks, dims), ctx, out=pinned)
    blocks[1][65535] = golden["uastc"][idx[1][65535]]
    blocks[2][40000] = golden["uastc"][idx[2][40000]]
    with pytest.raises(bu.BasisuError, match="block pattern is not valid"):
        bu.read_to_bc7(bb.uastc_file(blocks, dims), ctx, out=pinned)
    ctx.host_free(pinned)


def test_host_threads_share_a_context_and_use_their_own(ctx, golden):
    """the reference's functions are pure and re-entrant: concurrent callers on one context serialise on
    its staging buffers, separate contexts run side by side; every result must be right"""
    import threading

    from basisu_rs_amd import Context

    own = [Context(0) for _ in range(2)]
    errors = []

    def worker(c, seed):
        try:
            for it in range(6):
                n = 3000 + 517 * ((seed + it) % 7)
                idx = synth.gold_indices(n, seed=1000 * seed + it)
                blocks = golden["uastc"][idx]
                t = ("bc7", "astc", "etc1", "etc2")[(seed + it) % 4]
                got = c.transcode(FMT[t], blocks).reshape(n, -1)
                if not (got == golden[t][idx]).all():
                    errors.append((seed, it, t))
                img = c.decode_to_rgba(blocks[: 64 * 40], 64).reshape(40, 4, 64, 16)
                lin = np.ascontiguousarray(img.transpose(0, 2, 1, 3)).reshape(-1, 64)
                if not (lin == golden["rgba"][idx[: 64 * 40]]).all():
                    errors.append((seed, it, "rgba"))
        except Exception as e:  # noqa: BLE001
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(ctx, s)) for s in range(4)]
    threads += [threading.Thread(target=worker, args=(own[s % 2], 10 + s)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in own:
        c.close()
    assert not errors, errors[:5]
