"""Round-3 GPU tests: the store-policy canary (the shipped `sc1 nt` inline-asm stores against a build that lets the compiler emit
every store), the timed-window helper of bench.py, BASELINE config 4 at its stated size through the whole-file API, and both
sides of the rectangular-tile selection.  Everything goes through the C ABI; bit-exact.  Run on the GPU box: pytest -m gpu."""
import ctypes
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from basisu_rs_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CANARY_CHILD = r'''
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from basisu_rs_amd import Context, _lib, synth
g = synth.load_golden(os.path.join(%(root)r, "tests", "golden", "uastc_kat.bin"))
n = 1 << 20
idx = synth.gold_indices(n, seed=77)
blocks = g["uastc"][idx].copy()
blocks[12345] = 0
blocks[12345, 0] = 69  # the one invalid 7-bit mode code (uastc.rs:560-577): one failing block: the status word and the zeroed result travel the same stores
ctx = Context(0)
d_in = torch.from_numpy(blocks).cuda()
out = {}
for name, t, bb in (("astc", _lib.ASTC, 16), ("bc7", _lib.BC7, 16), ("etc1", _lib.ETC1, 8), ("etc2", _lib.ETC2, 16), ("rgba", _lib.RGBA32, 64)):
    d_out = torch.zeros((n, bb), dtype=torch.uint8, device="cuda")
    st = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(st)
    ctx.transcode_device(t, d_in, n, d_out, blocks_per_row=1024, d_status=st)
    torch.cuda.synchronize()
    out[name] = [hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest(), int(st.item()) & 0xFFFFFFFFFFFFFFFF]
ctx.close()
print(json.dumps(out))
'''


def test_intrinsic_store_build_matches_the_asm_store_build():
    """libbasisu_hip_st0.so (-DBU_ST_MODE=0: compiler-emitted nontemporal stores) and the shipped library (inline-asm
    `global_store_dwordx4 ... sc1 nt` + a hand-placed s_nop) must produce identical bytes and status words on 2^20 blocks x 5
    targets: the asm store's data-hazard padding is invisible to the compiler, this is the canary for a scheduling change."""
    import json

    from basisu_rs_amd import build

    st0 = build.LIB_ST0
    if not os.path.exists(st0):
        build.build_hip(canary=True)
    res = []
    for lib in (_lib.LIB_PATH, st0):
        env = dict(os.environ, BASISU_HIP_LIB=lib)
        r = subprocess.run([sys.executable, "-c", _CANARY_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
    for name in ("astc", "bc7", "etc1", "etc2", "rgba"):
        assert res[0][name][1] == (12345 << 8) | 1, (name, hex(res[0][name][1]))  # invalid mode at block 12345


def test_window_timing_helper_brackets_exactly_the_timed_launches():
    """bu_time_uastc_launches_window: the host bracket and the event pair agree, and every launch (lead + timed) really ran"""
    import torch

    from basisu_rs_amd import Context

    g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
    n, nbuf = 1 << 18, 4
    ctx = Context(0)
    lib = _lib.load()
    idx = [synth.gold_indices(n, seed=900 + k) for k in range(nbuf)]
    ins = [torch.from_numpy(g["uastc"][i]).cuda() for i in idx]
    outs = [torch.zeros((n, 16), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    A = ctypes.c_void_p * nbuf
    ev, host, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(-1)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    st = lib.bu_time_uastc_launches_window(ctx.handle, _lib.BC7, A(*[t.data_ptr() for t in ins]), A(*[t.data_ptr() for t in outs]), nbuf, 1, n, 512,
                                           64, 40, None, sp, ctypes.byref(ev), ctypes.byref(host), ctypes.byref(late))
    assert st == 0 and late.value in (0, 1)
    torch.cuda.synchronize()
    for k in range(nbuf):
        assert torch.equal(outs[k].cpu(), torch.from_numpy(g["bc7"][idx[k]]))
    assert ev.value > 0 and host.value > 0
    if not late.value:
        assert abs(host.value - ev.value) <= 0.25 * ev.value + 0.02, (host.value, ev.value)
    # argument checks
    assert lib.bu_time_uastc_launches_window(ctx.handle, _lib.BC7, A(*[t.data_ptr() for t in ins]), A(*[t.data_ptr() for t in outs]), nbuf, 0, n, 512,
                                             -1, 4, None, sp, ctypes.byref(ev), ctypes.byref(host), None) == _lib.ERR_ARGUMENT
    ctx.close()


@pytest.mark.parametrize("target", ["astc", "bc7", "rgba", "etc1", "etc2"])
def test_rectangular_tiles_on_both_sides_of_the_selection(golden, target):
    """the launcher takes the RECT kernels when blocks_per_row is a multiple of 64 (at least 128) and the slice is whole rows of
    64 x 16-block tiles (ETC1 / ETC2: of 64 x 64-block tiles, on the 4096-block shape of large slices), the strip kernels otherwise:
    same bytes either way, and the same first-error index (uastc.rs:157-165)"""
    import torch

    from basisu_rs_amd import BasisuError, Context

    ctx = Context(0)
    t, bb = {"astc": (_lib.ASTC, 16), "bc7": (_lib.BC7, 16), "rgba": (_lib.RGBA32, 64), "etc1": (_lib.ETC1, 8), "etc2": (_lib.ETC2, 16)}[target]
    # (blocks per row, block rows): rect 1 and 2 tiles per row, several tile rows, one big enough for every launch shape;
    # strip: ragged rows of tiles, width not a multiple of 32, and the block-linear call without a grid (blocks_per_row 0)
    shapes = [(128, 16, True), (128, 64, True), (192, 32, True), (1024, 512, True), (1024, 1024, True), (128, 63, False), (64, 64, False),
              (96, 64, False), (0, 4096, False)]
    if target in ("etc1", "etc2"):  # the 4096-block shape starts above 3 Ki blocks per CU; 1088 rows balance to 2176-block tiles (strips)
        shapes += [(2048, 1024, True), (512, 2048, True), (1024, 1088, False), (1024, 1056, False)]
    for bpr, rows, rect in shapes:
        n = (bpr or 1) * rows
        idx = synth.gold_indices(n, seed=31 + bpr + rows)
        blocks = golden["uastc"][idx].copy()
        d_in = torch.from_numpy(blocks).cuda()
        if target == "rgba":
            if bpr == 0:
                continue
            d_out = torch.zeros((rows * 4, bpr * 16), dtype=torch.uint8, device="cuda")
        else:
            d_out = torch.zeros((n, bb), dtype=torch.uint8, device="cuda")
        ctx.transcode_device(t, d_in, n, d_out, blocks_per_row=bpr)
        torch.cuda.synchronize()
        if target == "rgba":
            got = d_out.cpu().numpy().reshape(rows, 4, bpr, 16).transpose(0, 2, 1, 3).reshape(n, 64)
        else:
            got = d_out.cpu().numpy()
        assert (got == golden[target][idx]).all(), (target, bpr, rows, rect)
        # two failing blocks: the lower index is reported whichever tile (or strip) holds it
        bad_hi, bad_lo = n - 3, (n * 5) // 8 + 7
        blocks[bad_hi, 0] = 69  # the one invalid 7-bit mode code (uastc.rs:560-577)
        blocks[bad_lo, 0] = 69
        d_in = torch.from_numpy(blocks).cuda()
        st = torch.empty(1, dtype=torch.int64, device="cuda")
        ctx.status_word_reset(st)
        ctx.transcode_device(t, d_in, n, d_out, blocks_per_row=bpr, d_status=st)
        torch.cuda.synchronize()
        with pytest.raises(BasisuError) as e:
            ctx.status_word_check(int(st.item()))
        assert e.value.status == _lib.ERR_INVALID_MODE and e.value.first_bad_block == bad_lo, (target, bpr, rows)
    ctx.close()


@pytest.mark.parametrize("alpha", [False, True])
def test_config4_one_512x512_block_slice_end_to_end(ctx, oracle, alpha):
    """BASELINE config 4 at its stated size: ONE ETC1S slice of 2048 x 2048 px (512 x 512 blocks; with `alpha` a colour + alpha
    slice pair) in a .basis file, through the whole-file API -- host BasisLZ decode of 262 144 blocks against 4096-entry
    codebooks, GPU codebook lookup + repack -- to ETC1 and to RGBA32, against the oracle's whole-file path (basis.rs:17-70,
    102-124; basis_lz/mod.rs:97-186)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import basis_builder as bb
    import basisu_rs_amd as bu

    f, _, _ = bb.etc1s_file(np.random.default_rng(444 + alpha), [(512, 512)], n_codebook=4096, alpha=alpha)
    for target, read in (("etc1", bu.read_to_etc1), ("rgba", bu.read_to_rgba)):
        st, hdr, want = oracle.read_to(target, f)
        # (read_to_etc1 returns every slice as an image: colour and alpha slice of a pair alike, basis.rs:102-124; read_to_rgba
        # folds the pair into one image, basis.rs:17-70)
        assert st == 0 and len(want) == (2 if alpha and target == "etc1" else 1)
        res = read(f, ctx)
        got = res[1] if target == "rgba" else res
        assert len(got) == len(want)
        for g, (w, h, stride, data) in zip(got, want):
            assert (g.w, g.h, g.stride) == (w, h, stride) and (w, h) == (2048, 2048)
            assert g.data.tobytes() == data.tobytes(), (target, alpha)
    # the same through a page-locked output buffer (the kernel writes host memory directly)
    pinned = ctx.host_alloc(bu.read_query(0, f)[1])
    got = bu.read_to_rgba(f, ctx, out=pinned)[1]
    assert got[0].data.tobytes() == want[0][3].tobytes()
    ctx.host_free(pinned)


@pytest.mark.parametrize("n_ep,n_sel", [(4096, 8192), (16128, 16128)])
def test_etc1s_kernels_with_codebooks_staged_in_lds(ctx, oracle, n_ep, n_sel):
    """slices of 2^19 blocks and more run the kernels that stage both codebooks in LDS (bu_etc1s_staged_kernel): config-4
    codebooks (48 KiB) and the largest a .basis file can carry (126 KiB: dynamic LDS above the 64 KiB default), with and
    without an alpha slice, one block past a multiple of the grid stride, and the lowest failing block of two
    (basis_lz/mod.rs:122-181, 443-445)"""
    from basisu_rs_amd import BasisuError, etc1s_selector_from_rows

    ep, rows = synth.etc1s_codebooks(n_ep, n_sel, seed=3)
    sel = etc1s_selector_from_rows(rows)
    nbx, nby = 1024, 513  # 525 312 blocks
    n = nbx * nby
    idx = synth.etc1s_indices(n, n_ep, n_sel, seed=21)
    aidx = synth.etc1s_indices(n, n_ep, n_sel, seed=22)
    assert (ctx.etc1s_transcode_to_etc1(idx, ep, sel) == oracle.etc1s_to_etc1(idx, ep, sel)).all()
    assert (ctx.etc1s_decode_to_rgba(idx, None, nbx, nby, ep, sel) == oracle.etc1s_to_rgba(idx, None, nbx, nby, ep, sel)).all()
    assert (ctx.etc1s_decode_to_rgba(idx, aidx, nbx, nby, ep, sel) == oracle.etc1s_to_rgba(idx, aidx, nbx, nby, ep, sel)).all()
    bad = idx.copy()
    bad[n - 5] = n_ep            # endpoint index one past the codebook
    bad[400_000] = (n_sel << 16)  # selector index one past the codebook: the LOWER block is the one reported
    for call in (lambda: ctx.etc1s_transcode_to_etc1(bad, ep, sel), lambda: ctx.etc1s_decode_to_rgba(idx, bad, nbx, nby, ep, sel)):
        with pytest.raises(BasisuError) as e:
            call()
        assert e.value.status == _lib.ERR_INDEX_RANGE and e.value.first_bad_block == 400_000


def test_staged_etc1_kernel_on_ragged_sizes_and_unaligned_arrays(ctx, oracle):
    """the staged ETC1 kernel walks 256-block chunks with 8-byte index loads and 16-byte stores when the arrays are aligned for
    them and block by block otherwise: sizes that are not multiples of 256, an index array on a 4-byte boundary, an output on an
    8-byte boundary -- device entry point, same bytes as the oracle (basis_lz/mod.rs:163-181)"""
    import torch

    from basisu_rs_amd import etc1s_selector_from_rows

    lib = _lib.load()
    n_ep, n_sel = 4096, 8192
    ep, rows = synth.etc1s_codebooks(n_ep, n_sel, seed=5)
    sel = etc1s_selector_from_rows(rows)
    d_ep = torch.from_numpy(ep.view(np.int32)).cuda()
    d_sel = torch.from_numpy(sel).cuda()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n, idx_ofs, out_ofs in (((1 << 19) + 77, 0, 0), ((1 << 19) + 255, 1, 0), ((1 << 19) + 256, 0, 1), ((1 << 20) + 3, 1, 1), (1 << 19, 2, 2)):
        idx = synth.etc1s_indices(n, n_ep, n_sel, seed=40 + n % 97)
        want = oracle.etc1s_to_etc1(idx, ep, sel)
        d_idx = torch.zeros(n + 4, dtype=torch.int32, device="cuda")
        d_idx[idx_ofs:idx_ofs + n] = torch.from_numpy(idx.view(np.int32)).cuda()
        d_out = torch.full(((n + 4) * 8,), 0xA5, dtype=torch.uint8, device="cuda")
        assert lib.bu_etc1s_transcode_etc1_device(ctx.handle, d_idx.data_ptr() + 4 * idx_ofs, n, d_ep.data_ptr(), n_ep, d_sel.data_ptr(), n_sel,
                                                  d_out.data_ptr() + 8 * out_ofs, None, sp) == 0
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert (got[8 * out_ofs:8 * (out_ofs + n)].reshape(n, 8) == want.reshape(n, 8)).all(), (n, idx_ofs, out_ofs)
        assert (got[:8 * out_ofs] == 0xA5).all() and (got[8 * (out_ofs + n):] == 0xA5).all(), "wrote outside the result"


def test_batch_entry_point_merges_contiguous_slices_and_fans_out_the_rest(ctx, golden):
    """bu_uastc_transcode_batch_device: a loop over independent slices in one call.  Contiguous slices (one launch), slices in
    separate allocations (ONE launch per 96 runs on the caller's stream, the run table in the kernel arguments), a mix with an empty slice, RGBA32 slices of one pitch -- same bytes
    as slice-by-slice calls, block errors numbered through the whole batch (basis.rs:246-257)."""
    import torch

    from basisu_rs_amd import BasisuError

    lib = _lib.load()
    sizes = [4096, 65536, 0, 1024, 70000, 8]
    idx = [synth.gold_indices(n, seed=500 + k) for k, n in enumerate(sizes)]
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_s = len(sizes)
    VP, SZ = ctypes.c_void_p * n_s, ctypes.c_size_t * n_s

    def run(target, bb, ins, outs, bpr=0, status=None):
        st = lib.bu_uastc_transcode_batch_device(ctx.handle, target, n_s, VP(*[t.data_ptr() if t.numel() else None for t in ins]), SZ(*sizes),
                                                 VP(*[t.data_ptr() if t.numel() else None for t in outs]), bpr, None,
                                                 ctypes.c_void_p(status.data_ptr()) if status is not None else None, sp)
        assert st == 0
        torch.cuda.synchronize()

    # (a) separate allocations
    ins = [torch.from_numpy(golden["uastc"][i].copy()).cuda() if len(i) else torch.empty((0, 16), dtype=torch.uint8, device="cuda") for i in idx]
    for name, target, bb in (("bc7", _lib.BC7, 16), ("etc1", _lib.ETC1, 8)):
        outs = [torch.zeros((n, bb), dtype=torch.uint8, device="cuda") for n in sizes]
        run(target, bb, ins, outs)
        for k in range(n_s):
            assert (outs[k].cpu().numpy() == golden[name][idx[k]]).all(), (name, k)
    # (b) one contiguous buffer cut into the same slices: merged into a single launch
    cat_in = torch.cat(ins)
    cat_out = torch.zeros((sum(sizes), 16), dtype=torch.uint8, device="cuda")
    ofs = np.concatenate([[0], np.cumsum(sizes)])
    run(_lib.BC7, 16, [cat_in[ofs[k]:ofs[k + 1]] for k in range(n_s)], [cat_out[ofs[k]:ofs[k + 1]] for k in range(n_s)])
    assert (cat_out.cpu().numpy() == golden["bc7"][np.concatenate(idx)]).all()
    # (c) a failing block in slice 4 and one in slice 1: the lower batch-wide index is reported
    bad = [t.clone() for t in ins]
    bad[4][17, 0] = 69
    bad[1][60000, 0] = 69
    st = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(st)
    run(_lib.BC7, 16, bad, [torch.zeros((n, 16), dtype=torch.uint8, device="cuda") for n in sizes], status=st)
    with pytest.raises(BasisuError) as e:
        ctx.status_word_check(int(st.item()))
    assert e.value.first_bad_block == 4096 + 60000
    # (c2) the same through a target that keeps its statuses in a byte per block (ETC2), and a batch of more tiles than CUs
    st.fill_(0)
    ctx.status_word_reset(st)
    run(_lib.ETC2, 16, bad, [torch.zeros((n, 16), dtype=torch.uint8, device="cuda") for n in sizes], status=st)
    with pytest.raises(BasisuError) as e:
        ctx.status_word_check(int(st.item()))
    assert e.value.first_bad_block == 4096 + 60000
    many = 300  # 300 slices of 3 000 blocks = 900 tiles, every slice in its own allocation (tiles of 1024, 1024, 952 blocks)
    m_idx = [synth.gold_indices(3000, seed=9000 + k) for k in range(many)]
    m_in = [torch.from_numpy(golden["uastc"][i].copy()).cuda() for i in m_idx]
    m_out = [torch.zeros((3000, 16), dtype=torch.uint8, device="cuda") for _ in range(many)]
    VPm, SZm = ctypes.c_void_p * many, ctypes.c_size_t * many
    assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.ASTC, many, VPm(*[t.data_ptr() for t in m_in]), SZm(*([3000] * many)),
                                               VPm(*[t.data_ptr() for t in m_out]), 0, None, None, sp) == 0
    torch.cuda.synchronize()
    for k in range(many):
        assert (m_out[k].cpu().numpy() == golden["astc"][m_idx[k]]).all(), k
    # (d) RGBA32: whole block rows of one pitch per slice
    bpr = 8
    r_out = [torch.zeros((n // bpr * 4, bpr * 16), dtype=torch.uint8, device="cuda") for n in sizes]
    run(_lib.RGBA32, 64, ins, r_out, bpr=bpr)
    for k, n in enumerate(sizes):
        if n:
            got = r_out[k].cpu().numpy().reshape(n // bpr, 4, bpr, 16).transpose(0, 2, 1, 3).reshape(n, 64)
            assert (got == golden["rgba"][idx[k]]).all(), k


@pytest.mark.gpu
def test_batch_of_separate_allocations_replays_from_a_hip_graph(ctx, golden):
    """The multi-run launch carries its run table in the kernel arguments (no upload, no stream-ordered allocation), so a batch over
    slices in separate allocations is capturable like the single-slice entry point: capture, change the inputs, replay, compare."""
    import torch

    lib = _lib.load()
    sizes = [3000, 1024, 20000, 64, 5000]
    n_s = len(sizes)
    VP, SZ = ctypes.c_void_p * n_s, ctypes.c_size_t * n_s
    gu = torch.from_numpy(golden["uastc"]).cuda()
    ins = [torch.empty((n, 16), dtype=torch.uint8, device="cuda") for n in sizes]
    outs = [torch.empty((n, 16), dtype=torch.uint8, device="cuda") for n in sizes]
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()

    def fill(seed):
        idx = [torch.from_numpy(synth.gold_indices(n, seed=seed + k)).cuda() for k, n in enumerate(sizes)]
        for t, i in zip(ins, idx):
            t.copy_(gu[i])
        return idx

    def call():
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.ASTC, n_s, VP(*[t.data_ptr() for t in ins]), SZ(*sizes), VP(*[t.data_ptr() for t in outs]), 0,
                                                   None, ctypes.c_void_p(status.data_ptr()), ctypes.c_void_p(s.cuda_stream)) == 0

    fill(900)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ctx.status_word_reset(status, stream=s)
        call()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.status_word_reset(status, stream=s)
        call()
    for seed in (910, 920):
        idx = fill(seed)
        for t in outs:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
        for k in range(n_s):
            assert (outs[k].cpu().numpy() == golden["astc"][idx[k].cpu().numpy()]).all(), k
    ins[2][123, 0] = 69  # block 3000 + 1024 + 123 of the batch
    g.replay()
    torch.cuda.synchronize()
    assert (int(status.item()) & 0xFFFFFFFFFFFFFFFF) >> 8 == 3000 + 1024 + 123
