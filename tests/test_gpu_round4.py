"""Round 4: the streamed ETC1S front door against the one-launch path and the oracle (results and error order), the sharded
call's lock order, the BC7 big-shape launcher on both sides of its conditions.  Everything goes through the C ABI."""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest

from basisu_rs_amd import _lib, synth

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _read_all(bu, ctx, f, one_launch, one_thread=False):
    """(status or None, images) of read_to_rgba / read_to_etc1 with the streamed front door on or off (and, when on, with the first
    slice decoded on one thread instead of two)"""
    from basisu_rs_amd import BasisuError

    out = {}
    if one_launch:
        os.environ["BU_ETC1S_ONE_LAUNCH"] = "1"
    if one_thread:
        os.environ["BU_ETC1S_ONE_THREAD"] = "1"
    try:
        for name, fn in (("rgba", lambda: bu.read_to_rgba(f, ctx)[1]), ("etc1", lambda: bu.read_to_etc1(f, ctx))):
            try:
                out[name] = (0, [(g.w, g.h, g.stride, g.data.tobytes()) for g in fn()])
            except BasisuError as e:
                out[name] = (e.status, None)
    finally:
        os.environ.pop("BU_ETC1S_ONE_LAUNCH", None)
        os.environ.pop("BU_ETC1S_ONE_THREAD", None)
    return out


@pytest.mark.parametrize("dims,alpha,video", [([(256, 160)], False, False), ([(192, 192), (64, 64), (7, 5)], True, False),
                                              ([(96, 96)] * 5, False, False), ([(200, 170), (33, 31)], False, True),
                                              ([(181, 183)], True, False),
                                              # degenerate grids, large enough for the two-thread decode of the first slice: one block
                                              # column (every block is a left edge, block pairs are single blocks), one block row (no
                                              # odd rows at all: the saved predictor bits are never read), three columns (odd width)
                                              ([(1, 33000)], False, False), ([(33000, 1)], False, False), ([(3, 11001), (5, 3)], True, False),
                                              ([(1, 32800)], False, True)])
def test_streamed_etc1s_front_door_equals_one_launch_path_and_oracle(ctx, oracle, dims, alpha, video):
    """ETC1S files of 32 768 blocks and more take the streamed front door (bu_read_etc1s_streamed: tables first, codebooks / payload
    CRC / slices on pool threads, bands of finished rows launched while the rest is decoded, indices read from page-locked memory).
    Same images as the one-launch path (BU_ETC1S_ONE_LAUNCH=1) and as the oracle's whole-file path (basis.rs:8-143,
    basis_lz/mod.rs:97-186): single slice, several slices, alpha pairs, texture video, odd sizes (units that straddle rows)."""
    import basis_builder as bb
    import basisu_rs_amd as bu

    f, _, _ = bb.etc1s_file(np.random.default_rng(900 + len(dims) + 2 * alpha + video), dims, n_codebook=1024, alpha=alpha, is_video=video)
    a, b = _read_all(bu, ctx, f, False), _read_all(bu, ctx, f, True)
    assert a == b
    # a first slice of 32 768 blocks and more is decoded on two threads (slice_lex on the caller, slice_resolve on a pool thread)
    assert _read_all(bu, ctx, f, False, one_thread=True) == a
    for target in ("rgba", "etc1"):
        st, _, want = oracle.read_to(target, f)
        assert st == 0 and a[target][0] == 0
        assert a[target][1] == [(w, h, s, d.tobytes()) for (w, h, s, d) in want], target
    # page-locked output: the bands' results cross PCIe while the decode goes on
    pinned = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, f)[1])
    got = bu.read_to_rgba(f, ctx, out=pinned)[1]
    assert [(g.w, g.h, g.stride, g.data.tobytes()) for g in got] == a["rgba"][1]
    ctx.host_free(pinned)


def test_streamed_etc1s_front_door_reports_errors_in_the_reference_order(ctx, oracle):
    """Damaged files through both front doors and the oracle: the same status from all three, whatever thread found it first --
    payload CRC (basis.rs:338-341) before the endpoint / selector codebooks (basis_lz/mod.rs:69-76) before the tables (:77-83)
    before the first failing slice in file order; single-bit damage at 120 places of a resealed file (CRCs recomputed, so the
    damage reaches the decoders) and unsealed damage (the CRC must win)."""
    import basis_builder as bb
    import basisu_rs_amd as bu

    # (the first image's colour and alpha slices are large enough for the two-thread decode: its give-up path -- whichever half
    # finishes second decodes the slice again with the exact loop -- is what turns damage inside them into the reference's status)
    f, _, _ = bb.etc1s_file(np.random.default_rng(77), [(256, 160), (128, 96)], n_codebook=512, history_size=16, alpha=True)
    hdr = bu.read_header(f)
    rng = np.random.default_rng(5)
    spots = [hdr.endpoint_cb_file_ofs + 3, hdr.selector_cb_file_ofs + 1, hdr.tables_file_ofs + 2, hdr.tables_file_ofs + hdr.tables_file_size - 2]
    spots += [int(x) for x in rng.integers(hdr.endpoint_cb_file_ofs, len(f), 116)]
    seen = set()
    for k, pos in enumerate(spots):
        g = bytearray(f)
        g[pos] ^= 1 << (k % 8)
        for sealed in (True, False):
            h = bb.reseal(bytes(g)) if sealed else bytes(g)
            a, b = _read_all(bu, ctx, h, False), _read_all(bu, ctx, h, True)
            for target in ("rgba", "etc1"):
                st = oracle.read_to(target, h)[0]
                assert a[target][0] == b[target][0] == st, (pos, sealed, target, a[target][0], b[target][0], st)
                if st == 0:
                    assert a[target][1] == b[target][1]
                seen.add(st)
    assert len(seen) >= 3  # success, the CRC and at least one decoder error all occurred


@pytest.mark.timeout(600, method="thread")
def test_sharded_calls_over_the_same_contexts_in_opposite_orders_do_not_deadlock(golden):
    """bu_array_transcode_sharded locks every context it is given.  Two threads that list the same contexts in opposite orders
    used to be an A-then-B against B-then-A deadlock; the locks are now taken in one canonical order (by address)."""
    import torch

    from basisu_rs_amd import Context, sharded

    lib = _lib.load()
    n_slices, bps, n_ctx = 12, 1024, 3
    idx = synth.gold_indices(n_slices * bps, seed=91)
    blocks = golden["uastc"][idx]
    want = torch.from_numpy(golden["bc7"][idx]).cuda()
    ctxs = [Context(0) for _ in range(n_ctx)]
    PA = lambda vals: (ctypes.c_void_p * len(vals))(*vals)
    try:
        def setup(order):
            ins, fulls = [], []
            for r in range(n_ctx):
                lo, hi = sharded.partition(n_slices, n_ctx, r)
                ins.append(torch.from_numpy(blocks[lo * bps:hi * bps].copy()).cuda())
                fulls.append(torch.zeros((n_slices * bps, 16), dtype=torch.uint8, device="cuda"))
            return PA([ctxs[i].handle.value for i in order]), ins, fulls

        jobs = [setup([0, 1, 2]), setup([2, 1, 0])]
        torch.cuda.synchronize()
        errs = []

        def run(j):
            handles, ins, fulls = jobs[j]
            bad = ctypes.c_uint64(0)
            for _ in range(200):
                st = lib.bu_array_transcode_sharded(handles, n_ctx, _lib.BC7, PA([t.data_ptr() for t in ins]), n_slices, bps, PA([t.data_ptr() for t in fulls]), 1,
                                                    ctypes.byref(bad))
                if st != 0:
                    errs.append(st)
                    return

        th = [threading.Thread(target=run, args=(j,)) for j in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=240)
        assert not any(t.is_alive() for t in th), "the two sharded calls are stuck on each other's context locks"
        assert not errs
        torch.cuda.synchronize()
        for _, _, fulls in jobs:
            for full in fulls:
                assert torch.equal(full, want)
    finally:
        for c in ctxs:
            c.close()


def test_batch_entry_point_accepts_runs_of_any_length_beside_others(ctx, golden):
    """bu_uastc_transcode_batch_device used to refuse a batch of two or more runs as soon as one of them exceeded the run table's
    32-bit fields, although the same slice alone was accepted.  Here: the argument path only (a 2^32-block slice does not fit a
    test) -- a long run (4 Mi blocks: its own launch pieces) beside short ones in separate allocations, results as slice by slice."""
    import torch

    lib = _lib.load()
    sizes = [1 << 22, 4096, 70000]
    ins, outs, wants = [], [], []
    for k, n in enumerate(sizes):
        idx = synth.gold_indices(n, seed=300 + k)
        ins.append(torch.from_numpy(golden["uastc"][idx]).cuda())
        outs.append(torch.zeros((n, 16), dtype=torch.uint8, device="cuda"))
        wants.append(torch.from_numpy(golden["bc7"][idx]).cuda())
    VP, SZ = ctypes.c_void_p * 3, ctypes.c_size_t * 3
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    st = lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, 3, VP(*[t.data_ptr() for t in ins]), SZ(*sizes), VP(*[t.data_ptr() for t in outs]), 256, None, None, sp)
    assert st == 0
    torch.cuda.synchronize()
    for o, w in zip(outs, wants):
        assert torch.equal(o, w)


def test_device_entry_points_can_be_captured_in_a_hip_graph(ctx, golden):
    """bu_uastc_transcode_device, bu_uastc_transcode_batch_device and bu_status_word_reset only enqueue work on the caller's stream --
    nothing is allocated, copied from the host or synchronised behind them -- so a caller may record them into a HIP graph and
    replay it (tools/exp/graph_capture.py measures that: replaying 64 captured launches costs what launching them costs, the batch
    entry point is the fast form).  Capture, three replays on re-zeroed outputs, a bad block reported through the captured status
    word."""
    import torch

    lib = _lib.load()
    ns, nb = 4, 4096
    idx = [synth.gold_indices(nb, seed=70 + k) for k in range(ns)]
    blocks = [golden["uastc"][i].copy() for i in idx]
    blocks[2][100, 0] = 0x45  # the one invalid 7-bit mode code (69): block 2 * 4096 + 100 of the batch
    ins = [torch.from_numpy(b).cuda() for b in blocks]
    outs = [torch.zeros((nb, 16), dtype=torch.uint8, device="cuda") for _ in range(ns)]
    status = torch.zeros(1, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    def record():
        sp = ctypes.c_void_p(side.cuda_stream)
        assert lib.bu_status_word_reset(ctx.handle, ctypes.c_void_p(status.data_ptr()), sp) == 0
        for k in range(2):  # two slices as single launches, two through the batch entry point
            assert lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(ins[k].data_ptr()), nb, ctypes.c_void_p(outs[k].data_ptr()), 64, k * nb,
                                                 ctypes.c_void_p(status.data_ptr()), sp) == 0
        base = (ctypes.c_uint64 * 2)(2 * nb, 3 * nb)
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, 2, (ctypes.c_void_p * 2)(ins[2].data_ptr(), ins[3].data_ptr()),
                                                   (ctypes.c_size_t * 2)(nb, nb), (ctypes.c_void_p * 2)(outs[2].data_ptr(), outs[3].data_ptr()), 64, base,
                                                   ctypes.c_void_p(status.data_ptr()), sp) == 0

    with torch.cuda.stream(side):
        record()  # (first use outside the capture: anything lazily created exists afterwards)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        record()
    for _ in range(3):
        for o in outs:
            o.zero_()
        status.fill_(0)
        graph.replay()
        torch.cuda.synchronize()
        word = int(status.item()) & (2**64 - 1)
        assert word == ((2 * nb + 100) << 8 | 1)  # first failing block of the batch, BU_ERR_INVALID_MODE
        for k in (0, 1, 3):
            assert torch.equal(outs[k], torch.from_numpy(golden["bc7"][idx[k]]).cuda())
        want2 = golden["bc7"][idx[2]].copy()
        want2[100] = 0  # a failing block leaves zeros
        got2 = outs[2].cpu().numpy()
        assert (got2[:100] == want2[:100]).all() and (got2[101:] == want2[101:]).all()
