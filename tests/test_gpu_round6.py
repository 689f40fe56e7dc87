"""Round-6 GPU tests: the pipeline behind the product's own entry points and independent of the process environment.

  * the context's streams get a hardware queue each WITHOUT GPU_MAX_HW_QUEUES in the environment (child processes that unset it): CU-mask
    streams, reported by bu_context_query_in_flight; with BU_STREAM_MODE=plain (ordinary streams forced) the in-flight call degrades to the
    stream-ordered batch launch and says so -- results identical either way;
  * BU_LAUNCH_AUTO (the default): same bytes and status words as the explicit policies, alone and with four launches in flight;
  * tile tickets (dynamic tile assignment inside one persistent launch, from 16 tiles per workgroup on) and bu_uastc_transcode_device_sync:
    every target, ragged sizes, both sides of the threshold, lowest failing block; launches back to back on one stream (the counters reset themselves);
  * bu_array_transcode_sharded over ranges large enough to draw tickets (virtual ranks on one device).
Everything goes through the C ABI; bit-exact.  Run on the GPU box: pytest -m gpu."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from basisu_rs_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TB = {"astc": (_lib.ASTC, 16), "bc7": (_lib.BC7, 16), "rgba": (_lib.RGBA32, 64), "etc1": (_lib.ETC1, 8), "etc2": (_lib.ETC2, 16)}

# ---- a child process without GPU_MAX_HW_QUEUES: 64 atlases through bu_uastc_transcode_batch_in_flight -------------------------------------
_CHILD = r"""
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from basisu_rs_amd import Context, _lib, synth
assert "GPU_MAX_HW_QUEUES" not in os.environ
golden = synth.load_golden(os.path.join(%(root)r, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)            # torch's NULL stream exists and holds a queue of the pool, as in any torch process
ctx = Context(0)
lib = ctx._lib
g_u, g_b = torch.from_numpy(golden["uastc"]).to(dev), torch.from_numpy(golden["bc7"]).to(dev)
N, NB = 64, 1 << 20
idxs = [torch.randint(0, 608, (NB,), device=dev, generator=torch.Generator(device=dev).manual_seed(77 + k)) for k in range(N)]
ins = [g_u[i].contiguous() for i in idxs]
outs = [torch.zeros((NB, 16), dtype=torch.uint8, device=dev) for _ in range(N)]
status = torch.empty(1, dtype=torch.int64, device=dev)
ctx.status_word_reset(status)
torch.cuda.synchronize()
eff, mode = ctx.query_in_flight(4)
sharing_now = ctx.probe_streams(4)
VP, SZ = ctypes.c_void_p * N, ctypes.c_size_t * N
a_in, a_n, a_out, sp = VP(*[t.data_ptr() for t in ins]), SZ(*([NB] * N)), VP(*[t.data_ptr() for t in outs]), ctypes.c_void_p(status.data_ptr())
def call():  # (argument arrays built once: the calls and the wait are what is timed)
    assert lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, N, a_in, a_n, a_out, 1024, None, sp, 4) == 0
def run():
    call()
    assert lib.bu_context_synchronize(ctx.handle) == 0
run()                                  # correctness pass
ok = all(bool(torch.equal(outs[k], g_b[idxs[k]])) for k in range(N))
word = int(status.item()) & 0xFFFFFFFFFFFFFFFF
for _ in range(40): run()              # clocks (~15 ms)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); call(); call(); call(); run(); dt = (time.perf_counter() - t0) / (4 * N) * 1e6  # four calls back to back, one wait
    best = min(best, dt)
ctx.close()
print(json.dumps({"effective": eff, "mode": mode, "sharing_now": sharing_now, "ok": ok, "clear": word == _lib.STATUS_WORD_CLEAR, "us_per_atlas": best}))
"""


def _child(extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "BU_STREAM_MODE", "DEBUG_HIP_DYNAMIC_QUEUES")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_in_flight_without_the_environment_variable():
    """a host process that never heard of GPU_MAX_HW_QUEUES (a Rust binary): the library gives its streams a hardware queue each by itself, says so,
    and 64 atlases through ONE call of the in-flight entry point come out right at the pipeline's rate -- or the call reports fewer effective streams"""
    r = _child({})
    assert r["ok"] and r["clear"], r
    assert r["effective"] < 4 or r["us_per_atlas"] <= 6.6, r
    if r["effective"] == 4:
        assert r["sharing_now"] == 1, r
    print("in flight without GPU_MAX_HW_QUEUES:", r)


def test_in_flight_degrades_by_itself_when_the_streams_share_queues():
    """BU_STREAM_MODE=plain forces ordinary streams on the default pool of four queues (two of them taken): the context reports fewer effective
    streams than requested and the in-flight call goes out as the stream-ordered batch launch -- same bytes, and faster than a pipeline two deep"""
    r = _child({"BU_STREAM_MODE": "plain"})
    assert r["ok"] and r["clear"], r
    assert r["mode"] == "pool", r
    if r["effective"] < 4:  # (a runtime whose pool has room gives four queues anyway: then there is nothing to degrade)
        assert r["us_per_atlas"] <= 7.0, r
    print("in flight, plain streams on the default pool:", r)


# ---- BU_LAUNCH_AUTO ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("target", ["bc7", "astc", "etc1", "etc2", "rgba"])
def test_auto_policy_alone_and_in_flight_known_answers(golden, target):
    """the default policy picks a shape per call; whatever it picks, the bytes are the known answers: one launch at a time on torch's stream
    (exclusive), on one context stream (exclusive), and 12 launches round-robin on four context streams (the first exclusive, the others shared)"""
    import torch

    from basisu_rs_amd import Context

    ctx = Context(0)
    t, bb = TB[target]
    bpr, rows = 1024, 1040  # > 3 tiles per CU: the large shapes of every target; ragged against the 2048-block ETC tiles
    n = bpr * rows
    K = 12
    gu, gt = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden[target]).cuda()
    idxs = [torch.from_numpy(synth.gold_indices(n, seed=900 + k)).cuda() for k in range(K)]
    ins = [gu[i].contiguous() for i in idxs]
    shape = (rows * 4, bpr * 16) if target == "rgba" else (n, bb)
    outs = [torch.zeros(shape, dtype=torch.uint8, device="cuda") for _ in range(K)]
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    torch.cuda.synchronize()

    def check(k):
        got = outs[k]
        if target == "rgba":
            got = got.view(rows, 4, bpr, 16).permute(0, 2, 1, 3).reshape(n, 64)
        assert torch.equal(got, gt[idxs[k]]), (target, k)
        outs[k].zero_()

    ctx.transcode_device(t, ins[0], n, outs[0], blocks_per_row=bpr, d_status=status)  # torch's stream
    torch.cuda.synchronize()
    check(0)
    ctx.transcode_device(t, ins[1], n, outs[1], blocks_per_row=bpr, d_status=status, stream=ctx.stream(0))  # a lone launch on an own stream
    ctx.synchronize()
    check(1)
    torch.cuda.synchronize()
    for k in range(K):
        ctx.transcode_device(t, ins[k], n, outs[k], blocks_per_row=bpr, d_status=status, stream=ctx.stream(k % 4))
    ctx.synchronize()
    for k in range(K):
        check(k)
    ctx.status_word_check(int(status.item()))
    ctx.close()


@pytest.mark.parametrize("target", ["bc7", "astc"])
def test_auto_policy_two_streams_one_tile_workgroups_ragged_sizes(golden, target):
    """with one or two other launches in flight the default policy takes the shared kernels on one-tile workgroups (BU_POLICY_SHARED_FEW): strips and rectangles,
    ragged tails, sizes whose tiles exceed the grid (workgroups walk), the first-error index -- two and three streams round-robin"""
    import torch

    from basisu_rs_amd import BasisuError, Context

    ctx = Context(0)
    t, bb = TB[target]
    gu, gt = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden[target]).cuda()
    for n_streams in (2, 3):
        for bpr, n in [(0, (1 << 20) + 12345), (1024, 1024 * 1056), (96, 96 * 11000), (2048, 2048 * 1040)]:
            K = 6
            idxs = [torch.from_numpy(synth.gold_indices(n, seed=1200 + k + n_streams)).cuda() for k in range(K)]
            ins = [gu[i].contiguous() for i in idxs]
            outs = [torch.zeros((n, bb), dtype=torch.uint8, device="cuda") for _ in range(K)]
            status = torch.empty(1, dtype=torch.int64, device="cuda")
            ctx.status_word_reset(status)
            torch.cuda.synchronize()
            for k in range(K):
                ctx.transcode_device(t, ins[k], n, outs[k], blocks_per_row=bpr, block_index_base=k * n, d_status=status, stream=ctx.stream(k % n_streams))
            ctx.synchronize()
            ctx.status_word_check(int(status.item()))
            for k in range(K):
                assert torch.equal(outs[k], gt[idxs[k]]), (target, n_streams, bpr, k)
            ins[4][n - 7, 0] = 69
            ins[2][n // 3, 0] = 69
            torch.cuda.synchronize()
            for k in range(K):
                ctx.transcode_device(t, ins[k], n, outs[k], blocks_per_row=bpr, block_index_base=k * n, d_status=status, stream=ctx.stream(k % n_streams))
            ctx.synchronize()
            with pytest.raises(BasisuError) as e:
                ctx.status_word_check(int(status.item()))
            assert e.value.first_bad_block == 2 * n + n // 3, (target, n_streams, bpr)
            del ins, outs
    ctx.close()


# ---- bu_uastc_transcode_device_sync --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("target", ["bc7", "astc", "etc1", "etc2", "rgba"])
def test_device_sync_tile_tickets_known_answers_and_lowest_error(golden, target):
    import torch

    from basisu_rs_amd import BasisuError, Context

    ctx = Context(0)
    t, bb = TB[target]
    gu, gt = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden[target]).cuda()
    # below the ticket threshold (fixed walk), at it (16 tiles per workgroup: 2^24 blocks for BC7 / ASTC, 2^23 for RGBA32), ragged above it, strips (bpr 1000)
    for bpr, rows in [(1024, 1024), (1024, 8192 if target == "rgba" else 16384), (1024, 16384 + 1031), (2048, 8192 + 1024 + 7), (1000, 17000)]:
        if target == "rgba" and rows > 9000:
            rows //= 2  # (64 B per block of output)
        n = bpr * rows
        idx = torch.from_numpy(synth.gold_indices(n, seed=4000 + rows)).cuda()
        d_in = gu[idx].contiguous()
        shape = (rows * 4, bpr * 16) if target == "rgba" else (n, bb)
        d_out = torch.zeros(shape, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        word = ctx.transcode_device_sync(t, d_in, n, d_out, blocks_per_row=bpr, block_index_base=1000)
        assert word == _lib.STATUS_WORD_CLEAR
        got = d_out.view(rows, 4, bpr, 16).permute(0, 2, 1, 3).reshape(n, 64) if target == "rgba" else d_out
        assert torch.equal(got, gt[idx]), (target, bpr, rows)
        # failing blocks in the last piece, in the second and (lowest) in the first
        for bad in (n - 3, n // 2 + 17, n // 9 + 1):
            d_in[bad, 0] = 69
            torch.cuda.synchronize()
            word = ctx.transcode_device_sync(t, d_in, n, d_out, blocks_per_row=bpr, block_index_base=1000)
            with pytest.raises(BasisuError) as e:
                ctx.status_word_check(word)
            assert e.value.status == _lib.ERR_INVALID_MODE and e.value.first_bad_block == 1000 + bad, (target, bpr, rows, bad)
        del d_in, d_out
    # arguments
    lib = ctx._lib
    w = ctypes.c_uint64(0)
    assert lib.bu_uastc_transcode_device_sync(ctx.handle, t, None, 5, None, 0, 0, ctypes.byref(w)) == _lib.ERR_ARGUMENT
    assert lib.bu_uastc_transcode_device_sync(ctx.handle, t, None, 0, None, 1, 0, ctypes.byref(w)) == 0 and w.value == _lib.STATUS_WORD_CLEAR
    ctx.close()


def test_device_sync_against_the_oracle_on_random_blocks(oracle):
    """4.25 Mi random valid + high-contrast blocks, four times over (17 Mi blocks: a BC7 / ASTC launch that draws its tiles by ticket; ETC2 keeps the
    fixed walk), against the CPU restatement of the reference; then three such launches back to back on one stream -- the ticket counters reset themselves"""
    import torch

    from basisu_rs_amd import Context

    ctx = Context(0)
    n1 = (1 << 22) + (1 << 18)
    base = np.concatenate([synth.atlas_rand(1 << 22, seed=61), synth.atlas_contrast(1 << 18, seed=62)])
    n = 4 * n1
    d_in = torch.from_numpy(base).cuda().repeat(4, 1).contiguous()
    for target in ("bc7", "astc", "etc2"):
        t, bb = TB[target]
        want, st = oracle.batch(target, base)
        assert (st == 0).all()
        want = torch.from_numpy(want.reshape(n1, bb)).cuda()
        d_out = torch.zeros((n, bb), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert ctx.transcode_device_sync(t, d_in, n, d_out, blocks_per_row=1024) == _lib.STATUS_WORD_CLEAR
        for r in range(4):
            assert torch.equal(d_out[r * n1:(r + 1) * n1], want), (target, r)
        outs = [torch.zeros((n, bb), dtype=torch.uint8, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        for o in outs:  # back to back on one of the context's own streams (tickets), nothing in between
            ctx.transcode_device(t, d_in, n, o, blocks_per_row=1024, stream=ctx.stream(2))
        ctx.synchronize()
        for o in outs:
            assert torch.equal(o, d_out), target
    ctx.close()


def test_tile_tickets_on_the_callers_streams(golden):
    """a ticket set per STREAM: launches of 2^24 blocks on torch's NULL stream, on two side streams at the same time (each its own set), back to back on one, and on 40
    different streams in turn (the context has 32 sets for streams of the caller's: the rest walk fixed shares) -- same bytes as the known answers every time; a launch
    captured into a graph (no tickets: a graph may be replayed anywhere) replays correctly beside a ticketed launch on another stream"""
    import torch

    from basisu_rs_amd import Context

    ctx = Context(0)
    n, bpr = 1 << 24, 1024
    gu, gb = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden["bc7"]).cuda()
    idx = torch.randint(0, 608, (n,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(31))
    d_in = gu[idx].contiguous()
    want = gb[idx]
    outs = [torch.zeros((n, 16), dtype=torch.uint8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    ctx.transcode_device(_lib.BC7, d_in, n, outs[0], blocks_per_row=bpr)  # torch's current stream: the NULL stream
    torch.cuda.synchronize()
    assert torch.equal(outs[0], want)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    ctx.transcode_device(_lib.BC7, d_in, n, outs[0], blocks_per_row=bpr, stream=s1)
    ctx.transcode_device(_lib.BC7, d_in, n, outs[1], blocks_per_row=bpr, stream=s2)
    ctx.transcode_device(_lib.BC7, d_in, n, outs[2], blocks_per_row=bpr, stream=s1)  # back to back behind the first
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)
    many = [torch.cuda.Stream() for _ in range(40)]
    for k, st in enumerate(many):
        o = outs[k % 3]
        o.zero_()
        torch.cuda.synchronize()
        ctx.transcode_device(_lib.BC7, d_in, n, o, blocks_per_row=bpr, stream=st)
        torch.cuda.synchronize()
        assert torch.equal(o, want), k
    # a captured launch (fixed walk) replayed on s2 while a ticketed launch runs on s1
    g = torch.cuda.CUDAGraph()
    outs[1].zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s2):
        ctx.transcode_device(_lib.BC7, d_in, n, outs[1], blocks_per_row=bpr, stream=s2)
    outs[0].zero_()
    outs[1].zero_()
    torch.cuda.synchronize()
    ctx.transcode_device(_lib.BC7, d_in, n, outs[0], blocks_per_row=bpr, stream=s1)
    with torch.cuda.stream(s2):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], want) and torch.equal(outs[1], want)
    ctx.close()


@pytest.mark.parametrize("target", ["bc7", "rgba"])
def test_batch_device_long_walk_draws_tickets(golden, target):
    """bu_uastc_transcode_batch_device over 20 (RGBA32: 10) slices of about 2^20 blocks in separate allocations, ragged: ONE persistent launch whose workgroups walk
    the tiles of all runs -- 16 and more tiles per workgroup, so they draw them by ticket; known answers, twice back to back on one stream, and the lowest
    failing block of the batch"""
    import torch

    from basisu_rs_amd import BasisuError

    from basisu_rs_amd import Context

    ctx = Context(0)
    lib = ctx._lib
    t, bb = TB[target]
    bpr = 1024
    n_s = 10 if target == "rgba" else 20
    sizes = [bpr * (1024 + 3 * k) for k in range(n_s)]
    gu, gt = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden[target]).cuda()
    idxs = [torch.randint(0, 608, (n,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(77 + k)) for k, n in enumerate(sizes)]
    ins = [gu[i].contiguous() for i in idxs]
    outs = [torch.zeros((n // bpr * 4, bpr * 16) if target == "rgba" else (n, bb), dtype=torch.uint8, device="cuda") for n in sizes]
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    torch.cuda.synchronize()
    VP, SZ = ctypes.c_void_p * n_s, ctypes.c_size_t * n_s
    a = (n_s, VP(*[x.data_ptr() for x in ins]), SZ(*sizes), VP(*[x.data_ptr() for x in outs]))
    s = torch.cuda.Stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    for rep in range(2):
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, t, a[0], a[1], a[2], a[3], bpr, None, ctypes.c_void_p(status.data_ptr()), sp) == 0
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()))
    for k, n in enumerate(sizes):
        got = outs[k].view(n // bpr, 4, bpr, 16).permute(0, 2, 1, 3).reshape(n, 64) if target == "rgba" else outs[k]
        assert torch.equal(got, gt[idxs[k]]), (target, k)
    ins[7][sizes[7] - 1, 0] = 69
    ins[3][5, 0] = 69
    torch.cuda.synchronize()
    assert lib.bu_uastc_transcode_batch_device(ctx.handle, t, a[0], a[1], a[2], a[3], bpr, None, ctypes.c_void_p(status.data_ptr()), sp) == 0
    torch.cuda.synchronize()
    with pytest.raises(BasisuError) as e:
        ctx.status_word_check(int(status.item()))
    assert e.value.first_bad_block == sum(sizes[:3]) + 5
    ctx.close()


@pytest.mark.parametrize("bpr", [128, 256, 1024])
def test_rgba32_multi_run_launch_tiles_whole_runs_as_rectangles(golden, bpr):
    """bu_uastc_transcode_batch_device, RGBA32, slices in separate allocations: runs that are whole 64 x 16-block rectangles of the image's own power-of-two pitch are tiled that
    way inside the multi-run launch (as the plain launch tiles them; an image has no virtual pitch), ragged ones as strips, both kinds in one launch -- short walks and a long
    (ticketed) one, known answers per pixel row, lowest failing block"""
    import torch

    from basisu_rs_amd import BasisuError, Context

    ctx = Context(0)
    lib = ctx._lib
    gu, gt = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden["rgba"]).cuda()
    # (ragged images of 128 block rows or more go out as two table entries: the whole prefix as rectangles, the last rows as strips)
    for rows in ([64, 16, 160, 48, 1024 * 256 // bpr], [64, 40, 16, 7, 2048 * 256 // bpr, 33], [149, 64, 1003, 16 * 20 + 15, 128, 129], [4096 * 1024 // bpr] * 9):
        sizes = [r * bpr for r in rows]
        n_s = len(sizes)
        idxs = [torch.randint(0, 608, (n,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(900 + k)) for k, n in enumerate(sizes)]
        ins = [gu[i].contiguous() for i in idxs]
        outs = [torch.zeros((n // bpr * 4, bpr * 16), dtype=torch.uint8, device="cuda") for n in sizes]
        status = torch.empty(1, dtype=torch.int64, device="cuda")
        ctx.status_word_reset(status)
        torch.cuda.synchronize()
        VP, SZ = ctypes.c_void_p * n_s, ctypes.c_size_t * n_s
        a = (n_s, VP(*[x.data_ptr() for x in ins]), SZ(*sizes), VP(*[x.data_ptr() for x in outs]))
        for rep in range(2):
            assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.RGBA32, a[0], a[1], a[2], a[3], bpr, None, ctypes.c_void_p(status.data_ptr()), None) == 0
        torch.cuda.synchronize()
        ctx.status_word_check(int(status.item()))
        for k, n in enumerate(sizes):
            got = outs[k].view(n // bpr, 4, bpr, 16).permute(0, 2, 1, 3).reshape(n, 64)
            assert torch.equal(got, gt[idxs[k]]), (bpr, rows, k)
        ins[n_s - 1][sizes[-1] - 1, 0] = 69
        ins[1][3, 0] = 69
        torch.cuda.synchronize()
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.RGBA32, a[0], a[1], a[2], a[3], bpr, None, ctypes.c_void_p(status.data_ptr()), None) == 0
        torch.cuda.synchronize()
        with pytest.raises(BasisuError) as e:
            ctx.status_word_check(int(status.item()))
        assert e.value.first_bad_block == sizes[0] + 3
        del ins, outs
    ctx.close()


@pytest.mark.parametrize("target", ["bc7", "astc"])
def test_virtual_pitch_tiles_without_a_block_grid(golden, target):
    """BC7 / ASTC without a usable blocks_per_row (0, or no multiple of 64): slices that are whole multiples of 16 x {1024, 2048, 512, 256} blocks are tiled as 64 x 16-block
    rectangles of a VIRTUAL grid (faster loads and stores; never a different byte) -- small and large launches, both policies' shapes, lowest failing block -- and the same
    inside ONE multi-run launch: every run whole (the kernel variant without validity tests), and whole and ragged runs mixed"""
    import torch

    from basisu_rs_amd import BasisuError, Context

    ctx = Context(0)
    lib = ctx._lib
    t, bb = TB[target]
    gu, gt = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden[target]).cuda()
    for bpr, n in [(0, 1 << 20), (0, 3 * 16384), (0, 5 * 4096), (0, 7 * 8192), (100, 2 * 32768), (0, (1 << 21) + 16384), (0, 1 << 24)]:
        idx = torch.from_numpy(synth.gold_indices(n, seed=n & 0xFFFF)).cuda()
        d_in = gu[idx].contiguous()
        for shared in (False, True):
            ctx.set_launch_policy(shared)
            d_out = torch.zeros((n, bb), dtype=torch.uint8, device="cuda")
            status = torch.empty(1, dtype=torch.int64, device="cuda")
            ctx.status_word_reset(status)
            torch.cuda.synchronize()
            ctx.transcode_device(t, d_in, n, d_out, blocks_per_row=bpr, block_index_base=5, d_status=status)
            torch.cuda.synchronize()
            ctx.status_word_check(int(status.item()))
            assert torch.equal(d_out, gt[idx]), (target, bpr, n, shared)
        bad = n // 3 + 1
        d_in[bad, 0] = 69
        d_in[n - 1, 0] = 69
        torch.cuda.synchronize()
        ctx.transcode_device(t, d_in, n, d_out, blocks_per_row=bpr, block_index_base=5, d_status=status)
        torch.cuda.synchronize()
        with pytest.raises(BasisuError) as e:
            ctx.status_word_check(int(status.item()))
        assert e.value.first_bad_block == 5 + bad and not d_out[bad].any()
        del d_in, d_out
    ctx.set_launch_policy("auto")
    # one multi-run launch: (a) every run whole, (b) whole and ragged runs mixed; separate allocations
    for sizes in ([1 << 20, 3 << 18, 1 << 20, 5 << 18, 1 << 20, 1 << 19], [1 << 20, (1 << 20) + 1024, 3 << 18, 300000, 1 << 20, 77 * 1024]):
        n_s = len(sizes)
        idxs = [torch.randint(0, 608, (n,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(500 + k)) for k, n in enumerate(sizes)]
        ins = [gu[i].contiguous() for i in idxs]
        outs = [torch.zeros((n, bb), dtype=torch.uint8, device="cuda") for n in sizes]
        status = torch.empty(1, dtype=torch.int64, device="cuda")
        ctx.status_word_reset(status)
        torch.cuda.synchronize()
        VP, SZ = ctypes.c_void_p * n_s, ctypes.c_size_t * n_s
        a = (n_s, VP(*[x.data_ptr() for x in ins]), SZ(*sizes), VP(*[x.data_ptr() for x in outs]))
        for bpr in (0, 1024):
            for o in outs:
                o.zero_()
            assert lib.bu_uastc_transcode_batch_device(ctx.handle, t, a[0], a[1], a[2], a[3], bpr, None, ctypes.c_void_p(status.data_ptr()), None) == 0
            torch.cuda.synchronize()
            ctx.status_word_check(int(status.item()))
            for k in range(n_s):
                assert torch.equal(outs[k], gt[idxs[k]]), (target, sizes, bpr, k)
        ins[4][sizes[4] - 1, 0] = 69
        ins[1][12345, 0] = 69
        torch.cuda.synchronize()
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, t, a[0], a[1], a[2], a[3], 0, None, ctypes.c_void_p(status.data_ptr()), None) == 0
        torch.cuda.synchronize()
        with pytest.raises(BasisuError) as e:
            ctx.status_word_check(int(status.item()))
        assert e.value.first_bad_block == sizes[0] + 12345
    ctx.close()


# ---- bu_array_transcode_sharded with ranges that draw tickets ---------------------------------------------------------------------------
def _ptr_array(vals):
    return (ctypes.c_void_p * len(vals))(*vals)


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
def test_array_transcode_sharded_large_ranges(golden, n_ctx):
    """a 600-slice array of 65 536-block slices: 39 Mi blocks, 13-39 Mi per virtual rank -- ranges of 2^24 blocks and more draw their tiles by ticket;
    all contexts' full buffers equal the unsharded result, and the lowest failing block of the whole array is the one reported"""
    import torch

    from basisu_rs_amd import Context, sharded

    lib = _lib.load()
    n_slices, bps = 600, 65536
    gu, gb = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden["bc7"]).cuda()
    idx = torch.randint(0, 608, (n_slices * bps,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    blocks = gu[idx].contiguous()
    want = gb[idx]
    ctxs = [Context(0) for _ in range(n_ctx)]
    try:
        ins, fulls = [], []
        for r in range(n_ctx):
            lo, hi = sharded.partition(n_slices, n_ctx, r)
            ins.append(blocks[lo * bps:hi * bps].clone())
            fulls.append(torch.zeros((n_slices * bps, 16), dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        handles = _ptr_array([c.handle.value for c in ctxs])
        bad = ctypes.c_uint64(0)
        args = lambda: (handles, n_ctx, _lib.BC7, _ptr_array([t.data_ptr() for t in ins]), n_slices, bps, _ptr_array([t.data_ptr() for t in fulls]), 1, ctypes.byref(bad))  # noqa: E731
        st = lib.bu_array_transcode_sharded(*args())
        assert st == 0, (lib.bu_status_string(st), lib.bu_last_error(ctxs[0].handle))
        torch.cuda.synchronize()
        for r in range(n_ctx):
            assert torch.equal(fulls[r], want), r
        lo_last = sharded.partition(n_slices, n_ctx, n_ctx - 1)[0]
        n_last = ins[-1].shape[0]
        ins[-1][n_last - 2, 0] = 69       # in the last piece of the last range
        ins[0][3 * bps + 9, 0] = 69       # in the first piece of the first
        torch.cuda.synchronize()
        st = lib.bu_array_transcode_sharded(*args())
        assert st == _lib.ERR_INVALID_MODE and bad.value == 3 * bps + 9
        ins[0][3 * bps + 9, 0] = int(blocks[3 * bps + 9, 0].item())
        torch.cuda.synchronize()
        st = lib.bu_array_transcode_sharded(*args())
        assert st == _lib.ERR_INVALID_MODE and bad.value == lo_last * bps + n_last - 2
    finally:
        for c in ctxs:
            c.close()


def test_sharded_gpu_transcode_fn_on_a_large_shard(golden):
    """the torch.distributed driver's per-shard function (sharded.gpu_transcode_fn) on one rank: a 2^24-block shard through bu_uastc_transcode_device_sync"""
    import torch

    from basisu_rs_amd import Context, sharded

    ctx = Context(0)
    n_slices, bps = 256, 65536
    gu, gb = torch.from_numpy(golden["uastc"]).cuda(), torch.from_numpy(golden["bc7"]).cuda()
    idx = torch.randint(0, 608, (n_slices * bps,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    slices = gu[idx].view(n_slices, bps, 16).contiguous()
    out = sharded.transcode_array_sharded(slices, sharded.gpu_transcode_fn(ctx, _lib.BC7), gather=False)
    assert torch.equal(out.view(-1, 16), gb[idx])
    slices[100, 77, 0] = 69
    with pytest.raises(Exception) as e:
        sharded.transcode_array_sharded(slices, sharded.gpu_transcode_fn(ctx, _lib.BC7), gather=False)
    assert getattr(e.value, "first_bad_block", None) == 100 * bps + 77
    ctx.close()


def test_read_to_bc7_large_file(ctx, golden):
    """bu_read_to on a UASTC file whose slices form one run of 2^22 blocks (64 MiB), pageable output (one upload, one launch, one download: the four-stream
    pieces round 6 tried here lose 4-5 %, profiles/r06_read_to_pieces_pageable_output.txt) and page-locked output (two-stream pieces); also astc / etc1 and a
    damaged block in the third slice"""
    import basisu_rs_amd as bu
    from basisu_rs_amd import BasisuError

    nbx, nby, n_slices = 1024, 1024, 4
    idx = [synth.gold_indices(nbx * nby, seed=7100 + k) for k in range(n_slices)]
    slices = [golden["uastc"][i].copy() for i in idx]
    f = bu.write_uastc_file([dict(data=s, orig_w=4 * nbx, orig_h=4 * nby, nbx=nbx, nby=nby, image_index=k) for k, s in enumerate(slices)])
    for name, fn, bb in (("bc7", bu.read_to_bc7, 16), ("astc", bu.read_to_astc, 16), ("etc1", bu.read_to_etc1, 8)):
        imgs = fn(f, ctx)
        assert len(imgs) == n_slices
        for k in range(n_slices):
            assert (np.asarray(imgs[k].data).reshape(-1, bb) == golden[name][idx[k]]).all(), (name, k)
    pinned = ctx.host_alloc(n_slices * nbx * nby * 16)
    imgs = bu.read_to_bc7(f, ctx, out=pinned)
    for k in range(n_slices):
        assert (np.asarray(imgs[k].data).reshape(-1, 16) == golden["bc7"][idx[k]]).all(), k
    ctx.host_free(pinned)
    slices[2][4242, 0] = 69
    f = bu.write_uastc_file([dict(data=s, orig_w=4 * nbx, orig_h=4 * nby, nbx=nbx, nby=nby, image_index=k) for k, s in enumerate(slices)])
    with pytest.raises(BasisuError) as e:
        bu.read_to_bc7(f, ctx)
    assert e.value.status == _lib.ERR_INVALID_MODE


def test_bench_default_method_is_one_launch_per_64_atlases():
    """bench.py's default line (--method batch): a step is ONE launch over 64 atlases in separate allocations through bu_uastc_transcode_batch_device; the line's value, ms_per_step and
    roofline belong to that kernel (HIP events), round 5's pipeline is measured beside it, and with BENCH_FORCE_DIST the N > 1 branch runs the same steps"""
    import socket

    for force_dist in (False, True):
        env = dict(os.environ)
        if force_dist:
            env["BENCH_FORCE_DIST"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--headline-only", "--no-cpu", "--no-live-traffic", "--steps", "6", "--warmup", "2"], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["config"]["method"] == "batch" and line["config"]["atlases_per_step"] == 64 and line["steps"] == 6
        us_atlas = line["ms_per_step"] * 1e3 / 64
        assert 4.8 < us_atlas < 7.5, us_atlas
        assert abs(line["value"] - 64 * (1 << 20) / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]
        roof = line["roofline"]
        assert roof["atlases_per_launch"] == 64 and 0.55 < roof["frac"] < 0.9 and roof["bytes_per_launch"] == 64 * 32 << 20
        pipe = line["one_launch_per_atlas_in_flight"]
        assert 4.8 < pipe["us_per_atlas"] < 9.0 and pipe["roofline"]["bound"] == "hbm"
