"""soak checker (run by hand on the GPU box, not collected by pytest): launches that draw their tiles by TICKET (round 6: exclusive BC7 / ASTC / RGBA32 launches of 16 or
more tiles per workgroup on one of the context's own streams) against the oracle -- random valid, high-contrast and RAW blocks (statuses: lowest failing block),
ragged sizes, launches back to back on one stream (the counters reset themselves), two streams side by side (each has its own counter set), the blocking
entry point, and the multi-run launch of the batch entry point (24 slices in separate allocations: whole-rectangle and mixed runs).  TICKET_SEEDS rounds (default 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from basisu_rs_amd import Context, _lib, synth
from oracle.pyoracle import Oracle
ctx = Context(0); o = Oracle()
ctx.set_launch_policy(False)
FMT = {"astc": (_lib.ASTC, 16), "bc7": (_lib.BC7, 16), "rgba": (_lib.RGBA32, 64)}
t0 = time.time(); total = 0
n1 = 1 << 22
for seed in range(int(os.environ.get("TICKET_SEEDS", 3))):
    rng = np.random.Generator(np.random.PCG64(4242 + seed))
    base = np.concatenate([synth.atlas_rand(n1 - (1 << 19), seed=8000 + seed), synth.atlas_contrast(1 << 19, seed=8100 + seed)])
    rng.shuffle(base, axis=0)
    reps = 4 + seed % 2                      # 16 Mi / 20 Mi blocks
    extra_rows = int(rng.integers(1, 900))   # ragged: not a multiple of the tile
    bpr = 1024
    n = reps * n1 + extra_rows * bpr
    d_base = torch.from_numpy(base).cuda()
    d_in = torch.cat([d_base] * reps + [d_base[: extra_rows * bpr]]).contiguous()
    for name, (t, bb) in FMT.items():
        if name == "rgba":
            st, _, want = o.decode_to_rgba(base.tobytes(), bpr)
            assert st == 0
            want = torch.from_numpy(np.asarray(want).reshape(n1 // bpr, 4, bpr, 16)).cuda()  # image rows -> compare per repetition
        else:
            w_, st_ = o.batch(name, base)
            assert (st_ == 0).all()
            want = torch.from_numpy(w_.reshape(n1, bb)).cuda()
        shape = (n // bpr * 4, bpr * 16) if name == "rgba" else (n, bb)
        outs = [torch.zeros(shape, dtype=torch.uint8, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        # (1) the blocking entry point
        assert ctx.transcode_device_sync(t, d_in, n, outs[0], blocks_per_row=bpr) == _lib.STATUS_WORD_CLEAR
        # (2) back to back on own stream 1, and (3) one more on own stream 3 at the same time
        ctx.transcode_device(t, d_in, n, outs[1], blocks_per_row=bpr, stream=ctx.stream(1))
        ctx.transcode_device(t, d_in, n, outs[2], blocks_per_row=bpr, stream=ctx.stream(3))
        ctx.transcode_device(t, d_in, n, outs[1], blocks_per_row=bpr, stream=ctx.stream(1))
        ctx.synchronize()
        for k, out in enumerate(outs):
            for r in range(reps + 1):
                nr = n1 if r < reps else extra_rows * bpr
                if name == "rgba":
                    got = out[r * (n1 // bpr) * 4: r * (n1 // bpr) * 4 + nr // bpr * 4].view(nr // bpr, 4, bpr, 16)
                    assert torch.equal(got, want[: nr // bpr]), (seed, name, k, r)
                else:
                    assert torch.equal(out[r * n1: r * n1 + nr], want[:nr]), (seed, name, k, r)
        total += 4 * n
        del outs
    # RAW blocks: a few hundred thousand failing blocks among 16 Mi -- the status word is the LOWEST failing block, its status the oracle's
    raw = rng.integers(0, 256, size=(1 << 18, 16), dtype=np.uint8)
    _, st_raw = o.batch("bc7", raw)
    first = int(np.argmax(st_raw != 0))
    assert st_raw[first] != 0
    big = d_in.clone()
    at = int(rng.integers(1, reps)) * n1 + 12345
    big[at: at + (1 << 18)] = torch.from_numpy(raw).cuda()
    for name in ("bc7", "astc"):
        t, bb = FMT[name]
        out = torch.zeros((n, bb), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        word = ctx.transcode_device_sync(t, big, n, out, blocks_per_row=bpr, block_index_base=77)
        assert word >> 8 == 77 + at + first and (word & 0xFF) == int(st_raw[first]), (seed, name, word >> 8, 77 + at + first)
        w_raw, _ = o.batch(name, raw)
        ok_rows = torch.from_numpy(st_raw == 0).cuda()
        got = out[at: at + (1 << 18)]
        assert torch.equal(got[ok_rows], torch.from_numpy(w_raw.reshape(-1, bb)).cuda()[ok_rows]), (seed, name)
        assert not got[~ok_rows].any()
        total += n
    # the multi-run launch (bu_uastc_transcode_batch_device; the headline's kernel): 24 slices of 2^20 blocks in SEPARATE allocations -- every run whole rectangles (the
    # variant without validity tests), tiles by ticket -- then the same with three ragged slices among them (generic variant, strips and rectangles mixed), and raw blocks
    import ctypes
    lib = ctx._lib
    for ragged in (False, True):
        sizes = [1 << 20] * 24
        if ragged:
            sizes[3], sizes[11], sizes[20] = (1 << 20) + 1024 * 7, (1 << 19) + 333, 16384 * 5
        for name in ("bc7", "astc"):
            t, bb = FMT[name]
            w_, st_ = o.batch(name, base)
            want = torch.from_numpy(w_.reshape(n1, bb)).cuda()
            offs = [int(rng.integers(0, n1 - sz)) // 1024 * 1024 for sz in sizes]
            ins = [d_base[of: of + sz].clone() for of, sz in zip(offs, sizes)]
            outs = [torch.zeros((sz, bb), dtype=torch.uint8, device="cuda") for sz in sizes]
            n_s = len(sizes)
            VP, SZ = ctypes.c_void_p * n_s, ctypes.c_size_t * n_s
            a = (n_s, VP(*[x.data_ptr() for x in ins]), SZ(*sizes), VP(*[x.data_ptr() for x in outs]))
            status = torch.empty(1, dtype=torch.int64, device="cuda")
            ctx.status_word_reset(status)
            torch.cuda.synchronize()
            ctx.set_launch_policy("auto")
            for rep in range(2):
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, t, a[0], a[1], a[2], a[3], 1024, None, ctypes.c_void_p(status.data_ptr()), None) == 0
            torch.cuda.synchronize()
            ctx.status_word_check(int(status.item()))
            for k in range(n_s):
                assert torch.equal(outs[k], want[offs[k]: offs[k] + sizes[k]]), (seed, name, ragged, k)
            total += 2 * sum(sizes)
            # raw blocks in slice 9: the lowest failing block of the batch
            ins[9][1000: 1000 + (1 << 16)] = torch.from_numpy(raw[: 1 << 16]).cuda()
            first9 = int(np.argmax(st_raw[: 1 << 16] != 0))
            torch.cuda.synchronize()
            assert lib.bu_uastc_transcode_batch_device(ctx.handle, t, a[0], a[1], a[2], a[3], 1024, None, ctypes.c_void_p(status.data_ptr()), None) == 0
            torch.cuda.synchronize()
            word = int(status.item()) & 0xFFFFFFFFFFFFFFFF
            assert word >> 8 == sum(sizes[:9]) + 1000 + first9 and (word & 0xFF) == int(st_raw[first9]), (seed, name, ragged)
            ctx.set_launch_policy(False)
            del ins, outs
    print("seed %d ok: %d Mi blocks so far, %.0f s" % (seed, total >> 20, time.time() - t0), flush=True)
print("TICKET SOAK OK: %d Mi block transcodes against the oracle, statuses included, %.0f s" % (total >> 20, time.time() - t0))
