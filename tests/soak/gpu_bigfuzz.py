"""soak checker (run by hand on the GPU box, not collected by pytest): HIP path vs oracle on millions of random blocks (valid and raw), all targets, statuses included.
FUZZ_POLICY=shared: the context's launch policy (the half-CU shapes).  FUZZ_CONCURRENT=1: the five targets of every input are issued TOGETHER, each on its own
context stream (device entry point, block grid given), and compared afterwards -- launches of different kernels side by side on every CU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from basisu_rs_amd import Context, _lib, synth, BasisuError
from oracle.pyoracle import Oracle
ctx = Context(0); o = Oracle()
ctx.set_launch_policy({"shared": True, "auto": "auto"}.get(os.environ.get("FUZZ_POLICY"), False))  # FUZZ_POLICY=shared | auto | (exclusive)
FMT = {"astc": _lib.ASTC, "bc7": _lib.BC7, "etc1": _lib.ETC1, "etc2": _lib.ETC2}
t0 = time.time(); total = 0
SEED0 = int(os.environ.get("FUZZ_SEED0", 0))
for seed in range(SEED0, SEED0 + int(os.environ.get("FUZZ_SEEDS", 4))):
    n = 1 << 20
    for kind in os.environ.get("FUZZ_KINDS", "valid,raw").split(","):
        if kind == "valid":
            blocks = synth.atlas_rand(n, seed=500 + seed)
        elif kind == "contrast":  # endpoints at the extremes, skewed weights (synth.atlas_contrast)
            blocks = synth.atlas_contrast(n, seed=700 + seed)
        else:
            blocks = np.random.Generator(np.random.PCG64(900 + seed)).integers(0, 256, size=(n, 16), dtype=np.uint8)
        if os.environ.get("FUZZ_CONCURRENT"):
            import ctypes, torch
            use = blocks.copy()
            wants = {}
            for t in ("astc", "bc7", "etc1", "etc2", "rgba"):
                w_, st_ = o.batch(t, use)
                wants[t] = w_
            bad = np.where(o.batch("bc7", use)[1] != 0)[0]
            if bad.size:
                use[bad] = synth.atlas_rand(1, seed=1)[0]
                for t in wants: wants[t] = o.batch(t, use)[0]
            d_in = torch.from_numpy(use).cuda()
            outs = {t: torch.zeros((n // 1024 * 4, 1024 * 16) if t == "rgba" else (n, _lib.BLOCK_BYTES[FMT[t]]), dtype=torch.uint8, device="cuda") for t in wants}
            torch.cuda.synchronize()
            for rep in range(3):  # three rounds back to back: launches of every kernel overlap launches of every other
                for i, t in enumerate(wants):
                    ctx.transcode_device(_lib.RGBA32 if t == "rgba" else FMT[t], d_in, n, outs[t], blocks_per_row=1024, stream=ctx.stream(i % 4))
            torch.cuda.synchronize()
            for t in wants:
                got = outs[t].cpu().numpy()
                if t == "rgba": got = np.ascontiguousarray(got.reshape(n // 1024, 4, 1024, 16).transpose(0, 2, 1, 3)).reshape(n, 64)
                assert (got == wants[t].reshape(got.shape)).all(), (seed, kind, t)
            total += n
            continue
        for t in ("astc", "bc7", "etc1", "etc2", "rgba"):
            want, st = o.batch(t, blocks)
            bad = np.where(st != 0)[0]
            use = blocks
            if bad.size:  # the GPU call aborts at the first bad block: check that, then neutralise the bad blocks
                try:
                    ctx.transcode(FMT[t], blocks) if t != "rgba" else ctx.decode_to_rgba(blocks, 1024)
                    raise SystemExit("expected an error")
                except BasisuError as e:
                    assert e.first_bad_block == bad[0], (t, e.first_bad_block, bad[0])
                use = blocks.copy(); use[bad] = synth.atlas_rand(1, seed=1)[0]
                want, st2 = o.batch(t, use); assert (st2 == 0).all()
            if os.environ.get("FUZZ_RECT"):  # device entry point with the block grid given: the rectangular-tile kernels (BC7, ASTC, RGBA32)
                import torch
                d_in = torch.from_numpy(use).cuda()
                if t == "rgba":
                    d_out = torch.empty((n // 1024 * 4, 1024 * 16), dtype=torch.uint8, device="cuda")
                    ctx.transcode_device(_lib.RGBA32, d_in, n, d_out, blocks_per_row=1024)
                    got = np.ascontiguousarray(d_out.cpu().numpy().reshape(n // 1024, 4, 1024, 16).transpose(0, 2, 1, 3)).reshape(n, 64)
                else:
                    d_out = torch.empty((n, _lib.BLOCK_BYTES[FMT[t]]), dtype=torch.uint8, device="cuda")
                    ctx.transcode_device(FMT[t], d_in, n, d_out, blocks_per_row=1024)
                    got = d_out.cpu().numpy()
            elif t == "rgba":
                img = ctx.decode_to_rgba(use, 1024).reshape(n // 1024, 4, 1024, 16)
                got = np.ascontiguousarray(img.transpose(0, 2, 1, 3)).reshape(n, 64)
            else:
                got = ctx.transcode(FMT[t], use).reshape(n, -1)
            assert (got == want).all(), (seed, kind, t, np.where((got != want).any(axis=1))[0][:5])
        total += n
    print("seed", seed, "ok  %.0fs" % (time.time() - t0), flush=True)
print("blocks per target:", total)
