"""whole-file read_to_rgba on ETC1S files with several large slices, every large slice decoded on two host threads against one
(BU_ETC1S_ONE_THREAD=1) and against the one-launch path (BU_ETC1S_ONE_LAUNCH=1); page-locked output, median of 15 calls"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import basis_builder as bb
import basisu_rs_amd as bu
from basisu_rs_amd import _lib, Context
ctx = Context(0)
cases = [("one 512x512-block slice (config 4)", [(512, 512)], False),
         ("512x512 colour + alpha slices", [(512, 512)], True),
         ("mip chain 512 .. 1, colour + alpha", [(512 >> k, 512 >> k) for k in range(10)], True),
         ("four 256x256-block slices", [(256, 256)] * 4, False)]
for name, dims, alpha in cases:
    f, _, _ = bb.etc1s_file(np.random.default_rng(45), dims, n_codebook=4096, alpha=alpha)
    out = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, f)[1])
    res = []
    for env in (None, "BU_ETC1S_ONE_THREAD", "BU_ETC1S_ONE_LAUNCH"):
        if env: os.environ[env] = "1"
        for _ in range(3): bu.read_to_rgba(f, ctx, out=out)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); bu.read_to_rgba(f, ctx, out=out); ts.append(time.perf_counter() - t0)
        if env: os.environ.pop(env)
        res.append(sorted(ts)[7] * 1e3)
    ctx.host_free(out)
    print("%-40s two threads per large slice %.3f ms | one thread per slice %.3f ms | one launch after all decodes %.3f ms" % (name, *res), flush=True)
if os.environ.get("SPLIT_TRACE"):
    os.environ["BU_TRACE"] = "1"
    for name, dims, alpha in cases[1:3]:
        f, _, _ = bb.etc1s_file(np.random.default_rng(45), dims, n_codebook=4096, alpha=alpha)
        out = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, f)[1])
        for env in (None, "BU_ETC1S_ONE_THREAD"):
            if env: os.environ[env] = "1"
            bu.read_to_rgba(f, ctx, out=out)
            sys.stderr.flush()
            print("--- %s %s" % (name, env or "two threads"), file=sys.stderr, flush=True)
            bu.read_to_rgba(f, ctx, out=out)
            if env: os.environ.pop(env)
        ctx.host_free(out)
