"""config 4 end to end (one 512 x 512-block ETC1S slice through read_to_rgba): phase times of bu_read_to (BU_TRACE) + totals"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import basis_builder as bb
import basisu_rs_amd as bu
from basisu_rs_amd import _lib, Context
ctx = Context(0)
fone, _, _ = bb.etc1s_file(np.random.default_rng(45), [(512, 512)], n_codebook=4096)
out = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, fone)[1])
for which, fn in (("rgba", bu.read_to_rgba), ("etc1", bu.read_to_etc1)):
    o = out if which == "rgba" else ctx.host_alloc(bu.read_query(_lib.READ_ETC1, fone)[1])
    for _ in range(3): fn(fone, ctx, out=o)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); fn(fone, ctx, out=o); ts.append(time.perf_counter() - t0)
    print("read_to_%s: median %.3f ms, min %.3f ms" % (which, sorted(ts)[7] * 1e3, min(ts) * 1e3), flush=True)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); bu.basislz_decode(fone, 0); ts.append(time.perf_counter() - t0)
print("basislz_decode (host only): median %.3f ms min %.3f" % (sorted(ts)[4] * 1e3, min(ts) * 1e3), flush=True)
os.environ["BU_TRACE"] = "1"
sys.stderr.flush()
bu.read_to_rgba(fone, ctx, out=out)
bu.read_to_rgba(fone, ctx, out=out)
