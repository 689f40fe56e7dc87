"""EXPERIMENT: whole-file read_to_bc7 of a 4096^2 UASTC file (16 MiB) into a page-locked output: piece size of the upload pipeline"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, basisu_rs_amd as bu
from basisu_rs_amd import synth
g = synth.load_golden(os.path.join(ROOT, "tests/golden/uastc_kat.bin")); ctx = bu.Context(0)
idx = synth.gold_indices(1 << 20)
f = bu.write_uastc_file([dict(data=g["uastc"][idx], orig_w=4096, orig_h=4096, nbx=1024, nby=1024)])
out = ctx.host_alloc(16 << 20)
for piece in ("16", "8", "4", "2", "1"):
    os.environ["BU_RUN_PIECE_MIB"] = piece
    bu.read_to_bc7(f, ctx, out=out)
    ts = []
    for i in range(15):
        t0 = time.perf_counter(); imgs = bu.read_to_bc7(f, ctx, out=out); ts.append(time.perf_counter() - t0)
    ok = (np.asarray(imgs[0].data).reshape(-1, 16) == g["bc7"][idx]).all()
    print("piece %2s MiB: median %.3f ms  min %.3f  %s" % (piece, sorted(ts)[7] * 1e3, min(ts) * 1e3, "ok" if ok else "WRONG"))
