"""EXPERIMENT: read_to_bc7 on a texture-array file (config 5 shape, scaled): N slices of 256x256 blocks"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import basisu_rs_amd as bu
from basisu_rs_amd import synth, _lib
ctx = bu.Context(0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 128
idx = synth.gold_indices(NS * 65536, seed=3).reshape(NS, 65536)
f = bu.write_uastc_file([dict(data=g["uastc"][idx[k]], orig_w=1024, orig_h=1024, nbx=256, nby=256, image_index=k) for k in range(NS)])
pin = ctx.host_alloc(bu.read_query(_lib.READ_BC7, f)[1])
imgs = bu.read_to_bc7(f, ctx, out=pin)
ok = all((np.asarray(imgs[k].data).reshape(-1, 16) == g["bc7"][idx[k]]).all() for k in (0, NS // 2, NS - 1))
t0 = time.perf_counter()
for _ in range(3): bu.read_to_bc7(f, ctx, out=pin)
dt = (time.perf_counter() - t0) / 3
print("%d slices, %.0f MiB file: %.2f ms per file, %.0f Mblocks/s, verified=%s" % (NS, len(f) / 2**20, dt * 1e3, NS * 65536 / dt / 1e6, ok))
os.environ["BU_TRACE"] = "1"
bu.read_to_bc7(f, ctx, out=pin)
