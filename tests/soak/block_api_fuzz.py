"""soak checker (run by hand on the GPU box, not collected by pytest): the per-block API on its default route -- the library's own block
code on the calling thread -- against the oracle, random-valid and raw random blocks, every target, statuses included.
    FUZZ_BLOCKS=200000 python tests/soak/block_api_fuzz.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from basisu_rs_amd import Context, _lib, synth
from oracle.pyoracle import Oracle
ctx = Context(0); o = Oracle(); lib = _lib.load()
n = int(os.environ.get("FUZZ_BLOCKS", 200000))
fns = {"astc": (lib.bu_transcode_uastc_block_to_astc, 16), "bc7": (lib.bu_transcode_uastc_block_to_bc7, 16), "etc1": (lib.bu_transcode_uastc_block_to_etc1, 8),
       "etc2": (lib.bu_transcode_uastc_block_to_etc2, 16), "rgba": (lib.bu_unpack_uastc_block_to_rgba, 64)}
t0 = time.time()
for kind in ("valid", "raw", "contrast"):
    blocks = (synth.atlas_rand(n, seed=41) if kind == "valid" else synth.atlas_contrast(n, seed=42) if kind == "contrast"
              else np.random.Generator(np.random.PCG64(43)).integers(0, 256, size=(n, 16), dtype=np.uint8))
    blocks = np.ascontiguousarray(blocks)
    for t, (fn, bb) in fns.items():
        want, st = o.batch(t, blocks)
        got = np.zeros((n, bb), dtype=np.uint8)
        sts = np.zeros(n, dtype=np.int32)
        base_in, base_out, h = blocks.ctypes.data, got.ctypes.data, ctx.handle
        for i in range(n):
            sts[i] = fn(h, base_in + 16 * i, base_out + bb * i)
        assert (sts == st).all(), (kind, t, np.where(sts != st)[0][:5])
        ok = st == 0
        assert (got[ok] == want[ok]).all(), (kind, t)
    print(kind, "ok: %d blocks x 5 targets, %d with an error status, %.0f s" % (n, int((st != 0).sum()), time.time() - t0), flush=True)
