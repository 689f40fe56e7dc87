"""EXPERIMENT / check: ETC1, ETC2 and BC7 against the known answers at slice sizes that exercise the run-time tile size
(not multiples of 64, 1024 or 4096), and the first-error index in the last tile"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
from oracle.pyoracle import Oracle
ctx = Context(0); o = Oracle()
ctx.set_launch_policy({"shared": True, "auto": "auto"}.get(os.environ.get("FUZZ_POLICY"), False))  # FUZZ_POLICY=shared | auto | (exclusive)  # FUZZ_POLICY=shared: the half-CU shapes (fixed 1024- / 2048-block tiles, ragged tails)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
for n in (262145, 300001, 524289, 786433, 851968, 1000000, 1234567, 1572865, 3000001, 5000003):
    idx = synth.gold_indices(n, seed=n)
    blocks = g["uastc"][idx]
    for name, fmt in (("etc1", _lib.ETC1), ("etc2", _lib.ETC2), ("bc7", _lib.BC7), ("astc", _lib.ASTC)):
        got = ctx.transcode(fmt, blocks).reshape(n, -1)
        want = g[name][idx]
        assert (got == want).all(), (n, name, np.where((got != want).any(axis=1))[0][:5])
    # an error in the last tile must be reported with its index
    bad = blocks.copy(); bad[n - 3, 0] = 0xFF  # invalid mode code?
    want, st = o.batch("etc1", bad[n - 8:])
    if (st != 0).any():
        try:
            ctx.transcode(_lib.ETC1, bad); raise SystemExit("expected an error")
        except Exception as e:
            assert getattr(e, "first_bad_block", None) == n - 8 + int(np.where(st != 0)[0][0]), (n, e)
    print(n, "ok", flush=True)
