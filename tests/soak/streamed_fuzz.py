"""soak checker (run by hand on the GPU box, not collected by pytest): damaged ETC1S files through the streamed front door -- every
large slice on two host threads (default), on one thread (BU_ETC1S_ONE_THREAD), and the one-launch path (BU_ETC1S_ONE_LAUNCH) --
against the oracle: same status from all four, same images where the file is still accepted.  Bit flips anywhere behind the header,
byte stomps, truncations; sealed (CRCs recomputed, the damage reaches the decoders) and unsealed (the CRC must win)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import basis_builder as bb
import basisu_rs_amd as bu
from basisu_rs_amd import BasisuError, Context
from oracle.pyoracle import Oracle
ctx = Context(0); o = Oracle()
N = int(os.environ.get("FUZZ_FILES", 400))
bases = [("colour + alpha 256x160, 64x64", bb.etc1s_file(np.random.default_rng(1), [(256, 160), (64, 64)], n_codebook=700, alpha=True)[0]),
         ("mip chain 256 .. 1", bb.etc1s_file(np.random.default_rng(2), [(256 >> k, 256 >> k) for k in range(9)], n_codebook=2048, history_size=8)[0]),
         ("video 200x170 x 3 frames", bb.etc1s_file(np.random.default_rng(3), [(200, 170)] * 3, n_codebook=300, is_video=True)[0]),
         ("one column 1x33000", bb.etc1s_file(np.random.default_rng(4), [(1, 33000)], n_codebook=256)[0])]
def product(f, env):
    if env: os.environ[env] = "1"
    try:
        res = {}
        for name, fn in (("rgba", lambda: bu.read_to_rgba(f, ctx)[1]), ("etc1", lambda: bu.read_to_etc1(f, ctx))):
            try:
                res[name] = (0, [g.data.tobytes() for g in fn()])
            except BasisuError as e:
                res[name] = (e.status, None)
        return res
    finally:
        if env: os.environ.pop(env)
t0 = time.time(); seen = {}
for bname, base in bases:
    hdr = bu.read_header(base)
    rng = np.random.default_rng(len(base))
    for it in range(N):
        g = bytearray(base)
        kind = it % 4
        if kind == 0:
            for _ in range(1 + int(rng.integers(0, 3))):
                g[int(rng.integers(77, len(g)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            p = int(rng.integers(hdr.endpoint_cb_file_ofs, len(g))); g[p] = int(rng.integers(0, 256))
        elif kind == 2:
            g = g[: int(rng.integers(100, len(g)))]
        else:  # inside the first slice: the two-thread decode's give-up path
            p = int(rng.integers(len(g) // 2, len(g))); g[p] ^= 0xFF
        for sealed in (True, False):
            h = bb.reseal(bytes(g)) if sealed else bytes(g)
            want = {t: o.read_to(t, h) for t in ("rgba", "etc1")}
            for env in (None, "BU_ETC1S_ONE_THREAD", "BU_ETC1S_ONE_LAUNCH"):
                got = product(h, env)
                for t in ("rgba", "etc1"):
                    st, _, imgs = want[t]
                    assert got[t][0] == st, (bname, it, sealed, env, t, got[t][0], st)
                    if st == 0:
                        assert got[t][1] == [d.tobytes() for (_, _, _, d) in imgs], (bname, it, sealed, env, t)
                    seen[st] = seen.get(st, 0) + 1
    print("%s: %d damaged files x sealed/unsealed x 3 paths x 2 targets ok, %.0f s" % (bname, N, time.time() - t0), flush=True)
print("statuses seen:", dict(sorted(seen.items())))
