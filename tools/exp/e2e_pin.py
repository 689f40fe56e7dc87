"""EXPERIMENT: host-pointer UASTC->BC7 of one 4096^2 atlas, pageable vs page-locked, piece size / stream count sweep."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from basisu_rs_amd import Context, _lib, synth
    ctx = Context(0)
    g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
    N = 1 << 20
    idx = synth.gold_indices(N)
    host_in = g["uastc"][idx]
    pin_in, pin_out = ctx.host_alloc(N * 16), ctx.host_alloc(N * 16)
    pin_in[:] = host_in.reshape(-1)
    def t(fn, reps=8):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps * 1e3
    if sys.argv[1] == "pageable":
        print("pageable           %.3f ms" % t(lambda: ctx.transcode(_lib.BC7, host_in)))
        out = np.empty(N * 16, dtype=np.uint8)
        print("pageable, out=     %.3f ms" % t(lambda: ctx.transcode(_lib.BC7, host_in, out=out)))
    else:
        ms = t(lambda: ctx.transcode(_lib.BC7, pin_in, out=pin_out))
        ok = (pin_out.reshape(-1, 16) == g["bc7"][idx]).all()
        print("pinned piece=%s streams=%s  %.3f ms ok=%s" % (os.environ.get("BU_PIPE_PIECE"), os.environ.get("BU_PIPE_STREAMS"), ms, ok))
else:
    subprocess.call([sys.executable, __file__, "pageable"])
    for piece in (1 << 20, 1 << 19, 1 << 18, 1 << 17, 1 << 16):
        for ns in (1, 2, 3, 4):
            env = dict(os.environ, BU_PIPE_PIECE=str(piece), BU_PIPE_STREAMS=str(ns))
            subprocess.call([sys.executable, __file__, "pinned"], env=env)
