"""stress of the streamed ETC1S front door (bu_read_etc1s_streamed): thousands of calls on small and large files, pageable and
page-locked outputs, while a watchdog thread reports a call that does not return (a deadlock would otherwise hang silently)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import basis_builder as bb
import basisu_rs_amd as bu
from basisu_rs_amd import _lib, Context
ctx = Context(0)
state = {"i": -1, "t": time.time(), "what": ""}
def dog():
    while True:
        time.sleep(5)
        if time.time() - state["t"] > 20:
            print("STUCK in call %d (%s) for %.0f s" % (state["i"], state["what"], time.time() - state["t"]), flush=True)
            os._exit(7)
threading.Thread(target=dog, daemon=True).start()
files = [("small 256x160", bb.etc1s_file(np.random.default_rng(901), [(256, 160)], n_codebook=1024)[0], 1500),
         ("five slices", bb.etc1s_file(np.random.default_rng(902), [(96, 96)] * 5, n_codebook=1024)[0], 800),
         ("alpha pairs", bb.etc1s_file(np.random.default_rng(903), [(192, 192), (64, 64)], n_codebook=1024, alpha=True)[0], 500),
         ("config 4", bb.etc1s_file(np.random.default_rng(45), [(512, 512)], n_codebook=4096)[0], 200)]
# a wide pool job first (parks many threads), as the test suite does before it reaches these calls
wide = bb.etc1s_file(np.random.default_rng(7), [(64, 64)] * 40, n_codebook=512)[0]
os.environ["BU_ETC1S_ONE_LAUNCH"] = "1"; bu.read_to_rgba(wide, ctx); os.environ.pop("BU_ETC1S_ONE_LAUNCH")
n = 0
for name, f, reps in files:
    pinned = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, f)[1])
    want = bu.read_to_rgba(f, ctx)[1][0].data.tobytes()
    t0 = time.time()
    for r in range(reps):
        state.update(i=n, t=time.time(), what=name); n += 1
        if r % 3 == 0: got = bu.read_to_rgba(f, ctx, out=pinned)[1]
        elif r % 3 == 1: got = bu.read_to_rgba(f, ctx)[1]
        else: got = bu.read_to_etc1(f, ctx); continue
        assert got[0].data.tobytes() == want
    print("%s: %d calls ok, %.3f ms per call" % (name, reps, (time.time() - t0) / reps * 1e3), flush=True)
    ctx.host_free(pinned)
print("stress ok", flush=True)
