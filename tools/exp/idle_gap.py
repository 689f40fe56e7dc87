"""EXPERIMENT: how fast do the clocks fall when the GPU idles?  steady launches, a host-side pause, then 20 timed launches"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0)
gu = torch.from_numpy(g["uastc"]).to(dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg, NBUF, steps in ((20, 64, 20), (25, 4, 20)):
    N = 1 << lg
    ins = []
    for k in range(NBUF):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
    outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    A = ctypes.c_void_p * NBUF
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    def run(L):
        lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, NBUF, 0, N, 1024, L, None, sp, ctypes.byref(ms)); return ms.value / L * 1e3
    for gap_ms in (0, 0.05, 0.2, 1, 5, 20, 100):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15: run(64 if lg == 20 else 8)   # steady state
        if gap_ms: time.sleep(gap_ms / 1e3)
        print("2^%d blocks, idle %6.2f ms, then %d launches: %.2f us per launch" % (lg, gap_ms, steps, run(steps)), flush=True)
    del ins, outs
