"""EXPERIMENT: K = 20 timed launches right after a synchronize (the driver's bench command) against long runs: where do the extra
0.8 us per launch come from?  event pair around 20 launches, repeated; prewarm between or not"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 64
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = ctypes.c_void_p * NBUF
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
ms = ctypes.c_float(0); rot = [0]
def run(L):
    lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, NBUF, rot[0] % NBUF, N, 1024, L, None, sp, ctypes.byref(ms)); rot[0] += L
    return ms.value / L * 1e3
def warm(ms_):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms_: run(256)
for label, pre, gap in (("25 ms prewarm, 5 warm-up, sync, 20 timed", 25, 0), ("100 ms prewarm", 100, 0), ("25 ms prewarm, then 1 ms host pause", 25, 1e-3), ("no prewarm", 0, 0)):
    res = []
    for rep in range(6):
        time.sleep(0.05)
        if pre: warm(pre)
        run(5); torch.cuda.synchronize()
        if gap: time.sleep(gap)
        torch.cuda.synchronize()
        res.append(run(20))
    print("%-45s %s" % (label, " ".join("%.2f" % r for r in res)), flush=True)
print("K = 512 right after the same:               %.2f" % (warm(25), run(5), torch.cuda.synchronize(), run(512))[3])
each = (ctypes.c_float * 20)()
warm(25); run(5); torch.cuda.synchronize()
lib.bu_time_uastc_launches_each(ctx.handle, _lib.BC7, ip, op, NBUF, rot[0] % NBUF, N, 1024, 20, None, sp, each)
print("per launch (event to event):", " ".join("%.1f" % e for e in each))
# keep the GPU busy up to the synchronize in front of the timed window: untimed launches on a side stream
side = torch.cuda.Stream(device=dev); ssp = ctypes.c_void_p(side.cuda_stream)
for n_async in (0, 50, 200, 800):
    res = []
    for rep in range(8):
        time.sleep(0.05)
        warm(25); run(5)
        for i in range(n_async):
            k = (rot[0] + i) % NBUF
            lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(ip[k]), N, ctypes.c_void_p(op[k]), 1024, 0, None, ssp)
        torch.cuda.synchronize()
        res.append(run(20))
    print("%4d async launches before the synchronize:  %s   mean %.2f" % (n_async, " ".join("%.2f" % r for r in res), sum(sorted(res)[:7]) / 7), flush=True)
