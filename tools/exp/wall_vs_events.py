"""EXPERIMENT: where does the wall clock around K = 20 launches exceed the event time?"""
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
ms = ctypes.c_float(0)
lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, NBUF, 0, N, 1024, 3000, None, sp, ctypes.byref(ms))
for K in (1, 5, 20, 100):
    w, e = [], []
    for rep in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, NBUF, rep * K, N, 1024, K, None, sp, ctypes.byref(ms))
        torch.cuda.synchronize()
        w.append((time.perf_counter() - t0) * 1e6); e.append(ms.value * 1e3)
    w.sort(); e.sort()
    print("K=%3d  wall median %.1f us (%.2f/step)  events median %.1f us (%.2f/step)  overhead %.1f us" % (K, w[10], w[10] / K, e[10], e[10] / K, w[10] - e[10]))
