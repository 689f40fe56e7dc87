"""EXPERIMENT: the copy kernel's launch-to-launch time vs size (what is the fixed per-launch floor?) with the BC7 transcode of the
same buffers beside it: how far from a plain copy is the transcode at every size?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
dev = torch.device("cuda", 0)
gu = torch.from_numpy(synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))["uastc"]).to(dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {"bc7": _lib.BC7}
for lg in (6, 11, 14, 16, 18, 19, 20, 21, 22, 23, 24, 25):
    N = 1 << lg
    nbuf = max(2, min(64, (1 << 31) // (N * 16)))
    ins = [gu[torch.randint(0, 608, (N,), device=dev)].contiguous() for _ in range(nbuf)]  # A-gold style: uniform mix of the 19 modes
    outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    A = ctypes.c_void_p * nbuf
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    launches = 256
    import time
    def warm(fn):  # (clocks: launches of 2^22 blocks and more take ~100 ms to settle them, as bench.py's prewarm)
        t0 = time.perf_counter()
        while True:
            fn()
            if N < (1 << 22) or time.perf_counter() - t0 > 0.12: break
    warm(lambda: lib.bu_time_copy_launches(ctx.handle, ip, op, nbuf, 0, N, 16, sp, ctypes.byref(ms)))
    best = 1e9
    for _ in range(3):
        lib.bu_time_copy_launches(ctx.handle, ip, op, nbuf, 0, N, launches, sp, ctypes.byref(ms))
        best = min(best, ms.value / launches * 1e3)
    launches = max(8, min(256, (1 << 28) // N))
    bb = 1e9
    warm(lambda: lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, nbuf, 0, N, 1024 if N >= 1024 * 16 else 0, 8, None, sp, ctypes.byref(ms)))
    for _ in range(3):
        lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, nbuf, 0, N, 1024 if N >= 1024 * 16 else 0, launches, None, sp, ctypes.byref(ms))
        bb = min(bb, ms.value / launches * 1e3)
    print("2^%-2d blocks  copy %9.2f us %7.1f GB/s (%.3f of 8 TB/s)   UASTC->BC7 %9.2f us %7.1f GB/s (%.3f)   BC7 / copy %.2f" % (lg, best, 32 * N / best / 1e3, 32 * N / best / 8e6, bb, 32 * N / bb / 1e3, 32 * N / bb / 8e6, bb / best), flush=True)
