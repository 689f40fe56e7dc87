"""EXPERIMENT: UASTC->BC7 kernel time vs slice size (cold rotation over enough buffers to exceed the 256 MiB Infinity Cache)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0)
gu = torch.from_numpy(g["uastc"]).to(dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
target = int(os.environ.get("TARGET", _lib.BC7))
for lg in range(int(os.environ.get("LG_LO", 11)), int(os.environ.get("LG_HI", 26))):
    N = 1 << lg
    nbuf = max(2, min(64, (1 << 30) // (N * 16)))
    ins, outs = [], []
    for k in range(nbuf):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        idx = torch.randint(0, 608, (N,), device=dev, generator=gen)
        ins.append(torch.cat([gu[idx[lo:lo + (1 << 22)]] for lo in range(0, N, 1 << 22)]).contiguous())
        outs.append(torch.empty((N, 16), dtype=torch.uint8, device=dev))
    A = ctypes.c_void_p * nbuf
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    launches = max(8, min(256, (1 << 28) // N))
    lib.bu_time_uastc_launches(ctx.handle, target, ip, op, nbuf, 0, N, 1024, 8, None, sp, ctypes.byref(ms))
    best = 1e9
    for _ in range(3):
        lib.bu_time_uastc_launches(ctx.handle, target, ip, op, nbuf, 0, N, 1024, launches, None, sp, ctypes.byref(ms))
        best = min(best, ms.value / launches * 1e3)
    print("2^%-2d blocks  %9.2f us  %7.1f GB/s  %8.1f Mblocks/s  (nbuf %d)" % (lg, best, 32 * N / best / 1e3, N / best, nbuf), flush=True)
    del ins, outs
