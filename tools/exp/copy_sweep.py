"""EXPERIMENT: the copy kernel's launch-to-launch time vs size: what is the fixed per-launch floor?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from basisu_rs_amd import Context, _lib
ctx = Context(0); lib = _lib.load()
dev = torch.device("cuda", 0)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in (6, 11, 14, 16, 18, 19, 20, 21, 22):
    N = 1 << lg
    nbuf = max(2, min(64, (1 << 30) // (N * 16)))
    ins = [torch.randint(0, 255, (N, 16), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    A = ctypes.c_void_p * nbuf
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    launches = 256
    lib.bu_time_copy_launches(ctx.handle, ip, op, nbuf, 0, N, 16, sp, ctypes.byref(ms))
    best = 1e9
    for _ in range(3):
        lib.bu_time_copy_launches(ctx.handle, ip, op, nbuf, 0, N, launches, sp, ctypes.byref(ms))
        best = min(best, ms.value / launches * 1e3)
    print("copy 2^%-2d blocks  %8.2f us  %7.1f GB/s" % (lg, best, 32 * N / best / 1e3), flush=True)
