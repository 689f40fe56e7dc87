"""EXPERIMENT: UASTC->RGBA32 launch time on a 4096^2 atlas (cold rotation over 16 buffer pairs = 1.25 GiB), verified"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 16
gu = torch.from_numpy(g["uastc"]).to(dev); gr = torch.from_numpy(g["rgba"]).to(dev)
ins, outs, idxs = [], [], []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    idx = torch.randint(0, 608, (N,), device=dev, generator=gen)
    ins.append(gu[idx].contiguous()); outs.append(torch.empty((N, 64), dtype=torch.uint8, device=dev)); idxs.append(idx)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = ctypes.c_void_p * NBUF
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
ms = ctypes.c_float(0)
lib.bu_time_uastc_launches(ctx.handle, _lib.RGBA32, ip, op, NBUF, 0, N, 1024, 16, None, sp, ctypes.byref(ms))
torch.cuda.synchronize()
img = outs[3].view(1024, 4, 1024, 16).permute(0, 2, 1, 3).reshape(N, 64)
ok = bool(torch.equal(img, gr[idxs[3]]))
best = 1e9
for _ in range(3):
    lib.bu_time_uastc_launches(ctx.handle, _lib.RGBA32, ip, op, NBUF, 0, N, 1024, 128, None, sp, ctypes.byref(ms))
    best = min(best, ms.value / 128 * 1e3)
print("rgba32 %.2f us  %.0f GB/s  ok=%s" % (best, 80 * N / best / 1e3, ok))
