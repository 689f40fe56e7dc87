"""EXPERIMENT: does the allocator of the OUTPUT buffer matter at 2^25 blocks per launch?  torch caching allocator vs hipMalloc (bu_device_alloc)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 25
gu = torch.from_numpy(g["uastc"]).to(dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(ins, out_ptrs, nb, launches=64):
    A = ctypes.c_void_p * nb
    ip, op = A(*[x.data_ptr() for x in ins]), A(*out_ptrs)
    ms = ctypes.c_float(0)
    lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, nb, 0, N, 256, 400, None, sp, ctypes.byref(ms))
    best = 1e9
    for _ in range(3):
        lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, nb, 0, N, 256, launches, None, sp, ctypes.byref(ms))
        best = min(best, ms.value / launches * 1e3)
    return best
for nb in (2, 8):
    ins = []
    for k in range(nb):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        idx = torch.randint(0, 608, (1 << 22,), device=dev, generator=gen)
        ins.append(gu[idx].repeat(8, 1).contiguous())
    outs_t = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(nb)]
    print("nbuf %d torch outputs:    %.1f us" % (nb, t(ins, [x.data_ptr() for x in outs_t], nb)), flush=True)
    del outs_t; torch.cuda.empty_cache()
    raw = []
    for k in range(nb):
        p = ctypes.c_void_p(0); assert lib.bu_device_alloc(ctx.handle, N * 16, ctypes.byref(p)) == 0; raw.append(p.value)
    print("nbuf %d hipMalloc outputs: %.1f us" % (nb, t(ins, raw, nb)), flush=True)
    for p in raw: lib.bu_device_free(ctx.handle, ctypes.c_void_p(p))
    del ins; torch.cuda.empty_cache()
