"""EXPERIMENT: kernel time of every UASTC target over slice sizes (powers of two and 1.5 x), looking for shape-policy cliffs:
us per launch, cold rotation; ns per block in brackets"""
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
targets = [("bc7", _lib.BC7), ("astc", _lib.ASTC), ("etc1", _lib.ETC1), ("etc2", _lib.ETC2), ("rgba", _lib.RGBA32)]
sizes = []
for lg in range(int(os.environ.get("LG_LO", 10)), int(os.environ.get("LG_HI", 23))):
    sizes += [1 << lg, 3 << (lg - 1)]
for N in sizes:
    nbuf = max(2, min(64, (1 << 29) // (N * 16)))
    ins = []
    for k in range(nbuf):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
    row = []
    for nm, t in targets:
        ob = 64 if nm == "rgba" else 16
        no = min(nbuf, max(2, (1 << 30) // (N * ob)))
        outs = [torch.empty((N, ob), dtype=torch.uint8, device=dev) for _ in range(no)]
        ip = (ctypes.c_void_p * no)(*[ins[k].data_ptr() for k in range(no)])
        op = (ctypes.c_void_p * no)(*[x.data_ptr() for x in outs])
        ms = ctypes.c_float(0)
        launches = max(16, min(512, (1 << 27) // N))
        bpr = 1024 if N % 1024 == 0 else N
        lib.bu_time_uastc_launches(ctx.handle, t, ip, op, no, 0, N, bpr, launches, None, sp, ctypes.byref(ms))
        best = 1e9
        for _ in range(3):
            lib.bu_time_uastc_launches(ctx.handle, t, ip, op, no, 0, N, bpr, launches, None, sp, ctypes.byref(ms))
            best = min(best, ms.value / launches * 1e3)
        row.append("%s %8.2f (%5.2f)" % (nm, best, best * 1e3 / N))
        del outs
    print("%9d blocks  " % N + "  ".join(row), flush=True)
    del ins
