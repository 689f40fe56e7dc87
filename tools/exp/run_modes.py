"""how does the sorted BC7 kernel's time depend on the NUMBER of distinct mode paths in the atlas? (I-cache hypothesis)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 48
gu = torch.from_numpy(g["uastc"]).to(dev)
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(modes, launches=200):
    ins = []
    mt = torch.tensor(modes, device=dev)
    for k in range(NBUF):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        m = mt[torch.randint(0, len(modes), (N,), device=dev, generator=gen)]
        ins.append(gu[m * 32 + torch.randint(0, 32, (N,), device=dev, generator=gen)].contiguous())
    A = ctypes.c_void_p * NBUF
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    lib.bu_time_uastc_launches(ctx.handle, int(os.environ.get("TARGET", _lib.BC7)), ip, op, NBUF, 0, N, 1024, 32, None, sp, ctypes.byref(ms))
    best = 1e9
    for _ in range(3):
        lib.bu_time_uastc_launches(ctx.handle, int(os.environ.get("TARGET", _lib.BC7)), ip, op, NBUF, 0, N, 1024, launches, None, sp, ctypes.byref(ms))
        best = min(best, ms.value / launches * 1e3)
    return best
base = None
for modes in [[m] for m in range(19)] + [list(range(19))]:
    t = run(modes)
    print("%-40s %7.2f us" % (str(modes), t), flush=True)
