"""experiment driver (GPU box): time kernel variants on cold-rotated A-gold atlases + a wave-uniform atlas"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libbu_exp.so"))
vp = ctypes.c_void_p
lib.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
lib.bu_exp_time.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
h = vp(); assert lib.bu_context_create(0, ctypes.byref(h)) == 0
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 64
gu = torch.from_numpy(g["uastc"]).to(dev)
def mk(kind):
    ins = []
    for k in range(NBUF):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        if kind == "gold":
            idx = torch.randint(0, 608, (N,), device=dev, generator=gen)
        else:  # every aligned run of 64 blocks shares one mode
            mode = torch.randint(0, 19, (N // 64,), device=dev, generator=gen).repeat_interleave(64)
            idx = mode * 32 + torch.randint(0, 32, (N,), device=dev, generator=gen)
        ins.append(gu[idx].contiguous())
    return ins
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
sp = vp(torch.cuda.current_stream().cuda_stream)
def t(variant, ins, launches=256):
    A = vp * NBUF
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    lib.bu_exp_time(h, variant, ip, op, NBUF, N, 32, sp, ctypes.byref(ms))
    best = 1e9
    for _ in range(3):
        assert lib.bu_exp_time(h, variant, ip, op, NBUF, N, launches, sp, ctypes.byref(ms)) == 0
        best = min(best, ms.value / launches * 1e3)
    return best
gold = mk("gold")
names = {0: "WG256 BPT4 lds-out", 10: "WG256 BPT4 direct", 11: "WG512 BPT4 lds-out", 12: "WG512 BPT4 direct", 13: "WG1024 BPT4 lds-out",
         14: "WG1024 BPT4 direct", 15: "WG256 BPT8 direct", 16: "WG512 BPT2 direct"}
for v in sorted(names):
    print("%-28s %8.2f us" % (names[v], t(v, gold)), flush=True)
