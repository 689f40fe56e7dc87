"""EXPERIMENT: per-tile-iteration phase time line of a persistent shape (two tiles per workgroup):
python stamps2_run.py VARIANT WAVES_PER_WG N_WG"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
variant, wpw, n_wg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libbu_exp.so"))
vp = ctypes.c_void_p
lib.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
lib.bu_exp_time.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
lib.bu_exp_set_stamps.argtypes = [vp]
h = vp(); assert lib.bu_context_create(0, ctypes.byref(h)) == 0
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 64
gu = torch.from_numpy(g["uastc"]).to(dev)
gold = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    gold.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
sp = vp(torch.cuda.current_stream().cuda_stream)
A = vp * NBUF
ip, op = A(*[x.data_ptr() for x in gold]), A(*[x.data_ptr() for x in outs])
ms = ctypes.c_float(0)
lib.bu_exp_time(h, variant, ip, op, NBUF, N, 32, sp, ctypes.byref(ms))
best = 1e9
for _ in range(3):
    assert lib.bu_exp_time(h, variant, ip, op, NBUF, N, 256, sp, ctypes.byref(ms)) == 0
    best = min(best, ms.value / 256 * 1e3)
print("variant %d: %.2f us per launch (no stamps)" % (variant, best))
nw = n_wg * wpw
buf = torch.zeros(nw * 32 + 64, dtype=torch.int64, device=dev)
lib.bu_exp_set_stamps(vp(buf.data_ptr()))
lib.bu_exp_time(h, variant, ip, op, NBUF, N, 5, sp, ctypes.byref(ms))
torch.cuda.synchronize()
lib.bu_exp_set_stamps(None)
raw = buf.cpu().numpy()[: nw * 32].reshape(nw, 32).astype(np.float64)
# clock from the first iteration's realtime pair: stamp 0 (start, rt at 9) .. stamp 8 of iteration 0 (rt at 10)
rt0, rt1 = raw[:, 9], raw[:, 10]
mhz = np.median((raw[:, 8] - raw[:, 0]) / np.maximum(rt1 - rt0, 1)) * 100
start_us = (rt0 - rt0.min()) / 100.0
names = ["start", "tables", "A", "bar1", "B+scat", "bar2", "C", "bar3", "D/end"]
wg = np.arange(nw) // wpw
gen = wg // 256
print("clock %.0f MHz.  median time (us after the first wave) per generation and tile iteration:" % mhz)
print("                 " + "  ".join("%-6s" % n for n in names))
for q in sorted(set(gen)):
    sel = gen == q
    for it in range(2):
        t = start_us[sel, None] + (raw[sel][:, [0, 1] + [16 * it + k for k in range(2, 9)]] - raw[sel][:, :1]) / mhz
        if it == 1: t[:, :2] = np.nan
        print("  gen %d tile %d    " % (q, it) + "  ".join("%6.2f" % np.nanmedian(t[:, k]) if not np.all(np.isnan(t[:, k])) else "      " for k in range(9)))
