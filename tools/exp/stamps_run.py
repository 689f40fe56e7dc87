"""EXPERIMENT: in-kernel phase stamps (s_memtime) of the sorted BC7 kernel: python stamps_run.py VARIANT LOG2_BLOCKS WAVES_PER_WG TILE"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
variant, lg, wpw, tile = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", os.environ.get("BU_EXP_LIB", "libbu_exp.so")))
vp = ctypes.c_void_p
lib.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
lib.bu_exp_time.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
lib.bu_exp_set_stamps.argtypes = [vp]
h = vp(); assert lib.bu_context_create(0, ctypes.byref(h)) == 0
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << lg; NBUF = 64
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
print("variant %d, 2^%d blocks: %.2f us per launch (no stamps)" % (variant, lg, best))
n_wg = (N + tile - 1) // tile
nw = n_wg * wpw
buf = torch.zeros(nw * 32 + 64, dtype=torch.int64, device=dev)  # 32 words per wave: [0,16) first tile, [16,32) the second one (BU_STAMP)
lib.bu_exp_set_stamps(vp(buf.data_ptr()))
# one launch on a cold buffer, preceded by a few others so that clocks are up
lib.bu_exp_time(h, variant, ip, op, NBUF, N, 5, sp, ctypes.byref(ms))
torch.cuda.synchronize()
lib.bu_exp_set_stamps(None)
allraw = buf.cpu().numpy()[: nw * 32].reshape(nw, 32)
BLK = int(os.environ.get("STAMP_BLOCK", 0))  # 0: the first pass through the tile loop, 1: the second one
raw = allraw[:, 16 * BLK: 16 * BLK + 16].copy()
if BLK: raw[:, [0, 1, 9, 11]] = allraw[:, [0, 1, 9, 11]]  # stamps 0, 1, 11 and the start time are taken once, before the loop
s = raw[:, :9].astype(np.float64)
rt = (raw[:, 10] - raw[:, 9]).astype(np.float64)  # s_memrealtime ticks (100 MHz) over the wave's life
ok = rt > 0
print("shader clock from s_memtime / s_memrealtime over a wave's life: %.0f MHz (median), wave life %.2f us (median), %.2f us (max)" % (
    np.median((s[ok, 8] - s[ok, 0]) / rt[ok]) * 100, np.median(rt[ok]) / 100, rt[ok].max() / 100))
print("kernel span by s_memrealtime: %.2f us" % ((raw[:, 10].max() - raw[:, 9].min()) / 100))
names = ["start", "tables+loads", "A done", "bar1", "B done(bar2)", "scatter(bar3)", "C done", "bar4", "end"]
t0 = s[:, 0].min()
print("shader clocks since the first wave started: mean / min / p50 / max   (per-wave delta mean)")
for k in range(9):
    a = s[:, k] - t0
    d = (s[:, k] - s[:, k - 1]).mean() if k else 0
    print("  %-14s %8.0f %8.0f %8.0f %8.0f   (%6.0f)" % (names[k], a.mean(), a.min(), np.median(a), a.max(), d))
print("kernel span first start -> last end: %.0f clocks" % (s[:, 8].max() - t0))

st_us = (raw[:, 9] - raw[:, 9].min()) / 100.0
en_us = (raw[:, 10] - raw[:, 9].min()) / 100.0
print("wave START times (us after the first wave): p1 %.2f p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(st_us, [1, 10, 25, 50, 75, 90, 99, 100])))
print("wave END   times (us after the first wave): p1 %.2f p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(en_us, [1, 10, 25, 50, 75, 90, 99, 100])))
wg = np.arange(nw) // wpw
for q in range(4):
    sel = (wg // 256) == q
    if sel.any():
        print("  CU slot %d (blockIdx %4d..%4d): start p50 %.2f max %.2f | end p50 %.2f max %.2f | life p50 %.2f us" % (
            q, q * 256, q * 256 + 255, np.median(st_us[sel]), st_us[sel].max(), np.median(en_us[sel]), en_us[sel].max(), np.median(en_us[sel] - st_us[sel])))
# phase boundaries in real time for the median wave: scale shader clocks by the measured clock

# per-slot Gantt: median real time (us after the first wave) at which waves of each CU slot pass each stamp
mhz = np.median((s[ok, 8] - s[ok, 0]) / rt[ok]) * 100
t11 = st_us + (raw[:, 11].astype(np.float64) - s[:, 0]) / mhz
t12 = st_us + (raw[:, 12].astype(np.float64) - s[:, 0]) / mhz
for q in range(4):
    sel = (wg // 256) == q
    if sel.any():
        print("  slot %d: tables in LDS (before barrier 0) p50 %.2f p90 %.2f | barrier 0 passed p50 %.2f | keys known (tile data arrived) p10 %.2f p50 %.2f p90 %.2f | A done p50 %.2f" % (
            q, np.median(t11[sel]), np.percentile(t11[sel], 90), np.median(st_us[sel] + (s[sel, 1] - s[sel, 0]) / mhz), np.percentile(t12[sel], 10), np.median(t12[sel]), np.percentile(t12[sel], 90),
            np.median(st_us[sel] + (s[sel, 2] - s[sel, 0]) / mhz)))
t_us = st_us[:, None] + (s - s[:, :1]) / mhz
print("median time (us) at each stamp, per CU slot:   " + "  ".join("%-6s" % n[:6] for n in names))
for q in range(4):
    sel = (wg // 256) == q
    if sel.any():
        print("  slot %d                                       " % q + "  ".join("%6.2f" % np.median(t_us[sel, k]) for k in range(9)))
