"""experiment driver (GPU box): time kernel variants / collect in-kernel stamps on cold-rotated A-gold atlases"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
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
ONLY = os.environ.get("ONLY_MODE")
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    if ONLY is None:
        gold.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
    else:
        gold.append(gu[int(ONLY) * 32 + torch.randint(0, 32, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
sp = vp(torch.cuda.current_stream().cuda_stream)
A = vp * NBUF
ip, op = A(*[x.data_ptr() for x in gold]), A(*[x.data_ptr() for x in outs])
def t(variant, launches=256):
    ms = ctypes.c_float(0)
    lib.bu_exp_time(h, variant, ip, op, NBUF, N, 32, sp, ctypes.byref(ms))
    best = 1e9
    for _ in range(3):
        assert lib.bu_exp_time(h, variant, ip, op, NBUF, N, launches, sp, ctypes.byref(ms)) == 0
        best = min(best, ms.value / launches * 1e3)
    return best
def chunk_stamps(variant, nwaves_per_wg, n_wg):
    nw = n_wg * nwaves_per_wg
    buf = torch.zeros(nw * 16 + nw * 12 * 8, dtype=torch.int64, device=dev)
    lib.bu_exp_set_stamps(vp(buf.data_ptr()))
    ms = ctypes.c_float(0)
    lib.bu_exp_time(h, variant, ip, op, NBUF, N, 3, sp, ctypes.byref(ms))
    torch.cuda.synchronize()
    lib.bu_exp_set_stamps(None)
    allb = buf.cpu().numpy()
    s = allb[: nw * 16].reshape(nw, 16)[:, :9].astype(np.float64)
    names_ = ["start", "tables+loads", "A done", "bar1", "B done(bar2)", "scatter(bar3)", "C done", "bar4", "end"]
    print("per-wave phase deltas (shader clocks): mean / p10 / p50 / p90")
    for k in range(1, 9):
        d = s[:, k] - s[:, k - 1]
        print("  %-14s %8.0f %8.0f %8.0f %8.0f" % (names_[k], d.mean(), np.percentile(d, 10), np.median(d), np.percentile(d, 90)))
    rec = allb[nw * 16:].reshape(nw, 12, 8)
    t0, t1, m, t2 = rec[:, :, 0], rec[:, :, 2], rec[:, :, 3], rec[:, :, 4]
    ok = (t0 > 0) & (t2 > t0)
    print("chunks recorded", ok.sum(), "per wave", ok.sum() / nw)
    if ok.sum() == 0:
        return
    fetch = (t1 - t0)[ok]; comp = (t2 - t1)[ok]; mm = m[ok]
    print("fetch (desc + block read): mean %.0f p50 %.0f p90 %.0f" % (fetch.mean(), np.median(fetch), np.percentile(fetch, 90)))
    print("transcode: mean %.0f p50 %.0f p90 %.0f" % (comp.mean(), np.median(comp), np.percentile(comp, 90)))
    for k in range(20):
        sel = mm == k
        if sel.any():
            print("  mode %2d: n %5d  transcode mean %6.0f  p50 %6.0f" % (k, sel.sum(), comp[sel].mean(), np.median(comp[sel])))
    gaps = []
    for w in range(0, nw, 5):
        for k in range(11):
            if ok[w, k] and ok[w, k + 1]:
                gaps.append(rec[w, k + 1, 0] - rec[w, k, 4])
    gaps = np.array(gaps); print("gap end-of-transcode -> next chunk start (result write + atomic grab): mean %.0f p50 %.0f" % (gaps.mean(), np.median(gaps)))
print("1024x4 time", t(13))
chunk_stamps(13, 16, 256)
