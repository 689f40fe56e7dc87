"""Time series of consecutive multi-stream windows: python series.py LIB --streams 4 --policy 1 --lead 0 --launches 40 --calls 60
prints us per atlas of every call (to see bimodal behaviour, drifts with time, clock effects)."""
import argparse, ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("lib"); ap.add_argument("--streams", type=int, default=4); ap.add_argument("--policy", type=int, default=1)
ap.add_argument("--lead", type=int, default=0); ap.add_argument("--launches", type=int, default=40); ap.add_argument("--calls", type=int, default=60)
ap.add_argument("--tail", type=int, default=0); ap.add_argument("--status", type=int, default=0); ap.add_argument("--sleep_ms", type=float, default=0.0); ap.add_argument("--prewarm_ms", type=float, default=0.0); ap.add_argument("--target", type=int, default=1)
a = ap.parse_args()
vp = ctypes.c_void_p
N, NBUF = 1 << 20, 64
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
A = vp * NBUF
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
torch.cuda.synchronize()
L = ctypes.CDLL(os.path.abspath(a.lib))
L.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
L.bu_context_set_launch_policy.argtypes = [vp, ctypes.c_int]
L.bu_time_uastc_launches_streams_window.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                                    ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float),
                                                    ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
h = vp(); assert L.bu_context_create(0, ctypes.byref(h)) == 0
stt = torch.full((1,), -1, dtype=torch.int64, device=dev); torch.cuda.synchronize()
STP = vp(stt.data_ptr()) if a.status else None
assert L.bu_context_set_launch_policy(h, a.policy) == 0
first, res = 0, []
def win(lead, launches):
    global first
    ev, host, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
    assert L.bu_time_uastc_launches_streams_window(h, a.target, ip, op, NBUF, first, N, 1024, lead, launches, 0, a.streams, STP, ctypes.byref(ev), ctypes.byref(host), None, ctypes.byref(late)) == 0
    first = (first + lead + launches) % NBUF
for c in range(a.calls):
    t0 = time.perf_counter()
    while a.prewarm_ms and (time.perf_counter() - t0) * 1e3 < a.prewarm_ms: win(0, 256)
    ev, host, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
    assert L.bu_time_uastc_launches_streams_window(h, a.target, ip, op, NBUF, first, N, 1024, a.lead, a.launches, a.tail, a.streams, STP, ctypes.byref(ev), ctypes.byref(host), None, ctypes.byref(late)) == 0
    first = (first + a.lead + a.launches) % NBUF
    res.append(max(ev.value, host.value) / a.launches * 1e3)
    if a.sleep_ms: time.sleep(a.sleep_ms / 1e3)
print(" ".join("%.2f" % x for x in res))
s = sorted(res)
print("min %.2f  median %.2f  mean %.2f  max %.2f" % (s[0], s[len(s) // 2], sum(s) / len(s), s[-1]))
