"""Launches in flight x kernel shape, whole-library variants inside ONE process (atlases built once, cells interleaved round by round):
    python tools/exp/ab_streams.py [--target bc7] [--streams 1,2,3] [--rounds 3] [--launches 240] lib_a.so lib_b.so ...
Each library is a full libbasisu_hip.so loaded under its own path with its own context.  One line per library, stream count and round:
us per atlas = bu_time_uastc_launches_streams_window (lead launches, event 0, `launches` timed launches round-robin over the streams,
end event per stream; max over streams), cold rotation over 64 atlases.  After the timed rounds every output buffer is compared with the
reference's known answers (all 64 atlases, not just the first)."""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--target", default="bc7")
ap.add_argument("--streams", default="1,2,3")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--bpr", type=int, default=1024)
ap.add_argument("--launches", type=int, default=240)
ap.add_argument("--lead", type=int, default=64)
ap.add_argument("--tail", type=int, default=-1)  # untimed launches behind the end events; -1 = one per stream
ap.add_argument("--prewarm_ms", type=float, default=0.0)  # untimed windows of the same cell for this long in front of every measurement (clocks)
ap.add_argument("--policy", default="")  # comma list of launch policies to set per cell (libraries that export bu_context_set_launch_policy)
a = ap.parse_args()
vp = ctypes.c_void_p
TGT = {"astc": 0, "bc7": 1, "etc1": 2, "etc2": 3, "rgba": 4, "copy": 100}
t = TGT[a.target]
N = a.n; NBUF = 64 if N <= (1 << 20) else 8
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).to(dev)
gw = torch.from_numpy(g[a.target]).to(dev) if a.target != "copy" else None
ins, idxs = [], []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    idx = torch.randint(0, 608, (N,), device=dev, generator=gen)
    ins.append(gu[idx].contiguous()); idxs.append(idx)
OB = {"etc1": 8, "rgba": 64}.get(a.target, 16)
outs = [torch.empty((N, OB), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
A = vp * NBUF
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
torch.cuda.synchronize()
libs = []
for path in a.libs:
    L = ctypes.CDLL(os.path.abspath(path))
    L.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.bu_time_uastc_launches_streams_window.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                                        ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float),
                                                        ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    h = vp(); assert L.bu_context_create(0, ctypes.byref(h)) == 0, path
    pols = [None]
    if a.policy and hasattr(L, "bu_context_set_launch_policy"):
        L.bu_context_set_launch_policy.argtypes = [vp, ctypes.c_int]
        pols = [int(x) for x in a.policy.split(",")]
    for p in pols:
        libs.append((os.path.basename(path) + ("" if p is None else ":p%d" % p), L, h, p))
def run(L, h, pol, ns, first, lead, launches):
    if pol is not None: assert L.bu_context_set_launch_policy(h, pol) == 0
    ev, host, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
    st = L.bu_time_uastc_launches_streams_window(h, t, ip, op, NBUF, first, N, a.bpr, lead, launches, (ns if a.tail < 0 else a.tail) if lead else 0, ns, None, ctypes.byref(ev), ctypes.byref(host), None, ctypes.byref(late))
    assert st == 0, st
    # the window holds `launches` completions only if the streams ran in step: start events within a few periods of each other
    oos = False
    if hasattr(L, "bu_time_last_window_streams") and ns > 1:
        a_, b_, n_ = (ctypes.c_float * 8)(), (ctypes.c_float * 8)(), ctypes.c_int(0)
        L.bu_time_last_window_streams.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
        if L.bu_time_last_window_streams(h, a_, b_, ctypes.byref(n_)) == 0:
            st_ = [a_[i] for i in range(n_.value) if a_[i] >= 0]
            oos = bool(st_) and (max(st_) - min(st_)) * 1e3 > 16 * (ev.value / launches * 1e3)
    return max(ev.value, host.value) / launches * 1e3, ev.value / launches * 1e3, (late.value, oos)
def check():
    torch.cuda.synchronize()
    bad = 0
    for k in range(NBUF):
        got = outs[k].view(N // a.bpr, 4, a.bpr, 16).permute(0, 2, 1, 3).reshape(N, 64) if a.target == "rgba" else outs[k]
        if not torch.equal(got, ins[k] if a.target == "copy" else gw[idxs[k]]): bad += 1
    return bad
streams = [int(x) for x in a.streams.split(",")]
bad = {}
for name, L, h, p in libs:
    for ns in streams:
        for o in outs: o.zero_()
        torch.cuda.synchronize()  # (the context's streams do not wait for torch's stream)
        run(L, h, p, ns, 0, 0, NBUF)
        bad[(name, ns)] = check()
first = 0
for r in range(a.rounds):
    for name, L, h, p in libs:
        res = []
        for ns in streams:
            import time as _t
            t0 = _t.perf_counter()
            while a.prewarm_ms and (_t.perf_counter() - t0) * 1e3 < a.prewarm_ms: run(L, h, p, ns, first, 0, 256)
            us, ev, late = run(L, h, p, ns, first, a.lead, a.launches)
            first = (first + a.lead + a.launches) % NBUF
            res.append("S%d %.2f (ev %.2f%s%s)%s" % (ns, us, ev, " LATE" if late[0] else "", " OUT-OF-STEP: not a period" if late[1] else "", "" if bad[(name, ns)] == 0 else " WRONG:%d" % bad[(name, ns)]))
        print("%-22s %s" % (name, "  ".join(res)), flush=True)
