"""Single-mode atlases with several launches in flight: does the period still depend on the mode path's instruction count?
    GPU_MAX_HW_QUEUES=8 python tools/exp/run_modes_streams.py [--target 1] [--streams 4] [--policy 1]
2^20 blocks of ONE UASTC mode per atlas (every chunk full), then the uniform mix; us per atlas, best of 3 windows of 256 (lead 64, tail = streams)
after ~40 ms of the same work."""
import argparse, ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ap = argparse.ArgumentParser()
ap.add_argument("--target", type=int, default=1); ap.add_argument("--streams", type=int, default=4); ap.add_argument("--policy", type=int, default=1)
a = ap.parse_args()
ctx = Context(0); lib = _lib.load()
ctx.set_launch_policy(bool(a.policy))
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 48
gu = torch.from_numpy(g["uastc"]).to(dev)
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
def run(modes):
    ins = []
    mt = torch.tensor(modes, device=dev)
    for k in range(NBUF):
        gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
        m = mt[torch.randint(0, len(modes), (N,), device=dev, generator=gen)]
        ins.append(gu[m * 32 + torch.randint(0, 32, (N,), device=dev, generator=gen)].contiguous())
    torch.cuda.synchronize()
    A = ctypes.c_void_p * NBUF
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    def win(lead, launches, tail):
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        assert lib.bu_time_uastc_launches_streams_window(ctx.handle, a.target, ip, op, NBUF, 0, N, 1024, lead, launches, tail, a.streams, None, ctypes.byref(ev), ctypes.byref(host), None, None) == 0
        return max(ev.value, host.value) / launches * 1e3
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04: win(0, 256, 0)
    return min(win(64, 256, a.streams if a.streams > 1 else 0) for _ in range(3))
print("target %d, %d launches in flight, policy %d" % (a.target, a.streams, a.policy))
for modes in [[m] for m in range(19)] + [list(range(19))]:
    print("%-70s %7.2f us" % (str(modes), run(modes)), flush=True)
