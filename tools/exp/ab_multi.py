"""A/B of whole-library kernel variants inside ONE process (atlases built once, variants interleaved round by round):
    python tools/exp/ab_multi.py [--targets bc7,copy] [--rounds 3] [--n 1048576] lib_a.so lib_b.so ...
Each library is a full libbasisu_hip.so (tools/exp/build_variant.sh) loaded under its own path with its own context.
One line per library and round: us per launch (mean of 256 cold-rotated launches between two events); buffer 0 of every target is
checked against the reference's known answers."""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--targets", default="bc7")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--launches", type=int, default=256)
ap.add_argument("--coh", action="store_true")
a = ap.parse_args()
vp = ctypes.c_void_p
TGT = {"astc": 0, "bc7": 1, "etc1": 2, "etc2": 3, "rgba": 4}
BB = {0: 16, 1: 16, 2: 8, 3: 16, 4: 64}
N = a.n; NBUF = 64 if N <= (1 << 20) else 8
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).to(dev)
ins, idx0 = [], None
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    idx = torch.randint(0, 608, (N,), device=dev, generator=gen)
    if a.coh:
        idx = torch.from_numpy(synth.coh_indices(1024, N // 1024, seed=synth.GOLD_SEED + k)).to(dev)
    ins.append(gu[idx].contiguous())
    if k == 0: idx0 = idx
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
routs = [torch.empty((N, 64), dtype=torch.uint8, device=dev) for _ in range(min(NBUF, 16) if "rgba" in a.targets.split(",") else 1 if N <= (1 << 22) else 0)] or [outs[0]]
sp = vp(torch.cuda.current_stream().cuda_stream)
A = vp * NBUF
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
rop = (vp * len(routs))(*[x.data_ptr() for x in routs])
libs = []
for path in a.libs:
    L = ctypes.CDLL(os.path.abspath(path))
    L.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.bu_time_uastc_launches.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                         ctypes.c_size_t, ctypes.c_int, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.bu_time_copy_launches.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp,
                                        ctypes.POINTER(ctypes.c_float)]
    h = vp(); assert L.bu_context_create(0, ctypes.byref(h)) == 0, path
    libs.append((os.path.basename(path), L, h))
want = a.targets.split(",")
def run(L, h, nm, first, launches):
    ms = ctypes.c_float(0)
    if nm == "copy":
        assert L.bu_time_copy_launches(h, ip, op, NBUF, first, N, launches, sp, ctypes.byref(ms)) == 0
    else:
        o, nb = (rop, len(routs)) if nm == "rgba" else (op, NBUF)
        assert L.bu_time_uastc_launches(h, TGT[nm], ip, o, nb, first, N, 1024, launches, None, sp, ctypes.byref(ms)) == 0
    return ms.value / launches * 1e3
def check(nm):
    torch.cuda.synchronize()
    if nm == "copy": return bool(torch.equal(outs[0], ins[0]))
    t = TGT[nm]
    if nm == "rgba": got = routs[0].view(N // 1024, 4, 1024, 16).permute(0, 2, 1, 3).reshape(N, 64)
    else: got = outs[0] if BB[t] == 16 else outs[0].view(-1)[: N * 8].view(N, 8)
    return bool(torch.equal(got, torch.from_numpy(g[nm]).to(dev)[idx0]))
ok = {}
for name, L, h in libs:  # warm-up + verification (buffer 0 is written by the first launch)
    for nm in want:
        outs[0].zero_()
        run(L, h, nm, 0, 32)
        ok[(name, nm)] = check(nm)
for r in range(a.rounds):
    for name, L, h in libs:
        res = []
        for nm in want:
            L2 = a.launches if nm != "rgba" else a.launches // 2
            res.append("%s %.2f%s" % (nm, run(L, h, nm, 32 + r * L2, L2), "" if ok[(name, nm)] else " WRONG"))
        print(name, " ".join(res), flush=True)
