"""EXPERIMENT (round 4): persistent workgroups (the resident set walks the tiles with prefetch) against larger grids up to one
tile per workgroup (BU_X_BCAP = multiple of the resident set, 0 = unlimited), all targets' 16 B -> 16 B kernels, sizes 2^21..2^25.
Run once per BU_X_BCAP value (the knob is read once per process): BU_X_BCAP=k python tools/exp/bcap_sweep.py [targets...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).to(dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {"bc7": _lib.BC7, "astc": _lib.ASTC, "etc1": _lib.ETC1, "etc2": _lib.ETC2}
want = sys.argv[1:] or ["bc7"]
res = []
for lg in (21, 22, 23, 24, 25):
    N = 1 << lg
    nbuf = 3 if lg >= 24 else 8
    gen = torch.Generator(device=dev); gen.manual_seed(lg)
    idx0 = torch.randint(0, 608, (N,), device=dev, generator=gen)
    ins = [gu[idx0].contiguous()] + [gu[torch.randint(0, 608, (N,), device=dev)].contiguous() for _ in range(nbuf - 1)]
    outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    A = ctypes.c_void_p * nbuf
    ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
    ms = ctypes.c_float(0)
    for nm in want:
        t = names[nm]
        launches = max(12, min(128, (1 << 29) // N))
        assert lib.bu_time_uastc_launches(ctx.handle, t, ip, op, nbuf, 0, N, 1024, nbuf, None, sp, ctypes.byref(ms)) == 0
        torch.cuda.synchronize()
        key = nm
        got = outs[0] if _lib.BLOCK_BYTES[t] == 16 else outs[0].view(-1)[: N * 8].view(N, 8)
        ok = bool(torch.equal(got, torch.from_numpy(g[key]).to(dev)[idx0]))
        best = 1e9
        for _ in range(3):
            assert lib.bu_time_uastc_launches(ctx.handle, t, ip, op, nbuf, 0, N, 1024, launches, None, sp, ctypes.byref(ms)) == 0
            best = min(best, ms.value / launches * 1e3)
        bpb = 24 if nm == "etc1" else 32
        res.append("2^%d %s %.1f us %.0f GB/s%s" % (lg, nm, best, bpb * N / best / 1e3, "" if ok else " WRONG"))
    del ins, outs
print("BU_X_BCAP=%s | " % os.environ.get("BU_X_BCAP", "unset") + " | ".join(res), flush=True)
