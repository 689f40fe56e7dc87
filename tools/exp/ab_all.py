"""time every UASTC target of the library named by BASISU_HIP_LIB on cold-rotated A-gold atlases (2^20 blocks):
python tools/exp/ab_all.py [targets...]   -> one line: lib  bc7 astc etc1 etc2 rgba  (us per launch, best of 3 x 256)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = int(os.environ.get("AB_N", 1 << 20)); NBUF = 64 if N <= (1 << 20) else 8
gu = torch.from_numpy(g["uastc"]).to(dev)
ins, idxs = [], []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    idx = torch.randint(0, 608, (N,), device=dev, generator=gen)
    if os.environ.get("AB_COH"):  # mode per 8x8-block tile of a 1024-block-wide image (synth.coh_indices), texture-like
        idx = torch.from_numpy(synth.coh_indices(1024, N // 1024, seed=synth.GOLD_SEED + k)).to(dev)
    ins.append(gu[idx].contiguous()); idxs.append(idx if k == 0 else None)
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
routs = [torch.empty((N, 64), dtype=torch.uint8, device=dev) for _ in range(min(NBUF, 16))]
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = ctypes.c_void_p * NBUF
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
rop = (ctypes.c_void_p * len(routs))(*[x.data_ptr() for x in routs])
names = {"bc7": _lib.BC7, "astc": _lib.ASTC, "etc1": _lib.ETC1, "etc2": _lib.ETC2, "rgba": _lib.RGBA32}
want = sys.argv[1:] or list(names)
res = []
if "copy" in want:
    want.remove("copy")
    ms = ctypes.c_float(0)
    lib.bu_time_copy_launches(ctx.handle, ip, op, NBUF, 0, N, 32, sp, ctypes.byref(ms))
    best = 1e9
    for rep in range(3):
        lib.bu_time_copy_launches(ctx.handle, ip, op, NBUF, 32 + rep * 256, N, 256, sp, ctypes.byref(ms))
        best = min(best, ms.value / 256 * 1e3)
    res.append("copy %.2f" % best)
for nm in want:
    t = names[nm]
    o, nb = (rop, len(routs)) if nm == "rgba" else (op, NBUF)
    ms = ctypes.c_float(0)
    assert lib.bu_time_uastc_launches(ctx.handle, t, ip, o, nb, 0, N, 1024, 32, None, sp, ctypes.byref(ms)) == 0
    # verify buffer 0 against the known answers
    torch.cuda.synchronize()
    key = "rgba" if nm == "rgba" else nm
    if nm == "rgba":
        got = routs[0].view(N // 1024, 4, 1024, 16).permute(0, 2, 1, 3).reshape(N, 64)
    else:
        got = outs[0][:, : _lib.BLOCK_BYTES[t]] if _lib.BLOCK_BYTES[t] == 16 else outs[0].view(-1)[: N * 8].view(N, 8)
    ok = bool(torch.equal(got, torch.from_numpy(g[key]).to(dev)[idxs[0]]))
    best = 1e9
    L = 256 if nm != "rgba" else 128
    for rep in range(3):
        assert lib.bu_time_uastc_launches(ctx.handle, t, ip, o, nb, 32 + rep * L, N, 1024, L, None, sp, ctypes.byref(ms)) == 0
        best = min(best, ms.value / L * 1e3)
    res.append("%s %.2f%s" % (nm, best, "" if ok else " WRONG"))
print(os.path.basename(os.environ.get("BASISU_HIP_LIB", "shipped")), " ".join(res), flush=True)
