import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).cuda(); gb = torch.from_numpy(g["bc7"]).cuda()
for n in ((1 << 26), (1 << 26) + 4097, (1 << 25) + 123, 3 * (1 << 24)):
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    idx = torch.randint(0, 608, (n,), device="cuda", generator=gen)
    d_in = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    for lo in range(0, n, 1 << 22):
        d_in[lo:lo + (1 << 22)] = gu[idx[lo:lo + (1 << 22)]]
    d_out = torch.zeros((n, 16), dtype=torch.uint8, device="cuda")
    ctx.transcode_device(_lib.BC7, d_in, n, d_out)
    torch.cuda.synchronize()
    bad_total = 0; first = None; last = None
    for lo in range(0, n, 1 << 22):
        ne = (d_out[lo:lo + (1 << 22)] != gb[idx[lo:lo + (1 << 22)]]).any(dim=1)
        c = int(ne.sum())
        if c:
            w = torch.nonzero(ne).flatten()
            if first is None: first = lo + int(w[0])
            last = lo + int(w[-1]); bad_total += c
    print("n", n, "bad", bad_total, "first", first, "last", last, "first//4096", None if first is None else first // 4096, flush=True)
    del d_in, d_out, idx
