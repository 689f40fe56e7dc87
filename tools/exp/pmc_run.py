"""few launches of the shipped BC7 kernel on cold-rotated A-gold atlases, for rocprofv3 --pmc passes"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = 24
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
target = int(sys.argv[1]) if len(sys.argv) > 1 else _lib.BC7
torch.cuda.synchronize()
for k in range(NBUF):
    ctx.transcode_device(target, ins[k], N, outs[k] if target != 4 else torch.empty((N, 64), dtype=torch.uint8, device=dev), blocks_per_row=1024)
torch.cuda.synchronize()
print("done")
