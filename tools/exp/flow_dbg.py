"""EXPERIMENT: time line of the producer / consumer kernel (build with -DBU_FLOW_DBG): when sorter wave 0 starts ranking tile k,
publishes it, and starts writing it back (s_memrealtime, 100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = []
for k in range(16):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(16)]
for k in range(16):
    ctx.transcode_device(_lib.BC7, ins[k], N, outs[k])
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (1024 * 16))()
lib.bu_exp_flow_dbg.argtypes = [ctypes.c_void_p]
assert lib.bu_exp_flow_dbg(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
names = ["start"] + ["publish %d" % k for k in range(4)] + ["write-back %d" % k for k in range(4)] + ["rank %d" % k for k in range(4)]
print("%d workgroups; median / p10 / p90 time in us after the first workgroup's start" % a.shape[0])
for k, nm in enumerate(names):
    col = a[:, k]; col = col[col > 0]
    if col.size: print("  %-14s %6.2f %6.2f %6.2f" % (nm, np.median(col - t0) / 100, np.percentile(col - t0, 10) / 100, np.percentile(col - t0, 90) / 100))
