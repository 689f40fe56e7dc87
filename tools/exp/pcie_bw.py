"""EXPERIMENT: what PCIe allows for a 4096^2 atlas host -> device -> host (16 MiB each way) against the library's zero-copy path"""
import time, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import basisu_rs_amd as bu
from basisu_rs_amd import _lib, synth
n = 16 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory(); h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device="cuda"); d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, reps=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("H2D 16 MiB pinned      %.3f ms" % t(lambda: d_in.copy_(h_in, non_blocking=True)))
print("D2H 16 MiB pinned      %.3f ms" % t(lambda: h_out.copy_(d_out, non_blocking=True)))
def both():
    with torch.cuda.stream(s1): d_in.copy_(h_in, non_blocking=True)
    with torch.cuda.stream(s2): h_out.copy_(d_out, non_blocking=True)
print("H2D + D2H concurrently %.3f ms" % t(both))
ctx = bu.Context(0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
blocks = g["uastc"][synth.gold_indices(1 << 20)]
pin_in = ctx.host_alloc(n); pin_out = ctx.host_alloc(n)
np.asarray(pin_in)[:] = blocks.reshape(-1)
def zc(): ctx.transcode(_lib.BC7, np.asarray(pin_in), out=np.asarray(pin_out))
try:
    zc(); ts = []
    for _ in range(20):
        t0 = time.perf_counter(); zc(); ts.append(time.perf_counter() - t0)
    print("library, both buffers page-locked (zero-copy kernel): %.3f ms median" % (sorted(ts)[10] * 1e3))
except Exception as e:
    print("zero-copy call failed:", repr(e))
