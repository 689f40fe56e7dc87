"""EXPERIMENT: the kernels reading / writing page-locked host memory directly over PCIe (no staging copies)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
N = 1 << 20
idx = synth.gold_indices(N)
pin_in, pin_out = ctx.host_alloc(N * 16), ctx.host_alloc(N * 16)
pin_in[:] = g["uastc"][idx].reshape(-1)
s = torch.cuda.Stream()
sp = ctypes.c_void_p(s.cuda_stream)
def go(nb_piece):
    for lo in range(0, N, nb_piece):
        st = lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(pin_in.ctypes.data + lo * 16), nb_piece,
                                           ctypes.c_void_p(pin_out.ctypes.data + lo * 16), 1024, lo, None, sp)
        assert st == 0, st
    s.synchronize()
for piece in (N, N // 4, N // 16, N // 64):
    pin_out[:] = 0
    go(piece)
    t0 = time.perf_counter()
    for _ in range(5): go(piece)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("zero-copy piece=%8d  %.3f ms  ok=%s" % (piece, ms, (pin_out.reshape(-1, 16) == g["bc7"][idx]).all()), flush=True)
# reference points: device-resident
d_in = torch.from_numpy(g["uastc"][idx]).cuda(); d_out = torch.empty((N, 16), dtype=torch.uint8, device="cuda")
def dev(i, o):
    st = lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(i), N, ctypes.c_void_p(o), 1024, 0, None, sp); assert st == 0
    s.synchronize()
for name, i, o in (("host->dev", pin_in.ctypes.data, d_out.data_ptr()), ("dev->host", d_in.data_ptr(), pin_out.ctypes.data)):
    dev(i, o); t0 = time.perf_counter()
    for _ in range(5): dev(i, o)
    print("%s  %.3f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
