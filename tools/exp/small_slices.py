"""64 slices of 65 536 blocks: launches on 1 / 2 / 4 / 8 context streams (C loop, wall clock), the batch entry point, one launch"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); ns, nbs = 64, 65536
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = [gu[torch.randint(0, 608, (nbs,), device=dev)].contiguous() for _ in range(ns)]
outs = [torch.empty((nbs, 16), dtype=torch.uint8, device=dev) for _ in range(ns)]
A = ctypes.c_void_p * ns
ip, op = A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs])
ms = ctypes.c_float(0)
for n_streams in (1, 2, 4, 8):
    for _ in range(2):
        lib.bu_time_uastc_launches_streams(ctx.handle, _lib.BC7, ip, op, ns, nbs, 256, 64, n_streams, ctypes.byref(ms))
    best = 1e9
    for _ in range(5):
        lib.bu_time_uastc_launches_streams(ctx.handle, _lib.BC7, ip, op, ns, nbs, 256, 64, n_streams, ctypes.byref(ms)); best = min(best, ms.value)
    print("%d streams: %.1f us per 64 slices (%.2f us per slice)" % (n_streams, best * 1e3, best * 1e3 / 64))
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SZ = ctypes.c_size_t * ns
def batch():
    assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, ns, ip, SZ(*([nbs] * ns)), op, 256, None, None, sp) == 0
for _ in range(3): batch()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); batch(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("batch call, separate allocations: %.1f us" % (best * 1e6))
cat_in = torch.cat(ins); cat_out = torch.empty((ns * nbs, 16), dtype=torch.uint8, device=dev)
ms = ctypes.c_float(0)
one_in, one_out = (ctypes.c_void_p * 1)(cat_in.data_ptr()), (ctypes.c_void_p * 1)(cat_out.data_ptr())
for _ in range(3): lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, one_in, one_out, 1, 0, ns * nbs, 256, 8, None, sp, ctypes.byref(ms))
print("one launch over the concatenation: %.1f us" % (ms.value / 8 * 1e3))
def one():
    assert lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(cat_in.data_ptr()), ns * nbs, ctypes.c_void_p(cat_out.data_ptr()), 256, 0, None, sp) == 0
for _ in range(3): one()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); one(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("one launch over the concatenation, call + synchronize (as the batch row): %.1f us" % (best * 1e6))
