"""ETC1 / ETC2 shapes on LONG walks: python tools/exp/etc_long_walk.py LIB.so target
  (a) one bu_uastc_transcode_batch_device launch over 64 atlases of 2^20 blocks (separate allocations), us per atlas
  (b) one contiguous launch of 2^23 blocks (bu_uastc_transcode_device), us per 2^20 blocks
  (c) one launch per atlas, back to back on one stream, us per atlas
exclusive policy, one library per process, every output verified against the known answers afterwards."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
vp = ctypes.c_void_p
lib, tname = sys.argv[1], sys.argv[2]
pol = int(sys.argv[3]) if len(sys.argv) > 3 else 0
TGT = {"astc": 0, "bc7": 1, "etc1": 2, "etc2": 3, "rgba": 4}
t = TGT[tname]; OB = 8 if tname == "etc1" else (64 if tname == "rgba" else 16)
N = int(os.environ.get("LW_N", 1 << 20)); NBUF = 64  # (LW_N: blocks per atlas, e.g. 1045504 = 1021 rows of 1024: no whole rectangles, strips)
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).to(dev); gw = torch.from_numpy(g[tname]).to(dev) if tname != "rgba" else None  # (RGBA32: an image, timing only)
big_in = torch.empty((NBUF * N, 16), dtype=torch.uint8, device=dev); idxs = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    idx = torch.randint(0, 608, (N,), device=dev, generator=gen); idxs.append(idx)
    big_in[k * N:(k + 1) * N] = gu[idx]
SEP = os.environ.get("SEP", "0") == "1"  # every atlas in an allocation of its own (the multi-run kernel) instead of 64 adjacent ones (merged into one run)
ins = [big_in[k * N:(k + 1) * N].clone() if SEP else big_in[k * N:(k + 1) * N] for k in range(NBUF)]
big_out = torch.zeros((NBUF * N, OB), dtype=torch.uint8, device=dev)
outs = [torch.zeros((N, OB), dtype=torch.uint8, device=dev) if SEP else big_out[k * N:(k + 1) * N] for k in range(NBUF)]
L = ctypes.CDLL(os.path.abspath(lib))
L.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
L.bu_context_set_launch_policy.argtypes = [vp, ctypes.c_int]
L.bu_uastc_transcode_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(vp), ctypes.c_size_t, vp, vp, vp]
L.bu_uastc_transcode_device.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_uint64, vp, vp]
h = vp(); assert L.bu_context_create(0, ctypes.byref(h)) == 0
assert L.bu_context_set_launch_policy(h, pol) == 0
stream = torch.cuda.current_stream(); sp = vp(stream.cuda_stream)
A, S = vp * NBUF, ctypes.c_size_t * NBUF
a_in, a_n, a_out = A(*[x.data_ptr() for x in ins]), S(*([N] * NBUF)), A(*[x.data_ptr() for x in outs])
def batch():
    assert L.bu_uastc_transcode_batch_device(h, t, NBUF, a_in, a_n, a_out, 1024, None, None, sp) == 0
def big(k):
    o = (k % 8) * 8 * N
    assert L.bu_uastc_transcode_device(h, t, big_in.data_ptr() + o * 16, 8 * N, big_out.data_ptr() + o * OB, 1024, 0, None, sp) == 0
def lone(k):
    assert L.bu_uastc_transcode_device(h, t, ins[k % NBUF].data_ptr(), N, outs[k % NBUF].data_ptr(), 1024, 0, None, sp) == 0
def ok():
    torch.cuda.synchronize()
    return gw is None or all(bool(torch.equal(outs[k], gw[idxs[k]])) for k in range(NBUF))
def timed(fn, reps, units):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn(0) if fn is not batch else fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(reps):
            fn(k) if fn is not batch else fn()
        e1.record(stream); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / (reps * units))
    return sorted(best)[1]
res = []
[o.zero_() for o in outs]; r = timed(batch, 8, NBUF); res.append("batch64 %.3f %s" % (r, ok()))
if not SEP:
    big_out.zero_(); r = timed(big, 64, 8); res.append("one2^23 %.3f %s" % (r, ok()))
[o.zero_() for o in outs]; r = timed(lone, 512, 1); res.append("lone2^20 %.3f %s" % (r, ok()))
print("%-18s %s p%d  " % (os.path.basename(lib), tname, pol) + "   ".join(res))
