"""Batches of SMALL slices in separate allocations (BASELINE config 5's slices: 65 536 blocks each), per target:
    python tools/exp/slices_in_flight_ab.py LIB.so target [n_slices=64] [blocks=65536]
  (a) one bu_uastc_transcode_batch_device call per batch on one stream          us per batch
  (b) one bu_uastc_transcode_batch_in_flight call (4 streams) + bu_context_synchronize, host clock   us per batch
one library per process, outputs verified against the known answers."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
vp = ctypes.c_void_p
lib, tname = sys.argv[1], sys.argv[2]
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
N = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
TGT = {"astc": 0, "bc7": 1, "etc1": 2, "etc2": 3, "rgba": 4}
t = TGT[tname]; OB = 8 if tname == "etc1" else (64 if tname == "rgba" else 16)
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu = torch.from_numpy(g["uastc"]).to(dev); gw = torch.from_numpy(g[tname]).to(dev) if tname != "rgba" else None  # (RGBA32 is an image, not block-linear: timing only here)
ROT = 4 if NS * N <= (1 << 25) else 2  # batches rotated (cold inputs)
ins, outs, idxs = [], [], []
for k in range(ROT * NS):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    idx = torch.randint(0, 608, (N,), device=dev, generator=gen); idxs.append(idx)
    ins.append(gu[idx].contiguous()); outs.append(torch.zeros((N, OB), dtype=torch.uint8, device=dev))
L = ctypes.CDLL(os.path.abspath(lib))
L.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
L.bu_context_synchronize.argtypes = [vp]
L.bu_uastc_transcode_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(vp), ctypes.c_size_t, vp, vp, vp]
L.bu_uastc_transcode_batch_in_flight.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(vp), ctypes.c_size_t, vp, vp, ctypes.c_int]
h = vp(); assert L.bu_context_create(0, ctypes.byref(h)) == 0
stream = torch.cuda.current_stream(); sp = vp(stream.cuda_stream)
A, S = vp * NS, ctypes.c_size_t * NS
args = [(A(*[x.data_ptr() for x in ins[r * NS:(r + 1) * NS]]), S(*([N] * NS)), A(*[x.data_ptr() for x in outs[r * NS:(r + 1) * NS]])) for r in range(ROT)]
def dev_call(k):
    a = args[k % ROT]
    assert L.bu_uastc_transcode_batch_device(h, t, NS, a[0], a[1], a[2], 256, None, None, sp) == 0
def fl_call(k):
    a = args[k % ROT]
    assert L.bu_uastc_transcode_batch_in_flight(h, t, NS, a[0], a[1], a[2], 256, None, None, 4) == 0
def ok():
    torch.cuda.synchronize(); L.bu_context_synchronize(h)
    return gw is None or all(bool(torch.equal(outs[k], gw[idxs[k]])) for k in range(ROT * NS))
def run(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn(0)
    torch.cuda.synchronize(); L.bu_context_synchronize(h)
    res = []
    for _ in range(5):
        t0 = time.perf_counter()
        for k in range(reps):
            fn(k)
        torch.cuda.synchronize(); L.bu_context_synchronize(h)
        res.append((time.perf_counter() - t0) * 1e6 / reps)
    return sorted(res)[2]
def host_only(fn, reps):  # host time of the calls themselves (the GPU drains afterwards, untimed)
    res = []
    for _ in range(5):
        t0 = time.perf_counter()
        for k in range(reps):
            fn(k)
        res.append((time.perf_counter() - t0) * 1e6 / reps)
        torch.cuda.synchronize(); L.bu_context_synchronize(h)
    return sorted(res)[2]
out = []
for nm, fn in (("batch_device", dev_call), ("in_flight_4", fl_call)):
    [o.zero_() for o in outs]
    r = run(fn, 16); ho = host_only(fn, 16); out.append("%s %.1f us per batch (%.2f per 2^20 blocks; host %.1f per call) %s" % (nm, r, r / (NS * N / 2**20), ho, ok()))
print("GPU_MAX_HW_QUEUES=%s %-16s %s %dx%d  " % (os.environ.get("GPU_MAX_HW_QUEUES"), os.path.basename(lib), tname, NS, N) + "   ".join(out)); sys.exit(0)
print("%-20s %s %dx%d  " % (os.path.basename(lib), tname, NS, N) + "   ".join(out))
