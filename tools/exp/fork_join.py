"""What does a fork / join over the context's streams cost inside ONE stream-ordered call?  A shard of N blocks (a) as one launch on stream A (exclusive
policy), (b) as P pieces on the context's streams (shared policy), each waiting for an event recorded on A and A waiting for each piece's event.
Back-to-back repetitions on A, events around all of them; us per shard.  GPU_MAX_HW_QUEUES=8."""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0)
gu = torch.from_numpy(g["uastc"]).to(dev); gw = torch.from_numpy(g["bc7"]).to(dev)
print("streams on one queue (probe):", ctx.probe_streams(4))
A = torch.cuda.Stream(device=dev)
S = [torch.cuda.ExternalStream(ctx.stream(i), device=dev) for i in range(4)]
vp = ctypes.c_void_p
for lg in (20, 22, 23, 25):
    N = 1 << lg; NB = max(2, min(16, (1 << 27) >> lg))
    idx = [torch.randint(0, 608, (N,), device=dev) for _ in range(NB)]
    ins = [torch.cat([gu[i[lo:lo + (1 << 22)]] for lo in range(0, N, 1 << 22)]).contiguous() for i in idx]
    outs = [torch.zeros((N, 16), dtype=torch.uint8, device=dev) for _ in range(NB)]
    torch.cuda.synchronize()
    def one(k):
        ctx.set_launch_policy(False)
        lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, vp(ins[k % NB].data_ptr()), N, vp(outs[k % NB].data_ptr()), 1024, 0, None, vp(A.cuda_stream))
    def forked(k, P):
        ctx.set_launch_policy(True)
        ef = torch.cuda.Event(); ef.record(A)
        n = N // P
        for p in range(P):
            S[p].wait_event(ef)
            lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, vp(ins[k % NB].data_ptr() + p * n * 16), n, vp(outs[k % NB].data_ptr() + p * n * 16), 1024, 0, None, vp(S[p].cuda_stream))
            e = torch.cuda.Event(); e.record(S[p]); A.wait_event(e)
    def timed(fn, reps):
        for k in range(reps): fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(A)
        for k in range(reps): fn(k)
        e1.record(A); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    reps = max(16, min(512, (1 << 29) >> lg))
    for _ in range(3): timed(one, reps)
    r = ["one launch %.2f" % timed(one, reps)]
    for P in (2, 4):
        if N // P >= (1 << 19):
            timed(lambda k: forked(k, P), reps)
            r.append("%d pieces fork/join %.2f" % (P, timed(lambda k: forked(k, P), reps)))
    ok = all(bool(torch.equal(outs[k], gw[idx[k]])) for k in range(NB))
    print("2^%d blocks: %s   us per shard   %s" % (lg, "   ".join(r), "verified" if ok else "WRONG"), flush=True)
    del ins, outs, idx
