"""tools/exp/product_call_trace.py [atlases_per_call] [calls]: the headline pipeline through the PRODUCT call -- bu_uastc_transcode_batch_in_flight over N atlases in separate
allocations (one 2^20-block launch each, launch i on context stream i % 4, shared shapes; from 256 launches per call on, one enqueue thread per stream), calls back to
back, one bu_context_synchronize at the end.  Meant to run under `rocprofv3 --kernel-trace -- python3 tools/exp/product_call_trace.py`; prints its own host clock too."""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu, gb = torch.from_numpy(g["uastc"]).to(dev), torch.from_numpy(g["bc7"]).to(dev)
NB, NBUF = 1 << 20, 64
idxs = [torch.randint(0, 608, (NB,), device=dev, generator=torch.Generator(device=dev).manual_seed(21 + k)) for k in range(NBUF)]
ins = [gu[i].contiguous() for i in idxs]
outs = [torch.zeros((NB, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
ctx = Context(0)
lib = ctx._lib
status = torch.empty(1, dtype=torch.int64, device=dev)
ctx.status_word_reset(status)
torch.cuda.synchronize()
VP, SZ = ctypes.c_void_p * N, ctypes.c_size_t * N
a_in, a_n, a_out = VP(*[ins[k % NBUF].data_ptr() for k in range(N)]), SZ(*([NB] * N)), VP(*[outs[k % NBUF].data_ptr() for k in range(N)])
sp = ctypes.c_void_p(status.data_ptr())

def call():
    assert lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, N, a_in, a_n, a_out, 1024, None, sp, 4) == 0

call(); ctx.synchronize()
ok = all(bool(torch.equal(outs[k], gb[idxs[k]])) for k in range(NBUF))
for _ in range(6):
    call()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(CALLS):
    call()
ctx.synchronize()
dt = time.perf_counter() - t0
print("%d calls of bu_uastc_transcode_batch_in_flight over %d atlases, one wait: %.3f us per atlas by this process's host clock   verified %s   in flight %s"
      % (CALLS, N, dt / (CALLS * N) * 1e6, ok, ctx.query_in_flight(4)))
ctx.close()
