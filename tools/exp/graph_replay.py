"""tools/exp/graph_replay.py [launches_per_graph] [replays]: the headline's pipeline -- UASTC -> BC7, one 2^20-block launch per atlas, shared launch shapes, launch i on
context stream i % 4 -- captured ONCE into a HIP graph (fork from stream 0 to streams 1-3, the launches, join back) and replayed.  Under
`rocprofv3 --kernel-trace -- python3 tools/exp/graph_replay.py` the host enqueues one graph per 256 launches instead of one launch at a time: the profiler's
per-dispatch host cost (6-8 us, more than the pipeline's period) is off the critical path.  Prints the unprofiled wall clock per launch; trace_periods.py reads the trace."""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPLAYS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
gu, gb = torch.from_numpy(g["uastc"]).to(dev), torch.from_numpy(g["bc7"]).to(dev)
NB, NBUF = 1 << 20, 64
idxs = [torch.randint(0, 608, (NB,), device=dev, generator=torch.Generator(device=dev).manual_seed(21 + k)) for k in range(NBUF)]
ins = [gu[i].contiguous() for i in idxs]
outs = [torch.zeros((NB, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
ctx = Context(0)
lib = ctx._lib
ctx.set_launch_policy(True)
S = 4
ext = [torch.cuda.ExternalStream(ctx.stream(i)) for i in range(S)]
print("in flight:", ctx.query_in_flight(S), "sharing now:", ctx.probe_streams(S))
torch.cuda.synchronize()

def launches():
    for j in range(NL):
        k = j % NBUF
        st = lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(ins[k].data_ptr()), NB, ctypes.c_void_p(outs[k].data_ptr()), 1024, 0, None,
                                           ctypes.c_void_p(ext[j % S].cuda_stream))
        assert st == 0, st

graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=ext[0]):
    fork = torch.cuda.Event()
    fork.record(ext[0])
    for s in ext[1:]:
        s.wait_event(fork)
    launches()
    for s in ext[1:]:
        e = torch.cuda.Event()
        e.record(s)
        ext[0].wait_event(e)
for o in outs:
    o.zero_()
torch.cuda.synchronize()
graph.replay()
torch.cuda.synchronize()
ok = all(bool(torch.equal(outs[k], gb[idxs[k]])) for k in range(NBUF))
for _ in range(30):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPLAYS):
    graph.replay()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("graph of %d launches on %d streams, %d replays back to back: %.3f us per launch (host clock, fork / join of every replay included)   verified %s"
      % (NL, S, REPLAYS, dt / (REPLAYS * NL) * 1e6, ok))
# the same launches enqueued one by one (no graph), for reference
for _ in range(10):
    launches()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(REPLAYS):
    launches()
ctx.synchronize()
dt = time.perf_counter() - t0
print("the same %d launches enqueued one by one from Python, %d rounds: %.3f us per launch" % (NL, REPLAYS, dt / (REPLAYS * NL) * 1e6))
ctx.close()
