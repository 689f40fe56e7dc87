"""Are the device entry points capturable into a HIP graph?  64 slices of 65 536 blocks in separate allocations: a loop of
bu_uastc_transcode_device calls captured once (torch.cuda.graph) and replayed, against the same loop launched directly and against the
one-call batch entry point; status words included.  Results verified."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0)
NS, NB = 64, 65536
idx = [synth.gold_indices(NB, seed=40 + k) for k in range(NS)]
ins = [torch.from_numpy(g["uastc"][i]).to(dev) for i in idx]
outs = [torch.zeros((NB, 16), dtype=torch.uint8, device=dev) for _ in range(NS)]
status = torch.zeros(1, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
def loop(stream):
    sp = ctypes.c_void_p(stream.cuda_stream)
    assert lib.bu_status_word_reset(ctx.handle, ctypes.c_void_p(status.data_ptr()), sp) == 0
    for k in range(NS):
        st = lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(ins[k].data_ptr()), NB, ctypes.c_void_p(outs[k].data_ptr()), 256,
                                           k * NB, ctypes.c_void_p(status.data_ptr()), sp)
        assert st == 0, st
def check():
    torch.cuda.synchronize()
    assert int(status.item()) == -1, hex(int(status.item()) & (2**64 - 1))
    for k in (0, 17, NS - 1):
        assert torch.equal(outs[k], torch.from_numpy(g["bc7"][idx[k]]).to(dev))
    for o in outs: o.zero_()
def wall(fn, reps=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
with torch.cuda.stream(side):
    loop(side)
check()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    loop(side)
graph.replay(); check()
t_graph = wall(graph.replay)
graph.replay(); check()
def direct():
    with torch.cuda.stream(side): loop(side)
t_loop = wall(direct); direct(); check()
VP, SZ = ctypes.c_void_p * NS, ctypes.c_size_t * NS
pin, pout, pn = VP(*[t.data_ptr() for t in ins]), VP(*[t.data_ptr() for t in outs]), SZ(*([NB] * NS))
def batch():
    sp = ctypes.c_void_p(side.cuda_stream)
    lib.bu_status_word_reset(ctx.handle, ctypes.c_void_p(status.data_ptr()), sp)
    assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, NS, pin, pn, pout, 256, None, ctypes.c_void_p(status.data_ptr()), sp) == 0
t_batch = wall(batch); batch(); check()
# the batch call itself inside a graph
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=side):
    batch()
g2.replay(); check()
t_g2 = wall(g2.replay)
print("64 slices x 65 536 blocks, UASTC->BC7, wall clock per batch incl. synchronize: loop of launches %.1f us | the same loop captured in a HIP graph, replayed %.1f us | "
      "batch entry point %.1f us | batch entry point captured, replayed %.1f us" % (t_loop, t_graph, t_batch, t_g2))
