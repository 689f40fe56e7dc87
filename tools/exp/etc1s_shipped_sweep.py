"""ETC1S back-end of the library named by BASISU_HIP_LIB (default: shipped): size sweep of both device entry points on cold-rotated
index arrays (config-4 codebooks: 4096 endpoints, 8192 selectors); timing only (parity: tests/test_gpu_round3.py)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth, etc1s_selector_from_rows
ctx = Context(0); lib = _lib.load()
dev = torch.device("cuda", 0)
N_EP, N_SEL = 4096, 8192
ep, rows = synth.etc1s_codebooks(N_EP, N_SEL, seed=2)
sel = etc1s_selector_from_rows(rows)
d_ep = torch.from_numpy(ep.view(np.int32)).to(dev); d_sel = torch.from_numpy(sel).to(dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for lg in (18, 19, 20, 21, 22, 24):
    n = 1 << lg
    nbuf = max(2, min(16, (1 << 31) // (n * 68)))
    host_idx = [synth.etc1s_indices(n, N_EP, N_SEL, seed=100 + k) for k in range(nbuf)]
    idx = [torch.from_numpy(h.view(np.int32)).to(dev) for h in host_idx]
    for rgba in (0, 1):
        outs = [torch.zeros(n * (64 if rgba else 8), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        def go(k):
            if rgba: st = lib.bu_etc1s_decode_rgba_device(ctx.handle, idx[k].data_ptr(), None, 512, n // 512, d_ep.data_ptr(), N_EP, d_sel.data_ptr(), N_SEL, outs[k].data_ptr(), None, sp)
            else: st = lib.bu_etc1s_transcode_etc1_device(ctx.handle, idx[k].data_ptr(), n, d_ep.data_ptr(), N_EP, d_sel.data_ptr(), N_SEL, outs[k].data_ptr(), None, sp)
            assert st == 0
        for k in range(nbuf): go(k)
        torch.cuda.synchronize()
        reps = max(16, min(256, (1 << 26) // n)); best = 1e9
        for _ in range(3):
            e0.record()
            for i in range(reps): go(i % nbuf)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        res.append("2^%d %s %.2f" % (lg, "rgba" if rgba else "etc1", best))
        del outs
print(os.path.basename(os.environ.get("BASISU_HIP_LIB", "shipped")), " | ".join(res), flush=True)
