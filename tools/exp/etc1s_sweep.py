"""ETC1S back-end: size sweep 2^16 .. 2^24 blocks of both kernels (config-4 codebooks: 4096 endpoints, 8192 selectors), the shipped
L2-gather kernels (variant 0) against codebooks staged in LDS by one (1) or two (2) persistent 1024-thread workgroups per CU.
python tools/exp/etc1s_sweep.py   (needs tools/exp/libbu_exp.so: hipcc ... -o tools/exp/libbu_exp.so tools/exp/bu_exp.hip)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth, etc1s_selector_from_rows
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libbu_exp.so"))
vp = ctypes.c_void_p
lib.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
lib.bu_exp_etc1s.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_uint, ctypes.c_size_t, vp, ctypes.c_uint32, vp, ctypes.c_uint32, vp, vp]
h = vp(); assert lib.bu_context_create(0, ctypes.byref(h)) == 0
dev = torch.device("cuda", 0)
N_EP, N_SEL = 4096, 8192
ep, rows = synth.etc1s_codebooks(N_EP, N_SEL, seed=2)
sel = etc1s_selector_from_rows(rows)
d_ep = torch.from_numpy(ep.view(np.int32)).to(dev); d_sel = torch.from_numpy(sel).to(dev)
sp = vp(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("blocks      target  variant   us/launch   GB/s (12 / 68 B per block)   Mblocks/s   verified")
for lg in range(16, 25, 2):
    n = 1 << lg
    nbuf = max(2, min(16, (1 << 31) // (n * 68)))  # cold rotation within ~2 GiB of RGBA output
    idx = [torch.from_numpy(synth.etc1s_indices(n, N_EP, N_SEL, seed=100 + k).view(np.int32)).to(dev) for k in range(nbuf)]
    for rgba, bpb in ((0, 12), (1, 68)):
        outs = [torch.zeros(n * (64 if rgba else 8), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        ref = None
        for variant in (0, 1, 2):
            def go(k):
                assert lib.bu_exp_etc1s(h, variant, rgba, idx[k].data_ptr(), 512, n, d_ep.data_ptr(), N_EP, d_sel.data_ptr(), N_SEL, outs[k].data_ptr(), sp) == 0
            for k in range(nbuf): go(k)
            torch.cuda.synchronize()
            if variant == 0: ref = outs[0].clone()
            ok = bool(torch.equal(outs[0], ref))
            reps = max(16, min(256, (1 << 26) // n))
            best = 1e9
            for _ in range(3):
                e0.record()
                for i in range(reps): go(i % nbuf)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps * 1e3)
            print("2^%-2d %9d  %-5s   %d      %9.2f   %8.1f   %10.1f   %s" % (lg, n, "rgba" if rgba else "etc1", variant, best, bpb * n / best / 1e3, n / best, ok), flush=True)
        del outs
