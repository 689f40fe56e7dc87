"""a few launches of EVERY shipped kernel (five UASTC targets under BOTH launch policies -- the exclusive and the shared shapes are different
instantiations --, both ETC1S kernels, the copy ceiling, and `array512`: BASELINE config 5's 2^25-block BC7 launch) on cold-rotated inputs, for
the rocprofv3 --kernel-trace / --pmc passes of tools/gpu_pmc.sh.  One launch at a time on one stream (counter passes serialise dispatches anyway)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth, etc1s_selector_from_rows
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = 1 << 20; NBUF = int(os.environ.get("PMC_NBUF", 24))
only = sys.argv[1:]  # optional subset: bc7 astc etc1 etc2 rgba etc1s copy
REPS = int(os.environ.get("PMC_REPS", 1))  # the kernel-trace pass repeats every kernel's rotation (steady-state durations)
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for shared in (False, True):
    ctx.set_launch_policy(shared)
    if only == ["array512"]: break
    for name, t in (("bc7", _lib.BC7), ("astc", _lib.ASTC), ("etc1", _lib.ETC1), ("etc2", _lib.ETC2)):
        if only and name not in only: continue
        # back-to-back launches from the C timing helper (a Python call per launch leaves the GPU idle between kernels and the
        # traced durations come out ~15 % long)
        A = ctypes.c_void_p * NBUF
        ms = ctypes.c_float(0)
        assert lib.bu_time_uastc_launches(ctx.handle, t, A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs]), NBUF, 0, N, 1024, NBUF * REPS, None, sp, ctypes.byref(ms)) == 0
        torch.cuda.synchronize()
ctx.set_launch_policy(False)
if not only or "rgba" in only:
    ro = [torch.empty((N, 64), dtype=torch.uint8, device=dev) for _ in range(8)]
    A8 = ctypes.c_void_p * 8
    ms = ctypes.c_float(0)
    assert lib.bu_time_uastc_launches(ctx.handle, _lib.RGBA32, A8(*[x.data_ptr() for x in ins[:8]]), A8(*[x.data_ptr() for x in ro]), 8, 0, N, 1024, NBUF * REPS, None, sp, ctypes.byref(ms)) == 0
    torch.cuda.synchronize(); del ro
if not only or "batch" in only:
    # the headline's step (bench.py --method batch): ONE launch over all NBUF atlases in their separate allocations (bu_uastc_transcode_batch_device: the multi-run kernel,
    # whole rectangular tiles, tickets), a few launches back to back
    VPn, SZn = ctypes.c_void_p * NBUF, ctypes.c_size_t * NBUF
    a_in, a_n, a_out = VPn(*[x.data_ptr() for x in ins]), SZn(*([N] * NBUF)), VPn(*[x.data_ptr() for x in outs])
    ctx.set_launch_policy("auto")
    for _ in range(4 * REPS):
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, NBUF, a_in, a_n, a_out, 1024, None, None, sp) == 0
    torch.cuda.synchronize()
    ctx.set_launch_policy(False)
if "array512" in only:  # (its own rocprofv3 passes: the persistent grid is the 2^20-block launch's, the summaries could not tell them apart)
    # BASELINE config 5's kernel: the 512-slice array (2^25 blocks, 512 MiB in + 512 MiB out) in ONE launch, two pairs rotated
    nbig = 1 << 25
    bi, bo = [], []
    for k in range(2):
        gen = torch.Generator(device=dev); gen.manual_seed(9000 + k)
        bi.append(gu[torch.randint(0, 608, (nbig,), device=dev, generator=gen)].contiguous())
        bo.append(torch.empty((nbig, 16), dtype=torch.uint8, device=dev))
    A2 = ctypes.c_void_p * 2
    ms = ctypes.c_float(0)
    assert lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, A2(*[x.data_ptr() for x in bi]), A2(*[x.data_ptr() for x in bo]), 2, 0, nbig, 256, 4 * REPS if REPS > 1 else 4, None, sp, ctypes.byref(ms)) == 0
    torch.cuda.synchronize(); del bi, bo
if not only or "copy" in only:
    A = ctypes.c_void_p * NBUF
    ms = ctypes.c_float(0)
    assert lib.bu_time_copy_launches(ctx.handle, A(*[x.data_ptr() for x in ins]), A(*[x.data_ptr() for x in outs]), NBUF, 0, N, NBUF * REPS, sp, ctypes.byref(ms)) == 0
    torch.cuda.synchronize()
if not only or "etc1s" in only:
    ep, rows = synth.etc1s_codebooks(4096, 8192, seed=2)
    sel = etc1s_selector_from_rows(rows)
    nbl = 512 * 512
    d_ep = torch.from_numpy(ep.view(np.int32)).to(dev); d_sel = torch.from_numpy(sel).to(dev)
    d_idx = [torch.from_numpy(synth.etc1s_indices(nbl, 4096, 8192, seed=100 + k).view(np.int32)).to(dev) for k in range(NBUF)]
    o8 = [torch.empty((nbl, 8), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    o64 = [torch.empty((nbl, 64), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    for k in range(NBUF * REPS):
        lib.bu_etc1s_transcode_etc1_device(ctx.handle, d_idx[k % NBUF].data_ptr(), nbl, d_ep.data_ptr(), 4096, d_sel.data_ptr(), 8192, o8[k % NBUF].data_ptr(), None, sp)
    for k in range(NBUF * REPS):
        lib.bu_etc1s_decode_rgba_device(ctx.handle, d_idx[k % NBUF].data_ptr(), None, 512, 512, d_ep.data_ptr(), 4096, d_sel.data_ptr(), 8192, o64[k % NBUF].data_ptr(), None, sp)
    torch.cuda.synchronize()
    del o8, o64, d_idx
    # the LDS-staged kernels (from 2^19 blocks): 2^22 blocks, 6 cold index arrays
    nbig, nb6 = 1 << 22, 6
    d_idx = [torch.from_numpy(synth.etc1s_indices(nbig, 4096, 8192, seed=300 + k).view(np.int32)).to(dev) for k in range(nb6)]
    o8 = [torch.empty((nbig, 8), dtype=torch.uint8, device=dev) for _ in range(nb6)]
    o64 = [torch.empty((nbig, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
    for k in range(nb6 * 2 * REPS):
        lib.bu_etc1s_transcode_etc1_device(ctx.handle, d_idx[k % nb6].data_ptr(), nbig, d_ep.data_ptr(), 4096, d_sel.data_ptr(), 8192, o8[k % nb6].data_ptr(), None, sp)
    for k in range(nb6 * REPS):
        lib.bu_etc1s_decode_rgba_device(ctx.handle, d_idx[k % nb6].data_ptr(), None, 2048, 2048, d_ep.data_ptr(), 4096, d_sel.data_ptr(), 8192, o64[k % 2].data_ptr(), None, sp)
    torch.cuda.synchronize()
print("done")
