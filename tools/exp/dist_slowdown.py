"""EXPERIMENT: does initialising torch.distributed / RCCL slow the transcode kernels down?  (bench.py --gpus N > 1 vs N = 1)"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import Context, _lib, synth
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
if os.environ.get("INIT_FIRST"):
    torch.cuda.set_device(0)
    if os.environ.get("WITH_DEVICE_ID"): dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else: dist.init_process_group("nccl", rank=0, world_size=1)
    if os.environ.get("BARRIER_FIRST"): dist.barrier(); torch.cuda.synchronize()
ctx = Context(0); lib = _lib.load()
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0); N = int(os.environ.get("AB_N", 1 << 25)); NBUF = 4
gu = torch.from_numpy(g["uastc"]).to(dev)
ins = []
for k in range(NBUF):
    gen = torch.Generator(device=dev); gen.manual_seed(k + 1)
    ins.append(gu[torch.randint(0, 608, (N,), device=dev, generator=gen)].contiguous())
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = ctypes.c_void_p * NBUF
if os.environ.get("RAW_OUT"):
    ptrs = []
    for _ in range(NBUF):
        p = ctypes.c_void_p(0); assert lib.bu_device_alloc(ctx.handle, N * 16, ctypes.byref(p)) == 0; ptrs.append(p.value)
    op = A(*ptrs)
else:
    outs = [torch.empty((N, 16), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    op = A(*[x.data_ptr() for x in outs])
ip = A(*[x.data_ptr() for x in ins])
def t(label, L=64):
    ms = ctypes.c_float(0)
    lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, NBUF, 0, N, 1024, 32, None, sp, ctypes.byref(ms))
    best = 1e9
    for rep in range(3):
        lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, ip, op, NBUF, 0, N, 1024, L, None, sp, ctypes.byref(ms))
        best = min(best, ms.value / L * 1e3)
    print("%-60s %.1f us" % (label, best), flush=True)
t("plain")
if not os.environ.get("INIT_FIRST"):
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev) if os.environ.get("WITH_DEVICE_ID") else dist.init_process_group("nccl", rank=0, world_size=1)
t("after init_process_group(nccl)")
dist.barrier(); torch.cuda.synchronize()
t("after the first barrier (communicator exists)")
x = torch.ones(1 << 20, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
t("after an all_reduce")
dist.destroy_process_group()
t("after destroy_process_group")
