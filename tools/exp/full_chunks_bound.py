"""What would PERFECT lane utilisation buy?  python tools/exp/full_chunks_bound.py target
Two inputs with the same overall mix of the 19 UASTC modes, 64 atlases of 2^20 blocks adjacent in memory (one run: the plain ticketed launch), us per atlas:
  uniform      every block's mode drawn independently (the benches' input): a 1024-block tile holds ~54 blocks of each mode = 19-21 chunks, most of them partly filled (0.78 of the lanes)
  full chunks  every 1024-block tile (64 x 16 rectangle of the 1024-wide grid) holds exactly 64 blocks of each of 16 modes, the 16 rotating through the 19 from tile to tile,
               positions shuffled inside the tile: 16 full chunks per tile (1.00 of the lanes), the same code mix over the atlas
The difference bounds what carrying partly filled chunks from one tile into the next could gain (it would add LDS traffic on top)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import synth
vp = ctypes.c_void_p
tname = sys.argv[1]
TGT = {"astc": 0, "bc7": 1, "etc1": 2, "etc2": 3}
t = TGT[tname]; OB = 8 if tname == "etc1" else 16
N = 1 << 20; NBUF = 64; W = 1024
dev = torch.device("cuda", 0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
modes = synth.block_modes(g["uastc"])
by_mode = [np.nonzero(modes == m)[0] for m in range(19)]
assert all(len(b) > 0 for b in by_mode)
gu = torch.from_numpy(g["uastc"]).to(dev); gw = torch.from_numpy(g[tname]).to(dev)
rng = np.random.default_rng(3)
def atlas_full(k):
    idx = np.empty((N // W, W), dtype=np.int64)  # [block row][block column]
    tiles = N // 1024
    for tl in range(tiles):
        ty, tx = tl // 16, tl % 16
        first = (tl + 5 * k) % 19
        ms = [(first + j) % 19 for j in range(16)]
        v = np.concatenate([rng.choice(by_mode[m], 64) for m in ms])
        rng.shuffle(v)
        idx[16 * ty:16 * ty + 16, 64 * tx:64 * tx + 64] = v.reshape(16, 64)
    return idx.reshape(-1)
def atlas_uniform(k):
    m = rng.integers(0, 19, N)
    out = np.empty(N, dtype=np.int64)
    for mm in range(19):
        sel = np.nonzero(m == mm)[0]
        out[sel] = rng.choice(by_mode[mm], len(sel))
    return out
L = ctypes.CDLL(os.path.join(ROOT, "basisu_rs_amd", "libbasisu_hip.so"))
L.bu_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
L.bu_context_set_launch_policy.argtypes = [vp, ctypes.c_int]
L.bu_uastc_transcode_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(vp), ctypes.c_size_t, vp, vp, vp]
h = vp(); assert L.bu_context_create(0, ctypes.byref(h)) == 0
assert L.bu_context_set_launch_policy(h, 0) == 0
stream = torch.cuda.current_stream(); sp = vp(stream.cuda_stream)
for name, make in (("uniform", atlas_uniform), ("full chunks", atlas_full), ("uniform", atlas_uniform), ("full chunks", atlas_full)):
    NA = 8  # distinct atlases, each used eight times over the 64
    idxs = [torch.from_numpy(make(k)).to(dev) for k in range(NA)]
    big_in = torch.cat([gu[idxs[k % NA]] for k in range(NBUF)]).contiguous()
    big_out = torch.zeros((NBUF * N, OB), dtype=torch.uint8, device=dev)
    A, S = vp * NBUF, ctypes.c_size_t * NBUF
    a_in = A(*[big_in.data_ptr() + k * N * 16 for k in range(NBUF)]); a_n = S(*([N] * NBUF)); a_out = A(*[big_out.data_ptr() + k * N * OB for k in range(NBUF)])
    def batch():
        assert L.bu_uastc_transcode_batch_device(h, t, NBUF, a_in, a_n, a_out, 1024, None, None, sp) == 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        batch(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(8): batch()
        e1.record(stream); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / (8 * NBUF))
    ok = all(bool(torch.equal(big_out[k * N:(k + 1) * N], gw[idxs[k % NA]])) for k in (0, 1, 9, NBUF - 1))
    print("%-5s %-12s %.3f us per atlas (median of three windows)  verified %s" % (tname, name, sorted(res)[1], ok))
    del big_in, big_out
