"""tools/exp/big_file_time.py: bu_read_to on a UASTC file whose slices form ONE run of 2^22 / 2^23 blocks (64 / 128 MiB), PAGEABLE output: one upload + one launch + one download
(BU_RUN_PIECE_MIB=0) against round 6's pieces on four streams (upload and launch of a piece share a stream; shared launch shapes), ms per call"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, basisu_rs_amd as bu
from basisu_rs_amd import synth
g = synth.load_golden(os.path.join(ROOT, "tests/golden/uastc_kat.bin")); ctx = bu.Context(0)
for n_slices in (4, 8):
    idx = [synth.gold_indices(1 << 20, seed=10 + k) for k in range(n_slices)]
    f = bu.write_uastc_file([dict(data=g["uastc"][i], orig_w=4096, orig_h=4096, nbx=1024, nby=1024, image_index=k) for k, i in enumerate(idx)])
    out = np.empty(n_slices << 24, dtype=np.uint8)
    for piece in ("0", "16"):
        os.environ["BU_RUN_PIECE_MIB"] = piece
        bu.read_to_bc7(f, ctx, out=out)
        ts = []
        for i in range(9):
            t0 = time.perf_counter(); imgs = bu.read_to_bc7(f, ctx, out=out); ts.append(time.perf_counter() - t0)
        ok = all((np.asarray(imgs[k].data).reshape(-1, 16) == g["bc7"][idx[k]]).all() for k in range(n_slices))
        print("%d MiB file, %s: median %.3f ms  min %.3f  %s" % (n_slices * 16, "one upload + one launch" if piece == "0" else "pieces of <= 16 MiB on four streams", sorted(ts)[4] * 1e3, min(ts) * 1e3, "ok" if ok else "WRONG"))
