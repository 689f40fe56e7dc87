"""EXPERIMENT: where does read_to_rgba of a 16-slice ETC1S file spend its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import basis_builder as bb
import basisu_rs_amd as bu
ctx = bu.Context(0)
f, _, _ = bb.etc1s_file(np.random.default_rng(44), [(128, 128)] * 16, n_codebook=4096)
def t(fn, n=5):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
nbytes = bu.read_query(0, f)[1]
print("file %d bytes -> %d bytes out" % (len(f), nbytes))
print("read_to_rgba, fresh numpy output each call : %.3f ms" % t(lambda: bu.read_to_rgba(f, ctx)))
out = np.empty(nbytes, dtype=np.uint8); out[:] = 0
print("read_to_rgba, reused pageable output        : %.3f ms" % t(lambda: bu.read_to_rgba(f, ctx, out=out)))
pin = ctx.host_alloc(nbytes)
print("read_to_rgba, page-locked output (zero-copy): %.3f ms" % t(lambda: bu.read_to_rgba(f, ctx, out=pin)))
print("read_to_etc1, reused pageable output        : %.3f ms" % t(lambda: bu.read_to_etc1(f, ctx, out=out)))
print("crc16 of the file                           : %.3f ms" % t(lambda: bu.crc16(f[77:])))
print("codebooks only (basislz_decode, no slice)   : %.3f ms" % t(lambda: bu.basislz_decode(f)))
print("codebooks + slice 0                         : %.3f ms" % t(lambda: bu.basislz_decode(f, 0)))
