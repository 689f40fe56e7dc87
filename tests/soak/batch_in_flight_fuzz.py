"""Soak: bu_uastc_transcode_batch_in_flight on random slice tables against the reference's known answers -- random targets, stream counts 1..8,
slice counts 1..300 (beyond the 96-run table), sizes from one block to 2^23, random contiguity (slices carved out of shared allocations in order,
so they merge into runs), zero-length slices, pitches that allow rectangular tiles / none / no multiple of 64, one bad block per case in a random
slice with the batch-wide index expected back.  FUZZ_SECONDS (default 60), FUZZ_SEED."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from basisu_rs_amd import BasisuError, Context, _lib, synth
ctx = Context(0)
g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
dev = torch.device("cuda", 0)
gu = torch.from_numpy(g["uastc"]).to(dev)
TG = {"astc": (_lib.ASTC, 16), "bc7": (_lib.BC7, 16), "etc1": (_lib.ETC1, 8), "etc2": (_lib.ETC2, 16), "rgba": (_lib.RGBA32, 64)}
gw = {k: torch.from_numpy(g[k]).to(dev) for k in TG}
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
budget = float(os.environ.get("FUZZ_SECONDS", "60"))
t0 = time.time(); cases = 0; blocks_done = 0
status = torch.empty(1, dtype=torch.int64, device=dev)
while time.time() - t0 < budget:
    name = list(TG)[rng.integers(0, 5)]
    fmt, bb = TG[name]
    bpr = int(rng.choice([0, 64, 128, 192, 512, 1024, 100])) if name != "rgba" else int(rng.choice([64, 128, 192, 512, 1024, 100]))
    unit = bpr if bpr else 1
    kind = rng.integers(0, 4)
    if kind == 0:    # a few large slices
        sizes = [int(rng.integers(1 << 18, 1 << 22)) for _ in range(rng.integers(1, 7))]
    elif kind == 1:  # a crowd of small ones
        sizes = [int(rng.integers(0, 1 << 17)) for _ in range(rng.integers(1, 300))]
    elif kind == 2:  # one array, equal slices
        sizes = [int(rng.choice([4096, 65536, 1 << 18]))] * int(rng.integers(1, 128))
    else:            # everything
        sizes = [int(rng.choice([0, 1, 63, 1024, 65536, 1 << 20, int(rng.integers(1, 1 << 21))])) for _ in range(rng.integers(1, 40))]
    if name == "rgba":
        budget_blocks = 1 << 22  # 64 B per block
        sizes = [max(unit, s // unit * unit) if s else 0 for s in sizes]
    else:
        budget_blocks = 1 << 24
        if bpr and rng.integers(0, 2):
            sizes = [s // unit * unit for s in sizes]
    while sum(sizes) > budget_blocks:
        sizes.pop()
    if not sizes or not any(sizes):
        continue
    n = len(sizes)
    # allocations: consecutive slices share one with probability 0.6 (they then merge into a run); gaps otherwise
    groups, cur = [], [0]
    for i in range(1, n):
        if rng.random() < 0.6: cur.append(i)
        else: groups.append(cur); cur = [i]
    groups.append(cur)
    ins, outs, idxs = [None] * n, [None] * n, [None] * n
    for grp in groups:
        tot = sum(sizes[i] for i in grp)
        idx = torch.from_numpy(synth.gold_indices(max(tot, 1), seed=int(rng.integers(0, 1 << 30)))).to(dev)[:tot]
        gin = gu[idx].contiguous() if tot else torch.zeros((0, 16), dtype=torch.uint8, device=dev)
        gout = torch.zeros((max(tot, 1), bb), dtype=torch.uint8, device=dev)
        o = 0
        for i in grp:
            ins[i], outs[i], idxs[i] = gin[o:o + sizes[i]], gout[o:o + sizes[i]], idx[o:o + sizes[i]]
            o += sizes[i]
    ns = int(rng.integers(1, 9))
    use_base = bool(rng.integers(0, 2))
    base = None
    if use_base:
        base, b = [], int(rng.integers(0, 1 << 30))
        for s in sizes:
            base.append(b); b += s + int(rng.integers(0, 3)) * 1000
    ctx.status_word_reset(status); torch.cuda.synchronize()
    ctx.transcode_batch_in_flight(fmt, [x.data_ptr() if x.numel() else 0 for x in ins], sizes, [x.data_ptr() if x.numel() else 0 for x in outs], blocks_per_row=bpr,
                                  index_base=base, d_status=status, n_streams=ns)
    ctx.synchronize()
    ctx.status_word_check(int(status.item()))
    for i in range(n):
        if sizes[i] == 0: continue
        got = outs[i]
        if name == "rgba":
            got = got.view(sizes[i] // bpr, 4, bpr, 16).permute(0, 2, 1, 3).reshape(sizes[i], 64)
        assert torch.equal(got, gw[name][idxs[i]]), (cases, name, bpr, ns, i, sizes[i])
    # one bad block somewhere: the batch-wide index of the FIRST one comes back
    nz = [i for i in range(n) if sizes[i]]
    k = nz[int(rng.integers(0, len(nz)))]
    pos = int(rng.integers(0, sizes[k]))
    keep = ins[k][pos, 0].item()
    ins[k][pos, 0] = 69
    ctx.status_word_reset(status); torch.cuda.synchronize()
    ctx.transcode_batch_in_flight(fmt, [x.data_ptr() if x.numel() else 0 for x in ins], sizes, [x.data_ptr() if x.numel() else 0 for x in outs], blocks_per_row=bpr,
                                  index_base=base, d_status=status, n_streams=ns)
    ctx.synchronize()
    expect = (base[k] if use_base else sum(sizes[:k])) + pos
    try:
        ctx.status_word_check(int(status.item()))
        raise SystemExit("case %d: the bad block was not reported" % cases)
    except BasisuError as e:
        assert e.first_bad_block == expect, (cases, name, e.first_bad_block, expect)
    ins[k][pos, 0] = keep
    cases += 1; blocks_done += 2 * sum(sizes)
    if cases % 50 == 0:
        print("%d cases, %.1f M blocks, %.0f s" % (cases, blocks_done / 1e6, time.time() - t0), flush=True)
print("batch_in_flight_fuzz: %d cases, %.1f M blocks transcoded, every slice equal to the known answers, every first-error index right (seed %s)" % (
    cases, blocks_done / 1e6, os.environ.get("FUZZ_SEED", "1")))
