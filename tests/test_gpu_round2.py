"""Round-2 GPU parity tests: the known-answer blind spots on the DEVICE build (the host-emulation tests cover the same
inputs through a different compiler back-end), the claims of the header that had no test (graph capture, RGBA32 launch
splitting, ETC1S above 2^18 blocks), the multi-GPU entry points of the C ABI, and the ETC1S identity on HIP outputs.
Everything goes through the C ABI; bit-exact.  Run on the GPU box: pytest -m gpu."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

from basisu_rs_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL = ["astc", "bc7", "etc1", "etc2", "rgba"]
FMT = {"astc": _lib.ASTC, "bc7": _lib.BC7, "etc1": _lib.ETC1, "etc2": _lib.ETC2}


def _gpu_lenient(ctx, target, blocks):
    """per-block device results for a batch that may hold invalid blocks: device API + status word, outputs of the valid
    blocks are compared by the caller (the slice API would abort at the first bad block)"""
    import torch

    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    n = blocks.shape[0]
    d_in = torch.from_numpy(blocks).cuda()
    if target == "rgba":
        d_out = torch.empty((4, n, 16), dtype=torch.uint8, device="cuda")  # one block row: pixel row r of block i at [r, i]
        ctx.transcode_device(_lib.RGBA32, d_in, n, d_out, blocks_per_row=n)
        torch.cuda.synchronize()
        return np.ascontiguousarray(d_out.cpu().numpy().transpose(1, 0, 2)).reshape(n, 64)
    bb = _lib.BLOCK_BYTES[FMT[target]]
    d_out = torch.empty((n, bb), dtype=torch.uint8, device="cuda")
    ctx.transcode_device(FMT[target], d_in, n, d_out)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


def _compare(ctx, oracle, target, blocks):
    oo, ost = oracle.batch(target, blocks)
    out = _gpu_lenient(ctx, target, blocks)
    ok = ost == 0
    bad = np.nonzero((out[ok] != oo[ok]).any(axis=1))[0]
    assert bad.size == 0, "%s differs: block %s mode %d" % (target, blocks[ok][bad[0]].tobytes().hex(), synth.block_modes(blocks[ok][bad[:1]])[0])
    return out, ost


def _solid_blocks():
    """UASTC mode 8 (solid colour) over the 7^4 = 2 401 corner colours: 0, 1, 2, 127, 128, 254, 255 per channel"""
    vals = np.array([0, 1, 2, 127, 128, 254, 255], dtype=np.uint64)
    r, g, b, a = np.meshgrid(vals, vals, vals, vals, indexing="ij")
    rgba = (r | (g << 8) | (b << 16) | (a << 24)).reshape(-1)
    rng = np.random.default_rng(9)
    blocks = rng.integers(0, 256, size=(rgba.size, 16), dtype=np.uint8)
    lo = blocks[:, :8].copy().view("<u8").reshape(-1)
    lo = (lo & ~np.uint64((1 << 37) - 1)) | np.uint64(0x17) | (rgba << np.uint64(5))  # mode 8 code = 0b10111
    blocks[:, :8] = lo.view(np.uint8).reshape(-1, 8)
    return blocks, rgba


def test_solid_colour_corners_take_the_bc7_mode5_fallback_on_the_device(ctx, oracle):
    """bc7.rs:335-352: UASTC mode 8 -> BC7 mode 5 exactly when one channel is 0 and another 255.  The reference vectors
    never reach it and the random atlases hit it once; here 2 401 corner colours, every target, on the HIP build."""
    blocks, rgba = _solid_blocks()
    assert (synth.block_modes(blocks) == 8).all()
    for t in ALL:
        out, st = _compare(ctx, oracle, t, blocks)
        assert (st == 0).all()
        if t == "bc7":
            ch = np.stack([(rgba >> np.uint64(8 * k)) & np.uint64(0xFF) for k in range(4)], axis=1)
            want5 = (ch == 0).any(axis=1) & (ch == 255).any(axis=1)
            is5 = (out[:, 0] & 0x3F) == 0x20  # unary prefix of BC7 mode 5
            is6 = (out[:, 0] & 0x7F) == 0x40
            assert (is5 == want5).all() and (is6 == ~want5).all() and want5.sum() == 434  # 7^4 - 2*6^4 + 5^4


@pytest.mark.parametrize("target", ALL)
def test_per_mode_dense_blocks_match_oracle_on_the_device(ctx, golden, oracle, target):
    """every mode equally (12 800 blocks each): a golden block's mode code, everything behind it random -- invalid pattern
    indices included, whose status must match too"""
    rng = np.random.default_rng(5)
    base = np.repeat(golden["uastc"], 400, axis=0)
    noise = rng.integers(0, 256, size=base.shape, dtype=np.uint8)
    blocks = noise.copy()
    blocks[:, 0] = (base[:, 0] & 0x7F) | (noise[:, 0] & 0x80)
    out, ost = _compare(ctx, oracle, target, blocks)
    # the lowest failing block and its status through the slice-level error contract
    import torch

    d_in = torch.from_numpy(blocks).cuda()
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    n = blocks.shape[0]
    if target == "rgba":
        ctx.transcode_device(_lib.RGBA32, d_in, n, torch.empty((n, 64), dtype=torch.uint8, device="cuda"), blocks_per_row=n, d_status=status)
    else:
        ctx.transcode_device(FMT[target], d_in, n, torch.empty((n, _lib.BLOCK_BYTES[FMT[target]]), dtype=torch.uint8, device="cuda"), d_status=status)
    torch.cuda.synchronize()
    word = int(status.item()) & 0xFFFFFFFFFFFFFFFF
    first = int(np.nonzero(ost)[0][0])
    assert word >> 8 == first and word & 0xFF == int(ost[first])


def test_etc2_alpha_modes_with_a_zero_table_multiplier(ctx, golden, oracle):
    """etc.rs:278-279: etc2tm == 0 (multiplier 0 / table 0) on the alpha modes 9-17; no reference vector has it"""
    rng = np.random.default_rng(12)
    keep = np.isin(synth.block_modes(golden["uastc"]), np.arange(9, 18))
    base = np.repeat(golden["uastc"][keep], 200, axis=0).copy()
    modes = synth.block_modes(base)
    # the 8-bit etc2tm field sits behind code, bc1 hints (2 or 1), 8 ETC1 flag bits and the 5-bit bias (absent in 10-12)
    code_size = {9: 5, 10: 3, 11: 2, 12: 3, 13: 5, 14: 5, 15: 7, 16: 6, 17: 6}
    v = base[:, :8].copy().view("<u8").reshape(-1)
    for m, cs in code_size.items():
        m1012 = 10 <= m <= 12
        pos = cs + (1 if m1012 else 2) + 8 + (0 if m1012 else 5)
        sel = modes == m
        v[sel] &= ~(np.uint64(0xFF) << np.uint64(pos))
    base[:, :8] = v.view(np.uint8).reshape(-1, 8)
    noise = rng.integers(0, 256, size=(base.shape[0], 6), dtype=np.uint8)
    base[:, 10:] = noise  # different weights per copy
    assert (synth.block_modes(base) == modes).all()
    for t in ("etc2", "etc1", "rgba"):
        _, st = _compare(ctx, oracle, t, base)
        assert (st == 0).all()


def test_etc1s_hip_outputs_decode_to_the_rgba_outputs(ctx, oracle):
    """The ETC1S path has no reference vectors; exact cross-check on the HIP outputs themselves: the ETC1
    block bu_etc1s_transcode_etc1 emits, decoded with plain ETC1 rules (independent decoder), equals the texels
    bu_etc1s_decode_rgba writes -- at 2^19 blocks, above the per-slice sizes the other tests use."""
    from basisu_rs_amd import etc1s_selector_from_rows

    ep, rows = synth.etc1s_codebooks(4096, 8192, seed=3)
    sel = etc1s_selector_from_rows(rows)
    nbx, nby = 1024, 512
    n = nbx * nby
    idx = synth.etc1s_indices(n, 4096, 8192, seed=31)
    etc1 = ctx.etc1s_transcode_to_etc1(idx, ep, sel).reshape(n, 8)
    rgba = ctx.etc1s_decode_to_rgba(idx, None, nbx, nby, ep, sel).reshape(nby, 4, nbx, 16)
    lin = np.ascontiguousarray(rgba.transpose(0, 2, 1, 3)).reshape(n, 64)
    # against the oracle at full size
    assert (etc1.reshape(-1) == oracle.etc1s_to_etc1(idx, ep, sel)).all()
    assert (rgba.reshape(-1) == oracle.etc1s_to_rgba(idx, None, nbx, nby, ep, sel)).all()
    # the identity, on a sample (the per-block decoder call is a Python loop)
    rng = np.random.default_rng(1)
    for i in rng.choice(n, 4096, replace=False):
        out = np.zeros(64, dtype=np.uint8)
        blk = np.ascontiguousarray(etc1[i])
        oracle.lib.bu_oracle_decode_etc1_block(blk.ctypes.data, out.ctypes.data)
        assert (out == lin[i]).all(), i
        assert etc1[i][3] & 3 == 3  # diff = 1, flip = 1 (basis_lz/mod.rs:177)


def test_rgba32_launch_splitting_above_2_pow_26_blocks(ctx, golden):
    """RGBA32 pieces end on whole block rows so that image addressing stays launch-relative (bu_launch_uastc): 2^26 blocks +
    3 block rows of 4096, 4 GiB of output, an invalid block behind the split reported with its global index"""
    import torch

    from basisu_rs_amd import BasisuError

    bpr = 4096
    n = (1 << 26) + 3 * bpr
    gu = torch.from_numpy(golden["uastc"]).cuda()
    gr = torch.from_numpy(golden["rgba"]).cuda()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(16)
    idx = torch.randint(0, 608, (n,), device="cuda", generator=gen)
    d_in = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    for lo in range(0, n, 1 << 22):
        d_in[lo:lo + (1 << 22)] = gu[idx[lo:lo + (1 << 22)]]
    d_out = torch.empty((n // bpr, 4, bpr, 16), dtype=torch.uint8, device="cuda")  # block row, pixel row, block, 16 B
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    ctx.status_word_reset(status)
    ctx.transcode_device(_lib.RGBA32, d_in, n, d_out, blocks_per_row=bpr, d_status=status)
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    rows = n // bpr
    for by in (0, 1, (1 << 26) // bpr - 1, (1 << 26) // bpr, rows - 1):  # around the split and at both ends
        got = d_out[by].permute(1, 0, 2).reshape(bpr, 64)
        assert torch.equal(got, gr[idx[by * bpr:(by + 1) * bpr]]), by
    bad = (1 << 26) + bpr + 5
    d_in[bad, 0] = 69
    ctx.status_word_reset(status)
    ctx.transcode_device(_lib.RGBA32, d_in, n, d_out, blocks_per_row=bpr, d_status=status)
    torch.cuda.synchronize()
    with pytest.raises(BasisuError, match="invalid mode index") as e:
        ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    assert e.value.first_bad_block == bad


def test_rgba32_device_entry_rejects_a_ragged_last_block_row(ctx, golden):
    """include/basisu_hip.h: n_blocks must be whole block rows for RGBA32 (rows 4by+1..3 are stored at the image pitch)"""
    import torch

    from basisu_rs_amd import BasisuError

    d_in = torch.from_numpy(golden["uastc"][:5].copy()).cuda()
    guard = torch.full((8 * 64 + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
    with pytest.raises(BasisuError, match="invalid argument"):
        ctx.transcode_device(_lib.RGBA32, d_in, 5, guard, blocks_per_row=4)
    torch.cuda.synchronize()
    assert bool((guard == 0xA5).all())
    ctx.transcode_device(_lib.RGBA32, d_in[:4], 4, guard, blocks_per_row=4)
    torch.cuda.synchronize()
    assert bool((guard[4 * 64:] == 0xA5).all()) and not bool((guard[:4 * 64] == 0xA5).all())


def test_device_entry_points_replay_from_a_hip_graph(ctx, golden):
    """include/basisu_hip.h promises the *_device entry points are graph-capturable: capture status reset + two transcodes
    (BC7, RGBA32) into a hipGraph through torch's capture, replay on new input contents, compare with the vectors"""
    import torch

    n, bpr = 1 << 16, 256
    gu = torch.from_numpy(golden["uastc"]).cuda()
    d_in = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    d_bc7 = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    d_rgba = torch.empty((n // bpr, 4, bpr, 16), dtype=torch.uint8, device="cuda")
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    idx0 = torch.from_numpy(synth.gold_indices(n, seed=1)).cuda()
    d_in.copy_(gu[idx0])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):  # warm-up outside the capture (first launches load the code object)
        ctx.status_word_reset(status, stream=s)
        ctx.transcode_device(_lib.BC7, d_in, n, d_bc7, d_status=status, stream=s)
        ctx.transcode_device(_lib.RGBA32, d_in, n, d_rgba, blocks_per_row=bpr, d_status=status, stream=s)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.status_word_reset(status, stream=s)
        ctx.transcode_device(_lib.BC7, d_in, n, d_bc7, d_status=status, stream=s)
        ctx.transcode_device(_lib.RGBA32, d_in, n, d_rgba, blocks_per_row=bpr, d_status=status, stream=s)
    for seed in (2, 3):
        idx = torch.from_numpy(synth.gold_indices(n, seed=seed)).cuda()
        d_in.copy_(gu[idx])
        d_bc7.zero_()
        d_rgba.zero_()
        g.replay()
        torch.cuda.synchronize()
        ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
        assert (d_bc7.cpu().numpy() == golden["bc7"][idx.cpu().numpy()]).all()
        lin = d_rgba.permute(0, 2, 1, 3).reshape(n, 64).cpu().numpy()
        assert (lin == golden["rgba"][idx.cpu().numpy()]).all()
    # a replay over an invalid block reports it through the captured status word
    d_in[777, 0] = 69
    g.replay()
    torch.cuda.synchronize()
    assert (int(status.item()) & 0xFFFFFFFFFFFFFFFF) >> 8 == 777


# ---- multi-GPU entry points of the C ABI -----------------------------------------------------------------------------
def _ptr_array(vals):
    return (ctypes.c_void_p * len(vals))(*vals)


@pytest.mark.parametrize("n_ctx", [1, 2, 3, 8])
def test_array_transcode_sharded_with_virtual_ranks_on_one_device(golden, n_ctx):
    """bu_array_transcode_sharded, one process driving n contexts (here all on device 0 -- virtual ranks; distinct devices
    take the same path with hipMemcpyPeerAsync crossing xGMI): every context's full buffer must equal the unsharded result"""
    import torch

    from basisu_rs_amd import BasisuError, Context, sharded

    lib = _lib.load()
    n_slices, bps = 37, 2048  # ragged over 2, 3 and 8 ranks
    idx = synth.gold_indices(n_slices * bps, seed=55)
    blocks = golden["uastc"][idx]
    want = torch.from_numpy(golden["bc7"][idx]).cuda()
    ctxs = [Context(0) for _ in range(n_ctx)]
    try:
        ins, fulls = [], []
        for r in range(n_ctx):
            lo, hi = sharded.partition(n_slices, n_ctx, r)
            ins.append(torch.from_numpy(blocks[lo * bps:hi * bps].copy()).cuda() if hi > lo else torch.empty((0, 16), dtype=torch.uint8, device="cuda"))
            fulls.append(torch.zeros((n_slices * bps, 16), dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        handles = _ptr_array([c.handle.value for c in ctxs])
        bad = ctypes.c_uint64(0)
        st = lib.bu_array_transcode_sharded(handles, n_ctx, _lib.BC7, _ptr_array([t.data_ptr() if t.numel() else None for t in ins]), n_slices, bps,
                                            _ptr_array([t.data_ptr() for t in fulls]), 1, ctypes.byref(bad))
        assert st == 0, lib.bu_status_string(st)
        torch.cuda.synchronize()
        for r in range(n_ctx):
            assert torch.equal(fulls[r], want), r
        # an invalid block in the LAST shard and one in the first: the first (lowest array-wide index) is reported
        if n_ctx > 1:
            lo_last = sharded.partition(n_slices, n_ctx, n_ctx - 1)[0]
            ins[-1][5, 0] = 69
            ins[0][1234, 0] = 69
            st = lib.bu_array_transcode_sharded(handles, n_ctx, _lib.BC7, _ptr_array([t.data_ptr() if t.numel() else None for t in ins]), n_slices, bps,
                                                _ptr_array([t.data_ptr() for t in fulls]), 1, ctypes.byref(bad))
            assert st == _lib.ERR_INVALID_MODE and bad.value == 1234
            ins[0][1234, 0] = int(blocks[1234, 0])
            st = lib.bu_array_transcode_sharded(handles, n_ctx, _lib.BC7, _ptr_array([t.data_ptr() if t.numel() else None for t in ins]), n_slices, bps,
                                                _ptr_array([t.data_ptr() for t in fulls]), 1, ctypes.byref(bad))
            assert st == _lib.ERR_INVALID_MODE and bad.value == lo_last * bps + 5
    finally:
        for c in ctxs:
            c.close()


def test_rccl_inplace_allgather_with_one_rank(ctx, golden):
    """bu_comm_* / bu_allgather_inplace resolve RCCL at run time and run a (degenerate) world-size-1 collective in place"""
    import torch

    lib = _lib.load()
    ident = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
    st = lib.bu_comm_unique_id(ident)
    assert st == 0, lib.bu_status_string(st)
    comm = ctypes.c_void_p(0)
    st = lib.bu_comm_create(ctx.handle, 1, 0, ident, ctypes.byref(comm))
    assert st == 0, (lib.bu_status_string(st), lib.bu_last_error(ctx.handle))
    n = 1 << 16
    idx = synth.gold_indices(n, seed=8)
    d_in = torch.from_numpy(golden["uastc"][idx]).cuda()
    full = torch.zeros((n, 16), dtype=torch.uint8, device="cuda")
    ctx.transcode_device(_lib.BC7, d_in, n, full)
    st = lib.bu_allgather_inplace(comm, ctypes.c_void_p(full.data_ptr()), n * 16, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, lib.bu_last_error(ctx.handle)
    torch.cuda.synchronize()
    assert (full.cpu().numpy() == golden["bc7"][idx]).all()
    lib.bu_comm_destroy(comm)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ipc_worker(rank, world, port, q):
    """two PROCESSES, each with its own context (both on device 0 when the box has one GPU, on devices 0 and 1 otherwise):
    transcode the own shard, exchange HIP IPC handles, pull the peer's shard with bu_allgather_peer"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basisu_rs_amd import Context, sharded

        dev = rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        lib = _lib.load()
        c = Context(dev)
        g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
        n_slices, bps = 8, 4096
        idx = synth.gold_indices(n_slices * bps, seed=66)
        lo, hi = sharded.partition(n_slices, world, rank)
        shard_bytes = (n_slices // world) * bps * 16
        p = ctypes.c_void_p(0)
        assert lib.bu_device_alloc(c.handle, world * shard_bytes, ctypes.byref(p)) == 0
        d_in = torch.from_numpy(g["uastc"][idx[lo * bps:hi * bps]]).cuda()
        sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.bu_uastc_transcode_device(c.handle, _lib.BC7, d_in.data_ptr(), (hi - lo) * bps, p.value + rank * shard_bytes, 1, lo * bps, None, sp) == 0
        torch.cuda.synchronize()
        hb = (ctypes.c_uint8 * _lib.IPC_HANDLE_BYTES)()
        st = lib.bu_ipc_export(c.handle, p, hb)
        assert st == 0, lib.bu_last_error(c.handle)
        handles = [None] * world
        dist.all_gather_object(handles, bytes(hb))
        peers = (ctypes.c_void_p * world)()
        for r in range(world):
            if r == rank:
                peers[r] = p.value
                continue
            pp = ctypes.c_void_p(0)
            st = lib.bu_ipc_open(c.handle, (ctypes.c_uint8 * _lib.IPC_HANDLE_BYTES)(*handles[r]), ctypes.byref(pp))
            assert st == 0, lib.bu_last_error(c.handle)
            peers[r] = pp.value
        dist.barrier()  # the peers' shards are complete
        st = lib.bu_allgather_peer(c.handle, p, peers, world, rank, shard_bytes, sp)
        assert st == 0, lib.bu_last_error(c.handle)
        torch.cuda.synchronize()
        host = np.empty(world * shard_bytes, dtype=np.uint8)
        assert lib.bu_memcpy(c.handle, host.ctypes.data, p, host.size, 0) == 0
        ok = bool((host.reshape(-1, 16) == g["bc7"][idx]).all())
        dist.barrier()
        for r in range(world):
            if r != rank:
                lib.bu_ipc_close(c.handle, ctypes.c_void_p(peers[r]))
        dist.barrier()
        lib.bu_device_free(c.handle, p)
        c.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_peer_pull_allgather_between_two_processes():
    """bu_ipc_* + bu_allgather_peer across process boundaries (the one-process-per-GPU deployment)"""
    import torch.multiprocessing as mp

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_ipc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]


def _nccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from basisu_rs_amd import BasisuError, Context, sharded

        c = Context(rank)
        g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
        res = []
        for n_slices in (8, 7, 1):
            bps = 4096
            idx = synth.gold_indices(n_slices * bps, seed=3 + n_slices)
            slices = torch.from_numpy(g["uastc"][idx].reshape(n_slices, bps, 16).copy()).cuda()
            full = sharded.transcode_array_sharded(slices, sharded.gpu_transcode_fn(c, _lib.BC7))
            res.append(bool((full.cpu().numpy() == g["bc7"][idx].reshape(n_slices, bps, 16)).all()))
        # a failing block in rank 1's range raises the SAME error on both ranks
        blocks = synth.atlas_err(g["uastc"], 8 * 4096, bad_at=[5 * 4096 + 9])
        slices = torch.from_numpy(blocks.reshape(8, 4096, 16).copy()).cuda()
        try:
            sharded.transcode_array_sharded(slices, sharded.gpu_transcode_fn(c, _lib.BC7))
            res.append(False)
        except BasisuError as e:
            res.append(e.first_bad_block == 5 * 4096 + 9 and "invalid mode index" in str(e))
        c.close()
        q.put((rank, all(res)))
    finally:
        dist.destroy_process_group()


def test_two_rank_nccl_product_path():
    """the N = 2 product path: two processes, two devices, RCCL, sharded.gpu_transcode_fn, against the goldens
    (skipped on a one-GPU box; the gloo test covers the driver logic on CPU)"""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]


def test_bench_array512_mode_on_one_gpu():
    """bench.py --config array512 (BASELINE config 5 as a bench mode) runs to one JSON line; with BENCH_FORCE_DIST the N > 1
    branch (RCCL init, barriers, both gather transports) runs with a single rank"""
    import json
    import subprocess

    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "array512", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["config"]["blocks_per_step"] == 512 * 65536
    assert line["allgather"]["rccl_inplace"].get("verified") is True, line["allgather"]
    assert line["allgather"]["peer_pull"].get("verified") is True, line["allgather"]


def test_device_side_data_crc_folds_pieces_gaps_and_tails(ctx, golden, oracle):
    """Large UASTC files: the data CRC (basis.rs:338-341) is computed by bu_crc16_pieces_kernel over the uploaded slice bytes,
    one register per 64 KiB piece, and folded on the host with the bytes the device never sees.  A flipped bit in EVERY
    position class must be caught -- slice table, first piece, a middle piece, the tail behind the last whole piece, the
    second run of a file whose runs are separated by a gap -- and intact files of awkward sizes must pass."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import basis_builder as bb
    import basisu_rs_amd as bu

    def check(f, n_expected):
        imgs = bu.read_to_bc7(f, ctx)
        assert len(imgs) == n_expected
        st, _, want = oracle.read_to("bc7", f)
        assert st == 0
        for g_, (_, _, _, data) in zip(imgs, want):
            assert g_.data.tobytes() == data.tobytes()

    # one run: 1.5 MB + an odd tail, and an exact multiple of 64 KiB
    for nblk in (96_000 + 37, 1 << 17):
        blocks = golden["uastc"][synth.gold_indices(nblk, seed=nblk & 255)]
        f = bu.write_uastc_file([dict(data=blocks, orig_w=4, orig_h=4 * nblk, nbx=1, nby=nblk)])
        check(f, 1)
        ofs = bu.read_slice_descs(f)[0].file_ofs
        for pos in (80, ofs + 5, ofs + 70_000, ofs + 16 * nblk - 3, len(f) - 1):
            g = bytearray(f)
            g[pos] ^= 0x04
            with pytest.raises(bu.BasisuError, match="Data CRC16 failed"):
                bu.read_to_bc7(bytes(g), ctx)
    # several slices: back-to-back ones merge into one run; basis_builder can leave a gap between runs
    dims = [(256, 200), (128, 100), (64, 50), (300, 250)]
    blocks = [golden["uastc"][synth.gold_indices(x * y, seed=i + 3)] for i, (x, y) in enumerate(dims)]
    f = bb.uastc_file(blocks, dims)
    assert len(f) >= 1 << 20
    check(f, 4)
    descs = bu.read_slice_descs(f)
    for d in descs:
        for pos in (d.file_ofs, d.file_ofs + d.file_size - 1):
            g = bytearray(f)
            g[pos] ^= 0x80
            with pytest.raises(bu.BasisuError, match="Data CRC16 failed"):
                bu.read_to_bc7(bytes(g), ctx)
    # the whole-file RGBA path uploads the same runs
    st, hdr, want = oracle.read_to("rgba", f)
    h, got = bu.read_to_rgba(f, ctx)
    assert st == 0 and len(got) == 4 and all(a.data.tobytes() == w[3].tobytes() for a, w in zip(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [786433, 1000000, 1572865])
def test_etc_shapes_at_sizes_between_the_tile_multiples(ctx, golden, n):
    """just above the switch to the 4096-block shape, a size that is no multiple of 64, and 1.5 tiles per CU (the launcher
    sizes the ETC tile at run time: two rounds of 3072-block tiles): outputs against the known answers, and a bad block in
    the last tile reported with its own index"""
    from basisu_rs_amd import BasisuError

    idx = synth.gold_indices(n, seed=n)
    blocks = golden["uastc"][idx]
    for name, fmt in (("etc1", _lib.ETC1), ("etc2", _lib.ETC2)):
        got = ctx.transcode(fmt, blocks).reshape(n, -1)
        assert (got == golden[name][idx]).all(), name
    bad = blocks.copy()
    bad[n - 3, 0] = 0x45  # the one 7-bit prefix that is no mode code (uastc.rs:560-577)
    with pytest.raises(BasisuError) as e:
        ctx.transcode(_lib.ETC1, bad)
    assert e.value.first_bad_block == n - 3


@pytest.mark.gpu
@pytest.mark.parametrize("n", [262144, 263168, 786432, 787456, 3145728, 3146752])
def test_every_target_on_both_sides_of_the_launch_shape_thresholds(ctx, golden, n):
    """the launcher changes workgroup shape at one tile per CU (all targets), three tiles per CU (ETC1 / ETC2) and 3 Mi blocks
    (RGBA32), and switches the generation priorities off for uneven tile counts: device-resident slices of exactly those sizes
    and one block row more, all five targets against the known answers (torch-side comparison, nothing leaves the GPU)"""
    import torch

    gu = torch.from_numpy(golden["uastc"]).cuda()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(n)
    idx = torch.randint(0, 608, (n,), device="cuda", generator=gen)
    d_in = gu[idx].contiguous()
    status = torch.empty(1, dtype=torch.int64, device="cuda")
    for name, fmt, ob in (("bc7", _lib.BC7, 16), ("astc", _lib.ASTC, 16), ("etc1", _lib.ETC1, 8), ("etc2", _lib.ETC2, 16), ("rgba", _lib.RGBA32, 64)):
        want = torch.from_numpy(golden[name]).cuda()[idx]
        d_out = torch.zeros((n, ob), dtype=torch.uint8, device="cuda")
        ctx.status_word_reset(status)
        ctx.transcode_device(fmt, d_in, n, d_out, blocks_per_row=1024, d_status=status)
        torch.cuda.synchronize()
        ctx.status_word_check(int(status.item()))
        if name == "rgba":  # image rows -> per-block texel rows
            got = d_out.view(n // 1024, 4, 1024, 16).permute(0, 2, 1, 3).reshape(n, 64)
        else:
            got = d_out
        assert torch.equal(got, want), name
        del d_out, want
