"""The N>1 path on CPU: 2 processes, gloo backend, the per-shard transcode injected (oracle), checking
partitioning and the all-gather reassembly against the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_slices, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basisu_rs_amd import sharded, synth
        from oracle.pyoracle import Oracle

        oracle = Oracle()
        g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
        bps = 64
        idx = synth.gold_indices(n_slices * bps, seed=3)
        slices = torch.from_numpy(g["uastc"][idx].reshape(n_slices, bps, 16).copy())

        def fn(t):
            out, st = oracle.batch("bc7", t.numpy())
            assert (st == 0).all()
            return torch.from_numpy(out)

        full = sharded.transcode_array_sharded(slices, fn)
        want = g["bc7"][idx].reshape(n_slices, bps, 16)
        ok = bool((full.numpy() == want).all())
        lo, hi = sharded.partition(n_slices, world, rank)
        local = sharded.transcode_array_sharded(slices, fn, gather=False)
        ok = ok and bool((local.numpy() == want[lo:hi]).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _worker_failing(rank, world, port, q):
    """rank 1's shard holds an invalid block: BOTH ranks must raise the same error (lowest failing block of the whole
    array) instead of rank 0 waiting in the all-gather until the collective times out"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basisu_rs_amd import sharded, synth
        from oracle.pyoracle import Oracle

        oracle = Oracle()
        g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
        n_slices, bps = 4, 32
        blocks = synth.atlas_err(g["uastc"], n_slices * bps, bad_at=[70, 100])  # both in rank 1's range (blocks 64..127)
        slices = torch.from_numpy(blocks.reshape(n_slices, bps, 16).copy())

        def fn(t, out, base):  # the product function's protocol: write `out`, return the status word
            res, st = oracle.batch("bc7", t.numpy())
            out.copy_(torch.from_numpy(res))
            bad = np.nonzero(st)[0]
            return sharded._CLEAR if bad.size == 0 else ((base + int(bad[0])) << 8) | int(st[bad[0]])

        fn.block_bytes = 16
        try:
            sharded.transcode_array_sharded(slices, fn)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def _worker_raising(rank, world, port, q):
    """rank 1's transcode RAISES (not a block error): rank 0 must not be left waiting in a collective"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from basisu_rs_amd import sharded

        slices = torch.zeros((4, 8, 16), dtype=torch.uint8)

        def fn(t, out, base):
            if rank == 1:
                raise ValueError("device fell over on rank 1")
            out.zero_()
            return sharded._CLEAR

        fn.block_bytes = 16
        try:
            sharded.transcode_array_sharded(slices, fn)
            q.put((rank, "no error"))
        except (RuntimeError, ValueError) as e:
            q.put((rank, type(e).__name__ + ": " + str(e)))
    finally:
        dist.destroy_process_group()


def _spawn(target, args_of_rank, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args_of_rank) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(q.get(timeout=5) for _ in range(world))


def test_failing_shard_raises_on_every_rank():
    res = _spawn(_worker_failing, ())
    assert res[0][1] == res[1][1] and "block 70 failed" in res[0][1], res


def test_a_raising_rank_stops_every_rank_before_the_gather():
    res = _spawn(_worker_raising, ())
    assert res[0] == (0, "RuntimeError: the transcode raised on another rank; no rank enters the all-gather"), res
    assert res[1] == (1, "ValueError: device fell over on rank 1"), res


@pytest.mark.parametrize("n_slices", [8, 7, 1])
def test_two_rank_sharded_transcode_reassembles(n_slices):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slices, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, True), (1, True)]


def test_partition_covers_everything_once():
    from basisu_rs_amd import sharded

    for n in (0, 1, 7, 8, 512, 513):
        for w in (1, 2, 3, 8):
            ranges = [sharded.partition(n, w, r) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus N` without a launcher must spawn torch.distributed.run as a CHILD (never re-exec) before it
    touches a GPU, with the rendezvous on 127.0.0.1 and its own arguments passed through; under a launcher (WORLD_SIZE
    set) it must not spawn anything.  Dry run: the command is printed instead of executed."""
    import json
    import subprocess

    env = dict(os.environ, BENCH_SELF_LAUNCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "3", "--config", "array512"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["self_launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-8:] == ["--gpus", "8", "--steps", "7", "--warmup", "3", "--config", "array512"] and cmd[-9].endswith("bench.py")
    assert d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") not in (None, "0")
    assert d["parent_initialised_cuda"] is False  # the parent only counted devices: nothing that a child could not re-do
    # the ranks inherit one hardware queue per stream from the launcher's environment (set before torch is imported: the HIP runtime
    # reads it when it initialises; four launches in flight need four queues of their own, and a rank's RCCL communicator brings streams of
    # its own: 16 in the multi-rank branch, profiles/r05_dist_branch_hw_queues.txt)
    assert d["GPU_MAX_HW_QUEUES"] == "16"


def test_bench_live_traffic_falls_back_without_a_gpu():
    """bench.py measures roofline.traffic with child rocprofv3 passes (a kernel trace, then two counter passes); where they cannot
    run (no GPU here, no rocprofv3, a run that is itself profiled) it must return a reason, not raise -- the committed passes are
    quoted then.  Third element: what the trace pass saw (None, or a dict that may only carry an error here)."""
    sys.path.insert(0, ROOT)
    import bench

    got = bench.live_traffic(timeout_s=120)
    assert len(got) == 3 and got[0] is None and isinstance(got[1], str) and got[1]
    assert got[2] is None or "kernel_avg_ns" not in got[2]
    os.environ["ROCPROF_TEST_MARKER"] = "1"  # a profiled run never starts a nested profiler
    try:
        assert bench.live_traffic() in ((None, "this run is itself being profiled", None), (None, "rocprofv3 not on PATH", None))
    finally:
        del os.environ["ROCPROF_TEST_MARKER"]


def test_bench_in_step_rule_on_recorded_windows():
    """bench.streams_out_of_step on per-stream start / end events as the GPU runs recorded them (profiles/r05_dist_branch_hw_queues.txt): a
    pipeline in step, the same bench beside an RCCL communicator with 8 hardware queues (two streams 5.7 ms behind), config 5's four
    2^23-block launches, a probe that found two streams on one queue, a stream without timed launches, one launch in flight"""
    sys.path.insert(0, ROOT)
    import bench

    plain = {"start_us": [11501.1, 11518.8, 11529.5, 11534.5], "end_us": [11615.2, 11630.3, 11641.3, 11650.9]}
    dist8 = {"start_us": [10411.3, 10418.6, 16101.9, 16116.7], "end_us": [10508.2, 10514.4, 16216.7, 16231.5]}
    a512 = {"start_us": [2744.9, 2775.6, 2829.7, 2860.7], "end_us": [6214.3, 6270.5, 6299.9, 6350.5]}
    oos, spread = bench.streams_out_of_step(plain, 5.82, 4)
    assert not oos and abs(spread - 33.4) < 0.01
    spread_plain = spread
    oos, spread = bench.streams_out_of_step(dist8, 5.74, 4)
    assert oos and spread > 5000
    oos, spread = bench.streams_out_of_step(a512, 174.5 / 4, 4)
    assert not oos and abs(spread - 115.8) < 0.01
    assert bench.streams_out_of_step(plain, 5.82, 4, queue_sharing=2)[0]          # the probe overrides a small spread
    assert bench.streams_out_of_step({"start_us": [100.0, -1.0, 120.0, 110.0], "end_us": [200.0, -1.0, 220.0, 210.0]}, 5.8, 4) == (False, 20.0)
    assert bench.streams_out_of_step(dist8, 5.74, 1) == (False, 0.0)             # one launch at a time: nothing to be in step with
    assert bench.streams_out_of_step(None, 5.8, 4) == (False, 0.0)
    assert bench.IN_STEP_PERIODS == 16
    # the bound scales with the window (round 6): K = 20 -> 10 periods, K = 8 -> in_flight + 4, K = 512 -> 16
    assert [bench.in_step_periods(k, 4) for k in (None, 8, 20, 24, 512)] == [16, 8, 10, 12, 16]
    assert bench.streams_out_of_step(plain, 5.82, 4, steps=20) == (False, spread_plain)     # 33.4 us = 5.7 periods: in step
    lag = {"start_us": [100.0, 105.0, 111.0, 100.0 + 13 * 5.8], "end_us": [220.0, 226.0, 231.0, 300.0]}
    assert not bench.streams_out_of_step(lag, 5.8, 4)[0]                                     # 13 periods: inside the absolute bound ...
    assert bench.streams_out_of_step(lag, 5.8, 4, steps=20)[0]                               # ... but 65 % of a K = 20 window
