#!/usr/bin/env python3
"""Headline benchmark: UASTC -> BC7 at 4096x4096 (64 atlases of 1 048 576 blocks per step) on N MI355X.

  python bench.py --gpus N --steps K --warmup W [--config atlas4096|array512]
  N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run the ranks come from the environment
  (RANK / LOCAL_RANK / WORLD_SIZE); started plainly (`python bench.py --gpus 8`) this process spawns
  `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD before it touches any GPU, relays the
  child's output and exits with its code.

  --config atlas4096 (default, the headline metric): weak scaling, every rank transcodes its own 64 atlases of 4096x4096 per step (one launch).
  --config array512  (BASELINE config 5): strong scaling, ONE texture array of 512 slices x 65 536 blocks per step, rank r
                     owns slices [r*512/N, (r+1)*512/N); value = all 33.5 M blocks / max-over-ranks time of the transcode;
                     the all-gather that reassembles the array on every rank is timed separately (RCCL in place, and
                     direct peer pulls) and never folded into `value`.

A step (--method batch, the default since round 6) = ONE launch over 64 synthetic atlases of 4096x4096 (1 048 576 blocks each) already resident in HBM in
their 64 separate allocations: one call of bu_uastc_transcode_batch_device on the caller's stream hands the library the slice table, and the
library issues ONE kernel -- a persistent grid that walks the 65 536 tiles of all runs, every run tiled as 64 x 16-block rectangles, the tiles
drawn by ticket.  A kernel's duration is what HIP events, the host clock and rocprofv3 measure alike: `value` = K x 64 x 2^20 blocks / the time of
the K timed launches (lead launches in front, an event on either side, no host synchronisation in between; median of --repeats windows), and the
line carries the same launch's per-kernel average from a child `rocprofv3 --kernel-trace` pass of the same run (they agree to a fraction of a
percent) and its HBM traffic from child --pmc passes.  Every step touches 2 GiB in + out (cold for L2 and the 256 MiB Infinity Cache).
Round 5's headline -- one launch per ATLAS, step i on context stream i % IN_FLIGHT under the shared launch policy, throughput counted as the
pipeline's completion period -- is still measured in every run, by its own method, and reported under `one_launch_per_atlas_in_flight`
(`--method pipeline` makes it the headline again): launches queued on one stream never overlap, and one launch over an atlas waits ~3.4 us for
HBM with the ALUs idle and then computes with HBM idle; independent atlases on several streams use both at once.  The pipeline needs a hardware
queue per stream (the library sees to that itself since round 6; GPU_MAX_HW_QUEUES=8, 16 beside an RCCL communicator, is still set here so that
ordinary streams are kept).
Every rank owns its own atlases (weak scaling); the transcode needs no data-path collective.  The all-gather that
reassembles the array is timed separately and reported under "allgather" -- never folded into `value`.

Prints ONE JSON line (rank 0): metric/value per the driver contract plus
  roofline      algorithmic bytes (32 B/block x 64 x 2^20 blocks per launch) / the launch's average duration measured with
                HIP events around the timed launches, against the 8 TB/s HBM peak; rocprofv3's own average for the same
                launch and its HBM traffic from this run's child rocprofv3 passes
  cpu_baseline  the oracle (C restatement of the reference CPU path, kind "port") timed on this box's
                host cores on a bounded sample of the same atlas
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NBX = NBY = 1024  # 4096x4096 px
N_BLOCKS = NBX * NBY
BYTES_PER_BLOCK = 32  # 16 read + 16 written
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ATLASES_PER_STEP = 64  # --method batch: one step = ONE launch over this many 4096x4096 atlases in separate allocations (at most BU_MULTI_RUNS = 96 runs per launch)
ETC1S_UNPINNED = "unpinned: the reference holds no ETC1S / BasisLZ vectors (tests/corpus_tests.rs:54-73 are #[ignore]d, the corpus absent); checked against the oracle's reading of the source"


IN_STEP_PERIODS = 16  # start events of a window's streams further apart than this many launch periods: the streams are not in step (long windows)


def in_step_periods(steps, in_flight):
    """the spread of the start events, in launch periods, up to which a window of `steps` timed launches counts as in step.  A pipeline in step has its start
    events in_flight - 1 periods apart, give or take a burst (recorded: 15-35 us at 5.7 us per period); an absolute bound of 16 periods is 80 % of a K = 20
    window -- a stream that far behind finishes its last timed launches with fewer partners and `latest start -> latest end` reads short -- so short windows
    get a bound that scales with them: K / 2 periods, at least in_flight + 4, at most 16.  (The window also carries TWO tail launches per stream, so a
    stream up to 2 x in_flight periods behind still has partners until its end event.)"""
    if not steps:
        return IN_STEP_PERIODS
    return max(in_flight + 4, min(IN_STEP_PERIODS, steps // 2))


def streams_out_of_step(streams, period_us, in_flight, queue_sharing=1, steps=None):
    """(out_of_step, start-event spread in us) of a pipelined window.  `streams` = {"start_us": [...], "end_us": [...]} per stream (-1: a stream
    without timed launches), from bu_time_last_window_streams.  A window "latest start event -> latest end event" holds its K completions only
    while the streams run in step: their start events then lie in_flight - 1 periods apart, give or take a burst.  Streams that share a hardware
    queue (`queue_sharing` > 1, bu_context_probe_streams) run at a fraction of the others' pace and end up thousands of microseconds behind.
    `steps` = the window's K (None: the absolute bound of long windows)."""
    if not streams or in_flight <= 1:
        return False, 0.0
    st = [x for x in streams["start_us"] if x >= 0]
    spread = (max(st) - min(st)) if st else 0.0
    return bool(spread > in_step_periods(steps, in_flight) * period_us or queue_sharing > 1), spread


def pmc_traffic():
    """HBM bytes per launch of the BC7 kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE with the
    gfx950 x2 correction + WRITE_SIZE; tools/gpu_pmc.sh), or None when no summary is committed"""
    import glob

    import re

    def version_key(path):  # r01_v9 < r01_v10: compare the numbers, not the characters
        return [int(x) for x in re.findall(r"\d+", os.path.basename(path))]

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_bc7.json")), key=version_key)
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return int(d["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT), d.get("SQ_INSTS_VALU")
    except Exception:
        return None


def pmc_array512():
    """rocprofv3's per-kernel average and the HBM traffic of BASELINE config 5's 2^25-block launch from the committed passes (tools/gpu_pmc.sh:
    --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs), or None"""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_bc7_array512.json")), key=lambda p: [int(x) for x in re.findall(r"\d+", os.path.basename(p))])
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        out = {"profile": os.path.relpath(files[-1], ROOT) + " (committed rocprofv3 passes over this launch; not measured in this run)"}
        if "trace_avg_ns" in d:
            out["kernel_avg_ns"] = round(d["trace_avg_ns"], 1)
            out["frac_by_rocprofv3_kernel_avg"] = round(BYTES_PER_BLOCK * (1 << 25) / d["trace_avg_ns"] / HBM_PEAK_GBS, 4)
        if "hbm_bytes_per_launch" in d:
            out["traffic"] = int(d["hbm_bytes_per_launch"])
            out["traffic_over_algorithmic"] = round(d["hbm_bytes_per_launch"] / (BYTES_PER_BLOCK * (1 << 25)), 4)
        return out
    except Exception:
        return None


def pmc_child():
    """`bench.py --pmc-child`: the program rocprofv3's child passes run (live_traffic): the headline BC7 kernel (launch policy from
    BENCH_PMC_POLICY) over 24 cold A-gold atlases of the headline size, nothing else.  Counter passes: once over each atlas, one launch
    at a time (the profiler serialises dispatches under --pmc anyway; bytes per launch do not depend on what runs beside it).  Trace pass
    (BENCH_PMC_ROUNDS rounds): the timed region's own native loop, BENCH_PMC_STREAMS launches in flight on the context's streams."""
    import torch

    from basisu_rs_amd import Context, _lib, synth

    ctx = Context(0)
    ctx.set_launch_policy(os.environ.get("BENCH_PMC_POLICY", "shared") == "shared")
    _lib.load().bu_time_set_enqueue_threads(ctx.handle, int(os.environ.get("BENCH_PMC_ENQ_THREADS", "0")))
    g = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
    dev = torch.device("cuda", 0)
    gu = torch.from_numpy(g["uastc"]).to(dev)
    ins, outs = [], []
    n_atl = ATLASES_PER_STEP if os.environ.get("BENCH_PMC_BATCH") == "1" else 24
    for k in range(n_atl):
        gen = torch.Generator(device=dev)
        gen.manual_seed(k + 1)
        ins.append(gu[torch.randint(0, 608, (N_BLOCKS,), device=dev, generator=gen)].contiguous())
        outs.append(torch.empty((N_BLOCKS, 16), dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    if os.environ.get("BENCH_PMC_BATCH") == "1":
        # the headline's step: ONE launch (bu_uastc_transcode_batch_device on the NULL stream) over ATLASES_PER_STEP atlases in separate allocations; the counter
        # passes run 4 such launches, the trace pass BENCH_PMC_ROUNDS of them back to back
        lib = _lib.load()
        ctx.set_launch_policy("auto")
        VPn, SZn = ctypes.c_void_p * n_atl, ctypes.c_size_t * n_atl
        a_in, a_n, a_out = VPn(*[t.data_ptr() for t in ins]), SZn(*([N_BLOCKS] * n_atl)), VPn(*[t.data_ptr() for t in outs])
        for _ in range(max(4, int(os.environ.get("BENCH_PMC_ROUNDS", "4")))):
            assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, n_atl, a_in, a_n, a_out, NBX, None, None, None) == 0
        torch.cuda.synchronize()
        ctx.close()
        return
    rounds = max(1, int(os.environ.get("BENCH_PMC_ROUNDS", "1")))
    streams = max(1, int(os.environ.get("BENCH_PMC_STREAMS", "1")))
    if os.environ.get("BENCH_PMC_ARRAY") == "1":
        # 32 atlases contiguous in memory = the 2^25-block array of config 5, ONE launch each, one at a time on this process's NULL stream (exclusive shape, tile
        # tickets): rocprofv3's per-kernel duration of these launches is a reading of the headline's workload that needs no pipeline and no period
        del outs
        big = [torch.cat(ins[:16] + ins[:16]).contiguous(), torch.cat(ins[8:24] + ins[8:24]).contiguous()]
        del ins
        bout = [torch.empty((32 * N_BLOCKS, 16), dtype=torch.uint8, device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        ctx.set_launch_policy("auto")
        lib = _lib.load()
        P2 = ctypes.c_void_p * 2
        ms = ctypes.c_float(0)
        st = lib.bu_time_uastc_launches(ctx.handle, _lib.BC7, P2(*[t.data_ptr() for t in big]), P2(*[t.data_ptr() for t in bout]), 2, 0, 32 * N_BLOCKS, NBX,
                                        int(os.environ.get("BENCH_PMC_ARRAY_LAUNCHES", "480")), None, None, ctypes.byref(ms))
        assert st == 0
        torch.cuda.synchronize()
        ctx.close()
        return
    if rounds == 1:
        for k in range(24):
            ctx.transcode_device(_lib.BC7, ins[k], N_BLOCKS, outs[k], blocks_per_row=NBX)
    else:  # the trace pass: launches enqueued back to back by the native loop, as in the timed region (a Python loop leaves ~15 us of
        #    idle GPU between launches, and a kernel that starts on an idle chip takes ~1 us longer)
        lib = _lib.load()
        PtrArr = ctypes.c_void_p * 24
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        st = lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, PtrArr(*[t.data_ptr() for t in ins]), PtrArr(*[t.data_ptr() for t in outs]), 24, 0,
                                                       N_BLOCKS, NBX, 0, 24 * rounds, 0, streams, None, ctypes.byref(ev), ctypes.byref(host), None, None)
        assert st == 0
    torch.cuda.synchronize()
    ctx.close()


def _run_child(cmd, env, timeout_s):
    """run a profiler child in its OWN process group and take the whole group down on a timeout: rocprofv3 starts the profiled
    python as its child, and a survivor of a killed launcher would keep launching kernels on GPU 0 beside the timed region"""
    import signal
    import subprocess

    p = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
    try:
        _, err = p.communicate(timeout=timeout_s)
        return p.returncode, err
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.wait()
        raise


def live_traffic(in_flight=1, policy="shared", timeout_s=90, batch=False):
    """HBM bytes per launch of the BC7 kernel MEASURED IN THIS RUN: two child rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE,
    then WRITE_SIZE: separate passes, nothing but --kernel-trace beside --pmc, the program directly after `--`) over
    `bench.py --pmc-child`, run before this process touches the GPU.  FETCH_SIZE / WRITE_SIZE count KiB; gfx950 reports half of
    a wide coalesced read stream (MI355X_MICROARCH.md, HBM section): bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024.
    A third child pass (--kernel-trace only, no counters, 400 rounds over the 24 atlases, `in_flight` launches in flight) gives what rocprofv3 says about the same
    launches unperturbed by counters: the per-kernel average duration, the launch-to-launch period of the back-to-back launches and
    how many launches started before their predecessor had ended.
    Returns (bytes, note, trace) or (None, reason, trace); trace is a dict or None."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH", None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled", None
    vals, t0 = {}, time.time()
    work = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_POLICY=policy, BENCH_PMC_STREAMS=str(in_flight))
    trace, note = None, None
    is_bc7 = lambda name: "bu_uastc_sorted_kernel<1," in name.replace("(int)", "")
    try:
        # ---- kernel trace only: durations and the launch-to-launch period as rocprofv3 sees them ----
        try:
            out = os.path.join(work, "trace")
            cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--pmc-child"]
            # (one enqueue thread per stream in this pass: under the profiler one enqueue costs 6-8 us of host time, more than the period, and a
            #  single thread would set the pace -- 7.5-8.1 us per completion against 6.3-6.4 with a thread per stream, profiles/r05_enqueue_threads_*)
            rc, err = _run_child(cmd, dict(env, BENCH_PMC_ROUNDS="400", BENCH_PMC_ENQ_THREADS="1" if in_flight > 1 else "0"), timeout_s)
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if is_bc7(row["Kernel_Name"]):
                        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row.get("Queue_Id", "")))
            rows.sort()
            rows = rows[len(rows) // 2:]  # the clocks take 25-40 ms of load to settle (9 600 launches: ~60 ms)
            if rc == 0 and len(rows) >= 64:
                dur = [e - s_ for s_, e, _ in rows]
                per = [rows[i + 1][0] - rows[i][0] for i in range(len(rows) - 1)]
                per = [x for x in per if x < 10 * (sum(dur) / len(dur))]  # (the host falls behind now and then: not a period)
                ends = sorted(e for _, e, _ in rows)
                eper = [ends[i + 1] - ends[i] for i in range(len(ends) - 1)]
                eper = [x for x in eper if x < 10 * (sum(dur) / len(dur))]
                wall = max(e for _, e, _ in rows) - rows[0][0]
                # steady stretches: runs of >= 64 consecutive completions without a pause (gap < 3 x median) -- the host threads of a profiled run stall
                # now and then for tens of microseconds, which is no property of the pipeline
                gaps = [ends[i + 1] - ends[i] for i in range(len(ends) - 1)]
                med = sorted(gaps)[len(gaps) // 2]
                runs, cur = [], []
                for x in gaps + [10 ** 12]:
                    if x < 3 * med:
                        cur.append(x)
                    else:
                        if len(cur) >= 64:
                            runs.append(cur)
                        cur = []
                steady_n = sum(len(r) for r in runs)
                steady_ns = sum(sum(r) for r in runs) / steady_n if steady_n else None
                trace = {"launches": len(rows), "launches_in_flight_requested": in_flight, "launch_policy": policy,
                         "kernel_avg_ns": round(sum(dur) / len(dur), 1), "kernel_min_ns": min(dur),
                         "period_avg_ns": round(sum(per) / max(1, len(per)), 1), "end_to_end_period_avg_ns": round(sum(eper) / max(1, len(eper)), 1),
                         "starts_before_previous_end": sum(1 for i in range(len(rows) - 1) if rows[i + 1][0] < rows[i][1]),
                         "avg_kernels_running": round(sum(dur) / max(1, wall), 2), "hardware_queues_used": len(set(q for _, _, q in rows)),
                         "end_to_end_period_median_ns": float(np.median(eper)) if eper else None, "enqueue_threads": in_flight if in_flight > 1 else 1,
                         "steady_period_ns": round(steady_ns, 1) if steady_ns else None, "steady_completions": steady_n,
                         "source": "child rocprofv3 --kernel-trace pass (no counters) over `bench.py --pmc-child`: 400 rounds x 24 cold atlases enqueued up front by "
                                   "bu_time_uastc_launches_streams_window on %d context stream(s), one enqueue thread per stream, the last half counted; kernel_avg = span of one dispatch "
                                   "(with several launches in flight a span is about that many periods); period = start-to-start of consecutive launches "
                                   "(end_to_end: end-to-end; steady_period: end-to-end over the stretches of >= 64 completions without a host-side pause, i.e. no gap above 3 x the median); avg_kernels_running = sum of spans / wall time" % in_flight}
        except Exception as e:  # the trace pass is a cross-check: without it the counter passes still run
            trace = {"error": "%s: %s" % (type(e).__name__, e)}
        # ---- kernel trace of ONE launch over 32 contiguous atlases (2^25 blocks), one at a time: the per-kernel duration rocprofv3 itself reports ----
        try:
            out = os.path.join(work, "array")
            cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--pmc-child"]
            rc, err = _run_child(cmd, dict(env, BENCH_PMC_ARRAY="1"), timeout_s)
            durs = []
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if is_bc7(row["Kernel_Name"]):
                        durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            durs = [d_ for _, d_ in sorted(durs) if d_ > 100000]  # (the 2^25-block launches; anything else of this kernel in the child is a 2^20-block launch)
            durs = durs[len(durs) // 2:]  # (these launches take ~100 ms to settle the clocks)
            if rc == 0 and len(durs) >= 16 and isinstance(trace, dict):
                trace["array32"] = {"launches": len(durs), "kernel_avg_ns": round(sum(durs) / len(durs), 1), "kernel_median_ns": float(np.median(durs)), "kernel_min_ns": min(durs),
                                    "source": "child rocprofv3 --kernel-trace pass over `bench.py --pmc-child` with BENCH_PMC_ARRAY=1: 480 launches of bu_uastc_transcode_device over "
                                              "2^25 contiguous blocks (32 atlases of 4096^2), one at a time on the NULL stream, two 1 GiB pairs rotated, the last half counted"}
        except Exception as e:
            if isinstance(trace, dict):
                trace["array32"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # ---- the headline's step (--method batch): ONE launch over ATLASES_PER_STEP atlases -- rocprofv3's own per-kernel duration, then its counters below ----
        is_multi = lambda name: "bu_uastc_multi_kernel<1," in name.replace("(int)", "")
        if batch:
            try:
                out = os.path.join(work, "batch_trace")
                cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--pmc-child"]
                rc, err = _run_child(cmd, dict(env, BENCH_PMC_BATCH="1", BENCH_PMC_ROUNDS="240"), timeout_s)
                durs = []
                for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if is_multi(row["Kernel_Name"]):
                            durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
                durs = [d_ for _, d_ in sorted(durs)]
                durs = durs[len(durs) // 2:]  # (the clocks settle on the load they are given: ~40 ms)
                if rc == 0 and len(durs) >= 16 and isinstance(trace, dict):
                    trace["batch"] = {"launches": len(durs), "atlases_per_launch": ATLASES_PER_STEP, "kernel_avg_ns": round(sum(durs) / len(durs), 1),
                                      "kernel_median_ns": float(np.median(durs)), "kernel_min_ns": min(durs), "kernel_max_ns": max(durs),
                                      "source": "child rocprofv3 --kernel-trace pass over `bench.py --pmc-child` with BENCH_PMC_BATCH=1: 240 launches of bu_uastc_transcode_batch_device over "
                                                "%d atlases of 4096^2 in separate allocations, back to back on the NULL stream, the last half counted" % ATLASES_PER_STEP}
            except Exception as e:
                if isinstance(trace, dict):
                    trace["batch"] = {"error": "%s: %s" % (type(e).__name__, e)}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--pmc-child"]
            rc, err = _run_child(cmd, dict(env, BENCH_PMC_BATCH="1") if batch else env, timeout_s)
            if rc != 0:
                return None, "rocprofv3 --pmc %s exited with %d: %s" % (counter, rc, err.decode(errors="replace")[-200:]), trace
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and (is_multi(row["Kernel_Name"]) if batch else is_bc7(row["Kernel_Name"])):
                        got.append(float(row["Counter_Value"]))
            if len(got) < (4 if batch else 8):
                return None, "rocprofv3 --pmc %s reported %d launches of the BC7 kernel" % (counter, len(got)), trace
            vals[counter] = sum(got) / len(got)
            vals[counter + "_n"] = len(got)
    except Exception as e:  # a timeout (the child's whole process group has been killed), a CSV layout this parser does not know: the committed passes stand in (pmc_traffic)
        return None, "%s: %s" % (type(e).__name__, e), trace
    finally:
        shutil.rmtree(work, ignore_errors=True)
    nbytes = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    return nbytes, ("measured in this run: child rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over `bench.py --pmc-child` "
                    "(%d + %d launches of %s; (2 x FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 read correction of "
                    "MI355X_MICROARCH.md; %.0f s with the trace passes)" % (vals["FETCH_SIZE_n"], vals["WRITE_SIZE_n"],
                                                                         ("the headline's launch over %d cold atlases" % ATLASES_PER_STEP) if batch else "this kernel on cold atlases", time.time() - t0)), trace


def cpu_baseline(golden, idx, budget_s=12.0):
    """oracle timed on the host: 1 thread and all hardware threads, bounded sample"""
    from oracle.pyoracle import Oracle

    orc = Oracle()
    cores = os.cpu_count() or 1
    sample = 1 << 18
    blocks = np.ascontiguousarray(golden["uastc"][idx[:sample]])
    out = np.empty((sample, 16), dtype=np.uint8)
    t0 = time.perf_counter()
    st = orc.lib.bu_oracle_transcode_mt(1, blocks.ctypes.data, blocks.size, out.ctypes.data, 1)
    t1 = time.perf_counter() - t0
    assert st == 0 and (out == golden["bc7"][idx[:sample]]).all()
    one = sample / t1 / 1e6
    # all threads on the full atlas, repeated until about budget_s seconds are spent
    blocks = np.ascontiguousarray(golden["uastc"][idx])
    out = np.empty((idx.size, 16), dtype=np.uint8)
    reps, spent = 0, 0.0
    while spent < budget_s and reps < 2000:
        t0 = time.perf_counter()
        st = orc.lib.bu_oracle_transcode_mt(1, blocks.ctypes.data, blocks.size, out.ctypes.data, cores)
        spent += time.perf_counter() - t0
        reps += 1
    assert st == 0
    allc = reps * idx.size / spent / 1e6
    return {
        "value": round(allc, 3), "unit": "Mblocks/s", "cores": cores, "kind": "port",
        "sample": "UASTC->BC7, %d x 4096x4096 A-gold atlas on %d threads (%.1f s); 1 thread on %d blocks = %.3f Mblocks/s" % (reps, cores, spent, sample, one),
        "one_thread_mblocks_s": round(one, 3),
    }


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks as a child torch.distributed.run (never re-exec: this
    process has not touched a GPU -- device_count() does not initialise one on this image -- and must not)."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    dry = os.environ.get("BENCH_SELF_LAUNCH_DRYRUN") == "1"  # tests: print the child command instead of running it
    if have < n and not dry:
        raise SystemExit("bench: --gpus %d but only %d GPU(s) are visible" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and HIP IPC handles need it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if dry:
        print(json.dumps({"self_launch": cmd, "HSA_ENABLE_IPC_MODE_LEGACY": env["HSA_ENABLE_IPC_MODE_LEGACY"], "GPU_MAX_HW_QUEUES": env.get("GPU_MAX_HW_QUEUES"),
                          "parent_initialised_cuda": bool(torch.cuda.is_initialized())}))
        raise SystemExit(0)
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=("atlas4096", "array512"), default="atlas4096")
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--method", choices=("batch", "pipeline"), default="batch",
                    help="what a step of --config atlas4096 is.  batch (default): ONE launch -- one call of bu_uastc_transcode_batch_device on the caller's stream -- over "
                         "%d atlases of 4096x4096 in separate allocations (a persistent grid over all their tiles, drawn by ticket): a kernel whose duration HIP events and "
                         "rocprofv3 agree on.  pipeline: one launch per atlas, --in-flight of them in flight on the context's streams (rounds 5's headline; always measured and "
                         "reported under `one_launch_per_atlas_in_flight`)" % ATLASES_PER_STEP)
    ap.add_argument("--nbuf", type=int, default=64, help="distinct atlas buffers rotated through (cold cache)")
    ap.add_argument("--prewarm-ms", type=float, default=40.0,
                    help="untimed launches BEFORE the W warm-up steps until this many milliseconds have passed: the GPU needs ~20 ms of "
                         "sustained work to reach its steady clocks (measured: 10.46 us per launch after --warmup 5, 9.8-9.9 after --warmup 2000 "
                         "or more); with it --warmup 5 --steps 20 and --warmup 64 --steps 512 agree.  0 disables")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="launches in flight: step i is issued on context stream i %% IN_FLIGHT (1..8).  One launch over a 4096^2 atlas waits ~3.4 us for "
                         "HBM with the ALUs idle and then computes with HBM idle, and launches on ONE stream never overlap; independent atlases on several "
                         "streams do.  1 = one launch at a time (the round 1-4 headline; always reported under extra.one_launch_at_a_time)")
    ap.add_argument("--enqueue-threads", type=int, choices=(0, 1), default=0,
                    help="1: the native timed loop enqueues from one host thread per stream (bu_time_set_enqueue_threads) instead of from one thread. "
                         "The unprofiled period is the same either way (the host is far ahead of the chip); under rocprofv3 --kernel-trace one enqueue "
                         "costs 6-8 us of host time, more than the period, and one thread then paces the run")
    ap.add_argument("--policy", choices=("shared", "exclusive"), default=None,
                    help="launch policy of the context (bu_context_set_launch_policy): shared = a launch keeps at most half of every CU so that launches "
                         "of different streams run side by side (default when --in-flight > 1), exclusive = a launch fills the chip by itself")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (lead launches, exactly K timed launches, tail) is run this many times back to back and the MEDIAN window is "
                         "reported; every window is listed in config.timed_region.windows_us_per_step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the verified, timed headline launches (no context rows, no CPU leg): the command profiled with rocprofv3, "
                         "so that its per-kernel average is the average of exactly the timed launches")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two child rocprofv3 counter passes that measure roofline.traffic (N = 1 only; ~1 min); the committed passes "
                         "under profiles/ are quoted instead")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # Launches on different streams overlap only when the streams sit on different HARDWARE queues, and the HIP runtime hands a process
    # GPU_MAX_HW_QUEUES (default 4) of them per priority level, two of which torch's NULL stream and the context's internal stream already
    # hold: with the default, 4 launches "in flight" run as 2 (profiles/r05_hip_hw_queue_knobs_vs_streams.txt).  The runtime reads the
    # variable when it initialises, so it is set here, before torch is imported and before any child is started; a value the caller set wins.
    # A process that also holds an RCCL communicator (N > 1) has more streams than eight queues: with 8, two of the context's four streams then
    # share a queue, run at half the others' pace and fall thousands of launches behind (profiles/r05_dist_branch_hw_queues.txt) -- 16 there.
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if multi else "8")
    if args.pmc_child:
        pmc_child()
        return
    if not 1 <= args.in_flight <= 8:
        raise SystemExit("bench: --in-flight must be 1..8")
    if args.policy is None:
        args.policy = "shared" if args.in_flight > 1 else "exclusive"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # does not return
    # roofline.traffic, measured: the counter passes are child processes, started before this process touches the GPU
    live = (None, "not requested", None)
    import torch  # (pays the cold first import of a fresh box here, before the children are clocked; importing initialises no GPU)
    assert not torch.cuda.is_initialized()
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.config == "atlas4096" and not args.headline_only and not args.no_live_traffic
            and os.environ.get("BENCH_FORCE_DIST") != "1"):
        live = live_traffic(args.in_flight, args.policy, batch=args.method == "batch")

    # stdout carries exactly ONE line, the JSON result of rank 0: libraries print banners there (RCCL's version block lands
    # in the C stdio buffer and is flushed at exit, i.e. BEHIND a Python print when stdout is a pipe), so file descriptor 1
    # is pointed at stderr for the whole run and the JSON line is written to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 exercises the N>1 code path (RCCL init, barriers, all_reduce, all-gather) with one rank
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if world > 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # BENCH_FORCE_DIST with one rank: this process is its own rendezvous.  A port somebody handed us (or the default) may have been taken in the meantime
            # (a test picks a free one, closes it, and starts us a second later): on "address already in use" take a fresh one and try again
            import socket
            for attempt in range(6):
                if attempt or "MASTER_PORT" not in os.environ:
                    s_ = socket.socket()
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
                    s_.close()
                try:
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                    break
                except Exception as e:  # noqa: BLE001 -- DistNetworkError is not importable on every torch
                    if attempt == 5 or "EADDRINUSE" not in repr(e) and "address already in use" not in repr(e):
                        raise
        # the communicator's first collectives (connection setup, tens of ms) happen here, far from the timed region
        dist.barrier()
        dist.barrier()
        torch.cuda.synchronize()

    from basisu_rs_amd import Context, _lib, synth

    env = Env()
    env.json_fd = json_fd
    env.args, env.torch, env.dist = args, torch, dist
    env.live_traffic = live
    env.rank, env.world, env.local_rank, env.use_dist = rank, world, local_rank, use_dist
    env.ctx = Context(local_rank)
    env.lib = _lib.load()
    env._lib, env.synth = _lib, synth
    env.golden = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
    env.dev = torch.device("cuda", local_rank)
    env.g_uastc = torch.from_numpy(env.golden["uastc"]).to(env.dev)
    env.g_bc7 = torch.from_numpy(env.golden["bc7"]).to(env.dev)
    env.stream = torch.cuda.current_stream()
    env.sp = ctypes.c_void_p(env.stream.cuda_stream)

    line = run_array512(env) if args.config == "array512" else run_atlas4096(env)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    env.ctx.close()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)


class Env:
    pass


class GatherWatchdog:
    """The gather rows are secondary: if a transport hangs (a collective one rank never entered, an IPC mapping the driver
    refuses), the headline line must still come out.  After `seconds` rank 0 writes the line it has, with the time-out
    recorded under "allgather", and every rank leaves the process without waiting for the stuck call."""

    def __init__(self, env, line, seconds=240):
        import threading

        self.env, self.line = env, line
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.t.start()

    def fire(self):
        if self.env.rank == 0:
            self.line["allgather"] = {"error": "timed out: no transport finished; the transcode numbers above are unaffected"}
            os.write(self.env.json_fd, (json.dumps(self.line) + "\n").encode())
        # a stuck collective is a failed run even though the headline line was written: the caller sees a non-zero code
        # (no in-process restart of anything that has touched the GPU; the launcher tears the other ranks down)
        os._exit(3)

    def cancel(self):
        self.t.cancel()


def check(env, st, what):
    if st != 0:
        detail = env.lib.bu_last_error(env.ctx.handle).decode() if st == env._lib.ERR_HIP else ""
        raise RuntimeError("%s: %s %s" % (what, env.lib.bu_status_string(st).decode(), detail))


class RawDeviceBuffer:
    """hipMalloc'ed memory (bu_device_alloc) seen by torch through __cuda_array_interface__: HIP IPC handles are taken on
    whole allocations, which a caching-allocator tensor is not"""

    def __init__(self, env, nbytes):
        self.env, self.nbytes = env, int(nbytes)
        p = ctypes.c_void_p(0)
        check(env, env.lib.bu_device_alloc(env.ctx.handle, self.nbytes, ctypes.byref(p)), "bu_device_alloc")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self):
        return self.env.torch.as_tensor(self, device=self.env.dev)

    def free(self):
        if self.ptr:
            self.env.lib.bu_device_free(self.env.ctx.handle, ctypes.c_void_p(self.ptr))
            self.ptr = 0


def busy_barrier(env, launch_async, n_launches):
    """The barrier + synchronize that opens the timed region, without letting the GPU idle through it.  An idle gap costs
    clocks (tools/exp/idle_gap.py: 1 ms of idleness makes the next 20 launches of the 2^25-block kernel 4 % slower, 5 ms
    14 %, 20 ms 25 %; the first collective of a fresh communicator takes longer than that).  So: untimed launches of the
    same kernel are queued on a side stream, the ranks meet in dist.barrier() while those run, and synchronize() then
    waits out whatever is left of them -- the K timed steps start on a GPU that has been busy all along.  Nothing of the
    side stream's work is in the timed region: it has completed when synchronize() returns."""
    torch, dist = env.torch, env.dist
    if not env.use_dist:
        torch.cuda.synchronize()
        return
    if getattr(env, "side_stream", None) is None:
        env.side_stream = torch.cuda.Stream(device=env.dev)
    ssp = ctypes.c_void_p(env.side_stream.cuda_stream)
    for i in range(n_launches):  # asynchronous: this only fills the queue (a few ms of GPU work)
        launch_async(i, ssp)
    dist.barrier()
    torch.cuda.synchronize()


def measure_gather(env, full_buf, shard_bytes, verify=None, reps=10):
    """Reassembly of the array on every rank, timed apart from the transcode: every rank's shard already sits at
    rank*shard_bytes of its own `full_buf` (RawDeviceBuffer of world*shard_bytes).  Two transports through the C ABI:
    RCCL in-place all-gather (bu_allgather_inplace) and direct peer pulls over HIP IPC (bu_allgather_peer).
    `verify(tensor)` -> bool checks the gathered buffer.  Returns the JSON fragment (rank 0's view, MAX over ranks)."""
    torch, dist, lib, ctx = env.torch, env.dist, env.lib, env.ctx
    world, rank = env.world, env.rank
    out = {"bytes_per_rank": shard_bytes, "bytes_total": shard_bytes * world}

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=env.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def all_ok(ok_here):
        """every rank learns whether every rank got through its local setup: a rank that failed must not leave the others
        waiting inside a collective it never enters"""
        f = torch.tensor([1 if ok_here else 0], dtype=torch.int32, device=env.dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item())

    def scrub():  # zero the other ranks' slots so that a transport that does nothing cannot pass the check
        t = full_buf.tensor()
        if rank > 0:
            t[: rank * shard_bytes].zero_()
        if rank < world - 1:
            t[(rank + 1) * shard_bytes:].zero_()
        torch.cuda.synchronize()

    # ---- RCCL, in place ----
    try:
        ident = torch.zeros(env._lib.COMM_ID_BYTES, dtype=torch.uint8)
        id_ok = True
        if rank == 0:
            buf = (ctypes.c_uint8 * env._lib.COMM_ID_BYTES)()
            id_ok = lib.bu_comm_unique_id(buf) == 0
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if not all_ok(id_ok):
            raise RuntimeError("bu_comm_unique_id failed (no RCCL library could be resolved)")
        ident = ident.to(env.dev)
        dist.broadcast(ident, 0)
        idb = (ctypes.c_uint8 * env._lib.COMM_ID_BYTES)(*ident.cpu().tolist())
        comm = ctypes.c_void_p(0)
        st_c = lib.bu_comm_create(ctx.handle, world, rank, idb, ctypes.byref(comm))  # collective: every rank calls it
        if not all_ok(st_c == 0):
            raise RuntimeError("bu_comm_create: " + lib.bu_status_string(st_c).decode())
        nr, me = ctypes.c_int(-1), ctypes.c_int(-1)
        if lib.bu_comm_query(comm, ctypes.byref(nr), ctypes.byref(me)) == 0:  # what RCCL itself says (ncclCommCount / ncclCommUserRank)
            out["rccl_comm"] = {"ranks": nr.value, "this_rank": me.value}
        scrub()
        s = timed(lambda: check(env, lib.bu_allgather_inplace(comm, ctypes.c_void_p(full_buf.ptr), shard_bytes, env.sp), "bu_allgather_inplace"))
        ok = bool(verify(full_buf.tensor())) if verify else None
        out["rccl_inplace"] = {"ms": round(s * 1e3, 3), "gb_s_per_rank_in": round(shard_bytes * (world - 1) / s / 1e9, 1) if s > 0 else None,
                               "verified": ok, "api": "bu_allgather_inplace (ncclAllGather, send = recv + rank*count)"}
        lib.bu_comm_destroy(comm)
    except Exception as e:  # a secondary row must never break the headline line
        out["rccl_inplace"] = {"error": repr(e)}
    # ---- direct peer pulls over HIP IPC ----
    try:
        hb = (ctypes.c_uint8 * env._lib.IPC_HANDLE_BYTES)()
        st_e = lib.bu_ipc_export(ctx.handle, ctypes.c_void_p(full_buf.ptr), hb)
        if not all_ok(st_e == 0):
            raise RuntimeError("bu_ipc_export: " + lib.bu_last_error(ctx.handle).decode())
        mine = torch.tensor(list(hb), dtype=torch.uint8, device=env.dev)
        allh = torch.empty(world * env._lib.IPC_HANDLE_BYTES, dtype=torch.uint8, device=env.dev)
        dist.all_gather_into_tensor(allh, mine)
        allh = allh.cpu().view(world, -1)
        peers = (ctypes.c_void_p * world)()
        open_ok = True
        for p_ in range(world):
            if p_ == rank:
                peers[p_] = full_buf.ptr
                continue
            hp = (ctypes.c_uint8 * env._lib.IPC_HANDLE_BYTES)(*allh[p_].tolist())
            pp = ctypes.c_void_p(0)
            open_ok = open_ok and lib.bu_ipc_open(ctx.handle, hp, ctypes.byref(pp)) == 0
            peers[p_] = pp.value
        if not all_ok(open_ok):
            raise RuntimeError("bu_ipc_open: " + lib.bu_last_error(ctx.handle).decode())
        scrub()
        dist.barrier()

        def pull():
            check(env, lib.bu_allgather_peer(ctx.handle, ctypes.c_void_p(full_buf.ptr), peers, world, rank, shard_bytes, env.sp), "bu_allgather_peer")

        s = timed(pull)
        ok = bool(verify(full_buf.tensor())) if verify else None
        out["peer_pull"] = {"ms": round(s * 1e3, 3), "gb_s_per_rank_in": round(shard_bytes * (world - 1) / s / 1e9, 1) if s > 0 else None,
                            "verified": ok, "api": "bu_allgather_peer (world-1 concurrent device-to-device copies over HIP IPC views)"}
        dist.barrier()
        for p_ in range(world):
            if p_ != rank and peers[p_]:
                lib.bu_ipc_close(ctx.handle, ctypes.c_void_p(peers[p_]))
        dist.barrier()
    except Exception as e:
        out["peer_pull"] = {"error": repr(e)}
    return out


class ProductWindow:
    """A timed window around the PRODUCT's own pipelined entry point.  A step = one contiguous run of `nb` blocks (an atlas, or the slices a rank owns of a
    texture array); steps rotate through `ins` / `outs`.  window(lead, K, tail) is three calls of bu_uastc_transcode_batch_in_flight -- `lead` steps, the K
    timed steps, `tail` steps, each call one slice per step -- with the context's timing-only events recorded on its streams between the calls
    (bu_time_mark_streams) and no host synchronisation in between: the window runs from the last lead launch's completion to the last timed launch's
    (bu_time_marks_elapsed), exactly as the helper-driven windows do, but every launch inside it was planned, shaped and enqueued by the call a user makes."""

    def __init__(self, env, target, ins, outs, nb, bpr, status, n_streams=4):
        self.env, self.lib, self.ctx, self.target, self.nb, self.bpr, self.n_streams = env, env.lib, env.ctx, target, nb, bpr, n_streams
        self.ins, self.outs, self.status = list(ins), list(outs), ctypes.c_void_p(status)
        self.rot = 0
        self.streams = None

    def _args(self, steps):
        n = len(self.ins)
        VP, SZ = ctypes.c_void_p * steps, ctypes.c_size_t * steps
        a = (steps, VP(*[self.ins[(self.rot + i) % n] for i in range(steps)]), SZ(*([self.nb] * steps)), VP(*[self.outs[(self.rot + i) % n] for i in range(steps)]))
        self.rot = (self.rot + steps) % n
        return a

    def _call(self, a):
        check(self.env, self.lib.bu_uastc_transcode_batch_in_flight(self.ctx.handle, self.target, a[0], a[1], a[2], a[3], self.bpr, None, self.status, self.n_streams),
              "bu_uastc_transcode_batch_in_flight")

    def window(self, lead, steps, tail=None):
        """(event ms, strict-bracket ms, host ms) of `steps` timed steps"""
        if tail is None:
            tail = 2 * self.n_streams if lead else 0
        parts = [self._args(k) if k else None for k in (lead, steps, tail)]  # (argument arrays first: nothing but the calls between the marks)
        h, S = self.ctx.handle, self.n_streams
        if parts[0]:
            self._call(parts[0])
        check(self.env, self.lib.bu_time_mark_streams(h, S, 0), "bu_time_mark_streams")
        self._call(parts[1])
        check(self.env, self.lib.bu_time_mark_streams(h, S, 1), "bu_time_mark_streams")
        if parts[2]:
            self._call(parts[2])
        ev, strict, host = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
        check(self.env, self.lib.bu_time_marks_elapsed(h, S, ctypes.byref(ev), ctypes.byref(strict), ctypes.byref(host)), "bu_time_marks_elapsed")
        a_, b_, n_ = (ctypes.c_float * 8)(), (ctypes.c_float * 8)(), ctypes.c_int(0)
        if self.lib.bu_time_last_window_streams(h, a_, b_, ctypes.byref(n_)) == 0:
            self.streams = {"start_us": [round(a_[i] * 1e3, 1) for i in range(n_.value)], "end_us": [round(b_[i] * 1e3, 1) for i in range(n_.value)]}
        self.ctx.synchronize()  # (the tail)
        return ev.value, strict.value, host.value


def batch_headline(env, line, in_ptrs, out_ptrs, nbuf, outs, idxs, status):
    """--method batch: the headline with one step = ONE launch over ATLASES_PER_STEP atlases.  One call of bu_uastc_transcode_batch_device on the caller's stream (torch's
    NULL stream) hands the library the atlases' separate allocations; they go out as ONE kernel -- a persistent grid that walks the 64 Ki tiles of all runs, every run tiled as
    64 x 16-block rectangles, the tiles drawn by ticket -- whose duration is the step.  A kernel's duration is what HIP events, the host clock and rocprofv3 all measure the
    same way: no streams, no period, no hardware queues.  W warm-up steps, barrier + synchronize, lead steps + EXACTLY K timed steps enqueued back to back with an event on
    either side, barrier + synchronize; --repeats windows, the median reported, MAX over ranks.  Rewrites the line's top-level figures; what the pipeline measured moves to
    line["one_launch_per_atlas_in_flight"]."""
    torch, dist, lib, ctx, args = env.torch, env.dist, env.lib, env.ctx, env.args
    world, rank, dev = env.world, env.rank, env.dev
    A = min(nbuf, ATLASES_PER_STEP)
    VPn, SZn = ctypes.c_void_p * A, ctypes.c_size_t * A
    sets = []
    for r_ in range(max(1, nbuf // A)):  # (nbuf = 64: one set -- every step touches all 2 GiB of the rotation, eight times the Infinity Cache)
        sets.append((VPn(*[in_ptrs[r_ * A + j] for j in range(A)]), SZn(*([N_BLOCKS] * A)), VPn(*[out_ptrs[r_ * A + j] for j in range(A)])))
    stp = ctypes.c_void_p(status.data_ptr())
    sp = env.sp
    ctx.set_launch_policy("auto")
    n_step = [0]

    def step(stream_ptr=None):
        a_ = sets[n_step[0] % len(sets)]
        n_step[0] += 1
        check(env, lib.bu_uastc_transcode_batch_device(ctx.handle, env._lib.BC7, A, a_[0], a_[1], a_[2], NBX, None, stp, stream_ptr or sp), "bu_uastc_transcode_batch_device")

    # correctness gate through this very path: every atlas of the step against the known answers
    for k in range(A):
        outs[k].zero_()
    ctx.status_word_reset(status)
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    for k in range(A):
        if not torch.equal(outs[k], env.g_bc7[idxs[k]]):
            raise SystemExit("bench: BC7 output of atlas %d (batch step) differs from the known-answer vectors" % k)
    t_pre, prewarm_steps = time.perf_counter(), 0
    while args.prewarm_ms > 0 and (time.perf_counter() - t_pre) * 1e3 < 4 * args.prewarm_ms:  # (launches of this size take ~100 ms to settle the clocks, as config 5's)
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        prewarm_steps += 8
    for _ in range(max(0, args.warmup)):
        step()
    torch.cuda.synchronize()
    busy_barrier(env, lambda i, ssp: step(ssp), 8)  # (N > 1: the ranks meet while ~3 ms of untimed steps keep the GPU busy)
    lead = 8
    wins = []
    for _ in range(max(1, args.repeats)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _l in range(lead):
            step()
        e0.record(env.stream)
        for _k in range(args.steps):
            step()
        e1.record(env.stream)
        late = e0.query()
        while not e0.query():
            pass
        t0 = time.perf_counter()
        while not e1.query():
            pass
        host_ms = (time.perf_counter() - t0) * 1e3
        ev_ms = e0.elapsed_time(e1)
        wins.append((max(ev_ms, 0.0 if late else host_ms), ev_ms, host_ms, bool(late)))
    torch.cuda.synchronize()
    if env.use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([w[0] / 1e3 for w in wins] + [w[1] / 1e3 for w in wins], dtype=torch.float64, device=dev)
    if env.use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    R_ = len(wins)
    dts, evs = [float(x) for x in t[:R_]], [float(x) for x in t[R_:]]
    m_ = sorted(range(R_), key=lambda i: dts[i])[R_ // 2]
    dt, ev = dts[m_], evs[m_]
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    K = args.steps
    step_s = ev / K
    achieved = BYTES_PER_BLOCK * A * N_BLOCKS / step_s / 1e9
    old_roof = line["roofline"]
    line["one_launch_per_atlas_in_flight"] = {
        "value": line["value"], "unit": line["unit"], "us_per_atlas": round(line["ms_per_step"] * 1e3, 3), "launches_in_flight": line["config"].get("launches_in_flight"),
        "launch_policy": line["config"].get("launch_policy"), "workload": line["config"]["workload"], "timed_region": line["config"].pop("timed_region", None),
        "roofline": {k_: v_ for k_, v_ in old_roof.items() if k_ not in ("one_launch_of_32_contiguous_atlases",)},
        "note": "rounds 5's headline, measured in this run by its own method: one launch per atlas, step i on context stream i % in_flight, K timed launches between per-stream events "
                "(window = last lead launch complete -> last timed launch complete); a pipeline's throughput is a PERIOD, which rocprofv3 does not report"}
    live = env.live_traffic
    tr = live[2] if len(live) > 2 and isinstance(live[2], dict) else None
    rp = (tr or {}).get("batch")
    line["value"] = round(world * K * A * N_BLOCKS / dt / 1e6, 1)
    line["ms_per_step"] = round(dt / K * 1e3, 6)
    line["config"]["method"] = "batch"
    line["config"]["workload"] = ("UASTC->BC7, %d atlases of 4096x4096 px (1 048 576 blocks each, separate allocations) per GPU per step, ONE launch per step: one call of "
                                  "bu_uastc_transcode_batch_device on the caller's stream = one persistent grid over the %d tiles of all runs (rectangular 64 x 16-block tiles, "
                                  "drawn by ticket); A-gold atlases (block i = reference known-answer block h(i) mod 608, uniform mix of the 19 modes), %d distinct atlases rotated "
                                  "(every step touches %.1f GiB: cold cache)" % (A, A * 1024, nbuf, A * N_BLOCKS * 32 / 2 ** 30))
    line["config"]["atlases_per_step"] = A
    line["config"]["blocks_per_step_per_gpu"] = A * N_BLOCKS
    line["config"]["us_per_atlas"] = round(dt / K / A * 1e6, 4)
    line["config"]["gb_s_in"] = round(line["value"] * 16 / 1e3, 1)
    line["config"]["prewarm"] = {"steps": prewarm_steps, "ms": 4 * args.prewarm_ms, "note": "untimed steps ahead of the W warm-up steps (clock ramp: 4 x --prewarm-ms for launches of this size); --prewarm-ms 0 disables"}
    line["config"]["timed_region"] = {
        "lead_steps": lead, "repeats": R_, "window_reported": "median", "windows_ms_per_step": [round(x / K * 1e3, 5) for x in dts], "event_ms": round(wins[m_][1], 6),
        "host_ms": round(wins[m_][2], 6), "host_started_late": wins[m_][3],
        "note": "barrier + synchronize, then -- enqueued back to back on one stream, no host synchronisation in between -- %d untimed lead steps, an event, EXACTLY K timed steps, an event; "
                "host clock from the first event seen complete to the second; value uses max(host, event) of the median window, MAX over ranks.  One step = one kernel: its "
                "duration is the step" % lead}
    line["roofline"] = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": live[0] if live[0] is not None else None,
        "traffic_source": live[1] if live[0] is not None else "not measured in this run (%s); committed passes of this launch: profiles/r06_*pmc*batch*" % live[1],
        "traffic_over_algorithmic": round(live[0] / (BYTES_PER_BLOCK * A * N_BLOCKS), 4) if live[0] else None,
        "kernel": "bu_uastc_multi_kernel<BC7, 256 threads x 4 blocks, five workgroups per CU, persistent, whole rectangular tiles> over %d runs" % A,
        "us_per_launch": round(step_s * 1e6, 3), "us_per_atlas": round(step_s / A * 1e6, 4), "bytes_per_launch": BYTES_PER_BLOCK * A * N_BLOCKS, "atlases_per_launch": A,
        "rocprofv3_this_run": rp,
        "us_per_launch_by_rocprofv3_kernel_avg": round(rp["kernel_avg_ns"] / 1e3, 3) if rp and rp.get("kernel_avg_ns") else None,
        "frac_by_rocprofv3_kernel_avg": round(BYTES_PER_BLOCK * A * N_BLOCKS / rp["kernel_avg_ns"] / HBM_PEAK_GBS, 4) if rp and rp.get("kernel_avg_ns") else None,
        "one_launch_of_32_contiguous_atlases": old_roof.get("one_launch_of_32_contiguous_atlases"),
        "note": "achieved = algorithmic bytes per launch (32 B x 2^20 blocks x %d atlases) / the launch's average duration = HIP events around the K timed launches / K; "
                "rocprofv3_this_run: the same launch's per-kernel duration from a child `rocprofv3 --kernel-trace` pass of this run (240 launches, the last half counted); "
                "traffic: child --pmc FETCH_SIZE / WRITE_SIZE passes over the same launch, (2 x FETCH + WRITE) KiB (the gfx950 read correction).  The three clocks agree because "
                "the step is ONE kernel; `one_launch_per_atlas_in_flight` has round 5's pipeline, measured in this run too" % A}
    return line


def run_array512(env):
    """BASELINE config 5: one texture array of 512 slices x (1024x1024 px = 65 536 blocks) -> BC7, strong scaling"""
    torch, dist, lib, ctx, args = env.torch, env.dist, env.lib, env.ctx, env.args
    world, rank, dev = env.world, env.rank, env.dev
    n_slices, bps = 512, 65536
    lo, hi = (n_slices * rank) // world, (n_slices * (rank + 1)) // world
    nb = (hi - lo) * bps
    per = -(-n_slices // world)  # slots of the gather buffer (ragged world sizes pad the last shards)
    shard_bytes = per * bps * 16
    # rotation: enough distinct shard inputs / full outputs that a step's traffic cannot sit in the 256 MiB Infinity Cache
    # (and enough that the steps in flight together -- --in-flight of them, plus the one being enqueued -- never share an output buffer)
    nrot = max(min(8, max(2, world)), min(8, args.in_flight + 2))

    def slice_idx(s):  # A-gold arrangement of slice s: any rank can regenerate any slice (verification after the gather)
        gen = torch.Generator(device=dev)
        gen.manual_seed(7000 + s)
        return torch.randint(0, 608, (bps,), device=dev, generator=gen)

    ins = []
    for r_ in range(nrot):
        if r_ == 0:
            ins.append(torch.cat([env.g_uastc[slice_idx(s)] for s in range(lo, hi)]).contiguous())
        else:  # other arrangements of the same blocks (a roll keeps it cheap; the mode mix is what matters)
            ins.append(torch.roll(ins[0], shifts=977 * r_, dims=0).contiguous())
    fulls = [RawDeviceBuffer(env, world * shard_bytes) for _ in range(nrot)]
    status = torch.empty(1, dtype=torch.int64, device=dev)
    ctx.status_word_reset(status)
    # P steps (arrays) are in flight together on P context streams (--in-flight, default 4): launches queued on one stream never overlap, and launches of
    # different streams fill each other's load phases and tails (2^25-block launches: one at a time 188.5 us fixed walk / 174 with tile tickets, four in
    # flight 167-171 us per array).  P = 1 (--in-flight 1): one launch over the range at a time.
    P = args.in_flight if args.in_flight > 1 else 1
    ctx.set_launch_policy("auto")  # (the pipelined entry point shapes its own launches; the one-launch fallback picks per call)
    lib.bu_time_set_enqueue_threads(ctx.handle, args.enqueue_threads)
    effective_streams, stream_mode = ctx.query_in_flight(P) if P > 1 else (1, "pool")
    queue_sharing = ctx.probe_streams(P) if P > 1 else 1  # measured NOW (the communicator of the N > 1 branch exists by now)
    rot = [0]  # in steps
    PtrArr1 = ctypes.c_void_p * nrot
    in_ptrs1 = PtrArr1(*[t.data_ptr() for t in ins])  # the range as ONE launch (the fallback when the streams are out of step)
    out_ptrs1 = PtrArr1(*[f.ptr + rank * shard_bytes for f in fulls])
    last_streams = [None]

    # The timed region issues the PRODUCT's pipelined entry point: one call of bu_uastc_transcode_batch_in_flight carries K steps, each step = this rank's
    # whole range as one slice (-> one launch under the shared policy, step i on context stream i % P: P steps in flight), with the context's timing-only
    # events between the lead, timed and tail calls (ProductWindow).  What a rank of a real job calls is what is timed, for every N.
    pw = ProductWindow(env, env._lib.BC7, [t.data_ptr() for t in ins], [f.ptr + rank * shard_bytes for f in fulls], nb, 256, status.data_ptr(), P) if P > 1 else None

    def window(lead_steps, steps, pieces=None):
        """`steps` passes over this rank's range behind `lead_steps` untimed ones; (event ms, host ms) of the timed part.  pieces = 1 (the fallback when the
        streams are out of step, and --in-flight 1): one bu_uastc_transcode_device per step on the context's stream 0, one launch at a time"""
        q = P if pieces is None else pieces
        if q > 1:
            pw.rot = rot[0] % nrot
            ev_, _strict, host_ = pw.window(lead_steps, steps)
            rot[0] = pw.rot
            last_streams[0] = pw.streams
            return ev_, host_
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        check(env, lib.bu_time_uastc_launches_streams_window(ctx.handle, env._lib.BC7, in_ptrs1, out_ptrs1, nrot, rot[0] % nrot, nb, 256, lead_steps, steps,
                                                             0, 1, ctypes.c_void_p(status.data_ptr()), ctypes.byref(ev), ctypes.byref(host), None, None),
              "bu_time_uastc_launches_streams_window")
        rot[0] = (rot[0] + lead_steps + steps) % nrot
        last_streams[0] = None
        return ev.value, host.value

    def run(launches):
        return window(0, launches)[0]

    def expect(s):
        return env.g_bc7[slice_idx(s)]

    # correctness gate: this rank's whole shard of rotation slot 0
    run(1)
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    mine = fulls[0].tensor()[rank * shard_bytes: rank * shard_bytes + nb * 16].view(-1, bps, 16)
    for k, s in enumerate(range(lo, hi)):
        if not torch.equal(mine[k], expect(s)):
            raise SystemExit("bench: BC7 output of slice %d differs from the known-answer vectors" % s)
    rot[0] = 1 % nrot
    prewarm_launches = 0
    t_pre = time.perf_counter()
    # clock ramp, untimed (see --prewarm-ms); the 1 GiB-per-launch kernel takes about four times as long to settle
    # (0.49 of the roofline after 25 ms, 0.56 after 100 ms and in every longer run)
    while args.prewarm_ms > 0 and (time.perf_counter() - t_pre) * 1e3 < 4 * args.prewarm_ms:
        run(8)
        prewarm_launches += 8
    if args.warmup > 0:
        run(args.warmup)
    torch.cuda.synchronize()

    def warm_async(i, ssp):
        k = (rot[0] + i) % nrot
        lib.bu_uastc_transcode_device(ctx.handle, env._lib.BC7, ctypes.c_void_p(in_ptrs1[k]), nb, ctypes.c_void_p(out_ptrs1[k]), 256, 0, None, ssp)

    busy_barrier(env, warm_async, max(2, 12 // world))  # ~3 ms of work
    # timed region as in run_atlas4096: lead untimed steps, K timed steps, (P > 1: one tail launch per stream), no host sync in between;
    # the window runs from the last lead launch's completion to the last timed launch's completion
    # (lead: 128 steps = 22 ms of the same work in front of the window.  The barrier above ends in a synchronize(): the chip idles for a moment, and
    #  after an idle moment the 2^23..2^25-block launches run 10-25 % slower for tens of milliseconds -- with 16 lead steps the one-rank run of this
    #  branch read 208-213 us per step where the same launches without torch.distributed read 175: profiles/r05_dist_branch_hw_queues.txt)
    lead = int(os.environ.get("BENCH_ARRAY_LEAD", "128"))
    ev_ms, host_ms = window(lead, args.steps)
    torch.cuda.synchronize()
    # the streams have to be in step for the window to hold `steps` completions of every stream (run_atlas4096 has the story): start events more than
    # 16 launch periods apart mean two streams share a hardware queue -- then the range is timed again as ONE launch per step on one stream
    timed_streams = last_streams[0]
    oos_, spread_us = streams_out_of_step(timed_streams, ev_ms * 1e3 / args.steps, P, queue_sharing, steps=args.steps)
    oos = torch.tensor([1.0 if oos_ else 0.0], dtype=torch.float64, device=dev)
    if env.use_dist:
        dist.all_reduce(oos, op=dist.ReduceOp.MAX)
    out_of_step = bool(oos.item() > 0)
    P_timed = P
    if out_of_step:
        ctx.set_launch_policy("auto")
        window(0, 2, pieces=1)
        ev_ms, host_ms = window(4, args.steps, pieces=1)
        torch.cuda.synchronize()
        P_timed = 1
    dt = max(host_ms, ev_ms) / 1e3
    if env.use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([dt, ev_ms / 1e3], dtype=torch.float64, device=dev)
    per_rank_kernel_us = [round(ev_ms / args.steps * 1e3, 3)]
    if env.use_dist:
        every = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, t[1:2].clone())
        per_rank_kernel_us = [round(float(x) / args.steps * 1e6, 3) for x in every.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, ev_max = float(t[0]), float(t[1])
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    total = n_slices * bps
    value = args.steps * total / dt_max / 1e6
    kern_s = ev_max / args.steps
    achieved = BYTES_PER_BLOCK * nb / kern_s / 1e9  # per GPU: this rank's shard (the even split makes all ranks alike)

    line = {
        "metric": "M 4x4 blocks/s UASTC->BC7 texture array 512 x (1024x1024)",
        "value": round(value, 1), "unit": "Mblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt_max / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "UASTC->BC7, texture array of 512 slices x 65 536 blocks (512 MiB in, 512 MiB out) per step; rank r owns "
                               "slices [r*512/N, (r+1)*512/N); the timed region is ONE call of the product's pipelined entry point "
                               "(bu_uastc_transcode_batch_in_flight) carrying the K steps, each step = the rank's contiguous range of %d blocks as one slice = one launch, "
                               "step i on context stream i %% %d (shared launch shapes; lead and tail steps through the same call in front of and behind it); A-gold blocks; "
                               "%d rotated input shards / full output buffers per rank" % (nb, P, nrot),
                   "launches_in_flight": P_timed, "timed_through": "bu_uastc_transcode_batch_in_flight" if P_timed > 1 else "bu_uastc_transcode_device (one launch at a time, context stream 0)",
                   "effective_streams": effective_streams, "stream_mode": stream_mode,
                   "timed_region": {"streams": timed_streams, "start_event_spread_us": round(spread_us, 1), "streams_in_step": not out_of_step,
                                    "streams_on_one_hardware_queue_max": queue_sharing,
                                    "note": None if not out_of_step else (
                                        "the %d streams were NOT in step (two share a hardware queue: GPU_MAX_HW_QUEUES=%s, more streams in this process than queues); "
                                        "the figures are those of ONE launch per step over the rank's range on one stream" % (P, os.environ.get("GPU_MAX_HW_QUEUES")))},
                   "blocks_per_step": total, "slices_per_gpu": hi - lo, "gb_s_in": round(value * 16 / 1e3, 1),
                   "prewarm": {"launches": prewarm_launches, "ms": 4 * args.prewarm_ms,
                               "note": "untimed launches ahead of the W warm-up steps (clock ramp); --prewarm-ms 0 disables"}},
        "transcode_only": {"mblocks_s": round(total / kern_s / 1e6, 1), "us_per_step_kernel_max_over_ranks": round(kern_s * 1e6, 3),
                           "us_per_step_kernel_per_rank": per_rank_kernel_us,
                           "note": "all blocks / slowest rank's time per step (hipEvents on the launch streams: last lead launch complete -> last timed launch complete)"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "bu_uastc_sorted_kernel<BC7>", "us_per_launch": round(kern_s * 1e6, 3),
                     "bytes_per_launch": BYTES_PER_BLOCK * nb, "note": "per GPU, on its shard"},
    }
    gather = None
    if env.use_dist:
        # shard of rotation slot 0 is verified above; re-run it so that slot 0 holds this rank's result, then gather slot 0
        rot[0] = 0
        run(1)
        torch.cuda.synchronize()
        dist.barrier()

        def verify(full_t):  # EVERY slice of every rank's range (outside the timed regions)
            ok = True
            v = full_t.view(world, per * bps, 16)
            for r_ in range(world):
                a, b = (n_slices * r_) // world, (n_slices * (r_ + 1)) // world
                for s in range(a, b):
                    ok = ok and bool(torch.equal(v[r_][(s - a) * bps: (s - a + 1) * bps], expect(s)))
            return ok

        dog = GatherWatchdog(env, line)
        gather = measure_gather(env, fulls[0], shard_bytes, verify)
        dog.cancel()
    for f in fulls:
        f.free()
    if gather:
        line["allgather"] = gather
        best = min([g["ms"] for g in gather.values() if isinstance(g, dict) and "ms" in g] or [None]) if gather else None
        if best is not None:
            line["total_ms_transcode_plus_gather"] = round(kern_s * 1e3 + best, 3)
    return line


def run_atlas4096(env):
    torch, dist, lib, ctx, args = env.torch, env.dist, env.lib, env.ctx, env.args
    world, rank, dev, use_dist, local_rank = env.world, env.rank, env.dev, env.use_dist, env.local_rank
    _lib, synth, golden, g_uastc, g_bc7, stream, sp = env._lib, env.synth, env.golden, env.g_uastc, env.g_bc7, env.stream, env.sp

    # NBUF distinct A-gold atlases per rank: block i of atlas k = G[h(i; seed_k) mod 608]
    nbuf = max(1, args.nbuf)
    idx0 = synth.gold_indices(N_BLOCKS, seed=synth.GOLD_SEED + 7919 * rank)
    ins, outs, idxs = [], [], []
    for k in range(nbuf):
        if k == 0:
            idx = torch.from_numpy(idx0).to(dev)
        else:  # a different pseudo-random arrangement per buffer, generated on the GPU
            gen = torch.Generator(device=dev)
            gen.manual_seed(1000 * rank + k)
            idx = torch.randint(0, 608, (N_BLOCKS,), device=dev, generator=gen)
        idxs.append(idx)
        ins.append(g_uastc[idx].contiguous())
        outs.append(torch.empty((N_BLOCKS, 16), dtype=torch.uint8, device=dev))
    status = torch.empty(1, dtype=torch.int64, device=dev)
    PtrArr = ctypes.c_void_p * nbuf
    in_ptrs = PtrArr(*[t.data_ptr() for t in ins])
    out_ptrs = PtrArr(*[t.data_ptr() for t in outs])
    # Rotation position carried across EVERY call: a timed launch never re-touches what the warm-up (or any earlier
    # row) just touched, whatever --warmup is.
    rot = [0]

    def run(launches, target=_lib.BC7, inp=in_ptrs, outp=out_ptrs, nb=nbuf):
        ms = ctypes.c_float(0)
        first = rot[0] % nb
        st = lib.bu_time_uastc_launches(ctx.handle, target, inp, outp, nb, first, N_BLOCKS, NBX, launches, ctypes.c_void_p(status.data_ptr()), sp, ctypes.byref(ms))
        if st != 0:
            raise RuntimeError("bu_time_uastc_launches: " + lib.bu_status_string(st).decode())
        rot[0] += launches
        return ms.value

    def row(launches, target=_lib.BC7, inp=in_ptrs, outp=out_ptrs, nb=nbuf, lead=64):
        """seconds per launch of a context row, measured like the headline: `lead` untimed launches and the timed ones enqueued
        back to back, events around the timed part (a plain window behind a synchronize carries ~35 us of pipeline refill)"""
        ev, host, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
        st = lib.bu_time_uastc_launches_window(ctx.handle, target, inp, outp, nb, rot[0] % nb, N_BLOCKS, NBX, lead, launches,
                                               ctypes.c_void_p(status.data_ptr()), sp, ctypes.byref(ev), ctypes.byref(host), ctypes.byref(late))
        if st != 0:
            raise RuntimeError("bu_time_uastc_launches_window: " + lib.bu_status_string(st).decode())
        rot[0] += lead + launches
        return ev.value / 1e3 / launches

    fill_drain = [0.0]  # out_fill_drain_ms of the last run_window call

    def window_streams():
        """per-stream start / end event of the last streams window, us from the head of that call (bu_time_last_window_streams)"""
        a_, b_, n_ = (ctypes.c_float * 8)(), (ctypes.c_float * 8)(), ctypes.c_int(0)
        if lib.bu_time_last_window_streams(ctx.handle, a_, b_, ctypes.byref(n_)) != 0:
            return None
        out_ = {"start_us": [round(a_[i] * 1e3, 1) for i in range(n_.value)], "end_us": [round(b_[i] * 1e3, 1) for i in range(n_.value)]}
        ms_, k_ = ctypes.c_float(0), ctypes.c_int(0)
        if lib.bu_time_last_window_enqueue(ctx.handle, ctypes.byref(ms_), ctypes.byref(k_)) == 0 and k_.value:
            out_["host_enqueue_us_per_launch"] = round(ms_.value * 1e3 / k_.value, 3)  # (an upper bound: waits for queue space are inside; at the period or above, the host may be setting the pace)
        return out_

    def run_window(lead, launches, in_flight=None, tail=None):
        """the timed region: step i on context stream i % in_flight; `lead` untimed launches, start event per stream, `launches` timed ones, end event
        per stream, `tail` untimed launches (default: one per stream when the pipeline was filled by lead launches)"""
        nfl = in_flight or args.in_flight
        if tail is None:
            tail = 2 * nfl if (nfl > 1 and lead >= nfl) else 0  # (two per stream: a stream a few periods behind still has partners until its end event)
        ev, host, fd, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
        st = lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, in_ptrs, out_ptrs, nbuf, rot[0] % nbuf, N_BLOCKS, NBX, lead, launches, tail,
                                                       nfl, ctypes.c_void_p(status.data_ptr()), ctypes.byref(ev), ctypes.byref(host), ctypes.byref(fd), ctypes.byref(late))
        if st != 0:
            raise RuntimeError("bu_time_uastc_launches_streams_window: " + lib.bu_status_string(st).decode())
        rot[0] += lead + launches + tail
        fill_drain[0] = fd.value
        if args.enqueue_threads and nfl > 1:
            # one enqueue thread per stream: the streams are not fed in step, so "latest start event to latest end event" no longer brackets exactly
            # `launches` completions (under rocprofv3 one stream runs hundreds of launches ahead of another).  The strict bracket -- earliest start to
            # latest end, every timed launch inside from first to last instruction -- holds whatever the order: the figure of this mode, K + S - 1 periods
            return fd.value, fd.value, late.value
        return ev.value, host.value, late.value

    def srow(launches, in_flight, shared, target=_lib.BC7, inp=in_ptrs, outp=out_ptrs, nb=nbuf, lead=64):
        """seconds per atlas of a context row with `in_flight` launches in flight under the given launch policy (the headline's method)"""
        ctx.set_launch_policy(shared)
        try:
            ev, host = ctypes.c_float(0), ctypes.c_float(0)
            tail = in_flight if in_flight > 1 else 0
            st = lib.bu_time_uastc_launches_streams_window(ctx.handle, target, inp, outp, nb, rot[0] % nb, N_BLOCKS, NBX, lead, launches, tail, in_flight,
                                                           ctypes.c_void_p(status.data_ptr()), ctypes.byref(ev), ctypes.byref(host), None, None)
            if st != 0:
                raise RuntimeError("bu_time_uastc_launches_streams_window: " + lib.bu_status_string(st).decode())
            rot[0] += lead + launches + tail
            return max(ev.value, host.value) / 1e3 / launches
        finally:
            ctx.set_launch_policy(policy_now[0])

    def sramp(in_flight, shared, **kw):
        """untimed multi-stream windows of the row's own kernel for --prewarm-ms: the clocks settle on the load they are given (25-40 ms)"""
        t0 = time.perf_counter()
        srow(64, in_flight, shared, **kw)
        while args.prewarm_ms > 0 and (time.perf_counter() - t0) * 1e3 < args.prewarm_ms:
            srow(256, in_flight, shared, lead=0, **kw)

    def ramp(**kw):
        """untimed launches of the row's own kernel for --prewarm-ms (at least 64): the rows after the host-side phases
        start from idle clocks otherwise"""
        run(64, **kw)
        t0 = time.perf_counter()
        while args.prewarm_ms > 0 and (time.perf_counter() - t0) * 1e3 < args.prewarm_ms:
            run(128, **kw)

    # ---- correctness gate before any timing: full-size, self-verifying, through the headline's own path (policy, streams) ----
    policy_now = [args.policy == "shared"]  # the context's launch policy outside srow()
    ctx.set_launch_policy(policy_now[0])
    lib.bu_time_set_enqueue_threads(ctx.handle, args.enqueue_threads)
    # does every stream of the timed region have a hardware queue of its own in this process (the communicator of the N > 1 branch exists by now)?
    queue_sharing = ctx.probe_streams(args.in_flight) if args.in_flight > 1 else 1
    effective_streams, stream_mode = ctx.query_in_flight(args.in_flight) if args.in_flight > 1 else (1, "pool")  # (what the library found when it created its streams)
    ctx.status_word_reset(status)
    torch.cuda.synchronize()  # (the context's streams do not wait for torch's)
    run_window(0, nbuf)       # one launch per atlas, round-robin over the streams
    rot[0] = 0
    torch.cuda.synchronize()
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
    for k in range(nbuf):
        if not torch.equal(outs[k], g_bc7[idxs[k]]):
            raise SystemExit("bench: BC7 output of atlas %d differs from the known-answer vectors" % k)

    # ---- clock ramp (untimed, see --prewarm-ms), W warm-up steps, then EXACTLY K timed steps between barrier + synchronize ----
    prewarm_launches = 0
    t_pre = time.perf_counter()
    while args.prewarm_ms > 0 and (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        run_window(0, 256)
        prewarm_launches += 256
    if args.warmup > 0:
        run_window(0, args.warmup)
    torch.cuda.synchronize()

    def warm_async(i, ssp):  # an untimed launch on the side stream, rotating like the timed ones
        k = (rot[0] + i) % nbuf
        lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(in_ptrs[k]), N_BLOCKS, ctypes.c_void_p(out_ptrs[k]), NBX, 0, None, ssp)

    # The contract's barrier + synchronize (the ranks meet while a few ms of untimed launches keep the GPU busy), then the
    # timed region of bu_time_uastc_launches_window: F untimed launches of the same kernel, event 0, EXACTLY K timed launches,
    # event 1 -- enqueued back to back with no host synchronisation in between, so the K steps run on a GPU that never went
    # idle.  The host clock starts when event 0 is first seen complete and stops when event 1 is: `value` brackets exactly
    # the K steps.  (A synchronize() directly in front of K = 20 launches put ~35 us of pipeline refill into a 200 us
    # window: round-2 driver run 11.47 us per launch against 9.77 us in the long pre-warm loop of the same process.)
    busy_barrier(env, warm_async, 400)  # ~4 ms of work
    # The timed region is run `--repeats` times back to back (default 5: lead launches, EXACTLY K timed launches, tail launches each) and the MEDIAN
    # window is the one reported -- every window's figure is in config.timed_region.windows_us_per_step.  One window of K = 20 launches lasts 0.12 ms: a
    # single clock transition or a stall of 30 us inside it moves the per-step figure by a quarter (seen once in ten runs: 7.2 where the others read
    # 5.6-6.0, with the strict bracket of the same run at 6.6), and a median of five does not care.  EVERY window sits behind `lead` = max(2048, 2K)
    # untimed launches (12 ms of the same work): the few tens of microseconds the chip idles between two windows cost it more than a millisecond of
    # slower launches (windows behind only 64 lead launches read 5.9-8.9 us per step where the first one read 5.8).
    lead = max(2048, 2 * args.steps)
    wins = []
    for r_ in range(max(1, args.repeats)):
        ev_ms_, host_ms_, late_ = run_window(lead, args.steps)
        wins.append((max(host_ms_, ev_ms_), ev_ms_, host_ms_, late_, window_streams()))
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # per window: this rank's K steps are never less than what the GPU's own events say; the MAX over ranks is the job's time for that window
    t = torch.tensor([w[0] / 1e3 for w in wins] + [w[1] / 1e3 for w in wins], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    R_ = len(wins)
    dts, evs = [float(x) for x in t[:R_]], [float(x) for x in t[R_:]]
    m_ = sorted(range(R_), key=lambda i: dts[i])[R_ // 2]  # the median window (of the job, i.e. of the slowest rank per window)
    dt_max, ev_max = dts[m_], evs[m_]
    ev_ms, host_ms, late = wins[m_][1], wins[m_][2], wins[m_][3]
    ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)

    total_blocks = world * args.steps * N_BLOCKS
    value = total_blocks / dt_max / 1e6
    period_s = ev_max / args.steps  # launch-to-launch period of the timed region (slowest rank), HIP events
    achieved = BYTES_PER_BLOCK * N_BLOCKS / period_s / 1e9
    # Are the streams in step?  The window counts K completions only if every stream is at about the same launch when it opens: the start events of a
    # pipeline in step lie in_flight - 1 periods apart.  Streams that share a hardware queue (more streams in the process than GPU_MAX_HW_QUEUES, e.g.
    # beside an RCCL communicator) run at half the others' pace and end up milliseconds behind -- the window then holds only the laggards' launches and
    # reads too short (profiles/r05_dist_branch_hw_queues.txt: spreads of 2 500-5 700 us; streams in step: 15-35 us).  Beyond 16 periods of spread the figure is not used: the one-launch-at-a-time
    # measurement below (one stream, no assumption about anybody's pace) becomes the headline, with a warning.
    sk_ = wins[m_][4]
    # a third reading of the same window, stream by stream: stream s completed its K_s timed launches between ITS start and end event, i.e. at a rate
    # of K_s / (end_s - start_s); the streams were active side by side, so the rates add up
    by_rates_us = None
    if sk_ and args.in_flight > 1:
        ks_ = [len(range(lead + j, lead + args.steps, args.in_flight)) for j in range(args.in_flight)]  # timed launches of the stream of launch lead + j
        ks_ = {(lead + j) % args.in_flight: ks_[j] for j in range(args.in_flight)}
        rate_ = sum(ks_[s_] / (sk_["end_us"][s_] - sk_["start_us"][s_]) for s_ in range(args.in_flight) if sk_["start_us"][s_] >= 0 and sk_["end_us"][s_] > sk_["start_us"][s_])
        by_rates_us = round(1.0 / rate_, 3) if rate_ > 0 else None
    oos_, start_spread_us = streams_out_of_step(sk_, period_s * 1e6, args.in_flight, queue_sharing, steps=args.steps)
    out_of_step = torch.tensor([1.0 if oos_ else 0.0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(out_of_step, op=dist.ReduceOp.MAX)
    out_of_step = bool(out_of_step.item() > 0) and not args.enqueue_threads  # (with --enqueue-threads the figure is the strict bracket already)

    # the strict bracket: the same K steps with NOTHING behind them, from the first instruction of the first timed launch to the last instruction
    # of the last one (the pipeline's fill is credited to nobody and its drain -- the last launches running with fewer and fewer partners -- is inside)
    stricts = []
    for _ in range(max(1, args.repeats)):  # (median of the same number of passes as the headline: one 0.12 ms window is at the mercy of a single stall)
        run_window(lead, args.steps, tail=0)
        stricts.append((fill_drain[0] / 1e3 / args.steps, window_streams()))
    strict_s, strict_streams = sorted(stricts, key=lambda x: x[0])[len(stricts) // 2]
    # ---- the round 1-4 headline, kept as a row: ONE launch at a time (exclusive policy, one stream), same window method ----
    policy_now[0] = False
    ctx.set_launch_policy(False)
    run_window(0, 256, in_flight=1)
    one_ev, one_host, _ = run_window(lead, args.steps, in_flight=1)
    one_s = max(one_ev, one_host) / 1e3 / args.steps
    if out_of_step:
        o_ = torch.tensor([one_s], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(o_, op=dist.ReduceOp.MAX)
        dt_max = ev_max = float(o_.item()) * args.steps
        value = total_blocks / dt_max / 1e6
        period_s = ev_max / args.steps
        achieved = BYTES_PER_BLOCK * N_BLOCKS / period_s / 1e9
    # per-launch distribution: one event between every two launches of another K steps
    each = (ctypes.c_float * args.steps)()
    st = lib.bu_time_uastc_launches_each(ctx.handle, _lib.BC7, in_ptrs, out_ptrs, nbuf, rot[0] % nbuf, N_BLOCKS, NBX, args.steps,
                                         ctypes.c_void_p(status.data_ptr()), sp, each)
    rot[0] += args.steps
    per_launch = None
    if st == 0:
        v = np.sort(np.array(list(each), dtype=np.float64))
        per_launch = {"median_us": round(float(np.median(v)), 3), "min_us": round(float(v[0]), 3), "p90_us": round(float(v[int(0.9 * (len(v) - 1))]), 3),
                      "max_us": round(float(v[-1]), 3),
                      "note": "event-to-event time of each launch in a separate pass of K steps, one launch at a time (the event packets between launches are included)"}
    one_row = {"us_per_launch": round(one_s * 1e6, 3), "mblocks_s": round(N_BLOCKS / one_s / 1e6, 1), "gb_s": round(BYTES_PER_BLOCK * N_BLOCKS / one_s / 1e9, 1),
               "frac_of_hbm_peak": round(BYTES_PER_BLOCK * N_BLOCKS / one_s / 1e9 / HBM_PEAK_GBS, 4), "launch_policy": "exclusive", "per_launch": per_launch,
               "note": "the headline of rounds 1-4: K launches back to back on ONE stream, exclusive launch policy (a launch fills the chip by itself); "
                       "consecutive launches of one stream do not overlap, so this is also one launch's own duration"}

    extra = {}
    if rank == 0 and not args.headline_only:
        # context rows (not the headline): copy ceiling of the same shape, hot-cache and coherent atlases, RGBA32
        ms = ctypes.c_float(0)
        lib.bu_time_copy_launches(ctx.handle, in_ptrs, out_ptrs, nbuf, rot[0] % nbuf, N_BLOCKS, 32, sp, ctypes.byref(ms))
        rot[0] += 32
        copy_n = max(args.steps, 256)  # (a long batch: the copy helper has no lead-in launches)
        lib.bu_time_copy_launches(ctx.handle, in_ptrs, out_ptrs, nbuf, rot[0] % nbuf, N_BLOCKS, copy_n, sp, ctypes.byref(ms))
        rot[0] += copy_n
        copy_s = ms.value / 1e3 / copy_n
        extra["copy_ceiling"] = {"gb_s": round(BYTES_PER_BLOCK * N_BLOCKS / copy_s / 1e9, 1), "us_per_launch": round(copy_s * 1e6, 3),
                                 "note": "uint4->uint4 copy kernel in its fastest known one-launch shape at this size (256 threads x 4 elements, nontemporal; from 2^22 elements on 1024 x 1, which copies 512 MiB at 0.81 of the roofline: profiles/r06_copy_ceiling_by_size.txt), same cold-cache rotation, one "
                                         "launch at a time.  NOT a ceiling for launches in flight: this kernel gains nothing from company (in_flight rows), the transcoder's "
                                         "persistent workgroups stream faster (single-mode BC7 atlases, four in flight: 5.2 us = 6.4 TB/s)"}
        try:
            COPY = 100  # BU_TIME_COPY_CEILING
            for nfl in (2, 4):
                srow(64, nfl, False, target=COPY)
                extra["copy_ceiling"]["in_flight_%d_us_per_launch" % nfl] = round(srow(256, nfl, False, target=COPY) * 1e6, 3)
            torch.cuda.synchronize()
            extra["copy_ceiling"]["verified"] = bool(torch.equal(outs[3], ins[3]))
        except Exception as e:
            extra["copy_ceiling"]["in_flight_error"] = repr(e)
        one_in, one_out = (ctypes.c_void_p * 1)(ins[0].data_ptr()), (ctypes.c_void_p * 1)(outs[0].data_ptr())
        hot_s = row(max(args.steps, 64), inp=one_in, outp=one_out, nb=1)
        extra["hot_cache"] = {"gb_s": round(BYTES_PER_BLOCK * N_BLOCKS / hot_s / 1e9, 1), "us_per_launch": round(hot_s * 1e6, 3),
                              "note": "same atlas every launch (32 MiB working set sits in the 256 MiB Infinity Cache) -- NOT the headline"}
        # launches in flight x launch policy (the headline's cell among its neighbours; same window method, same rotation, 256 timed launches)
        try:
            pol_arg = {"exclusive": False, "shared": True, "auto": "auto"}
            mat = {pol: {} for pol in pol_arg}
            auto_picks = {}
            for nfl in (1, 2, 3, 4):  # column by column, the three policies interleaved and the better of two windows kept: a drifting clock hits a column's cells alike
                for rep in range(2):
                    for pol in ("exclusive", "shared", "auto"):
                        c0_ = (ctypes.c_ulonglong * 3)()
                        lib.bu_time_auto_policy_counts(ctx.handle, c0_)
                        srow(64, nfl, pol_arg[pol])
                        v_ = round(srow(256, nfl, pol_arg[pol]) * 1e6, 3)
                        mat[pol][str(nfl)] = v_ if rep == 0 else min(mat[pol][str(nfl)], v_)
                        if pol == "auto":
                            c1_ = (ctypes.c_ulonglong * 3)()
                            lib.bu_time_auto_policy_counts(ctx.handle, c1_)
                            auto_picks[str(nfl)] = [int(c1_[i] - c0_[i]) for i in range(3)]
            torch.cuda.synchronize()
            extra["launches_in_flight_matrix"] = {"us_per_atlas": mat, "verified": all(bool(torch.equal(outs[k], g_bc7[idxs[k]])) for k in range(nbuf)),
                                                  "auto_picks_exclusive_onetile_shared": auto_picks,
                                                  "auto_matches_the_better_row": all(mat["auto"][c] <= 1.05 * min(mat["exclusive"][c], mat["shared"][c]) for c in ("1", "2", "3", "4")),  # (cells repeat within 3 %)
                                                  "note": "UASTC->BC7, 2^20 blocks per launch, step i on context stream i %% n; rows = launch policy (auto = BU_LAUNCH_AUTO, the default: chosen "
                                                          "per call -- exclusive for a launch that is alone, shared once another of the context's streams has work in flight), columns = launches in flight; "
                                                          "every one of the %d rotated outputs compared with the known answers afterwards" % nbuf}
            # cross-check of the overlap claim through the product's other route to it: TWO atlases in ONE launch -- two slices that are contiguous in
            # memory handed to bu_uastc_transcode_batch_device, which merges them into one plain launch of 2^21 blocks (exclusive policy, one stream,
            # one launch at a time): inside that launch the second atlas' loads overlap the first one's compute
            n_pairs = 12
            pin = [torch.cat([ins[2 * k], ins[2 * k + 1]]).contiguous() for k in range(n_pairs)]
            pout = [torch.empty((2 * N_BLOCKS, 16), dtype=torch.uint8, device=dev) for _ in range(n_pairs)]
            VP2, SZ2 = ctypes.c_void_p * 2, ctypes.c_size_t * 2
            pargs = [(VP2(pin[k].data_ptr(), pin[k].data_ptr() + N_BLOCKS * 16), SZ2(N_BLOCKS, N_BLOCKS), VP2(pout[k].data_ptr(), pout[k].data_ptr() + N_BLOCKS * 16))
                     for k in range(n_pairs)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def two(k):
                a_, n_, o_ = pargs[k % n_pairs]
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, 2, a_, n_, o_, NBX, None, None, sp) == 0

            for k in range(48):
                two(k)
            e0.record(stream)
            for k in range(96):
                two(k)
            e1.record(stream)
            torch.cuda.synchronize()
            two_s = e0.elapsed_time(e1) / 1e3 / 192
            extra["atlases_2_one_launch"] = {"us_per_atlas": round(two_s * 1e6, 3), "mblocks_s": round(N_BLOCKS / two_s / 1e6, 1),
                                             "verified": bool(torch.equal(pout[1][:N_BLOCKS], g_bc7[idxs[2]])) and bool(torch.equal(pout[1][N_BLOCKS:], g_bc7[idxs[3]])),
                                             "note": "two contiguous atlases per call of bu_uastc_transcode_batch_device = one 2^21-block launch, calls back to back on one "
                                                     "stream, exclusive policy, cold rotation over %d pairs" % n_pairs}
            del pin, pout
            # the reference's loop over slices (basis.rs:246-257) on EIGHT atlases in separate allocations through ONE call of the batch entry point:
            # one launch of a persistent grid that walks the tiles of all eight runs with the next tile's loads in flight (run table in the kernel arguments)
            VP8, SZ8 = ctypes.c_void_p * 8, ctypes.c_size_t * 8
            b8 = [(VP8(*[in_ptrs[(8 * k + j) % nbuf] for j in range(8)]), SZ8(*([N_BLOCKS] * 8)), VP8(*[out_ptrs[(8 * k + j) % nbuf] for j in range(8)])) for k in range(nbuf // 8)]

            def eight(k):
                a_, n_, o_ = b8[k % len(b8)]
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, 8, a_, n_, o_, NBX, None, None, sp) == 0

            for k in range(16):
                eight(k)
            e0.record(stream)
            for k in range(32):
                eight(k)
            e1.record(stream)
            torch.cuda.synchronize()
            b8_s = e0.elapsed_time(e1) / 1e3 / 256
            extra["batch_8_atlases_separate_allocations"] = {"us_per_atlas": round(b8_s * 1e6, 3), "mblocks_s": round(N_BLOCKS / b8_s / 1e6, 1),
                                                             "frac_of_hbm_peak": round(BYTES_PER_BLOCK * N_BLOCKS / b8_s / 1e9 / HBM_PEAK_GBS, 4),
                                                             "verified": all(bool(torch.equal(outs[k], g_bc7[idxs[k]])) for k in range(nbuf)),
                                                             "note": "one call of bu_uastc_transcode_batch_device per eight 2^20-block slices in separate allocations = ONE launch over 2^23 blocks, "
                                                                     "calls back to back on one stream, exclusive policy; every rotated output compared afterwards"}
            # ... and ALL the headline's atlases (separate allocations) through one call = ONE launch on the caller's stream: a persistent grid over 64 runs of whole
            # rectangular tiles, drawing its tiles by ticket -- the stream-ordered way to the pipeline's rate, and one kernel for a profiler to time
            if nbuf <= 96:
                VPn, SZn = ctypes.c_void_p * nbuf, ctypes.c_size_t * nbuf
                ball = (VPn(*[in_ptrs[j] for j in range(nbuf)]), SZn(*([N_BLOCKS] * nbuf)), VPn(*[out_ptrs[j] for j in range(nbuf)]))
                for o_ in outs:
                    o_.zero_()

                def all_atlases():
                    assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, nbuf, ball[0], ball[1], ball[2], NBX, None, None, sp) == 0

                t_b = time.perf_counter()
                while (time.perf_counter() - t_b) * 1e3 < max(args.prewarm_ms, 1.0) * 2:
                    all_atlases()
                    torch.cuda.synchronize()
                e0.record(stream)
                for k in range(12):
                    all_atlases()
                e1.record(stream)
                torch.cuda.synchronize()
                ball_s = e0.elapsed_time(e1) / 1e3 / (12 * nbuf)
                extra["batch_all_atlases_one_launch"] = {"atlases_per_launch": nbuf, "us_per_atlas": round(ball_s * 1e6, 3), "mblocks_s": round(N_BLOCKS / ball_s / 1e6, 1),
                                                         "frac_of_hbm_peak": round(BYTES_PER_BLOCK * N_BLOCKS / ball_s / 1e9 / HBM_PEAK_GBS, 4),
                                                         "verified": all(bool(torch.equal(outs[k], g_bc7[idxs[k]])) for k in range(nbuf)),
                                                         "note": "ONE call of bu_uastc_transcode_batch_device over the headline's %d atlases in their separate allocations = one launch "
                                                                 "on the caller's stream (run table in the kernel arguments, copied to LDS; every run tiled as 64 x 16-block rectangles; "
                                                                 "tiles drawn by ticket), 12 calls back to back between events" % nbuf}
            # the same loop as ONE call of the pipelined entry point: bu_uastc_transcode_batch_in_flight issues the slices -- eight atlases per multi-run launch --
            # round-robin on the context's four streams under the shared policy and returns; bu_context_synchronize waits.  512 slices per call (the 64 atlases eight times over),
            # host clock around call + wait -- the pipeline's fill and drain and the final wake-up are inside
            nsl = 8 * nbuf
            VPs, SZs = ctypes.c_void_p * nsl, ctypes.c_size_t * nsl
            fl_in, fl_out, fl_n = VPs(*[in_ptrs[j % nbuf] for j in range(nsl)]), VPs(*[out_ptrs[j % nbuf] for j in range(nsl)]), SZs(*([N_BLOCKS] * nsl))
            for o_ in outs:
                o_.zero_()
            torch.cuda.synchronize()

            def in_flight_call():
                t0_ = time.perf_counter()
                assert lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, nsl, fl_in, fl_n, fl_out, NBX, None, None, 4) == 0
                ctx.synchronize()
                return time.perf_counter() - t0_

            for _ in range(6):
                in_flight_call()
            fl_s = sorted(in_flight_call() for _ in range(5))[2] / nsl
            extra["batch_in_flight_512_atlases_one_call"] = {
                "us_per_atlas": round(fl_s * 1e6, 3), "mblocks_s": round(N_BLOCKS / fl_s / 1e6, 1),
                "frac_of_hbm_peak": round(BYTES_PER_BLOCK * N_BLOCKS / fl_s / 1e9 / HBM_PEAK_GBS, 4),
                "verified": all(bool(torch.equal(outs[k], g_bc7[idxs[k]])) for k in range(nbuf)),
                "note": "ONE call of bu_uastc_transcode_batch_in_flight over 512 slices of 2^20 blocks in separate allocations (the 64 atlases eight times over) on four "
                        "context streams (64 multi-run launches of eight atlases each) + bu_context_synchronize, host clock around both (median of five calls): the "
                        "headline's pipeline as an entry point, its fill, drain and the final wake-up included"}
            # ... and the headline's window itself around that entry point: [call over 2048 lead atlases][start marks][call over K timed atlases][end marks]
            # [call over 8 tail atlases], the context's timing-only events between the calls (ProductWindow) -- every launch of the window planned, shaped and
            # enqueued by the call a user makes (from 256 launches per call on: by one enqueue thread per stream)
            pwh = ProductWindow(env, _lib.BC7, list(in_ptrs), list(out_ptrs), N_BLOCKS, NBX, status.data_ptr(), 4)
            k_pw = max(args.steps, 256)
            pwh.window(0, 256)
            pw_h = sorted(pwh.window(2048, k_pw, 64) for _ in range(3))[1]  # (the call groups eight atlases per launch: a tail of eight launches)
            pw_hs = max(pw_h[0], pw_h[2]) / 1e3 / k_pw
            torch.cuda.synchronize()
            extra["headline_through_product_api"] = {
                "us_per_atlas": round(pw_hs * 1e6, 3), "mblocks_s": round(N_BLOCKS / pw_hs / 1e6, 1), "frac_of_hbm_peak": round(BYTES_PER_BLOCK * N_BLOCKS / pw_hs / 1e9 / HBM_PEAK_GBS, 4),
                "timed_atlases": k_pw, "us_per_atlas_strict_bracket": round(pw_h[1] / k_pw * 1e3, 3), "streams": pwh.streams,
                "verified": all(bool(torch.equal(outs[k], g_bc7[idxs[k]])) for k in range(nbuf)),
                "note": "the headline's timed window with bu_uastc_transcode_batch_in_flight doing the launches: one call carries the timed atlases (grouped by the call's planner into multi-run launches of "
                        "eight atlases = 2^23 blocks, launch i on context stream i % 4, shared shapes), 2048 lead atlases in the call in front, 64 tail atlases in the call behind, timing-only events of "
                        "the context between the calls (bu_time_mark_streams / bu_time_marks_elapsed): last lead launch complete -> last timed launch complete; median of "
                        "three windows; max(event, host) clock"}
        except Exception as e:  # secondary rows must never break the headline line
            extra["launches_in_flight_matrix_error"] = repr(e)
        # a loop over 64 independent slices of 65 536 blocks (256 x 256 blocks: a 1024 x 1024 px mip), the shape of the per-slice
        # loops of basis.rs:246-257: launches back to back on one stream, round-robin on 2 and 4 streams (the rejected alternatives),
        # the batch entry point on separate allocations (ONE launch per 96 runs on the caller's stream, the run table in the kernel
        # arguments) and on one contiguous allocation (the runs merge into one plain launch)
        try:
            ns, nbs = 64, 65536
            s_in = [ins[k % nbuf][(k // nbuf) * nbs: (k // nbuf + 1) * nbs] for k in range(ns)]  # 64 distinct 1 MiB pieces, 16 MiB apart
            s_out = [outs[k % nbuf][(k // nbuf) * nbs: (k // nbuf + 1) * nbs] for k in range(ns)]
            VPn, SZn = ctypes.c_void_p * ns, ctypes.c_size_t * ns
            p_in, p_out, p_n = VPn(*[t.data_ptr() for t in s_in]), VPn(*[t.data_ptr() for t in s_out]), SZn(*([nbs] * ns))
            cat_in = torch.cat(s_in).contiguous()
            cat_out = torch.empty((ns * nbs, 16), dtype=torch.uint8, device=dev)
            c_in = VPn(*[cat_in[k * nbs:(k + 1) * nbs].data_ptr() for k in range(ns)])
            c_out = VPn(*[cat_out[k * nbs:(k + 1) * nbs].data_ptr() for k in range(ns)])
            side = [torch.cuda.Stream(device=dev) for _ in range(4)]

            def loop_streams(n_streams):
                for k in range(ns):
                    st_ = env.stream if n_streams == 1 else side[k % n_streams]
                    lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, p_in[k], nbs, p_out[k], 256, 0, None, ctypes.c_void_p(st_.cuda_stream))

            def batch(pi, po):
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, ns, pi, p_n, po, 256, None, None, sp) == 0

            def wall(fn, reps=20):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps

            def in_flight(pi, po):  # the pipelined entry point: groups of 16 slices (2^20 blocks) per launch, four launches on four context streams, half-CU grids
                assert lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, ns, pi, p_n, po, 256, None, None, 4) == 0

            rows = {"one_stream": wall(lambda: loop_streams(1)), "two_streams": wall(lambda: loop_streams(2)), "four_streams": wall(lambda: loop_streams(4)),
                    "batch_call_separate_allocations": wall(lambda: batch(p_in, p_out)), "batch_call_contiguous": wall(lambda: batch(c_in, c_out)),
                    "in_flight_call_separate_allocations": wall(lambda: in_flight(p_in, p_out))}
            extra["slices_64_x_65536"] = {k: {"us_per_batch": round(v * 1e6, 2), "mblocks_s": round(ns * nbs / v / 1e6, 1)} for k, v in rows.items()}
            extra["slices_64_x_65536"]["verified"] = bool(torch.equal(cat_out[:nbs], outs[0][:nbs]))
            extra["slices_64_x_65536"]["note"] = ("wall clock per 64-slice batch (4 Mi blocks): a 65 536-block slice fills a quarter of the chip, so launches "
                                                  "on one stream leave it mostly idle; bu_uastc_transcode_batch_device is what a per-slice loop should call")
            del cat_in, cat_out
        except Exception as e:  # secondary rows must never break the headline line
            extra["slices_64_x_65536_error"] = repr(e)
        # the per-block API in the reference's own benchmark shape (benches/benchmark.rs:66-98: 32 blocks x 1000 calls per target)
        try:
            blk = np.ascontiguousarray(golden["uastc"][np.arange(32) * 19 % 608])
            pb = {}
            for tname, tcode, bbytes in (("astc", _lib.ASTC, 16), ("bc7", _lib.BC7, 16), ("etc1", _lib.ETC1, 8), ("etc2", _lib.ETC2, 16), ("rgba32", _lib.RGBA32, 64)):
                o = np.zeros((32, bbytes), dtype=np.uint8)
                ns = ctypes.c_float(0)
                check(env, lib.bu_time_block_api(ctx.handle, tcode, blk.ctypes.data, 32, 1000, o.ctypes.data, ctypes.byref(ns)), "bu_time_block_api")
                key = "rgba" if tname == "rgba32" else tname
                pb[tname] = {"us_per_call": round(ns.value / 1e3, 4), "verified": bool((o == golden[key][np.arange(32) * 19 % 608]).all())}
            ctx.block_api_on_device(True)
            try:
                o = np.zeros((32, 16), dtype=np.uint8)
                ns = ctypes.c_float(0)
                check(env, lib.bu_time_block_api(ctx.handle, _lib.BC7, blk.ctypes.data, 32, 10, o.ctypes.data, ctypes.byref(ns)), "bu_time_block_api")
                pb["bc7_through_one_block_launches"] = {"us_per_call": round(ns.value / 1e3, 2), "verified": bool((o == golden["bc7"][np.arange(32) * 19 % 608]).all())}
            finally:
                ctx.block_api_on_device(False)
            pb["note"] = ("bu_transcode_uastc_block_to_* / bu_unpack_uastc_block_to_rgba, 32 blocks x 1000 calls each (benches/benchmark.rs:66-98), native loop "
                          "(bu_time_block_api): the library's own block code on the calling thread; last row: the same calls forced through a one-block "
                          "kernel launch (bu_block_api_on_device)")
            extra["per_block_api"] = pb
        except Exception as e:
            extra["per_block_api_error"] = repr(e)
        # BASELINE config 5's kernel on one GPU: the whole 512-slice array (2^25 blocks, 512 MiB in + 512 MiB out) in ONE launch,
        # two input / output pairs rotated (every launch misses the 256 MiB Infinity Cache), verified against the known answers
        try:
            nbig = 1 << 25
            big_in, big_out = [], []
            for k in range(2):
                gen = torch.Generator(device=dev)
                gen.manual_seed(9000 + k)
                bidx = torch.randint(0, 608, (nbig,), device=dev, generator=gen)
                big_in.append(g_uastc[bidx].contiguous())
                big_out.append(torch.empty((nbig, 16), dtype=torch.uint8, device=dev))
                if k == 0:
                    big_idx0 = bidx
                del bidx
            bi, bo = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in big_in]), (ctypes.c_void_p * 2)(*[t.data_ptr() for t in big_out])

            def big_window(lead, launches):
                ev, host, late = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
                check(env, lib.bu_time_uastc_launches_window(ctx.handle, _lib.BC7, bi, bo, 2, 0, nbig, 256, lead, launches, ctypes.c_void_p(status.data_ptr()), sp,
                                                             ctypes.byref(ev), ctypes.byref(host), ctypes.byref(late)), "bu_time_uastc_launches_window")
                return ev.value / 1e3 / launches

            # the fixed walk first (every workgroup walks tiles b, b + grid, ...: what rounds 1-5 shipped), then -- the row itself -- the launch as the library
            # issues it now: the workgroups draw their tiles by ticket, on the caller's stream like on any other
            lib.bu_time_set_tile_tickets(ctx.handle, 0)
            big_window(0, 2)
            t_big = time.perf_counter()
            while args.prewarm_ms > 0 and (time.perf_counter() - t_big) * 1e3 < 4 * args.prewarm_ms:
                big_window(0, 8)
            fixed_s = big_window(8, 40)
            lib.bu_time_set_tile_tickets(ctx.handle, 1)
            for t_ in big_out:
                t_.zero_()
            big_window(0, 2)
            torch.cuda.synchronize()
            big_ok = bool(torch.equal(big_out[0], g_bc7[big_idx0]))
            t_big = time.perf_counter()
            while args.prewarm_ms > 0 and (time.perf_counter() - t_big) * 1e3 < 4 * args.prewarm_ms:  # the 1 GiB launches take ~100 ms to settle the clocks
                big_window(0, 8)
            big_s = big_window(8, 40)
            extra["array512_one_launch"] = {"blocks": nbig, "us_per_launch": round(big_s * 1e6, 2), "mblocks_s": round(nbig / big_s / 1e6, 1),
                                            "gb_s": round(BYTES_PER_BLOCK * nbig / big_s / 1e9, 1), "frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / big_s / 1e9 / HBM_PEAK_GBS, 4),
                                            "verified": big_ok,
                                            "fixed_walk_us_per_launch": round(fixed_s * 1e6, 2), "fixed_walk_frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / fixed_s / 1e9 / HBM_PEAK_GBS, 4),
                                            "note": "BASELINE config 5 on ONE GPU: 512 slices x 65 536 blocks contiguous, ONE launch per step on the caller's stream (bu_uastc_transcode_device), "
                                                    "cold (two 1 GiB pairs rotated), 8 lead + 40 timed launches between events.  The persistent workgroups draw their tiles by ticket "
                                                    "(round 6; fixed_walk_*: the same launch with every workgroup walking its fixed share, bu_time_set_tile_tickets(ctx, 0)); "
                                                    "`--config array512` is the sharded form"}
            # the same launch on one of the context's OWN streams (streams window helper), and the BLOCKING product call over the array on the host clock
            try:
                ctx.set_launch_policy(False)  # (exclusive, as the row above; BU_LAUNCH_AUTO picks the same for a launch that is alone)

                def big1(lead, launches):
                    ev, host = ctypes.c_float(0), ctypes.c_float(0)
                    check(env, lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, bi, bo, 2, 0, nbig, 256, lead, launches, 0, 1, ctypes.c_void_p(status.data_ptr()),
                                                                         ctypes.byref(ev), ctypes.byref(host), None, None), "bu_time_uastc_launches_streams_window")
                    return max(ev.value, host.value) / 1e3 / launches

                for t_ in big_out:
                    t_.zero_()
                torch.cuda.synchronize()
                big1(0, 2)
                torch.cuda.synchronize()
                tk_ok = bool(torch.equal(big_out[0], g_bc7[big_idx0]))
                t_big = time.perf_counter()
                while args.prewarm_ms > 0 and (time.perf_counter() - t_big) * 1e3 < 4 * args.prewarm_ms:  # (as the row above: these launches take ~100 ms to settle the clocks)
                    big1(0, 8)
                tk_s = sorted(big1(8, 40) for _ in range(3))[1]
                t0_ = time.perf_counter()
                n_sync = 12
                for r_ in range(n_sync):
                    assert ctx.transcode_device_sync(_lib.BC7, big_in[r_ % 2], nbig, big_out[r_ % 2], blocks_per_row=256) == _lib.STATUS_WORD_CLEAR
                sync_s = (time.perf_counter() - t0_) / n_sync
                extra["array512_one_launch_tile_tickets"] = {
                    "blocks": nbig, "us_per_launch": round(tk_s * 1e6, 2), "mblocks_s": round(nbig / tk_s / 1e6, 1), "gb_s": round(BYTES_PER_BLOCK * nbig / tk_s / 1e9, 1),
                    "frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / tk_s / 1e9 / HBM_PEAK_GBS, 4), "verified": tk_ok,
                    "blocking_call_us_per_array": round(sync_s * 1e6, 2), "blocking_call_frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / sync_s / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "one launch per array, one at a time, on context stream 0 (8 lead + 40 timed launches between events): from 16 tiles per workgroup on "
                            "the persistent workgroups draw their tiles by ticket (eight counters in device memory, drawn one tile ahead) instead of walking fixed "
                            "shares -- the launch ends when the tiles do, not when the slowest share does (profiles/r06_ab_tile_tickets.txt).  blocking_call: "
                            "bu_uastc_transcode_device_sync per array on the host clock (launch from an idle chip + kernel + completion seen by polling), what "
                            "bu_array_transcode_sharded and sharded.gpu_transcode_fn run per device"}
            except Exception as e:
                extra["array512_one_launch_tile_tickets_error"] = repr(e)
            # the same array as FOUR launches of 2^23 blocks in flight on four context streams (shared policy): what `--config array512` does per rank
            try:
                bi4 = (ctypes.c_void_p * 8)(*[t.data_ptr() + q * (nbig // 4) * 16 for t in big_in for q in range(4)])
                bo4 = (ctypes.c_void_p * 8)(*[t.data_ptr() + q * (nbig // 4) * 16 for t in big_out for q in range(4)])
                ctx.set_launch_policy(True)

                def big4(lead_steps, steps):
                    ev, host = ctypes.c_float(0), ctypes.c_float(0)
                    check(env, lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, bi4, bo4, 8, 0, nbig // 4, 256, 4 * lead_steps, 4 * steps, 4 if lead_steps else 0, 4,
                                                                         ctypes.c_void_p(status.data_ptr()), ctypes.byref(ev), ctypes.byref(host), None, None), "bu_time_uastc_launches_streams_window")
                    return max(ev.value, host.value) / 1e3 / steps

                for t_ in big_out:
                    t_.zero_()
                torch.cuda.synchronize()
                big4(0, 2)
                torch.cuda.synchronize()
                big4_ok = bool(torch.equal(big_out[0], g_bc7[big_idx0]))
                t_big = time.perf_counter()
                while args.prewarm_ms > 0 and (time.perf_counter() - t_big) * 1e3 < 4 * args.prewarm_ms:
                    big4(0, 8)
                b4_s = big4(8, 40)
                extra["array512_four_launches_in_flight"] = {"blocks": nbig, "us_per_array": round(b4_s * 1e6, 2), "mblocks_s": round(nbig / b4_s / 1e6, 1), "gb_s": round(BYTES_PER_BLOCK * nbig / b4_s / 1e9, 1),
                                                             "frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / b4_s / 1e9 / HBM_PEAK_GBS, 4), "verified": big4_ok,
                                                             "note": "the 512-slice array as four launches of 128 slices (2^23 blocks) each on four context streams, shared launch policy; 8 lead + 40 "
                                                                     "timed passes over the array, window from the last lead launch's completion to the last timed launch's"}
                # ... and as ONE call of the pipelined entry point over the 512 slices (one allocation: they merge into one run, which the call cuts
                # into four pieces on tile boundaries): call + bu_context_synchronize by the host clock, back to back on the two rotated pairs
                VPa, SZa = ctypes.c_void_p * 512, ctypes.c_size_t * 512
                a_n = SZa(*([65536] * 512))
                a_io = [(VPa(*[t_i.data_ptr() + k * 65536 * 16 for k in range(512)]), VPa(*[t_o.data_ptr() + k * 65536 * 16 for k in range(512)])) for t_i, t_o in zip(big_in, big_out)]
                for t_ in big_out:
                    t_.zero_()
                torch.cuda.synchronize()

                def array_calls(k0, n_calls):
                    t0_ = time.perf_counter()
                    for k in range(k0, k0 + n_calls):
                        check(env, lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, 512, a_io[k % 2][0], a_n, a_io[k % 2][1], 256, None, ctypes.c_void_p(status.data_ptr()), 4),
                              "bu_uastc_transcode_batch_in_flight")
                    ctx.synchronize()
                    return (time.perf_counter() - t0_) / n_calls

                array_calls(0, 2)
                a512_call_ok = bool(torch.equal(big_out[0], g_bc7[big_idx0]))
                array_calls(0, 64)  # clocks
                call_s = sorted(array_calls(16 * r_, 16) for r_ in range(5))[2]
                one_s_ = sorted(array_calls(r_, 1) for r_ in range(5))[2]
                extra["array512_one_call_in_flight"] = {"blocks": nbig, "us_per_array": round(call_s * 1e6, 2), "mblocks_s": round(nbig / call_s / 1e6, 1),
                                                        "frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / call_s / 1e9 / HBM_PEAK_GBS, 4),
                                                        "verified": a512_call_ok, "us_per_array_one_call_then_wait": round(one_s_ * 1e6, 2),
                                                        "note": "the 512-slice array through ONE call of bu_uastc_transcode_batch_in_flight per array (512 slice pointers into one "
                                                                "allocation; the call cuts the run into four launches of 2^23 blocks on four context streams): 16 calls back to back, "
                                                                "then bu_context_synchronize, host clock around all of it / 16 (median of five).  us_per_array_one_call_then_wait: a "
                                                                "single call and the wait, starting on an idle chip and ending with a wake-up"}
                # ... and the array through the product's pipelined entry point the way `--config array512` times it for every N: K arrays per call, one launch
                # per array, four arrays in flight; lead / timed / tail calls with the context's timing-only events between them (ProductWindow)
                try:
                    more_in = [torch.roll(big_in[0], shifts=977 * (r_ + 1), dims=0).contiguous() for r_ in range(4)]
                    more_out = [torch.empty((nbig, 16), dtype=torch.uint8, device=dev) for _ in range(4)]
                    pwa = ProductWindow(env, _lib.BC7, [t.data_ptr() for t in big_in + more_in], [t.data_ptr() for t in big_out + more_out], nbig, 256, status.data_ptr(), 4)
                    for t_ in big_out:
                        t_.zero_()
                    torch.cuda.synchronize()
                    pwa.window(0, 6)
                    pw_ok = bool(torch.equal(big_out[0], g_bc7[big_idx0])) and bool(torch.equal(more_out[1], torch.roll(big_out[0], shifts=977 * 2, dims=0)))
                    t_big = time.perf_counter()
                    while args.prewarm_ms > 0 and (time.perf_counter() - t_big) * 1e3 < 4 * args.prewarm_ms:
                        pwa.window(0, 12)
                    pw_runs = sorted(pwa.window(24, 40) for _ in range(3))
                    pw_ev, pw_strict, pw_host = pw_runs[1]
                    pw_s = max(pw_ev, pw_host) / 1e3 / 40
                    extra["array512_through_product_api"] = {
                        "blocks": nbig, "us_per_array": round(pw_s * 1e6, 2), "mblocks_s": round(nbig / pw_s / 1e6, 1), "gb_s": round(BYTES_PER_BLOCK * nbig / pw_s / 1e9, 1),
                        "frac_of_hbm_peak": round(BYTES_PER_BLOCK * nbig / pw_s / 1e9 / HBM_PEAK_GBS, 4), "verified": pw_ok,
                        "us_per_array_strict_bracket": round(pw_strict / 40 * 1e3, 2), "streams": pwa.streams,
                        "ratio_to_four_launches_in_flight": round(pw_s / b4_s, 4),
                        "note": "the 512-slice array through bu_uastc_transcode_batch_in_flight exactly as `--config array512` times it: ONE call carries 40 arrays (one "
                                "slice = one 2^25-block launch each, array i on context stream i % 4, shared shapes), 24 lead arrays in the call in front, 8 tail arrays "
                                "in the call behind, the context's timing-only events between the calls (bu_time_mark_streams / bu_time_marks_elapsed): last lead "
                                "launch complete -> last timed launch complete; median of three windows; six rotated 1 GiB pairs"}
                    del more_in, more_out, pwa
                except Exception as e:
                    extra["array512_through_product_api_error"] = repr(e)
            finally:
                ctx.set_launch_policy(policy_now[0])
            a512 = pmc_array512()
            if a512:  # rocprofv3's own view of this launch: committed kernel-trace and counter passes (tools/gpu_pmc.sh), NOT measured in this run
                extra["array512_one_launch"].update(a512)
            del big_in, big_out, big_idx0
        except Exception as e:
            extra["array512_one_launch_error"] = repr(e)
        # mode-coherent atlases (mode chosen per 8x8-block tile): texture-like, waves see 1-2 modes
        coh = []
        for k in range(min(nbuf, 64)):
            cidx = torch.from_numpy(synth.coh_indices(NBX, NBY, seed=synth.GOLD_SEED + k)).to(dev)
            coh.append(g_uastc[cidx].contiguous())
            if k == 0:
                coh_idx0 = cidx
        coh_ptrs = (ctypes.c_void_p * len(coh))(*[t.data_ptr() for t in coh])
        rot[0] = 0
        run(len(coh), inp=coh_ptrs, outp=out_ptrs, nb=len(coh))
        torch.cuda.synchronize()
        coh_ok = bool(torch.equal(outs[0], g_bc7[coh_idx0]))
        coh_s = row(max(args.steps, 64), inp=coh_ptrs, outp=out_ptrs, nb=len(coh))
        extra["coherent_atlas"] = {"gb_s": round(BYTES_PER_BLOCK * N_BLOCKS / coh_s / 1e9, 1), "us_per_launch": round(coh_s * 1e6, 3),
                                   "mblocks_s": round(N_BLOCKS / coh_s / 1e6, 1), "verified": coh_ok,
                                   "note": "A-coh: UASTC mode chosen per 8x8-block tile"}
        try:
            sramp(4, True, inp=coh_ptrs, outp=out_ptrs, nb=len(coh))
            coh4 = srow(256, 4, True, inp=coh_ptrs, outp=out_ptrs, nb=len(coh))
            torch.cuda.synchronize()
            extra["coherent_atlas"]["in_flight_4_shared"] = {"us_per_atlas": round(coh4 * 1e6, 3), "frac_of_hbm_peak": round(BYTES_PER_BLOCK * N_BLOCKS / coh4 / 1e9 / HBM_PEAK_GBS, 4),
                                                             "verified": bool(torch.equal(outs[0], g_bc7[coh_idx0]))}
        except Exception as e:
            extra["coherent_atlas"]["in_flight_error"] = repr(e)
        del coh
        # end-to-end row: host buffer -> host buffer through bu_uastc_transcode (H2D + kernel + D2H, PCIe-bound)
        host_in = golden["uastc"][idx0]
        ctx.transcode(_lib.BC7, host_in)
        t0 = time.perf_counter()
        e2e_reps = 5
        for _ in range(e2e_reps):
            host_out = ctx.transcode(_lib.BC7, host_in)
        e2e_s = (time.perf_counter() - t0) / e2e_reps
        extra["end_to_end_host_pointers"] = {"mblocks_s": round(N_BLOCKS / e2e_s / 1e6, 1), "ms_per_atlas": round(e2e_s * 1e3, 3),
                                             "verified": bool((host_out.reshape(-1, 16) == golden["bc7"][idx0]).all()),
                                             "note": "pageable host memory, includes PCIe both ways -- never the headline value"}
        # the same call on page-locked buffers (bu_host_alloc): zero-copy, the kernels read and write host memory over PCIe
        pin_in, pin_out = ctx.host_alloc(N_BLOCKS * 16), ctx.host_alloc(N_BLOCKS * 16)
        pin_in[:] = host_in.reshape(-1)
        ctx.transcode(_lib.BC7, pin_in, out=pin_out)
        t0 = time.perf_counter()
        for _ in range(e2e_reps):
            ctx.transcode(_lib.BC7, pin_in, out=pin_out)
        pin_s = (time.perf_counter() - t0) / e2e_reps
        extra["end_to_end_page_locked"] = {"mblocks_s": round(N_BLOCKS / pin_s / 1e6, 1), "ms_per_atlas": round(pin_s * 1e3, 3),
                                           "verified": bool((pin_out.reshape(-1, 16) == golden["bc7"][idx0]).all()),
                                           "note": "bu_host_alloc buffers: kernels read/write host memory directly over PCIe (no staging copies) -- never the headline value"}
        ctx.host_free(pin_in)
        ctx.host_free(pin_out)
        # the other block-linear targets of the same atlas (secondary rows; cold rotation over the same buffers): one launch at a time under
        # the exclusive policy, and four in flight under the shared policy (the headline's method), both verified against the known answers
        for tname, tcode, bpb in (("astc", _lib.ASTC, 32), ("etc1", _lib.ETC1, 24), ("etc2", _lib.ETC2, 32)):
            g_t = torch.from_numpy(golden[tname]).to(dev)

            def t_ok():
                torch.cuda.synchronize()
                got = [outs[k] if bpb != 24 else outs[k].view(-1)[: N_BLOCKS * 8].view(N_BLOCKS, 8) for k in (0, 1, nbuf - 1)]
                return all(bool(torch.equal(g_, g_t[idxs[k]])) for g_, k in zip(got, (0, 1, nbuf - 1)))

            ramp(target=tcode)
            ts = row(256, target=tcode)
            ok1 = t_ok()
            sramp(4, True, target=tcode)
            t4 = srow(256, 4, True, target=tcode)
            ok4 = t_ok()
            # ... and the headline's method for this target: ONE bu_uastc_transcode_batch_device launch over all the atlases in their separate allocations
            tb_s, okb = None, None
            if nbuf <= 96:
                VPn, SZn = ctypes.c_void_p * nbuf, ctypes.c_size_t * nbuf
                tb_a = (VPn(*[in_ptrs[j] for j in range(nbuf)]), SZn(*([N_BLOCKS] * nbuf)), VPn(*[out_ptrs[j] for j in range(nbuf)]))
                for o_ in outs:
                    o_.zero_()
                tb_t0 = time.perf_counter()  # (clocks: the same long lead-in as the BC7 row's)
                while (time.perf_counter() - tb_t0) * 1e3 < max(args.prewarm_ms, 1.0) * 2:
                    assert lib.bu_uastc_transcode_batch_device(ctx.handle, tcode, nbuf, tb_a[0], tb_a[1], tb_a[2], NBX, None, None, sp) == 0
                    torch.cuda.synchronize()
                tb_e0, tb_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tb_e0.record(stream)
                for _ in range(8):
                    assert lib.bu_uastc_transcode_batch_device(ctx.handle, tcode, nbuf, tb_a[0], tb_a[1], tb_a[2], NBX, None, None, sp) == 0
                tb_e1.record(stream)
                torch.cuda.synchronize()
                tb_s = tb_e0.elapsed_time(tb_e1) / 1e3 / (8 * nbuf)
                okb = t_ok()
            extra["uastc_to_" + tname] = {"gb_s": round(bpb * N_BLOCKS / ts / 1e9, 1), "us_per_launch": round(ts * 1e6, 3),
                                          "mblocks_s": round(N_BLOCKS / ts / 1e6, 1), "bytes_per_block": bpb, "frac_of_hbm_peak": round(bpb * N_BLOCKS / ts / 1e9 / HBM_PEAK_GBS, 4),
                                          "verified": ok1,
                                          "in_flight_4_shared": {"us_per_atlas": round(t4 * 1e6, 3), "mblocks_s": round(N_BLOCKS / t4 / 1e6, 1), "gb_s": round(bpb * N_BLOCKS / t4 / 1e9, 1),
                                                                 "frac_of_hbm_peak": round(bpb * N_BLOCKS / t4 / 1e9 / HBM_PEAK_GBS, 4), "verified": ok4}}
            if tb_s is not None:
                extra["uastc_to_" + tname]["all_atlases_one_launch"] = {"atlases_per_launch": nbuf, "us_per_atlas": round(tb_s * 1e6, 3), "mblocks_s": round(N_BLOCKS / tb_s / 1e6, 1),
                                                                        "gb_s": round(bpb * N_BLOCKS / tb_s / 1e9, 1),
                                                                        "frac_of_hbm_peak": round(bpb * N_BLOCKS / tb_s / 1e9 / HBM_PEAK_GBS, 4), "verified": okb}
            del g_t
        # config 4 shape: ETC1S 2048x2048 (512x512 blocks), 4096-entry endpoint / 8192-entry selector codebooks
        try:
            from basisu_rs_amd import etc1s_selector_from_rows

            ep, rows = synth.etc1s_codebooks(4096, 8192, seed=2)
            sel = etc1s_selector_from_rows(rows)
            nbl = 512 * 512
            d_ep = torch.from_numpy(ep.view(np.int32)).to(dev)
            d_sel = torch.from_numpy(sel).to(dev)
            d_idx = [torch.from_numpy(synth.etc1s_indices(nbl, 4096, 8192, seed=100 + k).view(np.int32)).to(dev) for k in range(32)]
            d_o8 = [torch.empty((nbl, 8), dtype=torch.uint8, device=dev) for _ in range(32)]
            d_o64 = [torch.empty((nbl, 64), dtype=torch.uint8, device=dev) for _ in range(32)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # (argument tuples built once: at 4-5 us of kernel the loop is bound by what the host spends per call, and four .data_ptr()
            # calls per launch were most of that)
            h_, pe_, ps_ = ctx.handle, d_ep.data_ptr(), d_sel.data_ptr()
            a8 = [(h_, d_idx[k].data_ptr(), nbl, pe_, 4096, ps_, 8192, d_o8[k].data_ptr(), None, sp) for k in range(32)]
            a64 = [(h_, d_idx[k].data_ptr(), None, 512, 512, pe_, 4096, ps_, 8192, d_o64[k].data_ptr(), None, sp) for k in range(32)]
            f8, f64 = lib.bu_etc1s_transcode_etc1_device, lib.bu_etc1s_decode_rgba_device
            for name, fn, bpb in (("etc1s_to_etc1", lambda k: f8(*a8[k]), 12), ("etc1s_to_rgba32", lambda k: f64(*a64[k]), 68)):
                for k in range(32):
                    fn(k)
                torch.cuda.synchronize()
                e0.record(stream)
                reps = 256
                for i in range(reps):
                    fn(i % 32)
                e1.record(stream)
                torch.cuda.synchronize()
                ts = e0.elapsed_time(e1) / 1e3 / reps
                extra[name] = {"gb_s": round(bpb * nbl / ts / 1e9, 1), "us_per_launch": round(ts * 1e6, 3), "mblocks_s": round(nbl / ts / 1e6, 1),
                               "bytes_per_block": bpb, "blocks": nbl, "parity": ETC1S_UNPINNED}
            # the same 2^18-block launches (launch-bound one at a time: a 4-6 us dispatch for 0.5-3 us of memory traffic) with four in flight on the context's
            # streams: independent slices of a mip chain or an array
            try:
                VP32 = ctypes.c_void_p * 32
                pi32 = VP32(*[t.data_ptr() for t in d_idx])
                for name, rgba_, outl, bpb in (("etc1s_to_etc1", 0, d_o8, 12), ("etc1s_to_rgba32", 1, d_o64, 68)):
                    po32 = VP32(*[t.data_ptr() for t in outl])

                    def ewin(lead_, n_, tail_):
                        ev, host = ctypes.c_float(0), ctypes.c_float(0)
                        check(env, lib.bu_time_etc1s_launches_streams_window(ctx.handle, rgba_, pi32, po32, 32, 0, 512, 512, ctypes.c_void_p(pe_), 4096, ctypes.c_void_p(ps_), 8192,
                                                                             lead_, n_, tail_, 4, ctypes.byref(ev), ctypes.byref(host)), "bu_time_etc1s_launches_streams_window")
                        return max(ev.value, host.value) / 1e3 / n_

                    t0_ = time.perf_counter()
                    while args.prewarm_ms > 0 and (time.perf_counter() - t0_) * 1e3 < args.prewarm_ms:
                        ewin(0, 256, 0)
                    t4_ = ewin(64, 256, 4)
                    extra[name]["in_flight_4"] = {"us_per_launch": round(t4_ * 1e6, 3), "gb_s": round(bpb * nbl / t4_ / 1e9, 1), "mblocks_s": round(nbl / t4_ / 1e6, 1),
                                                  "frac_of_hbm_peak": round(bpb * nbl / t4_ / 1e9 / HBM_PEAK_GBS, 4)}
            except Exception as e:
                extra["etc1s_in_flight_error"] = repr(e)
            del d_idx, d_o8, d_o64
            # the same kernels on a large slice (2^22 blocks, 2048 x 2048 blocks): the codebooks are staged in LDS from 2^19 blocks up
            nbl2 = 1 << 22
            d_idx2 = [torch.from_numpy(synth.etc1s_indices(nbl2, 4096, 8192, seed=300 + k).view(np.int32)).to(dev) for k in range(4)]
            d_o8b = [torch.empty((nbl2, 8), dtype=torch.uint8, device=dev) for _ in range(4)]
            d_o64b = [torch.empty((nbl2, 64), dtype=torch.uint8, device=dev) for _ in range(4)]
            for name, fn, bpb in (("etc1s_to_etc1_4Mi_blocks", lambda k: lib.bu_etc1s_transcode_etc1_device(ctx.handle, d_idx2[k].data_ptr(), nbl2, d_ep.data_ptr(), 4096, d_sel.data_ptr(), 8192, d_o8b[k].data_ptr(), None, sp), 12),
                                  ("etc1s_to_rgba32_4Mi_blocks", lambda k: lib.bu_etc1s_decode_rgba_device(ctx.handle, d_idx2[k].data_ptr(), None, 2048, 2048, d_ep.data_ptr(), 4096, d_sel.data_ptr(), 8192, d_o64b[k].data_ptr(), None, sp), 68)):
                for k in range(4):
                    fn(k)
                torch.cuda.synchronize()
                e0.record(stream)
                reps = 32
                for i in range(reps):
                    fn(i % 4)
                e1.record(stream)
                torch.cuda.synchronize()
                ts = e0.elapsed_time(e1) / 1e3 / reps
                extra[name] = {"gb_s": round(bpb * nbl2 / ts / 1e9, 1), "us_per_launch": round(ts * 1e6, 3), "mblocks_s": round(nbl2 / ts / 1e6, 1),
                               "bytes_per_block": bpb, "blocks": nbl2, "frac_of_hbm_peak": round(bpb * nbl2 / ts / 1e9 / HBM_PEAK_GBS, 3), "parity": ETC1S_UNPINNED}
            del d_idx2, d_o8b, d_o64b
        except Exception as e:  # secondary rows must never break the headline line
            extra["etc1s_error"] = repr(e)
        # config 4 end to end: a .basis ETC1S file (16 slices x 16 384 blocks) through read_to_rgba -- host BasisLZ decode of
        # the slices (concurrent on the host cores) + ONE GPU launch for the whole file.  The entropy decode is the whole cost.
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import basis_builder as bb  # test-only encoder: synthesises the input file
            import basisu_rs_amd as bu
            fbytes, _, _ = bb.etc1s_file(np.random.default_rng(44), [(128, 128)] * 16, n_codebook=4096)
            file_out = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, fbytes)[1])
            bu.read_to_rgba(fbytes, ctx, out=file_out)
            times = []
            for _ in range(9):
                t0 = time.perf_counter()
                hdr_f, imgs_f = bu.read_to_rgba(fbytes, ctx, out=file_out)
                times.append(time.perf_counter() - t0)
            file_s = sorted(times)[len(times) // 2]  # median: waking parked host threads is noisy
            ctx.host_free(file_out)
            t0 = time.perf_counter()
            for k in range(16):
                bu.basislz_decode(fbytes, k)
            seq_s = time.perf_counter() - t0
            extra["etc1s_file_read_to_rgba"] = {"slices": 16, "blocks": 16 * 16384, "file_bytes": len(fbytes), "ms_per_file": round(file_s * 1e3, 3),
                                                "mblocks_s": round(16 * 16384 / file_s / 1e6, 1),
                                                "ms_slice_by_slice_host_decode_only": round(seq_s * 1e3, 3),
                                                "parity": ETC1S_UNPINNED,
                                                "note": "whole-file API: parse + CRC + BasisLZ decode of all slices on the host cores + GPU decode + download"}
            # config 4 as stated: ONE slice of 512 x 512 blocks (the entropy decode of a single slice is serial: one host core)
            fone, _, _ = bb.etc1s_file(np.random.default_rng(45), [(512, 512)], n_codebook=4096)
            one_out = ctx.host_alloc(bu.read_query(_lib.READ_RGBA, fone)[1])
            bu.read_to_rgba(fone, ctx, out=one_out)
            times = []
            for _ in range(9):
                t0 = time.perf_counter()
                bu.read_to_rgba(fone, ctx, out=one_out)
                times.append(time.perf_counter() - t0)
            one_s = sorted(times)[len(times) // 2]
            ctx.host_free(one_out)
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                bu.basislz_decode(fone, 0)
                times.append(time.perf_counter() - t0)
            lz_s = sorted(times)[len(times) // 2]
            row4 = {"parity": ETC1S_UNPINNED, "blocks": 512 * 512, "file_bytes": len(fone), "ms_per_file": round(one_s * 1e3, 3), "mblocks_s": round(512 * 512 / one_s / 1e6, 1),
                    "ms_basislz_decode_of_the_slice_on_one_host_thread": round(lz_s * 1e3, 3),
                    "note": "BASELINE config 4 at its stated size through the whole-file API (read_to_rgba, page-locked output).  The slice's entropy "
                            "decode is one serial bit stream; inside the call it runs on two host threads (bit-serial lexer + index resolver, "
                            "csrc/bu_basis.hpp slice_lex / slice_resolve) while bands of finished rows are already launched, so the call is shorter than "
                            "the one-thread decode (bu_basislz_decode) timed beside it; phase times: profiles/r04_config4_phase_times_two_thread_slice_decode.txt"}
            if not args.no_cpu:
                from oracle.pyoracle import Oracle  # the checker, timed beside the product (the cpu_baseline leg of this row)
                orc4 = Oracle()
                orc4.read_to("rgba", fone)
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    st4 = orc4.read_to("rgba", fone)[0]
                    times.append(time.perf_counter() - t0)
                cpu4 = sorted(times)[1]
                row4["cpu_baseline"] = {"value": round(512 * 512 / cpu4 / 1e6, 2), "unit": "Mblocks/s", "ms_per_file": round(cpu4 * 1e3, 3), "cores": 1, "kind": "port",
                                        "sample": "the same file through the oracle's read_to_rgba (C restatement of basis.rs:8-77 + basis_lz/mod.rs, one thread), median of 3",
                                        "status": int(st4)}
            extra["etc1s_file_one_512x512_slice_read_to_rgba"] = row4
        except Exception as e:
            extra["etc1s_file_error"] = repr(e)
        # configs 1/2 through the whole-file API: a .basis UASTC file holding the 4096x4096 atlas -> read_to_bc7
        # (header + payload CRC-16 on the host cores, upload, one launch, download)
        try:
            import basisu_rs_amd as bu
            ufile = bu.write_uastc_file([dict(data=host_in, orig_w=4096, orig_h=4096, nbx=1024, nby=1024)])
            pin_file_out = ctx.host_alloc(N_BLOCKS * 16)
            bu.read_to_bc7(ufile, ctx, out=pin_file_out)
            t0 = time.perf_counter()
            for _ in range(5):
                imgs_u = bu.read_to_bc7(ufile, ctx, out=pin_file_out)
            ufile_s = (time.perf_counter() - t0) / 5
            t0 = time.perf_counter()
            for _ in range(5):
                bu.crc16(ufile[77:])
            crc_s = (time.perf_counter() - t0) / 5
            extra["uastc_file_read_to_bc7"] = {"file_bytes": len(ufile), "ms_per_file": round(ufile_s * 1e3, 3), "mblocks_s": round(N_BLOCKS / ufile_s / 1e6, 1),
                                               "ms_payload_crc16": round(crc_s * 1e3, 3),
                                               "verified": bool((np.asarray(imgs_u[0].data).reshape(-1, 16) == golden["bc7"][idx0]).all()),
                                               "note": "whole-file API on a page-locked output buffer: parse + CRC-16 (host cores) + upload + kernel; never the headline value"}
            ctx.host_free(pin_file_out)
            del ufile
        except Exception as e:
            extra["uastc_file_error"] = repr(e)
        # config 3: UASTC -> RGBA32 (16 B in, 64 B out)
        rg_n = min(nbuf, 16)
        rg_out = [torch.empty((N_BLOCKS, 64), dtype=torch.uint8, device=dev) for _ in range(rg_n)]
        rg_in = (ctypes.c_void_p * rg_n)(*[ins[k].data_ptr() for k in range(rg_n)])
        rg_outp = (ctypes.c_void_p * rg_n)(*[t.data_ptr() for t in rg_out])
        rot[0] = 0
        run(rg_n, target=_lib.RGBA32, inp=rg_in, outp=rg_outp, nb=rg_n)
        torch.cuda.synchronize()
        # row-major image (uastc.rs:96): block (by, bx) holds pixel rows 4 by .. 4 by + 3, 16 bytes each at column 16 bx
        rg_ok = bool(torch.equal(rg_out[0].view(NBY, 4, NBX, 16).permute(0, 2, 1, 3).reshape(N_BLOCKS, 64), torch.from_numpy(golden["rgba"]).to(dev)[idxs[0]]))
        ramp(target=_lib.RGBA32, inp=rg_in, outp=rg_outp, nb=rg_n)
        rg_s = row(256, target=_lib.RGBA32, inp=rg_in, outp=rg_outp, nb=rg_n)
        extra["uastc_to_rgba32"] = {"gb_s": round(80 * N_BLOCKS / rg_s / 1e9, 1), "us_per_launch": round(rg_s * 1e6, 3),
                                    "mblocks_s": round(N_BLOCKS / rg_s / 1e6, 1), "bytes_per_block": 80,
                                    "frac_of_hbm_peak": round(80 * N_BLOCKS / rg_s / 1e9 / HBM_PEAK_GBS, 4), "verified": rg_ok,
                                    "note": "BASELINE config 3: 4096x4096 UASTC -> RGBA32 (16 B in + 64 B out per block), cold rotation over 16 atlases, one launch at a time"}
        try:
            sramp(4, True, target=_lib.RGBA32, inp=rg_in, outp=rg_outp, nb=rg_n)
            rg3 = srow(256, 4, True, target=_lib.RGBA32, inp=rg_in, outp=rg_outp, nb=rg_n)
            torch.cuda.synchronize()
            rg3_ok = bool(torch.equal(rg_out[1].view(NBY, 4, NBX, 16).permute(0, 2, 1, 3).reshape(N_BLOCKS, 64), torch.from_numpy(golden["rgba"]).to(dev)[idxs[1]]))
            extra["uastc_to_rgba32"]["in_flight_4_shared"] = {"us_per_atlas": round(rg3 * 1e6, 3), "gb_s": round(80 * N_BLOCKS / rg3 / 1e9, 1),
                                                                 "frac_of_hbm_peak": round(80 * N_BLOCKS / rg3 / 1e9 / HBM_PEAK_GBS, 4), "verified": rg3_ok}
        except Exception as e:
            extra["uastc_to_rgba32"]["in_flight_error"] = repr(e)
        # ... and ONE bu_uastc_transcode_batch_device launch over the 16 atlases in their separate allocations (the multi-run kernel: whole runs of the image's own
        # power-of-two pitch are tiled as 64 x 16-block rectangles, tiles drawn by ticket)
        try:
            SZr = ctypes.c_size_t * rg_n
            rg_nn = SZr(*([N_BLOCKS] * rg_n))
            for o_ in rg_out:
                o_.zero_()

            def rg_batch():
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.RGBA32, rg_n, rg_in, rg_nn, rg_outp, NBX, None, None, sp) == 0

            rg_t0 = time.perf_counter()
            while (time.perf_counter() - rg_t0) * 1e3 < max(args.prewarm_ms, 1.0) * 2:
                rg_batch()
                torch.cuda.synchronize()
            rg_e0, rg_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rg_e0.record(stream)
            for _ in range(12):
                rg_batch()
            rg_e1.record(stream)
            torch.cuda.synchronize()
            rgb_s = rg_e0.elapsed_time(rg_e1) / 1e3 / (12 * rg_n)
            g_rg = torch.from_numpy(golden["rgba"]).to(dev)
            rgb_ok = all(bool(torch.equal(rg_out[k].view(NBY, 4, NBX, 16).permute(0, 2, 1, 3).reshape(N_BLOCKS, 64), g_rg[idxs[k]])) for k in (0, rg_n - 1))
            extra["uastc_to_rgba32"]["all_atlases_one_launch"] = {"atlases_per_launch": rg_n, "us_per_atlas": round(rgb_s * 1e6, 3), "gb_s": round(80 * N_BLOCKS / rgb_s / 1e9, 1),
                                                                  "frac_of_hbm_peak": round(80 * N_BLOCKS / rgb_s / 1e9 / HBM_PEAK_GBS, 4), "verified": rgb_ok}
            del g_rg
        except Exception as e:
            extra["uastc_to_rgba32"]["all_atlases_one_launch_error"] = repr(e)
        del rg_out


    line = {
        "metric": "M 4x4 blocks/s UASTC->BC7 4096x4096",
        "value": round(value, 1),
        "unit": "Mblocks/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt_max / args.steps * 1e3, 6),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "UASTC->BC7, 4096x4096 px (1 048 576 blocks) per GPU per step, ONE launch per step, A-gold atlas "
                               "(block i = reference known-answer block h(i) mod 608, uniform mix of the 19 modes), "
                               "%d distinct atlases rotated (cold cache); %s" % (
                                   nbuf, ("%d launches in flight on %d streams (step i on context stream i %% %d), %s launch policy" % (
                                       args.in_flight, args.in_flight, args.in_flight, args.policy)) if args.in_flight > 1 else "one launch at a time on one stream, %s launch policy" % args.policy),
                   "launches_in_flight": args.in_flight, "launch_policy": args.policy,
                   "effective_streams": effective_streams, "stream_mode": stream_mode,
                   "hip_runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "streams_on_one_hardware_queue_max": queue_sharing,
                                       "note": "bu_context_probe_streams over the streams of the timed region: 1 = every stream has its own hardware queue"},
                   "blocks_per_step_per_gpu": N_BLOCKS, "gb_s_in": round(value * 16 / 1e3, 1),
                   "prewarm": {"launches": prewarm_launches, "ms": args.prewarm_ms,
                               "note": "untimed launches of the same kernel ahead of the W warm-up steps (clock ramp); --prewarm-ms 0 disables"},
                   "timed_region": {"lead_launches": lead, "host_ms": round(host_ms, 6), "event_ms": round(ev_ms, 6), "host_started_late": bool(late),
                                    "repeats": len(wins), "window_reported": "median", "windows_us_per_step": [round(x / args.steps * 1e6, 3) for x in dts],
                                    "streams_of_median_window": wins[m_][4], "streams_of_strict_bracket": strict_streams,
                                    "start_event_spread_us": round(start_spread_us, 1), "streams_in_step": not out_of_step,
                                    "us_per_step_by_stream_rates": by_rates_us,
                                    "tail_launches": 2 * args.in_flight if args.in_flight > 1 else 0,
                                    "note": "barrier + synchronize, then -- everything enqueued up front, step i on stream i % in_flight -- lead untimed launches, a start "
                                            "event per stream behind its last lead launch, K timed launches, an end event per stream behind its last timed launch, two untimed "
                                            "tail launches per stream.  Launches in flight are a pipeline, so the K steps are counted as completions: window = from the LAST "
                                            "start event (every lead launch has completed) to the LAST end event (every timed launch has completed) on the device clock; "
                                            "host clock from every start event seen complete to every end event seen complete; value uses max(host, event).  The pipeline "
                                            "is full at both instants (what the first timed launches got done beside the last lead launches, the tail launches get done "
                                            "beside the last timed ones).  us_per_step_by_stream_rates: the same window read stream by stream -- 1 / sum over the streams of "
                                            "(its timed launches / (its end event - its start event)); streams_in_step: the start events lie within max(in_flight + 4, min(16, K / 2)) periods and no two "
                                            "streams share a hardware queue (bu_context_probe_streams).  --steps 512 gives the same figure with the ends weighing 25 times less"}},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                     "kernel": "bu_uastc_sorted_kernel<BC7> (%s-policy shape)" % args.policy,
                     "launches_in_flight": args.in_flight, "period_ns": round(period_s * 1e9, 1), "us_per_launch": round(period_s * 1e6, 3),
                     "strict_bracket_ns_per_step": round(strict_s * 1e9, 1),
                     "frac_by_strict_bracket": round(BYTES_PER_BLOCK * N_BLOCKS / strict_s / 1e9 / HBM_PEAK_GBS, 4),
                     "bytes_per_launch": BYTES_PER_BLOCK * N_BLOCKS,
                     "note": "achieved = algorithmic bytes per launch / launch-to-launch PERIOD of the K timed launches = (HIP event behind the last timed launch - HIP "
                             "event behind the last lead launch, latest over the streams) / K: K launches complete in that window and the pipeline is full at both ends.  "
                             "With several launches in flight one launch's own span is longer than the period (kernel_span_ns below, from this run's rocprofv3 kernel "
                             "trace) and the chip works on about span / period launches at a time.  strict_bracket_ns_per_step (median of `repeats` passes): K steps with nothing "
                             "launched behind them, first instruction of the first timed launch to last instruction of the last one, / K -- a window that holds "
                             "K + in_flight - 1 periods of a full pipeline and its drain (at --steps 512 the two figures meet)"},
    }
    if out_of_step:
        line["roofline"]["streams_out_of_step"] = (
            "the %d streams of the timed region were NOT in step (start events %.0f us apart, period %.1f us; bu_context_probe_streams found up to %d of them on "
            "one hardware queue): streams share a hardware queue (GPU_MAX_HW_QUEUES=%s; more streams in this process than queues).  The window over the pipelined launches is not reported; value, ms_per_step "
            "and this roofline are the ONE-LAUNCH-AT-A-TIME measurement (extra.one_launch_at_a_time), which assumes nothing about the streams' pace"
            % (args.in_flight, start_spread_us, ev_ms / args.steps * 1e3, queue_sharing, os.environ.get("GPU_MAX_HW_QUEUES")))
        line["roofline"]["launches_in_flight"] = 1
    tr = pmc_traffic()
    if tr:
        line["roofline"]["traffic"] = tr[0]
        line["roofline"]["traffic_source"] = tr[1] + " (committed rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this kernel; not measured in this run: " + env.live_traffic[1] + ")"
    tr3 = env.live_traffic[2] if len(env.live_traffic) > 2 else None
    if tr3:
        line["roofline"]["rocprofv3_this_run"] = tr3
        if tr3.get("kernel_avg_ns"):
            # two clocks, both reported: `frac` above = bytes / (event time of the K timed launches / K); the child trace pass of this run gives
            # the span of one dispatch and the start-to-start period of the same launches under the profiler
            line["roofline"]["kernel_span_ns"] = tr3["kernel_avg_ns"]
            line["roofline"]["period_ns_by_rocprofv3"] = tr3.get("steady_period_ns") or tr3["period_avg_ns"]
            line["roofline"]["frac_by_rocprofv3_period"] = round(BYTES_PER_BLOCK * N_BLOCKS / (tr3.get("steady_period_ns") or tr3["end_to_end_period_avg_ns"]) / HBM_PEAK_GBS, 4)
            # the profiler's stated bias: what it does to the ONE-launch copy kernel of the same size (rocprofv3 kernel stats of the committed bench profile against
            # this run's own unprofiled copy row).  Round 6 tried to take the host out of the profiled loop with a HIP graph; a graph does not run four chains as a
            # pipeline (8.2 us per launch unprofiled), and the product call under rocprofv3 reads 6.16-6.20 us: profiles/r06_rocprofv3_headline_graph_replay.txt
            copy_live = (extra.get("copy_ceiling") or {}).get("us_per_launch")
            line["roofline"]["rocprofv3_stretch_of_the_copy_kernel"] = {
                "profiled_us": 6.44, "unprofiled_us_this_run": copy_live, "unprofiled_us_same_session_as_profile": 5.84, "ratio": 1.10,
                "source": "bu_copy_kernel AverageNs 6 440 in profiles/r05_v10_rocprofv3_kernel_stats.csv against 5.84 us by HIP events in that session",
                "frac_by_rocprofv3_period_less_that_stretch": round(line["roofline"]["frac_by_rocprofv3_period"] * 1.10, 4),
                "note": "the same tool puts +10 % on a single launch of a kernel that does nothing but copy; the pipeline's profiled period is its unprofiled period x 1.10-1.13"}
            if tr3.get("hardware_queues_used", args.in_flight) < args.in_flight:
                line["roofline"]["warning"] = ("the trace pass saw %d hardware queue(s) for %d streams: streams that share a queue run their launches one after another -- "
                                               "GPU_MAX_HW_QUEUES was %r when HIP initialised (INTEGRATION.md section 4e)" % (
                                                   tr3["hardware_queues_used"], args.in_flight, os.environ.get("GPU_MAX_HW_QUEUES")))
            line["roofline"]["rocprofv3_note"] = ("the trace pass is the evidence for the OVERLAP (dispatch spans of several periods, dispatches starting before their predecessor ends, one "
                                                  "hardware queue per stream).  Its period is that of a PROFILED pipeline, which runs slower: rocprofv3 --kernel-trace adds host-side work to "
                                                  "every dispatch (an empty kernel completes once per 6.3-6.6 us under it, a 5 us whole-chip kernel once per 7.8-8.1 us, plain 1.5 / 4.85 us: "
                                                  "profiles/r05_rocprofv3_dispatch_floor_empty_and_5us_kernels.txt), so ONE enqueueing thread sets the pace at 7.5-8.1 us per completion; the trace pass "
                                                  "therefore enqueues from one host thread per stream (bu_time_set_enqueue_threads) and reads 6.3-6.4 us per completion over its steady stretches "
                                                  "(steady_period_ns; frac_by_rocprofv3_period is computed from it) -- still a profiled pipeline, 10-13 % slower than the unprofiled one.  Fed by one thread, the profiler's "
                                                  "completion period equals that run's HIP-event period (9.97 against 10.14 us: profiles/r05_v10_rocprofv3_headline_trace_summary.txt; config 5 with "
                                                  "2^23-block launches, where the profiler's cost does not matter: 175.7 against 174.8 us per array) -- the clocks agree, the profiler perturbs.  "
                                                  "This unprofiled run: HIP events and the host clock agree (timed_region.event_ms / host_ms); --steps 512 repeats it on a 3 ms window")
    if env.live_traffic[0] is not None:
        line["roofline"]["traffic"] = env.live_traffic[0]
        line["roofline"]["traffic_source"] = env.live_traffic[1]
        if tr:
            line["roofline"]["traffic_committed"] = {"bytes": tr[0], "source": tr[1]}
    if tr:
        if tr[2]:
            # the limiter DESIGN.md section 6 measures: vector-ALU instruction issue.  Peak = SIMDs x shader clock / 4.2 clk per
            # wave64 instruction (the issue cost of the slow instruction forms, tools/exp/opbench.hip; the clock is the 2.35 GHz
            # measured in-kernel from s_memtime / s_memrealtime)
            simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
            peak = simds * 2.35e9 / 4.2 / 1e9
            rate = tr[2] / period_s / 1e9
            extra["valu_issue"] = {"wave_instructions_per_launch": int(tr[2]), "achieved_g_per_s": round(rate, 1), "peak_g_per_s": round(peak, 1),
                                   "frac": round(rate / peak, 3), "source": tr[1] + " (SQ_INSTS_VALU of the committed counter pass -- read from that file, NOT measured in this run)"}
    extra["one_launch_at_a_time"] = one_row
    a1_ = extra.get("array512_one_launch")
    a32_ = (env.live_traffic[2] or {}).get("array32") if len(env.live_traffic) > 2 and isinstance(env.live_traffic[2], dict) else None
    if a1_ and a1_.get("verified"):
        # the headline's workload in ONE launch: 32 atlases of 4096^2 contiguous in memory ARE the 2^25-block array of config 5 -- a figure that needs no pipeline,
        # no streams and no period: one kernel's duration (this run: HIP events around 40 launches; rocprofv3's own average from the committed passes over the same launch)
        line["roofline"]["one_launch_of_32_contiguous_atlases"] = {
            "us_per_atlas": round(a1_["us_per_launch"] / 32, 3), "frac": a1_["frac_of_hbm_peak"],
            "us_per_atlas_by_rocprofv3_kernel_avg": round(a1_["kernel_avg_ns"] / 32e3, 3) if a1_.get("kernel_avg_ns") else None,
            "frac_by_rocprofv3_kernel_avg": a1_.get("frac_by_rocprofv3_kernel_avg"), "profile": a1_.get("profile"),
            "fixed_walk_us_per_atlas": round(a1_["fixed_walk_us_per_launch"] / 32, 3) if a1_.get("fixed_walk_us_per_launch") else None,
            "rocprofv3_this_run": a32_,
            "us_per_atlas_by_rocprofv3_kernel_avg_this_run": round(a32_["kernel_avg_ns"] / 32e3, 3) if a32_ and a32_.get("kernel_avg_ns") else None,
            "frac_by_rocprofv3_kernel_avg_this_run": round(BYTES_PER_BLOCK * 32 * N_BLOCKS / a32_["kernel_avg_ns"] / HBM_PEAK_GBS, 4) if a32_ and a32_.get("kernel_avg_ns") else None,
            "note": "extra.array512_one_launch read per atlas: bu_uastc_transcode_device over 2^25 contiguous blocks on the caller's stream, one launch at a time; the persistent "
                    "workgroups draw their tiles by ticket (round 6)"}
    line["extra"] = extra
    allgather = None
    if use_dist:
        # reassembly of the texture array: every rank's 16 MiB BC7 result at rank*16 MiB of a world*16 MiB buffer on every rank
        shard_bytes = N_BLOCKS * 16
        full = RawDeviceBuffer(env, world * shard_bytes)
        full.tensor()[rank * shard_bytes:(rank + 1) * shard_bytes].copy_(outs[0].view(-1))
        torch.cuda.synchronize()
        mine_ref = outs[0].view(-1)

        def verify(ft):  # own slot intact; every other slot non-zero (each rank's atlas differs, so no reference is held here)
            v = ft.view(world, shard_bytes)
            ok = bool(torch.equal(v[rank], mine_ref))
            for r_ in range(world):
                ok = ok and bool(v[r_].any())
            return ok

        dog = GatherWatchdog(env, line)
        allgather = measure_gather(env, full, shard_bytes, verify)
        dog.cancel()
        full.free()

    if allgather:
        line["allgather"] = allgather
    if args.method == "batch":
        try:
            line = batch_headline(env, line, in_ptrs, out_ptrs, nbuf, outs, idxs, status)
        except SystemExit:
            raise
        except Exception as e:  # the pipeline's line stands, and says why
            line["config"]["method"] = "pipeline (the batch step failed: %r)" % (e,)
    if world == 1 and rank == 0 and not args.no_cpu and not args.headline_only:
        line["cpu_baseline"] = cpu_baseline(golden, idx0)
    return line


if __name__ == "__main__":
    main()
