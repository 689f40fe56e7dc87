#!/usr/bin/env python3
"""tools/exp/r6_host.py [matrix] [range] [enqueue] : round-6 host-side measurements on one MI355X (UASTC -> BC7).
  matrix   launch policy {exclusive, shared, auto} x launches in flight {1..4}: us per 2^20-block atlas (streams window, 256 timed launches)
  range    bu_uastc_transcode_device_sync over 2^20..2^25 blocks: one launch (threshold off) against pieces in flight (threshold at 2^20), host clock per call
  enqueue  bu_uastc_transcode_batch_in_flight over 512 atlases in separate allocations, call + synchronize: enqueue threads on / off
Run with GPU_MAX_HW_QUEUES=8 for the pool-stream figures, without it for the CU-mask ones."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from basisu_rs_amd import Context, _lib, synth  # noqa: E402

what = set(sys.argv[1:]) or {"matrix", "range", "enqueue"}
dev = torch.device("cuda", 0)
golden = synth.load_golden(os.path.join(ROOT, "tests", "golden", "uastc_kat.bin"))
g_u, g_b = torch.from_numpy(golden["uastc"]).to(dev), torch.from_numpy(golden["bc7"]).to(dev)
NB = 1 << 20
print("GPU_MAX_HW_QUEUES=%s BU_STREAM_MODE=%s" % (os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("BU_STREAM_MODE")))


def atlases(n):
    idxs = [torch.randint(0, 608, (NB,), device=dev, generator=torch.Generator(device=dev).manual_seed(11 + k)) for k in range(n)]
    return idxs, [g_u[i].contiguous() for i in idxs], [torch.zeros((NB, 16), dtype=torch.uint8, device=dev) for _ in range(n)]


if "matrix" in what:
    ctx = Context(0)
    lib = ctx._lib
    print("in flight:", ctx.query_in_flight(4), "sharing now:", ctx.probe_streams(4))
    nbuf = 64
    idxs, ins, outs = atlases(nbuf)
    PA = ctypes.c_void_p * nbuf
    ip, op = PA(*[t.data_ptr() for t in ins]), PA(*[t.data_ptr() for t in outs])
    status = torch.empty(1, dtype=torch.int64, device=dev)
    ctx.status_word_reset(status)
    torch.cuda.synchronize()
    rot = [0]

    def win(lead, k, nfl):
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        tail = nfl if nfl > 1 else 0
        st = lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, ip, op, nbuf, rot[0] % nbuf, NB, 1024, lead, k, tail, nfl,
                                                       ctypes.c_void_p(status.data_ptr()), ctypes.byref(ev), ctypes.byref(host), None, None)
        assert st == 0
        rot[0] += lead + k + tail
        return max(ev.value, host.value) * 1e3 / k

    for _ in range(20):
        win(0, 256, 4)
    for rnd in range(2):
        for pol in ("exclusive", "shared", "auto"):
            ctx.set_launch_policy({"exclusive": False, "shared": True, "auto": "auto"}[pol])
            row = []
            for nfl in (1, 2, 3, 4):
                win(64, 256, nfl)
                row.append(min(win(512, 256, nfl) for _ in range(3)))
            print("%-10s " % pol + "  ".join("S%d %.2f" % (i + 1, v) for i, v in enumerate(row)))
    torch.cuda.synchronize()
    ok = all(bool(torch.equal(outs[k], g_b[idxs[k]])) for k in range(nbuf))
    print("verified:", ok)
    ctx.close()
    del ins, outs

if "range" in what:
    big = 1 << 25
    idx = torch.randint(0, 608, (big,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    d_in = g_u[idx].contiguous()
    d_out = torch.zeros((big, 16), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    import subprocess

    if "--child" not in sys.argv:  # (the threshold is read once per process: each setting runs in a child)
        del d_in, d_out
        for thr in ("30", "20"):
            env = dict(os.environ, BU_RANGE_IN_FLIGHT_MIN_LOG2=thr)
            print("-- threshold 2^%s (%s)" % (thr, "one launch" if thr == "30" else "pieces in flight"))
            sys.stdout.flush()
            subprocess.run([sys.executable, os.path.abspath(__file__), "range", "--child"], env=env, check=True)
    else:
        ctx = Context(0)
        for lg in (20, 21, 22, 23, 24, 25):
            n = 1 << lg
            reps = max(8, (1 << 27) >> lg)
            # cold rotation inside the 2^25-block buffers
            slots = big // n
            for _ in range(3):
                ctx.transcode_device_sync(_lib.BC7, d_in, n, d_out, blocks_per_row=1024)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for r in range(reps):
                    o = (r % slots) * n
                    assert ctx.transcode_device_sync(_lib.BC7, d_in[o:o + n], n, d_out[o:o + n], blocks_per_row=1024) == _lib.STATUS_WORD_CLEAR
                best = min(best, (time.perf_counter() - t0) / reps * 1e6)
            print("2^%d blocks: %8.2f us per call  %.3f of 8 TB/s" % (lg, best, 32.0 * n / best / 1e6 / 8000.0))
        ok = bool(torch.equal(d_out, g_b[idx]))
        print("verified:", ok)
        ctx.close()

if "enqueue" in what and "--child" not in sys.argv:
    n = 512
    idxs, ins, outs = atlases(64)
    # 512 slices over 64 buffers would alias outputs: use 512 distinct (in, out) views of 8 big allocations instead -> separate runs (gaps between them)
    ins = ins * 8
    outs2 = [torch.zeros((NB, 16), dtype=torch.uint8, device=dev) for _ in range(64)]
    for label, envv in (("enqueue threads (default)", None), ("one enqueue thread", "0")):
        if envv is None:
            os.environ.pop("BU_ENQUEUE_THREADS", None)
        else:
            os.environ["BU_ENQUEUE_THREADS"] = envv
        ctx = Context(0)
        status = torch.empty(1, dtype=torch.int64, device=dev)
        ctx.status_word_reset(status)
        torch.cuda.synchronize()
        outs_all = (outs + outs2) * 4
        ins_all = ins[:128] * 4

        VP, SZ = ctypes.c_void_p * n, ctypes.c_size_t * n
        a_in, a_n, a_out = VP(*[t.data_ptr() for t in ins_all]), SZ(*([NB] * n)), VP(*[t.data_ptr() for t in outs_all])
        sp = ctypes.c_void_p(status.data_ptr())

        def run():  # (the argument arrays are built once: the call, not Python, is what is timed)
            assert ctx._lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, n, a_in, a_n, a_out, 1024, None, sp, 4) == 0
            assert ctx._lib.bu_context_synchronize(ctx.handle) == 0

        for _ in range(4):
            run()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            run()
            ts.append((time.perf_counter() - t0) / n * 1e6)
        ok = all(bool(torch.equal(outs[k], g_b[idxs[k]])) for k in range(64))
        print("%-28s %.3f us per atlas (median of 7; min %.3f)  verified %s  in flight %s" % (label, sorted(ts)[3], min(ts), ok, ctx.query_in_flight(4)))
        ctx.close()

if "pw" in what:
    # the same 2^25-block launches, four in flight, shared shapes: (a) the measurement helper's window (bu_uastc_transcode_device per launch), (b) the product's
    # pipelined entry point (three calls of bu_uastc_transcode_batch_in_flight, timing marks between them) -- alternating, one process
    ctx = Context(0)
    lib = ctx._lib
    big, nbuf = 1 << 25, int(os.environ.get("PW_NBUF", "8"))
    idx0 = torch.randint(0, 608, (big,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    ins = [g_u[idx0].contiguous()]
    ins += [torch.roll(ins[0], shifts=977 * (k + 1), dims=0).contiguous() for k in range(nbuf - 1)]
    outs = [torch.zeros((big, 16), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    status = torch.empty(1, dtype=torch.int64, device=dev)
    ctx.status_word_reset(status)
    torch.cuda.synchronize()
    PA = ctypes.c_void_p * nbuf
    ip, op = PA(*[t.data_ptr() for t in ins]), PA(*[t.data_ptr() for t in outs])
    sp = ctypes.c_void_p(status.data_ptr())
    bpr = int(os.environ.get("PW_BPR", "256"))

    def helper(lead, k):
        ctx.set_launch_policy(True)
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        assert lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, ip, op, nbuf, 0, big, bpr, lead, k, 8 if lead else 0, 4, sp, ctypes.byref(ev), ctypes.byref(host), None, None) == 0
        return ev.value * 1e3 / k

    def args_for(first, steps):
        VP, SZ = ctypes.c_void_p * steps, ctypes.c_size_t * steps
        return steps, VP(*[ins[(first + i) % nbuf].data_ptr() for i in range(steps)]), SZ(*([big] * steps)), VP(*[outs[(first + i) % nbuf].data_ptr() for i in range(steps)])

    def product(lead, k):
        ctx.set_launch_policy("auto")
        parts = [args_for(0, lead) if lead else None, args_for(lead, k), args_for(lead + k, 8) if lead else None]
        call = lambda a: lib.bu_uastc_transcode_batch_in_flight(ctx.handle, _lib.BC7, a[0], a[1], a[2], a[3], bpr, None, sp, 4)  # noqa: E731
        if parts[0]:
            assert call(parts[0]) == 0
        assert lib.bu_time_mark_streams(ctx.handle, 4, 0) == 0
        assert call(parts[1]) == 0
        assert lib.bu_time_mark_streams(ctx.handle, 4, 1) == 0
        if parts[2]:
            assert call(parts[2]) == 0
        ev, strict, host = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
        assert lib.bu_time_marks_elapsed(ctx.handle, 4, ctypes.byref(ev), ctypes.byref(strict), ctypes.byref(host)) == 0
        ctx.synchronize()
        return ev.value * 1e3 / k

    for _ in range(4):
        helper(0, 16)
    for rnd in range(3):
        print("round %d: helper window %.2f us per array   product calls %.2f   helper %.2f   product %.2f" % (rnd, helper(24, 40), product(24, 40), helper(24, 40), product(24, 40)))
    torch.cuda.synchronize()
    print("verified:", bool(torch.equal(outs[0], g_b[idx0])))
    ctx.close()

if "one" in what:
    # ONE exclusive 2^25-block launch at a time on context stream 0 (tile tickets unless BU_TILE_TICKETS=0), PW_NBUF rotated pairs, PW_BPR blocks per row
    ctx = Context(0)
    lib = ctx._lib
    big, nbuf = 1 << 25, int(os.environ.get("PW_NBUF", "8"))
    bpr = int(os.environ.get("PW_BPR", "256"))
    idx0 = torch.randint(0, 608, (big,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    ins = [g_u[idx0].contiguous()]
    ins += [torch.roll(ins[0], shifts=977 * (k + 1), dims=0).contiguous() for k in range(nbuf - 1)]
    outs = [torch.zeros((big, 16), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    status = torch.empty(1, dtype=torch.int64, device=dev)
    ctx.status_word_reset(status)
    torch.cuda.synchronize()
    PA = ctypes.c_void_p * nbuf
    ip, op = PA(*[t.data_ptr() for t in ins]), PA(*[t.data_ptr() for t in outs])
    sp = ctypes.c_void_p(status.data_ptr())
    ctx.set_launch_policy(False)

    def one(lead, k, st=sp):
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        assert lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, ip, op, nbuf, 0, big, bpr, lead, k, 0, 1, st, ctypes.byref(ev), ctypes.byref(host), None, None) == 0
        return ev.value * 1e3 / k

    one(0, 24)
    print("nbuf %d bpr %d tickets %s: with status word %.2f %.2f   without %.2f %.2f us per launch" % (nbuf, bpr, os.environ.get("BU_TILE_TICKETS", "1"), one(8, 40), one(8, 40), one(8, 40, None), one(8, 40, None)))
    torch.cuda.synchronize()
    print("verified:", bool(torch.equal(outs[0], g_b[idx0])))
    ctx.close()

if "picks" in what:
    # what BU_LAUNCH_AUTO picks with 1..4 launches in flight, window shapes of bench.py's matrix (lead 64, 256 timed, tail = in flight)
    ctx = Context(0)
    lib = ctx._lib
    nbuf = 64
    idxs, ins, outs = atlases(nbuf)
    PA = ctypes.c_void_p * nbuf
    ip, op = PA(*[t.data_ptr() for t in ins]), PA(*[t.data_ptr() for t in outs])
    torch.cuda.synchronize()
    ctx.set_launch_policy("auto")
    for nfl in (1, 2, 3, 4, 4, 4):
        c0 = (ctypes.c_ulonglong * 3)(); lib.bu_time_auto_policy_counts(ctx.handle, c0)
        ev, host = ctypes.c_float(0), ctypes.c_float(0)
        assert lib.bu_time_uastc_launches_streams_window(ctx.handle, _lib.BC7, ip, op, nbuf, 0, NB, 1024, 64, 256, nfl if nfl > 1 else 0, nfl, None, ctypes.byref(ev), ctypes.byref(host), None, None) == 0
        c1 = (ctypes.c_ulonglong * 3)(); lib.bu_time_auto_policy_counts(ctx.handle, c1)
        ms_, k_ = ctypes.c_float(0), ctypes.c_int(0)
        lib.bu_time_last_window_enqueue(ctx.handle, ctypes.byref(ms_), ctypes.byref(k_))
        print("S%d: %.2f us per atlas; picks exclusive / one-tile / shared = %s; host enqueue %.2f us per launch" % (nfl, max(ev.value, host.value) * 1e3 / 256, [int(c1[i] - c0[i]) for i in range(3)], ms_.value * 1e3 / max(k_.value, 1)))
    ctx.close()

if "multi" in what:
    # ONE call of bu_uastc_transcode_batch_device over N slices of 2^20 blocks in separate allocations (a persistent grid walks all runs' tiles), tickets off / on
    ctx = Context(0)
    lib = ctx._lib
    nbuf = 64
    idxs, ins, outs = atlases(nbuf)
    s = torch.cuda.Stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    ctx.set_launch_policy(False)
    for n in (8, 16, 32, 64):
        VP, SZ = ctypes.c_void_p * n, ctypes.c_size_t * n
        a_in, a_n, a_out = VP(*[ins[k].data_ptr() for k in range(n)]), SZ(*([NB] * n)), VP(*[outs[k].data_ptr() for k in range(n)])
        res = []
        for tk in (0, 1, 0, 1):
            lib.bu_time_set_tile_tickets(ctx.handle, tk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, n, a_in, a_n, a_out, 1024, None, None, sp) == 0
            reps = max(4, 256 // n)
            e0.record(s)
            for _ in range(reps):
                assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, n, a_in, a_n, a_out, 1024, None, None, sp) == 0
            e1.record(s)
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / reps / n)
        ok = all(bool(torch.equal(outs[k], g_b[idxs[k]])) for k in range(n))
        print("%2d slices per call: fixed walk %.2f %.2f   tickets %.2f %.2f us per slice   verified %s" % (n, res[0], res[2], res[1], res[3], ok))
    ctx.close()

if "multi2" in what:
    # 32 atlases per launch: (a) contiguous in one allocation -- the plain persistent launch (bu_uastc_transcode_device), (b) in 32 separate allocations -- ONE multi-run launch
    # (bu_uastc_transcode_batch_device); exclusive policy, tickets on; events around 24 launches after a long warm-up, alternating
    ctx = Context(0)
    lib = ctx._lib
    ctx.set_launch_policy(False)
    n = 32
    idxs, ins, outs = atlases(64)
    cat_in = [torch.cat(ins[:32]).contiguous(), torch.cat(ins[32:]).contiguous()]
    cat_out = [torch.zeros((32 * NB, 16), dtype=torch.uint8, device=dev) for _ in range(2)]
    s = torch.cuda.Stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    VP, SZ = ctypes.c_void_p * n, ctypes.c_size_t * n
    sets = [(VP(*[ins[h * 32 + k].data_ptr() for k in range(n)]), SZ(*([NB] * n)), VP(*[outs[h * 32 + k].data_ptr() for k in range(n)])) for h in range(2)]

    def contiguous(k):
        assert lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(cat_in[k % 2].data_ptr()), 32 * NB, ctypes.c_void_p(cat_out[k % 2].data_ptr()), 1024, 0, None, sp) == 0

    def separate(k):
        a = sets[k % 2]
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, n, a[0], a[1], a[2], 1024, None, None, sp) == 0

    def timed(fn, reps=24):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(8):
            fn(k)
        e0.record(s)
        for k in range(reps):
            fn(k)
        e1.record(s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps / n

    # (c) the multi-run launch over the CONTIGUOUS buffers: 32 slices that would merge, kept apart by gaps in the block numbering -- the kernel's share of the difference
    U64 = ctypes.c_uint64 * n
    gaps = U64(*[k * 2 * NB for k in range(n)])
    csets = [(VP(*[cat_in[h].data_ptr() + k * NB * 16 for k in range(n)]), SZ(*([NB] * n)), VP(*[cat_out[h].data_ptr() + k * NB * 16 for k in range(n)])) for h in range(2)]

    def separate_runs_contiguous_memory(k):
        a = csets[k % 2]
        assert lib.bu_uastc_transcode_batch_device(ctx.handle, _lib.BC7, n, a[0], a[1], a[2], 1024, gaps, None, sp) == 0

    def contiguous_strips(k):  # the plain launch without a block grid: tiles are strips of 1024 consecutive blocks, as in the multi-run kernel
        assert lib.bu_uastc_transcode_device(ctx.handle, _lib.BC7, ctypes.c_void_p(cat_in[k % 2].data_ptr()), 32 * NB, ctypes.c_void_p(cat_out[k % 2].data_ptr()), 0, 0, None, sp) == 0

    for _ in range(40):
        contiguous(_)
    torch.cuda.synchronize()
    for rnd in range(3):
        print("round %d: contiguous %.3f us per atlas (strips: %.3f)   32 separate allocations, one launch %.3f   32 runs in contiguous memory, one launch %.3f" % (
            rnd, timed(contiguous), timed(contiguous_strips), timed(separate), timed(separate_runs_contiguous_memory)))
    ok = all(bool(torch.equal(outs[k], g_b[idxs[k]])) for k in range(64)) and bool(torch.equal(cat_out[0][:NB], g_b[idxs[0]]))
    print("verified:", ok)
    ctx.close()
