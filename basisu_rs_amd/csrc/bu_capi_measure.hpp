// C ABI, measurement helpers of bench.py: copy ceiling and back-to-back timed launches (hipEvents on the launch stream).
// Part of the single translation unit bu_hip.hip.
#pragma once
extern "C" {

// ---- measurement helpers ---------------------------------------------------------------------------
bu_status bu_copy_ceiling_device(bu_context* ctx, const void* d_in, size_t n_blocks, void* d_out, void* stream)
{
    if (!ctx || (n_blocks && (!d_in || !d_out))) return BU_ERR_ARGUMENT;
    if (n_blocks == 0) return BU_OK;
    if (n_blocks >= ((size_t)1 << 22))
        hipLaunchKernelGGL((bu_copy_kernel<1024, 1>), dim3((unsigned)((n_blocks + 1023) / 1024)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint4*>(d_in), static_cast<uint4*>(d_out), n_blocks);
    else
        hipLaunchKernelGGL((bu_copy_kernel<256, 4>), dim3((unsigned)((n_blocks + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint4*>(d_in), static_cast<uint4*>(d_out), n_blocks);
    BU_HIP(ctx, hipGetLastError());
    return BU_OK;
}

// spin until `ev` has completed, at most BU_SPIN_SECONDS: a GPU hang must not become an endless 100 % CPU loop
constexpr double BU_SPIN_SECONDS = 120.0;
static bu_status bu_spin_event(bu_context* ctx, hipEvent_t ev, std::chrono::steady_clock::time_point* when)
{
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(BU_SPIN_SECONDS);
    for (unsigned n = 0;; n++) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return bu_fail(ctx, q, "hipEventQuery");
        if ((n & 1023u) == 1023u && std::chrono::steady_clock::now() > deadline) {
            snprintf(ctx->err, sizeof(ctx->err), "hipEventQuery: event still pending after %.0f s", BU_SPIN_SECONDS);
            return BU_ERR_HIP;
        }
    }
    if (when) *when = std::chrono::steady_clock::now();
    (void)hipGetLastError();
    return BU_OK;
}

bu_status bu_time_uastc_launches(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                 size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int launches, uint64_t* d_status, void* stream,
                                 float* out_ms)
{
    if (!ctx || !d_in || !d_out || n_buffers == 0 || launches <= 0 || !out_ms) return BU_ERR_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BU_HIP(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < launches; i++) {
        const size_t k = (first_buffer + (size_t)i) % n_buffers;
        bu_status st = bu_uastc_transcode_device(ctx, target, d_in[k], n_blocks, d_out[k], blocks_per_row, 0, d_status, stream);
        if (st) return st;
    }
    BU_HIP(ctx, hipEventRecord(ctx->ev1, s));
    // poll instead of a blocking wait: the caller's wall clock around this call should not carry the tens of microseconds a
    // sleeping host thread needs to be woken up -- they are as long as several steps
    bu_status st = bu_spin_event(ctx, ctx->ev1, nullptr);
    if (st) return st;
    BU_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return BU_OK;
}

// The timed region of bench.py on a GPU that never went idle: `lead` untimed launches, event 0, exactly `launches` timed
// launches, event 1 -- all enqueued back to back on `stream` with no host synchronisation in between (rotation continuing
// through both parts).  The host then watches the two events: host_ms = steady_clock time between "event 0 has completed"
// and "event 1 has completed" as seen by hipEventQuery polling, i.e. a wall-clock bracket around exactly the timed
// launches; event_ms = hipEventElapsedTime of the same pair.  *late = 1 when event 0 had already completed at the first
// query (the host was still enqueueing when the GPU got there: host_ms then starts late and the caller must use
// max(host_ms, event_ms)).
bu_status bu_time_uastc_launches_window(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                        size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int lead, int launches,
                                        uint64_t* d_status, void* stream, float* out_event_ms, float* out_host_ms, int* out_late)
{
    if (!ctx || !d_in || !d_out || n_buffers == 0 || launches <= 0 || lead < 0 || !out_event_ms || !out_host_ms) return BU_ERR_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < lead + launches; i++) {
        if (i == lead) BU_HIP(ctx, hipEventRecord(ctx->ev0, s));
        const size_t k = (first_buffer + (size_t)i) % n_buffers;
        bu_status st = bu_uastc_transcode_device(ctx, target, d_in[k], n_blocks, d_out[k], blocks_per_row, 0, d_status, stream);
        if (st) return st;
    }
    BU_HIP(ctx, hipEventRecord(ctx->ev1, s));
    const hipError_t first = hipEventQuery(ctx->ev0);
    if (first != hipSuccess && first != hipErrorNotReady) return bu_fail(ctx, first, "hipEventQuery");
    (void)hipGetLastError();
    if (out_late) *out_late = first == hipSuccess ? 1 : 0;
    std::chrono::steady_clock::time_point t0, t1;
    bu_status st = bu_spin_event(ctx, ctx->ev0, &t0);
    if (st) return st;
    st = bu_spin_event(ctx, ctx->ev1, &t1);
    if (st) return st;
    *out_host_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    BU_HIP(ctx, hipEventElapsedTime(out_event_ms, ctx->ev0, ctx->ev1));
    return BU_OK;
}

bu_status bu_time_uastc_launches_each(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                      size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int launches, uint64_t* d_status, void* stream,
                                      float* out_us)
{
    if (!ctx || !d_in || !d_out || n_buffers == 0 || launches <= 0 || !out_us) return BU_ERR_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<hipEvent_t> ev((size_t)launches + 1, nullptr);
    bu_status ret = BU_OK;
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) ret = BU_ERR_HIP;
    if (ret == BU_OK) {
        (void)hipEventRecord(ev[0], s);
        for (int i = 0; i < launches && ret == BU_OK; i++) {
            const size_t k = (first_buffer + (size_t)i) % n_buffers;
            ret = bu_uastc_transcode_device(ctx, target, d_in[k], n_blocks, d_out[k], blocks_per_row, 0, d_status, stream);
            if (hipEventRecord(ev[(size_t)i + 1], s) != hipSuccess) ret = BU_ERR_HIP;
        }
        if (hipStreamSynchronize(s) != hipSuccess) ret = BU_ERR_HIP;
        for (int i = 0; i < launches && ret == BU_OK; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]) != hipSuccess) ret = BU_ERR_HIP;
            out_us[i] = ms * 1000.0f;
        }
    }
    for (auto e : ev)
        if (e) (void)hipEventDestroy(e);
    return ret;
}

bu_status bu_time_uastc_launches_streams(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                         size_t n_blocks, size_t blocks_per_row, int launches, int n_streams, float* out_ms)
{
    if (!ctx || !d_in || !d_out || n_buffers == 0 || launches <= 0 || !out_ms || n_streams < 1 || n_streams > 8) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    {
        const bu_status sst = bu_ctx_streams(ctx, n_streams);
        if (sst) return sst;
    }
    BU_HIP(ctx, hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; i++) {
        const size_t k = (size_t)i % n_buffers;
        bu_status st = bu_uastc_transcode_device(ctx, target, d_in[k], n_blocks, d_out[k], blocks_per_row, 0, nullptr, ctx->extra_streams[i % n_streams]);
        if (st) return st;
    }
    BU_HIP(ctx, hipDeviceSynchronize());
    *out_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return BU_OK;
}

// The window of bu_time_uastc_launches_window with SEVERAL launches in flight: launch i (lead, timed and tail alike) goes to context
// stream i % n_streams, everything is enqueued up front with no host synchronisation in between.  n launches in flight are a pipeline:
// launch j starts about one period after launch j - 1 and is under way for about n periods, so "first instruction of the first timed
// launch to last instruction of the last" spans launches + n - 1 periods, whatever runs before and behind.  Throughput is counted the
// way a pipeline's is, in COMPLETIONS: every stream gets a start event behind its last lead launch (= in front of its first timed
// launch) and an end event behind its last timed launch, and
//     event_ms = (latest end event) - (latest start event)
// on the device's clock (each measured from one reference event at the head of the call, which every stream waits for before its
// first launch): from the moment the LAST LEAD launch has completed to the moment the LAST TIMED launch has completed -- in between,
// exactly the `launches` timed launches complete.  With lead >= n_streams and tail >= n_streams (untimed launches behind the end
// events, one per stream) the pipeline is full at both instants: the part of the first timed launches that was done before the window
// opened (they run beside the last lead launches) is what the tail launches get done before it closes, so event_ms / launches is the
// steady-state period.  (Without tail launches that balance is gone: the head start of the first timed launches is credited, nothing is
// debited, and the figure comes out too SHORT -- use the strict bracket below for a run with nothing behind it.)
// *out_fill_drain_ms (optional) = latest end event - EARLIEST start event: first instruction of the first timed launch to last
// instruction of the last one, the strict bracket around the timed launches; launches + n_streams - 1 periods in a full pipeline, and
// with lead = tail = 0 the whole run from an idle chip to an idle chip.
// host_ms = steady clock from "every start event seen complete" to "every end event seen complete".
// n_streams == 1, tail == 0 is bu_time_uastc_launches_window on a context stream.
}  // extern "C"
// the per-stream window events.  Timing-only: no system-scope fence when they complete (hipEventDisableSystemFence) -- a default event writes the
// caches back and invalidates them, under the launches that are running beside it on the other streams
static bu_status bu_window_events(bu_context* ctx, int n_streams)
{
    for (int i = 0; i < n_streams; i++) {
        if (!ctx->ev_start[i]) BU_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_start[i], hipEventDisableSystemFence));
        if (!ctx->ev_end[i]) BU_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_end[i], hipEventDisableSystemFence));
    }
    return BU_OK;
}

template <class LAUNCH>  // bu_status launch(int i, hipStream_t s): enqueue launch number i (lead, timed and tail launches are numbered through) on s
static bu_status bu_streams_window(bu_context* ctx, int lead, int launches, int tail, int n_streams, float* out_event_ms, float* out_host_ms,
                                   float* out_fill_drain_ms, int* out_late, LAUNCH launch)
{
    if (!ctx || launches <= 0 || lead < 0 || tail < 0 || !out_event_ms || !out_host_ms || n_streams < 1 || n_streams > 8) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    {
        const bu_status sst = bu_ctx_streams(ctx, n_streams);
        if (sst) return sst;
    }
    // (ev0, the per-stream events and the win_* fields are the context's: a whole-file or host-pointer call on the same context from another thread
    //  re-records ev0 -- one window at a time, and not beside those calls)
    std::lock_guard<std::mutex> window_guard(ctx->lock);
    {
        const bu_status est = bu_window_events(ctx, n_streams);
        if (est) return est;
    }
    BuDrain drain(ctx);
    BU_HIP(ctx, hipEventRecord(ctx->ev0, ctx->extra_streams[0]));  // the reference point of every time below
    for (int i = 1; i < n_streams; i++) BU_HIP(ctx, hipStreamWaitEvent(ctx->extra_streams[i], ctx->ev0, 0));
    bool used[8] = {false, false, false, false, false, false, false, false};
    int last_timed[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // the number of each stream's last timed launch
    int first_timed[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (int i = lead + launches - 1; i >= lead; i--) first_timed[i % n_streams] = i;
    for (int i = lead; i < lead + launches; i++) last_timed[i % n_streams] = i;
    for (int i = 0; i < n_streams; i++) used[i] = first_timed[i] >= 0;
    const auto enqueue = [&](int i) -> bu_status {  // launch number i with the events that belong in front of and behind it
        const int si = i % n_streams;
        hipStream_t s = ctx->extra_streams[si];
        if (i == first_timed[si]) BU_HIP(ctx, hipEventRecord(ctx->ev_start[si], s));
        const bu_status st = launch(i, s);
        if (st) return st;
        if (i == last_timed[si]) BU_HIP(ctx, hipEventRecord(ctx->ev_end[si], s));
        return BU_OK;
    };
    const int total = lead + launches + tail;
    const auto enq0 = std::chrono::steady_clock::now();
    if (ctx->time_enqueue_threads.load(std::memory_order_relaxed) && n_streams > 1) {
        // one host thread per stream (bu_time_set_enqueue_threads): the order inside every stream is the one below, the order between
        // streams is whatever the threads make it -- the way a caller with one thread per stream drives the context
        bu_status sts[8] = {BU_OK, BU_OK, BU_OK, BU_OK, BU_OK, BU_OK, BU_OK, BU_OK};
        std::vector<std::thread> th;
        bool spawned = true;
        for (int si = 0; si < n_streams && spawned; si++) {
            try {
                th.emplace_back([&, si] {
                    if (hipSetDevice(ctx->device) != hipSuccess) {
                        sts[si] = BU_ERR_HIP;
                        return;
                    }
                    for (int i = si; i < total && sts[si] == BU_OK; i += n_streams) sts[si] = enqueue(i);
                });
            } catch (const std::exception&) {  // (no thread to be had: what is already enqueued is drained by `drain`)
                spawned = false;
            }
        }
        for (auto& t : th) t.join();
        if (!spawned) {
            std::lock_guard<std::mutex> eg(ctx->err_lock);
            snprintf(ctx->err, sizeof(ctx->err), "bu_streams_window: could not start an enqueue thread");
            return BU_ERR_HIP;
        }
        for (int si = 0; si < n_streams; si++)
            if (sts[si]) return sts[si];
    } else {
        for (int i = 0; i < total; i++) {
            const bu_status st = enqueue(i);
            if (st) return st;
        }
    }
    ctx->win_enqueue_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - enq0).count();
    ctx->win_enqueued = total;
    // host bracket: from every start event seen complete to every end event seen complete
    std::chrono::steady_clock::time_point t0, t1, t;
    bool first_query = true;
    for (int i = 0; i < n_streams; i++) {
        if (!used[i]) continue;
        if (first_query) {
            const hipError_t q = hipEventQuery(ctx->ev_start[i]);
            if (q != hipSuccess && q != hipErrorNotReady) return bu_fail(ctx, q, "hipEventQuery");
            (void)hipGetLastError();
            if (out_late) *out_late = q == hipSuccess ? 1 : 0;
            first_query = false;
        }
        bu_status st = bu_spin_event(ctx, ctx->ev_start[i], &t);
        if (st) return st;
        if (i == 0 || t > t0) t0 = t;
    }
    t1 = t0;
    float first_start = 0, last_start = 0, last_end = 0;
    bool any = false;
    ctx->win_streams = n_streams;
    for (int i = 0; i < n_streams; i++) {
        ctx->win_start_ms[i] = ctx->win_end_ms[i] = -1.0f;
        if (!used[i]) continue;
        bu_status st = bu_spin_event(ctx, ctx->ev_end[i], &t);
        if (st) return st;
        if (t > t1) t1 = t;
        float ms_s = 0, ms_e = 0;
        BU_HIP(ctx, hipEventElapsedTime(&ms_s, ctx->ev0, ctx->ev_start[i]));
        BU_HIP(ctx, hipEventElapsedTime(&ms_e, ctx->ev0, ctx->ev_end[i]));
        ctx->win_start_ms[i] = ms_s;
        ctx->win_end_ms[i] = ms_e;
        if (!any || ms_s < first_start) first_start = ms_s;
        if (!any || ms_s > last_start) last_start = ms_s;
        if (!any || ms_e > last_end) last_end = ms_e;
        any = true;
    }
    // (tail launches, and lead launches on streams that carry no timed launch, finish before we return)
    for (int i = 0; i < n_streams; i++) BU_HIP(ctx, hipStreamSynchronize(ctx->extra_streams[i]));
    drain.armed = false;
    *out_host_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    *out_event_ms = last_end - last_start;
    if (out_fill_drain_ms) *out_fill_drain_ms = last_end - first_start;
    return BU_OK;
}

extern "C" {

// on != 0: the streams windows below enqueue from one host thread per stream instead of from the calling thread alone.  The GPU side
// is the same; what changes is how much host time one enqueue may take before the host, not the chip, sets the pace -- under
// rocprofv3 --kernel-trace an enqueue costs 6-8 us of host time (profiles/r05_rocprofv3_dispatch_floor_*), more than the period.
// the per-stream events of the LAST streams window of this context, in ms from the head of that call: out_start_ms[i] = stream i's start
// event (behind its last lead launch), out_end_ms[i] = its end event (behind its last timed launch), -1 for a stream without timed
// launches; both arrays hold 8 floats.  Streams that run in step start and end within a few periods of each other; streams that share a
// hardware queue with something else fall behind, and "latest start to latest end" then no longer brackets `launches` completions.
// Window events around PRODUCT calls (bench.py times bu_uastc_transcode_batch_in_flight with them): which = 0 records the context's start
// event, which = 1 its end event, on every own stream 0..n_streams-1, behind whatever the caller enqueued there so far.  The pattern is that of
// bu_streams_window: a call that enqueues lead work, mark 0, the call that enqueues the timed work, mark 1, a call that enqueues tail work,
// bu_time_marks_elapsed.  Only enqueues.
bu_status bu_time_mark_streams(bu_context* ctx, int n_streams, int which)
{
    if (!ctx || n_streams < 1 || n_streams > 8 || (which != 0 && which != 1)) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    bu_status st = bu_ctx_streams(ctx, n_streams);
    if (st) return st;
    std::lock_guard<std::mutex> g(ctx->lock);
    st = bu_window_events(ctx, n_streams);
    if (st) return st;
    for (int i = 0; i < n_streams; i++) BU_HIP(ctx, hipEventRecord(which ? ctx->ev_end[i] : ctx->ev_start[i], ctx->extra_streams[i]));
    return BU_OK;
}

// waits for both marks of streams 0..n_streams-1 (polling), then: *out_event_ms = latest end event - LATEST start event (the timed work's
// completions in a pipeline that is full at both instants), *out_strict_ms (optional) = latest end - EARLIEST start (first instruction of the
// timed work to its last), *out_host_ms (optional) = host clock from "every start event seen complete" to "every end event seen complete" (valid
// when the caller came here before the start events completed).  The per-stream times are left for bu_time_last_window_streams (ms from the earliest start).
bu_status bu_time_marks_elapsed(bu_context* ctx, int n_streams, float* out_event_ms, float* out_strict_ms, float* out_host_ms)
{
    if (!ctx || n_streams < 1 || n_streams > 8 || !out_event_ms) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> g(ctx->lock);
    for (int i = 0; i < n_streams; i++)
        if (!ctx->ev_start[i] || !ctx->ev_end[i]) return BU_ERR_ARGUMENT;
    std::chrono::steady_clock::time_point t0, t1, t;
    for (int i = 0; i < n_streams; i++) {
        const bu_status st = bu_spin_event(ctx, ctx->ev_start[i], &t);
        if (st) return st;
        if (i == 0 || t > t0) t0 = t;
    }
    t1 = t0;
    for (int i = 0; i < n_streams; i++) {
        const bu_status st = bu_spin_event(ctx, ctx->ev_end[i], &t);
        if (st) return st;
        if (t > t1) t1 = t;
    }
    float s[8], e[8], s_min = 0, s_max = 0, e_max = 0;
    for (int i = 0; i < n_streams; i++) {
        BU_HIP(ctx, hipEventElapsedTime(&s[i], ctx->ev_start[0], ctx->ev_start[i]));
        BU_HIP(ctx, hipEventElapsedTime(&e[i], ctx->ev_start[0], ctx->ev_end[i]));
        if (i == 0 || s[i] < s_min) s_min = s[i];
        if (i == 0 || s[i] > s_max) s_max = s[i];
        if (i == 0 || e[i] > e_max) e_max = e[i];
    }
    ctx->win_streams = n_streams;
    for (int i = 0; i < n_streams; i++) {
        ctx->win_start_ms[i] = s[i] - s_min;
        ctx->win_end_ms[i] = e[i] - s_min;
    }
    *out_event_ms = e_max - s_max;
    if (out_strict_ms) *out_strict_ms = e_max - s_min;
    if (out_host_ms) *out_host_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    return BU_OK;
}

bu_status bu_time_last_window_streams(bu_context* ctx, float* out_start_ms, float* out_end_ms, int* out_n_streams)
{
    if (!ctx || !out_start_ms || !out_end_ms || !out_n_streams) return BU_ERR_ARGUMENT;
    for (int i = 0; i < 8; i++) {
        out_start_ms[i] = i < ctx->win_streams ? ctx->win_start_ms[i] : -1.0f;
        out_end_ms[i] = i < ctx->win_streams ? ctx->win_end_ms[i] : -1.0f;
    }
    *out_n_streams = ctx->win_streams;
    return BU_OK;
}

// host time the context's LAST streams window spent enqueueing (launches and events, before it started to wait) and how many launches that
// was: *out_ms / *out_launches is an upper bound of what one enqueue costs this host (waits for space in a full hardware queue are inside) -- at
// or above the pipeline's period the host, not the chip, may be setting the pace
bu_status bu_time_last_window_enqueue(bu_context* ctx, float* out_ms, int* out_launches)
{
    if (!ctx || !out_ms || !out_launches) return BU_ERR_ARGUMENT;
    *out_ms = ctx->win_enqueue_ms;
    *out_launches = ctx->win_enqueued;
    return BU_OK;
}

// what BU_LAUNCH_AUTO has chosen for this context's large launches since it was created: out[0] exclusive (no other own stream busy), out[1] the shared
// kernels on one-tile workgroups (one or two busy), out[2] the shared shape (three or more)
bu_status bu_time_auto_policy_counts(bu_context* ctx, unsigned long long out[3])
{
    if (!ctx || !out) return BU_ERR_ARGUMENT;
    for (int i = 0; i < 3; i++) out[i] = ctx->auto_picks[i].load(std::memory_order_relaxed);
    return BU_OK;
}

// on == 0: every persistent launch of this context walks fixed shares of the tiles (what rounds 1-5 shipped); on != 0 (default): long walks draw their
// tiles by ticket.  Measurement only -- the bench shows both forms of the 2^25-block launch in one process; results never depend on it.
bu_status bu_time_set_tile_tickets(bu_context* ctx, int on)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    ctx->tickets_off.store(on == 0, std::memory_order_relaxed);
    return BU_OK;
}

bu_status bu_time_set_enqueue_threads(bu_context* ctx, int on)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    ctx->time_enqueue_threads.store(on ? 1 : 0, std::memory_order_relaxed);
    return BU_OK;
}

bu_status bu_time_uastc_launches_streams_window(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                                size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int lead, int launches, int tail,
                                                int n_streams, uint64_t* d_status, float* out_event_ms, float* out_host_ms, float* out_fill_drain_ms,
                                                int* out_late)
{
    if (!d_in || !d_out || n_buffers == 0) return BU_ERR_ARGUMENT;
    return bu_streams_window(ctx, lead, launches, tail, n_streams, out_event_ms, out_host_ms, out_fill_drain_ms, out_late, [&](int i, hipStream_t s) {
        const size_t k = (first_buffer + (size_t)i) % n_buffers;
        // (target BU_TIME_COPY_CEILING: the uint4 -> uint4 copy kernel instead of a transcode -- the HBM ceiling of the same pipeline)
        return (int)target == BU_TIME_COPY_CEILING ? bu_copy_ceiling_device(ctx, d_in[k], n_blocks, d_out[k], s)
                                                   : bu_uastc_transcode_device(ctx, target, d_in[k], n_blocks, d_out[k], blocks_per_row, 0, d_status, s);
    });
}

// the same window over the ETC1S codebook-lookup kernels: launch i decodes index array d_idx[(first_buffer + i) % n_buffers] (nbx x nby blocks each)
// against one pair of codebooks into d_out[...]; rgba = 0: bu_etc1s_transcode_etc1_device, 1: bu_etc1s_decode_rgba_device (no alpha slice)
bu_status bu_time_etc1s_launches_streams_window(bu_context* ctx, int rgba, const uint32_t* const* d_idx, void* const* d_out, size_t n_buffers, size_t first_buffer,
                                                size_t nbx, size_t nby, const uint32_t* d_endpoints, uint32_t n_endpoints, const void* d_selectors,
                                                uint32_t n_selectors, int lead, int launches, int tail, int n_streams, float* out_event_ms, float* out_host_ms)
{
    if (!d_idx || !d_out || n_buffers == 0) return BU_ERR_ARGUMENT;
    return bu_streams_window(ctx, lead, launches, tail, n_streams, out_event_ms, out_host_ms, nullptr, nullptr, [&](int i, hipStream_t s) {
        const size_t k = (first_buffer + (size_t)i) % n_buffers;
        return rgba ? bu_etc1s_decode_rgba_device(ctx, d_idx[k], nullptr, nbx, nby, d_endpoints, n_endpoints, d_selectors, n_selectors, d_out[k], nullptr, s)
                    : bu_etc1s_transcode_etc1_device(ctx, d_idx[k], nbx * nby, d_endpoints, n_endpoints, d_selectors, n_selectors, d_out[k], nullptr, s);
    });
}

// The reference's own micro-benchmark shape (benches/benchmark.rs:66-98: 32 blocks x 1000 calls per target): `reps` passes over
// `n_blocks` blocks, ONE per-block API call per block, timed with the host's steady clock around the whole loop.  target
// RGBA32 times bu_unpack_uastc_block_to_rgba.  The last pass's results stay in `out` (n_blocks x block bytes); a failing
// block's status is returned.
bu_status bu_time_block_api(bu_context* ctx, bu_target target, const uint8_t* blocks, size_t n_blocks, int reps, uint8_t* out, float* out_ns_per_call)
{
    if (!ctx || !blocks || !out || n_blocks == 0 || reps <= 0 || !out_ns_per_call) return BU_ERR_ARGUMENT;
    const size_t bb = bu_target_block_bytes(target);
    if (bb == 0) return BU_ERR_ARGUMENT;
    bu_status st = BU_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps && st == BU_OK; r++)
        for (size_t i = 0; i < n_blocks && st == BU_OK; i++) {
            const uint8_t* in = blocks + 16 * i;
            uint8_t* o = out + bb * i;
            switch (target) {
            case BU_TARGET_ASTC: st = bu_transcode_uastc_block_to_astc(ctx, in, o); break;
            case BU_TARGET_BC7: st = bu_transcode_uastc_block_to_bc7(ctx, in, o); break;
            case BU_TARGET_ETC1: st = bu_transcode_uastc_block_to_etc1(ctx, in, o); break;
            case BU_TARGET_ETC2: st = bu_transcode_uastc_block_to_etc2(ctx, in, o); break;
            default: st = bu_unpack_uastc_block_to_rgba(ctx, in, reinterpret_cast<uint32_t*>(o)); break;
            }
        }
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    *out_ns_per_call = (float)(ns / ((double)reps * (double)n_blocks));
    return st;
}

bu_status bu_time_copy_launches(bu_context* ctx, const void* const* d_in, void* const* d_out, size_t n_buffers, size_t first_buffer,
                                size_t n_blocks, int launches, void* stream, float* out_ms)
{
    if (!ctx || !d_in || !d_out || n_buffers == 0 || launches <= 0 || !out_ms) return BU_ERR_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BU_HIP(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < launches; i++) {
        const size_t k = (first_buffer + (size_t)i) % n_buffers;
        bu_status st = bu_copy_ceiling_device(ctx, d_in[k], n_blocks, d_out[k], stream);
        if (st) return st;
    }
    BU_HIP(ctx, hipEventRecord(ctx->ev1, s));
    BU_HIP(ctx, hipEventSynchronize(ctx->ev1));
    BU_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return BU_OK;
}

}  // extern "C"
