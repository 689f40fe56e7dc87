// The context's own streams: creation in groups of four with a hardware-queue check behind it, the launch policy picked per call
// (BU_LAUNCH_AUTO), and the host-joined pipeline the blocking entry points run a large contiguous range through.
// Part of the single translation unit bu_hip.hip (included there behind bu_context.hpp; not a stand-alone header).
#pragma once
namespace {

// ---- do streams 0..n-1 run side by side? ---------------------------------------------------------------------------------------
// One sleeping wave (BU_PROBE_TICKS of the 100 MHz clock) is launched on every stream behind a common event; streams on different
// hardware queues sleep together (the whole probe takes one sleep), streams that share a queue sleep one after the other.
// *out_sharing = round(time of the probe / one sleep) = the largest number of the probed streams on one queue.  The HIP runtime has no
// call that says which queue a stream is on.  Caller holds ctx->stream_lock (the probe events are the context's).
constexpr unsigned long long BU_PROBE_TICKS = 20000;  // 200 us
bu_status bu_probe_streams_locked(bu_context* ctx, const hipStream_t* streams, int n, int* out_sharing)
{
    if (!ctx->probe_ev0) BU_HIP(ctx, hipEventCreate(&ctx->probe_ev0));
    for (int i = 0; i < n; i++)
        if (!ctx->probe_ev[i]) BU_HIP(ctx, hipEventCreateWithFlags(&ctx->probe_ev[i], hipEventDisableSystemFence));  // (timing only)
    struct Drain {  // an error return must not leave sleeping waves behind events that are about to be reused
        const hipStream_t* s;
        int n;
        bool armed = true;
        ~Drain()
        {
            if (!armed) return;
            for (int i = 0; i < n; i++)
                if (s[i]) (void)hipStreamSynchronize(s[i]);
        }
    } drain{streams, n};
    float best = 0;
    for (int pass = 0; pass < 2; pass++) {  // (the first pass pays for loading the kernel)
        BU_HIP(ctx, hipEventRecord(ctx->probe_ev0, streams[0]));
        for (int i = 1; i < n; i++) BU_HIP(ctx, hipStreamWaitEvent(streams[i], ctx->probe_ev0, 0));
        for (int i = 0; i < n; i++) {
            hipLaunchKernelGGL(bu_sleep_kernel, dim3(1), dim3(64), 0, streams[i], BU_PROBE_TICKS);
            BU_HIP(ctx, hipGetLastError());
            BU_HIP(ctx, hipEventRecord(ctx->probe_ev[i], streams[i]));
        }
        float worst = 0;
        for (int i = 0; i < n; i++) {
            BU_HIP(ctx, hipEventSynchronize(ctx->probe_ev[i]));
            float ms = 0;
            BU_HIP(ctx, hipEventElapsedTime(&ms, ctx->probe_ev0, ctx->probe_ev[i]));
            if (ms > worst) worst = ms;
        }
        best = worst;
    }
    drain.armed = false;
    const int k = (int)(best / (BU_PROBE_TICKS * 1e-5f) + 0.5f);
    *out_sharing = k < 1 ? 1 : (k > n ? n : k);
    return BU_OK;
}

// BU_STREAM_MODE (diagnostic knob, read once): "plain" = never re-create a group with CU masks (what round 5 shipped; the tests use it to reach
// the degraded path of the in-flight call), "cumask" = create every group with CU masks straight away; anything else: decide by the probe
inline int bu_forced_stream_mode()
{
    static const int mode = [] {
        const char* e = getenv("BU_STREAM_MODE");
        if (!e) return 0;
        if (!strcmp(e, "plain")) return (int)BU_STREAMS_PLAIN;
        if (!strcmp(e, "cumask")) return (int)BU_STREAMS_CU_MASK;
        return 0;
    }();
    return mode;
}

// streams [g0, g0 + 4) in `mode`.  A CU-mask stream (hipExtStreamCreateWithCUMask, every CU enabled) gets a hardware queue of its OWN from the
// runtime instead of a share of the GPU_MAX_HW_QUEUES pool; it has default-stream semantics towards the process's NULL stream (it is not
// hipStreamNonBlocking -- the extension has no flags argument), which a process that launches nothing on the NULL stream never notices.
bu_status bu_make_stream_group(bu_context* ctx, hipStream_t* out4, int mode)
{
    uint32_t mask[32];
    const unsigned words = ((unsigned)ctx->cu_count + 31u) / 32u;
    for (unsigned w = 0; w < 32; w++) mask[w] = 0xFFFFFFFFu;
    if (ctx->cu_count % 32) mask[words - 1] = (1u << (ctx->cu_count % 32)) - 1u;
    for (int i = 0; i < 4; i++) {
        const hipError_t e = mode == BU_STREAMS_CU_MASK ? hipExtStreamCreateWithCUMask(&out4[i], words < 32u ? words : 32u, mask)
                                                        : hipStreamCreateWithFlags(&out4[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipStreamDestroy(out4[j]);
            for (int j = 0; j < 4; j++) out4[j] = nullptr;
            return bu_fail(ctx, e, mode == BU_STREAMS_CU_MASK ? "hipExtStreamCreateWithCUMask" : "hipStreamCreateWithFlags");
        }
    }
    return BU_OK;
}

// The context's own streams 0..n-1 (n <= 8).  They are created on first use in GROUPS OF FOUR, and a group is checked before anybody sees it:
// whether launches on two streams overlap is decided by the HIP runtime, which multiplexes all plain streams of a process over a pool of
// hardware queues per priority level (GPU_MAX_HW_QUEUES, 4 by default) -- two streams that share a queue run their kernels one after the other
// exactly as one stream would (profiles/r05_stream_creation_modes_queues_and_drift.txt: in a process whose NULL stream and context stream hold
// two of the four queues, four more streams land on the other two and "4 launches in flight" run as 2: 7.3 instead of 5.7 us per atlas).
// So: create the group as plain non-blocking streams, probe all streams made so far, and if two of them share a queue re-create the NEW group
// (nobody holds its handles yet) with full CU masks -- a dedicated queue each whatever the environment says -- and probe again.  What the last
// probe found is kept (ctx->stream_sharing[group]); bu_context_query_in_flight reports it and bu_uastc_transcode_batch_in_flight acts on it.
// A process that exports GPU_MAX_HW_QUEUES >= its number of streams before HIP initialises (bench.py: 8; 16 beside an RCCL communicator)
// keeps plain streams; one that does not (a Rust host that never heard of the variable) gets CU-mask streams; both run four launches side by side.
bu_status bu_ctx_streams(bu_context* ctx, int n)
{
    if (n < 0 || n > 8) return BU_ERR_ARGUMENT;
    std::lock_guard<std::mutex> g(ctx->stream_lock);
    while (ctx->streams_made < n) {
        const int g0 = ctx->streams_made, grp = g0 / 4, forced = bu_forced_stream_mode();
        int mode = forced == BU_STREAMS_CU_MASK ? (int)BU_STREAMS_CU_MASK : (int)BU_STREAMS_PLAIN;
        hipStream_t made[4] = {nullptr, nullptr, nullptr, nullptr};
        bu_status st = bu_make_stream_group(ctx, made, mode);
        if (st) return st;
        // the group is probed together with the streams that exist already, and PUBLISHED (made visible to readers that load the handles without the
        // lock: bu_context_synchronize, bu_auto_policy) only in its final form
        hipStream_t all[8];
        for (int i = 0; i < g0; i++) all[i] = ctx->extra_streams[i].load(std::memory_order_acquire);
        const auto destroy = [&](hipStream_t* four) {
            for (int i = 0; i < 4; i++) {
                if (four[i]) (void)hipStreamDestroy(four[i]);
                four[i] = nullptr;
            }
        };
        for (int i = 0; i < 4; i++) all[g0 + i] = made[i];
        int sharing = 1;
        st = bu_probe_streams_locked(ctx, all, g0 + 4, &sharing);
        if (st) {
            destroy(made);
            return st;
        }
        if (sharing > 1 && mode == BU_STREAMS_PLAIN && forced == 0) {
            hipStream_t masked[4] = {nullptr, nullptr, nullptr, nullptr};
            if (bu_make_stream_group(ctx, masked, BU_STREAMS_CU_MASK) == BU_OK) {  // (a runtime without the extension keeps its plain streams)
                destroy(made);
                for (int i = 0; i < 4; i++) all[g0 + i] = made[i] = masked[i];
                mode = BU_STREAMS_CU_MASK;
                st = bu_probe_streams_locked(ctx, all, g0 + 4, &sharing);
                if (st) {
                    destroy(made);
                    return st;
                }
            } else {
                (void)hipGetLastError();
            }
        }
        for (int i = 0; i < 4; i++) ctx->extra_streams[g0 + i].store(made[i], std::memory_order_release);
        ctx->stream_mode[grp] = mode;
        ctx->stream_sharing[grp] = sharing;
        ctx->streams_made = g0 + 4;
    }
    return BU_OK;
}

// streams 0..n-1 exist; *out_effective = how many launches they really keep in flight = n / (streams per hardware queue, by the creation-time
// probe), *out_mode = BU_STREAMS_* of the group that holds stream n-1
bu_status bu_ctx_in_flight_streams(bu_context* ctx, int n, int* out_effective, int* out_mode)
{
    const bu_status st = bu_ctx_streams(ctx, n);
    if (st) return st;
    std::lock_guard<std::mutex> g(ctx->stream_lock);
    const int grp = (n - 1) / 4, sharing = ctx->stream_sharing[grp] < 1 ? 1 : ctx->stream_sharing[grp];
    if (out_effective) *out_effective = (n + sharing - 1) / sharing;
    if (out_mode) *out_mode = ctx->stream_mode[grp];
    return BU_OK;
}

inline long long bu_now_ns() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// a large launch is being enqueued on `s` under an EXPLICIT policy: if `s` is one of the context's own streams, remember when (bu_auto_policy reads it
// for launches that are left to BU_LAUNCH_AUTO on the other streams -- e.g. a lone bu_uastc_transcode_device beside a running in-flight batch)
void bu_note_big_enqueue(bu_context* ctx, hipStream_t s)
{
    if (!s) return;
    for (int i = 0; i < 8; i++)
        if (ctx->extra_streams[i].load(std::memory_order_acquire) == s) {
            ctx->last_big_enqueue_ns[i].store(bu_now_ns(), std::memory_order_relaxed);
            return;
        }
}

// ---- the launch policy of ONE call (BU_LAUNCH_AUTO) ----------------------------------------------------------------------------
// A large launch shaped to fill the chip (exclusive) is the fastest way through one slice that is alone (8.4 us per 2^20 blocks against 11.6 for
// the half-CU shape) and the slower one as soon as launches of other streams run beside it (6.2 against 5.6 with four in flight).  Which of
// the two a call is in is known at the moment it enqueues: the launch goes to one of the context's OWN streams and another of them has work
// that has not completed.  With ONE or TWO others busy the launch takes the shared kernels on one-tile workgroups (BU_POLICY_SHARED_FEW, BC7 / ASTC),
// with three or more the shared policy's persistent shape.  Launches on the caller's own streams are exclusive: the library cannot see what runs
// beside them (bu_context_set_launch_policy(ctx, BU_LAUNCH_SHARED) is the override for such callers).
constexpr long long BU_AUTO_RECENT_NS = 40000;
int bu_auto_policy(bu_context* ctx, hipStream_t s)
{
    if (!s) return BU_POLICY_EXCLUSIVE;
    int me = -1;
    for (int i = 0; i < 8; i++)
        if (ctx->extra_streams[i].load(std::memory_order_acquire) == s) me = i;
    if (me < 0) return BU_POLICY_EXCLUSIVE;
    const long long now = bu_now_ns();
    // how many OTHER own streams have work in flight: enqueued there within the last BU_AUTO_RECENT_NS of host time (a caller that feeds n streams
    // round-robin comes back to each every n x ~5 us: no runtime call at all on that path) ...
    int busy = 0;
    bool stale[8] = {false, false, false, false, false, false, false, false};
    for (int j = 0; j < 8; j++) {
        if (j == me) continue;
        const long long t = ctx->last_big_enqueue_ns[j].load(std::memory_order_relaxed);
        if (t != 0 && now - t < BU_AUTO_RECENT_NS) busy++;
        else stale[j] = t != 0;
    }
    // ... else hipStreamQuery says so (one call per stream that was ever used, only when nothing was enqueued recently)
    if (busy == 0) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap == hipStreamCaptureStatusNone) {  // (a stream query is not something to issue in the middle of a capture)
            for (int j = 0; j < 8; j++) {
                if (!stale[j]) continue;
                hipStream_t o = ctx->extra_streams[j].load(std::memory_order_acquire);
                if (!o) continue;
                if (hipStreamQuery(o) == hipErrorNotReady) busy++;
                (void)hipGetLastError();  // (hipErrorNotReady is not an error)
            }
        }
    }
    ctx->last_big_enqueue_ns[me].store(now, std::memory_order_relaxed);
    ctx->auto_picks[busy == 0 ? 0 : (busy <= 2 ? 1 : 2)].fetch_add(1, std::memory_order_relaxed);
    return busy == 0 ? BU_POLICY_EXCLUSIVE : (busy <= 2 ? BU_POLICY_SHARED_FEW : BU_POLICY_SHARED);
}

// The tile-ticket set a persistent launch on `s` draws its tiles from (kernel, `ticket`), or nullptr for the fixed walk.  A set must never serve two
// launches at once, and the kernel zeroes it when its last workgroup leaves.  Launches of ONE stream run one after the other -- that is what a stream is
// -- so a set per stream is safe for any stream: the context's own streams and its internal one have fixed sets (0..8), a stream of the caller's gets
// the next free one of BU_FOREIGN_TICKET_SETS on its first large launch (a small table under a lock: only launches of 16 or more tiles per workgroup
// -- 90 us and up -- ever ask).  Excluded: hipStreamPerThread (one handle, a different stream on every host thread), a stream that is being captured
// (a graph may be replayed on any stream, several times at once), and the caller's 33rd stream (fixed walk).
unsigned* bu_ticket_for(bu_context* ctx, hipStream_t s)
{
    static const bool off = [] { const char* e = getenv("BU_TILE_TICKETS"); return e && e[0] == '0'; }();  // diagnostic knob: 0 = fixed walk everywhere
    if (off || ctx->tickets_off.load(std::memory_order_relaxed) || !ctx->d_tickets || s == hipStreamPerThread) return nullptr;
    int slot = s == ctx->stream ? 8 : -1;
    for (int i = 0; i < 8 && slot < 0; i++)
        if (s && ctx->extra_streams[i].load(std::memory_order_acquire) == s) slot = i;
    if (slot < 0) {
        std::lock_guard<std::mutex> g(ctx->ticket_lock);
        for (int i = 0; i < ctx->n_foreign_streams && slot < 0; i++)
            if (ctx->foreign_streams[i] == s) slot = 9 + i;
        if (slot < 0 && ctx->n_foreign_streams < BU_FOREIGN_TICKET_SETS) {
            ctx->foreign_streams[ctx->n_foreign_streams] = s;  // (the NULL stream is a stream like any other here: its launches run one after the other)
            slot = 9 + ctx->n_foreign_streams++;
        }
    }
    if (slot < 0) return nullptr;
    if (s) {  // (the legacy NULL stream cannot be captured)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) return nullptr;
    }
    return ctx->d_tickets + BU_TICKET_WORDS * slot;
}

}  // namespace
