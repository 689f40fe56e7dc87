// bu_context (device resources of one context), the error / drain helpers of the host side, the launcher that picks a kernel
// shape per target and size (bu_launch_uastc) and the host-pointer driver shared by the slice-level entry points.
// Part of the single translation unit bu_hip.hip (included there; not a stand-alone header).
#pragma once

// ================================================================================================
constexpr int BU_FOREIGN_TICKET_SETS = 32;
struct bu_context {
    int device = -1;
    int cu_count = 256;
    hipStream_t stream = nullptr;
    BuTablesAll* d_tables = nullptr;
    BuCrcTables* d_crc_tables = nullptr;  // bu_crc16_pieces_kernel
    void* d_in = nullptr;
    size_t in_cap = 0;
    void* d_out = nullptr;
    size_t out_cap = 0;
    void* d_aux = nullptr;  // codebooks / alpha indices of the host-pointer ETC1S calls
    size_t aux_cap = 0;
    void* lex_buf = nullptr;  // token buffer of the two-thread slice loop (bu_read_etc1s_streamed): malloc'ed, grows
    size_t lex_cap = 0;
    void* h_idx = nullptr;  // page-locked index buffer of the streamed ETC1S front door: the host decoder writes it, the kernels read it over PCIe
    size_t h_idx_cap = 0;
    unsigned long long* d_status = nullptr;
    unsigned* d_tickets = nullptr;  // tile-ticket sets of the persistent launches (kernel, `ticket`): own streams [0..7], `stream` [8], then the caller's streams in order of first use
    hipStream_t foreign_streams[32] = {};  // (bu_ticket_for, under ticket_lock)
    int n_foreign_streams = 0;
    std::mutex ticket_lock;
    std::atomic<unsigned long long> auto_picks[3] = {{0}, {0}, {0}};  // what BU_LAUNCH_AUTO chose so far: exclusive, one-tile shared, shared (bu_time_auto_policy_counts)
    std::atomic<bool> tickets_off{false};  // bu_time_set_tile_tickets (measurement: the fixed walk beside the ticketed one in one process)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_start[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // per-stream events of bu_time_uastc_launches_streams_window
    hipEvent_t ev_end[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // the context's own streams (bu_streams.hpp): created in groups of four under stream_lock, final once published (readers load without the lock)
    std::atomic<hipStream_t> extra_streams[8] = {{nullptr}, {nullptr}, {nullptr}, {nullptr}, {nullptr}, {nullptr}, {nullptr}, {nullptr}};
    int streams_made = 0;                // 0, 4 or 8 (stream_lock)
    int stream_mode[2] = {0, 0};         // BU_STREAMS_* of each group of four (stream_lock)
    int stream_sharing[2] = {0, 0};      // the creation-time probe over streams 0..3 / 0..7: the largest number of them on one hardware queue (stream_lock)
    hipEvent_t probe_ev0 = nullptr, probe_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // bu_probe_streams_locked (stream_lock)
    std::atomic<long long> last_big_enqueue_ns[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};  // host clock of the last large launch enqueued on each own stream (bu_auto_policy)
    std::atomic<int> launch_policy{2};  // BU_POLICY_*: how much of a CU one large launch of the mode-sorted kernel takes (bu_context_set_launch_policy); default BU_POLICY_AUTO
    // the blocking device-pointer entry points (bu_range_begin): their status word is page-locked host memory
    unsigned long long* h_status = nullptr;   // eight page-locked words, written by the host (reset) and by failing blocks (system-scope atomic min)
    unsigned long long* hd_status = nullptr;  // the same words as the device addresses them
    float win_start_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}, win_end_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the last streams window: per-stream event times (bu_time_last_window_streams)
    int win_streams = 0;
    float win_enqueue_ms = 0;  // host time the last streams window spent enqueueing its win_enqueued launches (and their events)
    int win_enqueued = 0;
    std::atomic<bool> single_thread_enqueue{false};  // BU_ENQUEUE_THREADS=0 (diagnostic knob, bu_context_create): the pipelined batch call enqueues from the calling thread alone
    std::atomic<int> time_enqueue_threads{0};  // bu_time_set_enqueue_threads: the streams windows enqueue from one host thread per stream
    std::atomic<bool> block_api_on_device{false};  // per-block API: host build of the block code (default) or a 1-block launch
    size_t etc1s_lds_limit = 0;  // what the device reports a workgroup may use, less a margin (bu_context_create)
    std::atomic<size_t> etc1s_lds_state[2] = {{0}, {0}};  // bu_etc1s_staged_kernel<false / true>: 0 not asked, 1 refused, else dynamic LDS bytes granted
    std::mutex stream_lock;  // creation of extra_streams (bu_ctx_streams)
    std::mutex lock;  // host-pointer entry points share the staging buffers
    std::mutex err_lock;  // `err` is written by whichever thread fails (device-pointer entry points run without `lock`)
    char err[256] = {0};
};

namespace {

bu_status bu_fail(bu_context* ctx, hipError_t e, const char* what)
{
    if (ctx) {
        std::lock_guard<std::mutex> g(ctx->err_lock);
        snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, hipGetErrorString(e));
    }
    return BU_ERR_HIP;
}
#define BU_HIP(ctx, call)                                       \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return bu_fail(ctx, e_, #call);   \
    } while (0)

// An early error return must not leave asynchronous copies in flight: they target the caller's stack frame (status
// words), vectors about to be freed, or the context's staging buffers the next caller will reuse.  Armed while work is
// queued; the success path disarms it after its own final synchronisation.
struct BuDrain {
    bu_context* ctx;
    bool armed = true;
    explicit BuDrain(bu_context* c) : ctx(c) {}
    ~BuDrain()
    {
        if (!armed || !ctx) return;
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        for (hipStream_t es : ctx->extra_streams)
            if (es) (void)hipStreamSynchronize(es);
    }
};

enum { BU_STREAMS_NONE = 0, BU_STREAMS_PLAIN = 1, BU_STREAMS_CU_MASK = 2 };
bu_status bu_ctx_streams(bu_context* ctx, int n);  // bu_streams.hpp

bu_status bu_reserve(bu_context* ctx, void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return BU_OK;
    if (*p) BU_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t sz = need < (1u << 20) ? (1u << 20) : need;
    BU_HIP(ctx, hipMalloc(p, sz));
    *cap = sz;
    return BU_OK;
}

// workgroups of the zero-copy launches: enough loads in flight to cover PCIe latency, few enough that every workgroup
// walks many tiles and reads overlap writes (measured on a 4096^2 atlas: 16 -> 0.52 ms, 64 -> 0.47, 256 -> 0.54, 1024 -> 0.56)
constexpr unsigned BU_ZEROCOPY_GRID = 64;

// Blocks per tile of a launch whose kernel takes its tile size at run time (bu_uastc_sorted_kernel, DYN_TILE): the smallest
// number of rounds the full tile allows, then equal tiles (a multiple of 64 blocks) so that every workgroup slot gets the
// same share.  1.5 Mi blocks on 256 slots of up to 4096: two rounds of 3072 instead of 4096 + 2048.
size_t bu_balanced_tile(size_t max_tile, size_t n_blocks, size_t slots, bool dynamic)
{
    if (!dynamic || slots == 0) return max_tile;
    const size_t per_slot = (n_blocks + slots - 1) / slots, rounds = (per_slot + max_tile - 1) / max_tile;
    size_t t = ((per_slot + rounds - 1) / rounds + 63) & ~(size_t)63;
    return t < 64 ? 64 : (t > max_tile ? max_tile : t);
}

// ---- shapes of the mode-sorted kernel ------------------------------------------------------------------------------------------
// One BuShape = one compiled instantiation of bu_uastc_sorted_kernel (strip layout, plus the rectangular layout where RECT is set):
// WGS threads x BPT blocks per thread = one tile; MINW = waves per SIMD the register allocation leaves room for; PER_CU = how many
// workgroups of ONE launch may be resident on a CU (the grid of a large launch is min(tiles, PER_CU x CUs); workgroups walk the
// remaining tiles, with the next tile's loads in flight where PREFETCH is set).
template <int W, int B, int MINW_, bool PF, bool RECT_, int PER_CU_>
struct BuShape {
    static constexpr int WGS = W, BPT = B, MINW = MINW_, PER_CU = PER_CU_, TILE = W * B;
    static constexpr bool PREFETCH = PF, RECT = RECT_;
};
// The shape of a LARGE launch (more than one 1024-block tile per CU; ETC: more than three) per target and launch policy.
//
// BU_LAUNCH_EXCLUSIVE -- the launch is alone on the chip and must fill it by itself (rounds 1-4; every figure an A/B inside one run
// on a 4096^2 atlas = 4096 blocks per CU, DESIGN_HISTORY.md section 4 and profiles/r04_ab_bc7_tile_shapes_and_upfront_loads.txt):
//   BC7 / ASTC  512 x 2, four workgroups per CU = 32 waves, <= 64 VGPRs.  Four INDEPENDENT sort chains per CU hide each other's
//               barriers and LDS round trips; 2048-block tiles +10 %, 4096 +35 %, 256 x 4 four per CU +18 %.
//   ETC1 / ETC2 one 1024-thread workgroup per CU on a tile of up to 4096 blocks (99 / 121 VGPRs: 16 waves are all that fit);
//               73 chunks per 4096 blocks where two 2048-block tiles have 83.
//               (Only below 2^20 blocks since the end of round 6: from there on ETC launches are one-tile workgroups of the SHARED shape, bu_launch_sorted.)
//   RGBA32      1024-block tiles (64 KiB of LDS for the four pixel rows), two workgroups per CU, 1024 x 1 up to 3 Mi blocks then 512 x 2.
// BU_LAUNCH_SHARED -- several launches from different streams are in flight and should run SIDE BY SIDE on every CU, so that one
// launch's load phase (3.4 us with the vector ALUs idle when it is alone) lies under another one's chunk phase (ALUs saturated, HBM
// idle).  A launch takes at most half of a CU's wave slots, registers and LDS (round 5, profiles/r05_ab_bc7_two_launches_in_flight.txt,
// r05_ab_etc_shared_shapes_x_streams.txt, r05_ab_wave_priorities_with_launches_in_flight.txt;
// us per 4096^2 atlas with 1 / 2 / 3 / 4 launches in flight):
//   BC7 / ASTC  256 x 4, two per CU (8 waves, 56 KiB), no wave priorities   11.8 / 6.8 / 6.0 / 5.45-5.55   (exclusive shape: 8.4 / 6.7 / 6.2 / 6.2)
//   ETC1        512 x 4, one per CU (8 waves, <= 128 VGPRs, 63 KiB) 20.1 / 13.1 / 12.1 / 12.2   (17.7 / 15.5 / 15.2 / 15.7)
//   ETC2        the same without the prefetch (115 VGPRs)           25.4 / 16.3 / 15.0 / 15.0   (22.1 / 19.7 / 19.3 / 20.4)
//   RGBA32      1024 x 1, one per CU (16 waves, 69 KiB)             19.6 / 14.7 / 13.4 / 13.1   (14.7 / 14.2 / 13.8 / 13.6)
// Alone on the chip a shared-policy launch is 15-40 % slower than an exclusive one: the policy is for callers that keep >= 2
// streams busy (bu_context_set_launch_policy).
enum { BU_POLICY_EXCLUSIVE = 0, BU_POLICY_SHARED = 1, BU_POLICY_AUTO = 2,
       BU_POLICY_SHARED_FEW = 3 };  // (internal, picked by bu_auto_policy only: the shared kernels on one-tile workgroups, for one or two other launches in flight)
int bu_auto_policy(bu_context* ctx, hipStream_t s);  // bu_streams.hpp: BU_POLICY_AUTO resolved for one launch on `s`
void bu_note_big_enqueue(bu_context* ctx, hipStream_t s);  // bu_streams.hpp: a large launch under an explicit policy goes to `s`
unsigned* bu_ticket_for(bu_context* ctx, hipStream_t s);  // bu_streams.hpp: the tile-ticket pair of an own stream, nullptr for anybody else's
template <int TARGET, int POLICY> struct BuBigShape;
template <> struct BuBigShape<BU_TGT_BC7, BU_POLICY_EXCLUSIVE> : BuShape<512, 2, 1, true, true, 4> {};
template <> struct BuBigShape<BU_TGT_BC7, BU_POLICY_SHARED> : BuShape<256, 4, 1, true, true, 2> {};
template <> struct BuBigShape<BU_TGT_ASTC, BU_POLICY_EXCLUSIVE> : BuShape<512, 2, 8, true, true, 4> {};  // (MINW 8: the strip form took 65 VGPRs = three per CU: a ragged 2^20-block slice 12.5 -> 9.5 us)
template <> struct BuBigShape<BU_TGT_ASTC, BU_POLICY_SHARED> : BuShape<256, 4, 1, true, true, 2> {};
template <> struct BuBigShape<BU_TGT_ETC1, BU_POLICY_EXCLUSIVE> : BuShape<1024, 4, 1, true, true, 1> {};
template <> struct BuBigShape<BU_TGT_ETC1, BU_POLICY_SHARED> : BuShape<512, 4, 4, true, true, 1> {};
template <> struct BuBigShape<BU_TGT_ETC2, BU_POLICY_EXCLUSIVE> : BuShape<1024, 4, 1, true, true, 1> {};
template <> struct BuBigShape<BU_TGT_ETC2, BU_POLICY_SHARED> : BuShape<512, 4, 4, false, true, 1> {};
// BC7 / ASTC use their large shape from the first tile beyond one per CU (8 waves on a 1024-block tile beat 4: 2^16 blocks 7.5 -> 5.7 us,
// 2^18 8.1 -> 6.3 us); ETC1 / ETC2 keep every tile of the 512 x 2 shape resident up to three 1024-block tiles per CU (2^19 blocks:
// 12.7 against 18.2 us for the 4096-block shape, 786 432: 16.7 / 18.6) and switch beyond it (917 504 blocks: 21.5 against 18.9 us)
constexpr bool bu_big_from_one_tile_per_cu(int target) { return target == BU_TGT_BC7 || target == BU_TGT_ASTC; }
// the shapes below the large ones, the same under both policies:
//   at most one 1024-block tile per CU: 16 waves on it (BC7 1 Ki blocks 4.32 -> 3.92 us, 2^16 5.16 -> 4.80, 2^18 5.70 -> 5.41; ETC1 6.76 -> 6.47, 7.95 -> 7.64, 8.82 -> 8.54)
template <int TARGET> using BuOneTileShape = BuShape<1024, 1, 1, false, (TARGET == BU_TGT_BC7 || TARGET == BU_TGT_ASTC), 1>;
//   ETC, up to three tiles per CU: 8 waves per tile, every tile resident (ETC1 at 2^16 blocks: 14.1 -> 11.3 us)
using BuEtcMidShape = BuShape<512, 2, 1, false, false, 3>;
//   zero-copy launches over PCIe (grid_cap): 256 x 4 on a small persistent grid
using BuZeroCopyShape = BuShape<256, 4, 1, true, false, 1>;
constexpr int BU_HOST_TILE = 1024;  // the tile the launcher counts in where the shape does not say otherwise

// one piece (<= 2^26 blocks) of a slice, as the mode-sorted kernel takes it
struct BuPiece {
    const uint4* in;
    void* out;
    size_t nb, bpr;
    unsigned long long base;
    unsigned long long* status;
    const BuTablesAll* tables;
    hipStream_t stream;
    unsigned* ticket;     // tile-ticket pair of the stream (bu_ticket_for), nullptr: every workgroup walks its fixed share of the tiles
    bool rect_rows;       // blocks_per_row allows rectangular tiles at all: a multiple of 64, at least two tiles wide, below 2^21
    size_t rect_quantum;  // every piece of the slice is a multiple of (rows per tile x blocks_per_row) for rows per tile dividing this
    unsigned rect_magic;  // ceil(2^32 / tiles per row): the kernel's tile -> (row, column) reciprocal
    // Rectangular tiles (kernel, RECT): the caller told us the block grid, it is a multiple of 64 wide and the piece -- and every
    // other piece of the slice -- is whole rows of tiles `rows` blocks high
    bool rect_ok(size_t rows) const { return rect_rows && nb % (rows * bpr) == 0 && (rect_quantum == 0 || rect_quantum % (rows * bpr) == 0); }
};

template <int TARGET, class S>
void bu_go(const BuPiece& p, unsigned grid, unsigned cus, unsigned tile_rt, unsigned* ticket = nullptr)
{
    if constexpr (S::RECT) {
        // (a shape that sizes its tile at run time is rectangular only when that size is the full tile)
        if (p.rect_ok((size_t)S::TILE / BU_RECT_W) && (!bu_dyn_tile(TARGET, S::TILE) || tile_rt == (unsigned)S::TILE)) {
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<TARGET, S::WGS, S::BPT, S::MINW, S::PREFETCH, BU_LAYOUT_RECT>), dim3(grid), dim3(S::WGS), 0, p.stream, p.in,
                               p.out, (unsigned)p.nb, (unsigned)p.bpr, p.base, p.status, p.tables, cus, p.rect_magic, ticket);
            return;
        }
    }
    hipLaunchKernelGGL((bu_uastc_sorted_kernel<TARGET, S::WGS, S::BPT, S::MINW, S::PREFETCH, BU_LAYOUT_STRIP>), dim3(grid), dim3(S::WGS), 0, p.stream, p.in, p.out,
                       (unsigned)p.nb, (unsigned)p.bpr, p.base, p.status, p.tables, cus, tile_rt, ticket);
}

// a large launch in shape S: persistent workgroups, PER_CU per CU, walking equal shares of the tiles.  `priorities`: the static wave
// priorities by residency generation (kernel, `cus`).  They serve a launch that is ALONE on the chip (BC7 8.57 -> 8.37 us) and hurt as
// soon as launches of several streams share the CUs -- the generations of different launches then compete through the same four levels:
// shared shape, four in flight 5.72-5.79 -> 5.44-5.56 us per atlas without them (the exclusive shape on two streams 6.70 -> 5.97:
// profiles/r05_ab_wave_priorities_with_launches_in_flight.txt) -- so the shared policy launches without.
// tiles per workgroup from which an exclusive BC7 / ASTC / RGBA32 launch draws its tiles by ticket (ETC1 / ETC2 are bound by vector-ALU issue on every
// CU alike: nothing to balance, +0.7 % with tickets)
constexpr size_t BU_TICKET_MIN_WALK = 16;
constexpr bool bu_ticket_target(int target) { return target == BU_TGT_BC7 || target == BU_TGT_ASTC || target == BU_TGT_RGBA; }
template <int TARGET, class S>
void bu_go_big(const BuPiece& p, unsigned cu_count, bool priorities)
{
    const size_t slots = (size_t)cu_count * S::PER_CU;
    const size_t tile_rt = bu_balanced_tile((size_t)S::TILE, p.nb, slots, bu_dyn_tile(TARGET, S::TILE));
    const size_t tiles = (p.nb + tile_rt - 1) / tile_rt;
    // generation priorities (kernel, `cus`) only when every workgroup walks the same number of tiles: with 1.25 tiles per
    // slot the one-tile generations run ahead of the two-tile ones (1.25 Mi blocks BC7 13.06 -> 11.57 us, ASTC 13.5 -> 11.0)
    const unsigned cus = (priorities && (tiles <= slots || tiles % slots == 0)) ? cu_count : 0u;
    // Tile tickets (kernel, `ticket`) for the LONG walks of a launch that has the chip to itself: with a fixed share of 32 tiles per workgroup
    // a 2^25-block BC7 launch takes 188.5 us, with tickets 174 (ASTC 201 -> 184.5; the launch ends when the tiles do, not when the slowest share
    // does; 16 tiles per workgroup: BC7 -2.8 %, ASTC -5 %, RGBA32 -4 %; 8: +-0); a walk of 2-4 tiles loses to the atomics' round trips at its head
    // and tail (2^22 blocks: 26.7 -> 32.2 us), and launches that share the chip fill each other's tails anyway (four 2^25-block launches in flight
    // 167 -> 171): profiles/r06_ab_tile_tickets.txt
    unsigned* const ticket = (priorities && bu_ticket_target(TARGET) && tiles >= BU_TICKET_MIN_WALK * slots) ? p.ticket : nullptr;
    bu_go<TARGET, S>(p, (unsigned)(tiles < slots ? tiles : slots), cus, (unsigned)tile_rt, ticket);
}

template <int TARGET>
void bu_launch_sorted(const BuPiece& p, unsigned cu_count, int policy, unsigned grid_cap)
{
    const size_t tiles = (p.nb + BU_HOST_TILE - 1) / BU_HOST_TILE;
    if (grid_cap) {
        bu_go<TARGET, BuZeroCopyShape>(p, (unsigned)(tiles < grid_cap ? tiles : grid_cap), cu_count, (unsigned)BuZeroCopyShape::TILE);
    } else if (p.nb <= (size_t)BU_HOST_TILE * cu_count) {
        bu_go<TARGET, BuOneTileShape<TARGET>>(p, (unsigned)tiles, cu_count, (unsigned)BU_HOST_TILE);
    } else if (bu_big_from_one_tile_per_cu(TARGET) || p.nb > (size_t)3 * BU_HOST_TILE * cu_count) {
        // BU_POLICY_SHARED_FEW (BC7 / ASTC, from bu_auto_policy when one or two other launches are in flight): the shared policy's kernel with the grid at four
        // workgroups per CU -- 1024 one-tile workgroups dealt by the hardware dispatcher instead of 512 persistent ones walking two tiles each.  With two /
        // three launches in flight 6.07 / 5.7 us per 2^20-block atlas against 6.95 / 6.1 (shared) and 6.85 / 6.3 (exclusive); with four the persistent form wins
        // (5.60 against 5.77): profiles/r06_ab_bc7_shared_one_tile_workgroups.txt
        using SharedShape = BuBigShape<TARGET, BU_POLICY_SHARED>;
        if ((TARGET == BU_TGT_ETC1 || TARGET == BU_TGT_ETC2) && p.nb >= ((size_t)1 << 20))
            // ETC1 / ETC2 from 2^20 blocks on, under EVERY policy: ONE-TILE workgroups of the shared shape (512 x 4 on a 2048-block tile, two resident per CU) dealt by the
            // hardware dispatcher instead of a persistent grid -- the form in which four launches in flight reach 12.2 / 15.1 us per 2^20 blocks, in ONE launch.  Exclusive (was
            // 1024 x 4, one per CU): 2^20 blocks 17.8 / 22.2 -> 17.5 / 20.6 us, 1.5 x 2^20 28.2 / 33.9 -> 23.8 / 28.1, 2^22 59.5 / 75.6 -> 54.1 / 65.1, 2^25 433 / 556 -> 394 / 479
            // (12.3 / 15.0 per 2^20); shared (was 512 x 4 persistent, one per CU): one launch at a time 20.2 / 25.0 -> 17.6 / 20.6, two in flight 13.2 / 16.3 -> 12.1 / 15.0, three
            // and four +-1 %.  Below 2^20 blocks the persistent grids stay ahead (0.8 x 2^20 exclusive: 16.2 / 19.4 against 17.8 / 21.0).  Walking 2 / 4 / 8 tiles per workgroup
            // gives the gain back step by step (profiles/r06_ab_etc_one_tile_workgroups.txt; the copies of profiles/r06_copy_ceiling_by_size.txt behave the same way).
            bu_go<TARGET, SharedShape>(p, (unsigned)((p.nb + SharedShape::TILE - 1) / SharedShape::TILE), 0u, (unsigned)SharedShape::TILE);
        else if (policy == BU_POLICY_SHARED_FEW && bu_big_from_one_tile_per_cu(TARGET))
            bu_go_big<TARGET, BuShape<SharedShape::WGS, SharedShape::BPT, SharedShape::MINW, SharedShape::PREFETCH, SharedShape::RECT, 4>>(p, cu_count, false);
        else if (policy == BU_POLICY_SHARED || policy == BU_POLICY_SHARED_FEW) bu_go_big<TARGET, SharedShape>(p, cu_count, false);
        else if (TARGET == BU_TGT_ASTC && p.nb >= ((size_t)1 << 21))
            // ASTC from 2^21 blocks on: 256 x 4, five per CU (79 VGPRs, 24 KiB) -- 2^21 / 2^22 / 2^23 / 2^25 blocks 15.2 / 28.3 / 53.6 / 183.3 -> 14.3 / 27.5 / 51.2 / 179.9 us,
            // 2^24 level (97.5 / 98.0), a lone 2^20-block atlas 8.9 -> 9.9: profiles/r06_ab_astc_large_launch_256x4.txt.  (BC7 loses 0-5 % in that shape at every size.)
            bu_go_big<TARGET, BuShape<256, 4, 1, true, true, 5>>(p, cu_count, true);
        else bu_go_big<TARGET, BuBigShape<TARGET, BU_POLICY_EXCLUSIVE>>(p, cu_count, true);
    } else {
        bu_go<TARGET, BuEtcMidShape>(p, (unsigned)tiles, cu_count, (unsigned)BU_HOST_TILE);
    }
}

// RGBA32, 64 B of output per block: results return through a 64 KiB LDS tile (1024 blocks x 4 rows, the input tile aliased
// into row 0) so the image rows leave as coalesced 1 KiB stores; persistent workgroups walk their tiles with prefetch, two per
// CU (one under the shared policy).  Up to 3 Mi blocks 1024 threads per tile (32 waves per CU: 2^18 blocks 7.8 -> 7.2 us,
// 2^20 17.95 -> 16.9, 2^21 35.0 -> 33.75), above that 512 threads x 2 blocks (2^22 blocks 62.7 against 64.4 us, 2^24 252
// against 265).  The zero-copy launches (grid_cap) keep the 512 x 2 shape.
void bu_launch_sorted_rgba(const BuPiece& p, unsigned cu_count, int policy, unsigned grid_cap)
{
    const size_t tiles = (p.nb + BU_HOST_TILE - 1) / BU_HOST_TILE;
    if (policy == BU_POLICY_SHARED_FEW) policy = BU_POLICY_SHARED;
    const size_t cap = grid_cap ? (size_t)grid_cap : (size_t)cu_count * (policy == BU_POLICY_SHARED ? 1 : 2);
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    // generation priorities only when every workgroup walks at least two tiles (2^19 blocks 10.7 -> 10.3 us and
    // 786 432 blocks 15.75 -> 14.24 without them, 2^20 blocks 16.7 against 18.7 with them)
    const unsigned cus = (policy != BU_POLICY_SHARED && tiles >= 2 * (size_t)grid) ? cu_count : 0u;
    unsigned* const ticket = (policy != BU_POLICY_SHARED && tiles >= BU_TICKET_MIN_WALK * (size_t)grid) ? p.ticket : nullptr;  // (tile tickets for long walks, as bu_go_big)
    if (grid_cap == 0 && p.nb <= ((size_t)3 << 20)) bu_go<BU_TGT_RGBA, BuShape<1024, 1, 1, true, true, 2>>(p, grid, cus, (unsigned)BU_HOST_TILE, ticket);
    else bu_go<BU_TGT_RGBA, BuShape<512, 2, 1, true, true, 2>>(p, grid, cus, (unsigned)BU_HOST_TILE, ticket);
}

// grid_cap > 0 (zero-copy over PCIe): 1024-block tiles on at most grid_cap workgroups.
// policy = BU_POLICY_* of this launch, or -1 for the context's (bu_context_set_launch_policy; BU_POLICY_AUTO there is resolved per launch by
// bu_auto_policy): only the device-pointer slice entry points pass -1 -- the host-pointer and whole-file entry points issue their launches one
// after another on one stream, for which the exclusive shapes are the right ones whatever the context says.
bu_status bu_launch_uastc(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out, size_t bpr,
                          uint64_t base, uint64_t* d_status, hipStream_t stream, unsigned grid_cap = 0, int policy = BU_POLICY_EXCLUSIVE)
{
    if (n_blocks == 0) return BU_OK;
    const unsigned grid = bu_grid_for(n_blocks, ctx->cu_count);
    const uint4* in = static_cast<const uint4*>(d_in);
    unsigned long long* st = reinterpret_cast<unsigned long long*>(d_status);
    if (n_blocks >= (size_t)BU_SORT_MIN_BLOCKS) {
        // mode-sorted kernel.  The kernel indexes with 32 bits, so very large slices are cut into launches of <= 2^26 blocks
        // (1 GiB in); RGBA32 pieces end on whole block rows so the image addressing stays launch-relative.
        size_t piece = (size_t)1 << 26;
        if (target == BU_TARGET_RGBA32) piece = bpr <= piece ? (piece / bpr) * bpr : bpr;
        const size_t obytes = bu_target_block_bytes(target);
        if (policy < 0) policy = ctx->launch_policy.load(std::memory_order_relaxed);
        // BU_LAUNCH_AUTO: decided per call, and only where the shapes differ (a launch of more than one tile per CU)
        const bool big = grid_cap == 0 && n_blocks > (size_t)BU_HOST_TILE * ctx->cu_count;
        if (policy == BU_POLICY_AUTO) policy = big ? bu_auto_policy(ctx, stream) : (int)BU_POLICY_EXCLUSIVE;
        else if (big) bu_note_big_enqueue(ctx, stream);
        constexpr size_t RW = BU_RECT_W;
        BuPiece p;
        p.ticket = grid_cap == 0 ? bu_ticket_for(ctx, stream) : nullptr;
        p.status = st;
        p.tables = ctx->d_tables;
        p.stream = stream;
        p.bpr = bpr;
        // (one tile per row, blocks_per_row == 64: the strip IS the rectangle)
        p.rect_rows = grid_cap == 0 && bpr >= 2 * RW && bpr % RW == 0 && bpr < ((size_t)1 << 21);
        p.rect_quantum = n_blocks <= piece ? 0 : piece;
        // A VIRTUAL pitch for BC7 / ASTC when the caller gave no usable block grid (blocks_per_row 0, or no multiple of 64): for a block-linear target the
        // grid never changes a byte, it only decides which 1024 blocks form a tile -- and a tile that is 16 segments of 1 KiB at a pitch of 4 KiB or more
        // loads and stores measurably faster than 16 KiB in a row (its 16 segments sit on 16 different HBM channel groups; the workgroup waits for ALL of its
        // tile at barrier 1): strips 8.94 / 5.77 / 187.5 us against 8.45 / 5.56 / 177.7 for a lone 2^20-block launch / four in flight / one 2^25-block launch
        // (profiles/r06_tile_pitch_sweep.txt).  Needs whole tiles: the slice a multiple of 16 x pitch blocks.  (A real grid is kept whatever its pitch:
        // on texture-like content rectangles of the IMAGE keep regions of one mode whole, which is worth more.)
        if (!p.rect_rows && grid_cap == 0 && (target == BU_TARGET_BC7 || target == BU_TARGET_ASTC)) {
            for (const size_t v : {(size_t)1024, (size_t)2048, (size_t)512, (size_t)256}) {
                if (n_blocks % (16 * v) == 0) {
                    p.bpr = bpr = v;
                    p.rect_rows = true;
                    break;
                }
            }
        }
        p.rect_magic = p.rect_rows ? (unsigned)((((unsigned long long)1 << 32) + bpr / RW - 1) / (bpr / RW)) : 0u;  // ceil(2^32 / tiles per row)
        for (size_t done = 0; done < n_blocks; done += piece) {
            p.nb = n_blocks - done < piece ? n_blocks - done : piece;
            p.in = in + done;
            p.out = static_cast<uint8_t*>(d_out) + done * obytes;  // RGBA32: done is a multiple of bpr -> whole rows
            p.base = base + done;
            switch (target) {
            case BU_TARGET_ASTC: bu_launch_sorted<BU_TGT_ASTC>(p, (unsigned)ctx->cu_count, policy, grid_cap); break;
            case BU_TARGET_BC7: bu_launch_sorted<BU_TGT_BC7>(p, (unsigned)ctx->cu_count, policy, grid_cap); break;
            case BU_TARGET_ETC1: bu_launch_sorted<BU_TGT_ETC1>(p, (unsigned)ctx->cu_count, policy, grid_cap); break;
            case BU_TARGET_ETC2: bu_launch_sorted<BU_TGT_ETC2>(p, (unsigned)ctx->cu_count, policy, grid_cap); break;
            case BU_TARGET_RGBA32: bu_launch_sorted_rgba(p, (unsigned)ctx->cu_count, policy, grid_cap); break;
            default: return BU_ERR_ARGUMENT;
            }
            BU_HIP(ctx, hipGetLastError());
        }
        return BU_OK;
    }
    switch (target) {
    case BU_TARGET_ASTC: hipLaunchKernelGGL(bu_uastc_kernel<BU_TGT_ASTC>, dim3(grid), dim3(BU_WG), 0, stream, in, d_out, n_blocks, (unsigned)bpr, base, st, ctx->d_tables); break;
    case BU_TARGET_BC7: hipLaunchKernelGGL(bu_uastc_kernel<BU_TGT_BC7>, dim3(grid), dim3(BU_WG), 0, stream, in, d_out, n_blocks, (unsigned)bpr, base, st, ctx->d_tables); break;
    case BU_TARGET_ETC1: hipLaunchKernelGGL(bu_uastc_kernel<BU_TGT_ETC1>, dim3(grid), dim3(BU_WG), 0, stream, in, d_out, n_blocks, (unsigned)bpr, base, st, ctx->d_tables); break;
    case BU_TARGET_ETC2: hipLaunchKernelGGL(bu_uastc_kernel<BU_TGT_ETC2>, dim3(grid), dim3(BU_WG), 0, stream, in, d_out, n_blocks, (unsigned)bpr, base, st, ctx->d_tables); break;
    case BU_TARGET_RGBA32: hipLaunchKernelGGL(bu_uastc_kernel<BU_TGT_RGBA>, dim3(grid), dim3(BU_WG), 0, stream, in, d_out, n_blocks, (unsigned)bpr, base, st, ctx->d_tables); break;
    default: return BU_ERR_ARGUMENT;
    }
    BU_HIP(ctx, hipGetLastError());
    return BU_OK;
}

// device-side address of a page-locked host buffer; false for ordinary (pageable) memory
bool bu_device_view(const void* p, void** dev)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory reports an error on some runtimes: not sticky
        return false;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    if (reinterpret_cast<uintptr_t>(a.devicePointer) % 16 != 0) return false;  // the kernels move 16-byte vectors
    *dev = a.devicePointer;
    return true;
}

// host-pointer UASTC driver shared by transcode / decode_to_rgba / the per-block API
bu_status bu_uastc_host(bu_context* ctx, bu_target target, const uint8_t* in, size_t in_bytes, size_t bpr, uint8_t* out,
                        size_t out_bytes, uint64_t* first_bad)
{
    if (!ctx || (!in && in_bytes) || !out) return BU_ERR_ARGUMENT;
    const size_t bb = bu_target_block_bytes(target);
    if (bb == 0) return BU_ERR_ARGUMENT;
    if (in_bytes % 16 != 0) return BU_ERR_LENGTH;  // uastc.rs:54-59
    const size_t n = in_bytes / 16;
    if (out_bytes < n * bb) return BU_ERR_OUTPUT_SIZE;
    if (target == BU_TARGET_RGBA32 && bpr == 0) return BU_ERR_ARGUMENT;
    if (n == 0) return BU_OK;
    std::lock_guard<std::mutex> g(ctx->lock);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    bu_status st;
    // Page-locked caller buffers (bu_host_alloc, or anything the caller page-locked with the HIP runtime) are visible to
    // the GPU: the kernels read the slice and / or write the result straight over PCIe -- no staging copy on that side.
    // A small persistent grid walks the tiles with prefetch, so tile k's posted writes travel upstream while tile k+1's
    // reads come down (PCIe is full duplex): 0.45 ms per 4096^2 atlas with both sides mapped, against 0.69 ms for upload +
    // kernel + download.  Ordinary pageable memory cannot be mapped and is staged through the context's device buffers.
    void *zin = nullptr, *zout = nullptr;
    const bool map_in = bu_device_view(in, &zin);
    // RGBA32 with a ragged last block row stores whole image rows, past the 64*n bytes the caller sized: keep that staged
    const bool map_out = !(target == BU_TARGET_RGBA32 && n % bpr != 0) && bu_device_view(out, &zout);
    if (!map_in) {
        st = bu_reserve(ctx, &ctx->d_in, &ctx->in_cap, in_bytes);
        if (st) return st;
    }
    if (!map_out) {
        size_t out_need = n * bb;
        if (target == BU_TARGET_RGBA32) out_need = ((n + bpr - 1) / bpr) * bpr * 64;
        st = bu_reserve(ctx, &ctx->d_out, &ctx->out_cap, out_need);
        if (st) return st;
    }
    const void* din = map_in ? zin : ctx->d_in;
    void* dout = map_out ? zout : ctx->d_out;
    uint64_t word = 0;
    BuDrain drain(ctx);
    if (!map_in) BU_HIP(ctx, hipMemcpyAsync(ctx->d_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    BU_HIP(ctx, hipMemsetAsync(ctx->d_status, 0xFF, sizeof(uint64_t), ctx->stream));
    st = bu_launch_uastc(ctx, target, din, n, dout, bpr, 0, reinterpret_cast<uint64_t*>(ctx->d_status), ctx->stream,
                         (map_in || map_out) ? BU_ZEROCOPY_GRID : 0);
    if (st) return st;
    BU_HIP(ctx, hipMemcpyAsync(&word, ctx->d_status, sizeof(word), hipMemcpyDeviceToHost, ctx->stream));
    if (!map_out) BU_HIP(ctx, hipMemcpyAsync(out, ctx->d_out, n * bb, hipMemcpyDeviceToHost, ctx->stream));
    BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain.armed = false;
    return bu_status_word_decode(word, first_bad);
}

// ---- the per-block API's host side (bu_capi_slice.hpp): the product's block code compiled for the host, over a host copy of the tables
const BuTablesAll& bu_host_tables()
{
    static const BuTablesAll* const tables = [] {
        BuTablesAll* t = new BuTablesAll();
        bu_build_tables(t);
        return t;
    }();
    return *tables;
}

template <int TARGET>
bu_status bu_block_on_host(const uint8_t in[16], void* out)
{
    const BuTables& T = bu_host_tables().t;
    BuBlk b;
    memcpy(b.w, in, 16);
    constexpr int NO = TARGET == BU_TGT_RGBA ? 16 : (TARGET == BU_TGT_ETC1 ? 2 : 4);
    uint32_t o[16] = {0};  // a failing block leaves its zeros (every path checks before it writes), as the kernels' result slots do
    const int st = bu_block_any<TARGET>(T, T.mode_lut[b.w[0] & 127u], b, o);
    memcpy(out, o, NO * sizeof(uint32_t));
    return st == BU_ST_OK ? BU_OK : (st == BU_ST_BAD_PATTERN ? BU_ERR_INVALID_PATTERN : BU_ERR_INVALID_MODE);
}

}  // namespace

