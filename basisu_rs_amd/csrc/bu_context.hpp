// bu_context (device resources of one context), the error / drain helpers of the host side, the launcher that picks a kernel
// shape per target and size (bu_launch_uastc) and the host-pointer driver shared by the slice-level entry points.
// Part of the single translation unit bu_hip.hip (included there; not a stand-alone header).
#pragma once

// ================================================================================================
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
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t extra_streams[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::atomic<bool> block_api_on_device{false};  // per-block API: host build of the block code (default) or a 1-block launch
    size_t etc1s_lds_limit = 0;  // what the device reports a workgroup may use, less a margin (bu_context_create)
    std::atomic<size_t> etc1s_lds_state[2] = {{0}, {0}};  // bu_etc1s_staged_kernel<false / true>: 0 not asked, 1 refused, else dynamic LDS bytes granted
    std::mutex lock;  // host-pointer entry points share the staging buffers
    char err[256] = {0};
};

namespace {

bu_status bu_fail(bu_context* ctx, hipError_t e, const char* what)
{
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, hipGetErrorString(e));
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

// experiment knob (tools/exp): BU_X_BCAP = workgroups per launch of the big shapes as a multiple of the resident set (0 = one tile
// per workgroup, whatever the size); unset = the resident set (persistent workgroups walk their tiles with prefetch)
inline size_t bu_x_bcap(size_t resident)
{
    static const long mult = [] {
        const char* e = getenv("BU_X_BCAP");
        return e ? atol(e) : 1L;
    }();
    return mult <= 0 ? (size_t)1 << 40 : resident * (size_t)mult;
}

// shapes that are also compiled with rectangular tiles (bu_uastc_sorted_kernel, RECT): the 1024-block tiles (64 x 16 blocks) of
// BC7, ASTC and RGBA32, the 4096-block tiles (64 x 64) of ETC1 and ETC2
constexpr bool bu_rect_compiled(int target, int tile)
{
    if (target == BU_TGT_BC7 && tile == BuBigCfg<BU_TGT_BC7>::WGS * BuBigCfg<BU_TGT_BC7>::BPT) return true;
    if ((target == BU_TGT_ETC1 || target == BU_TGT_ETC2) && tile == BuBigCfg<BU_TGT_ETC1>::WGS * BuBigCfg<BU_TGT_ETC1>::BPT) return true;
    return tile == 1024 ? (target == BU_TGT_BC7 || target == BU_TGT_ASTC || target == BU_TGT_RGBA)
                        : (tile == 4096 && (target == BU_TGT_ETC1 || target == BU_TGT_ETC2));
}

// experiment (round 4): BC7 / ASTC launches of one tile per workgroup load their tile straight into LDS (kernel, GLDS)
#ifdef BU_X_GLDS
constexpr bool BU_GLDS_ON = true;
#else
constexpr bool BU_GLDS_ON = false;
#endif
// grid_cap > 0 (zero-copy over PCIe): 1024-block tiles on at most grid_cap workgroups
bu_status bu_launch_uastc(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out, size_t bpr,
                          uint64_t base, uint64_t* d_status, hipStream_t stream, unsigned grid_cap = 0)
{
    if (n_blocks == 0) return BU_OK;
    const unsigned grid = bu_grid_for(n_blocks, ctx->cu_count);
    const uint4* in = static_cast<const uint4*>(d_in);
    unsigned long long* st = reinterpret_cast<unsigned long long*>(d_status);
    if (n_blocks >= (size_t)BU_SORT_MIN_BLOCKS) {
        // mode-sorted kernel: one tile per workgroup, grid-stride beyond 7 workgroups per CU.  The kernel
        // indexes with 32 bits, so very large slices are cut into launches of <= 2^26 blocks (1 GiB in);
        // RGBA32 pieces end on whole block rows so the image addressing stays launch-relative.
        constexpr int BU_TILE = BU_SORT_WGS * BU_SORT_BPT;
        size_t piece = (size_t)1 << 26;
        if (target == BU_TARGET_RGBA32) piece = bpr <= piece ? (piece / bpr) * bpr : bpr;
        const size_t obytes = bu_target_block_bytes(target);
        for (size_t done = 0; done < n_blocks; done += piece) {
            const size_t nb = n_blocks - done < piece ? n_blocks - done : piece;
            const uint4* pin = in + done;
            void* pout = static_cast<uint8_t*>(d_out) + done * obytes;  // RGBA32: done is a multiple of bpr -> whole rows
            const size_t tiles = (nb + BU_TILE - 1) / BU_TILE;
            const size_t cap = grid_cap ? (size_t)grid_cap : (size_t)ctx->cu_count * 7;
            const unsigned sgrid = (unsigned)(tiles < cap ? tiles : cap);
            const unsigned long long pbase = base + done;
            // Rectangular tiles (kernel, RECT): the caller told us the block grid (blocks_per_row), it is a multiple of 64 wide and
            // the piece is whole rows of 64 x 16-block tiles (BC7, ASTC, RGBA32: the fixed 1024-block shapes) or of 64 x 64-block
            // tiles (ETC1 / ETC2: the 4096-block shape, when the balanced tile size is the full 4096 anyway).
            constexpr size_t RW = BU_RECT_W;
            // (one tile per row, blocks_per_row == 64: the strip IS the rectangle)
            auto rect_ok = [&](size_t rh) {
                return grid_cap == 0 && bpr >= 2 * RW && bpr % RW == 0 && bpr < ((size_t)1 << 21) && nb % (rh * bpr) == 0 &&
                       (n_blocks <= piece || piece % (rh * bpr) == 0);
            };
            const unsigned rect_magic = (bpr >= 2 * RW && bpr % RW == 0) ? (unsigned)((((unsigned long long)1 << 32) + bpr / RW - 1) / (bpr / RW)) : 0u;  // ceil(2^32 / tiles per row)
#define BU_GO(T, W, B, MINW, PF, DIR, SK, GRID, CUS, TRT)                                                                                       \
    do {                                                                                                                                        \
        constexpr int tile_ = (W) * (B);                                                                                                        \
        if (bu_rect_compiled(T, tile_) && rect_ok((size_t)tile_ / RW) && ((T) == BU_TGT_BC7 || tile_ == 1024 || (TRT) == (unsigned)tile_))        \
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, W, B, MINW, PF, DIR, SK, bu_rect_compiled(T, tile_)>), dim3(GRID), dim3(W), 0, stream, pin, pout, \
                               (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, CUS, rect_magic BU_STAMP_PASS);                           \
        else                                                                                                                                    \
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, W, B, MINW, PF, DIR, SK, false>), dim3(GRID), dim3(W), 0, stream, pin, pout,          \
                               (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, CUS, TRT BU_STAMP_PASS);                                  \
    } while (0)
            // large inputs: the per-target BuBigCfg configuration, see its definition
#define BU_LAUNCH_SORTED(T)                                                                                                             \
    if (grid_cap == 0 && nb <= (size_t)1024 * ctx->cu_count) {                                                                         \
        /* at most one tile per CU: 16 waves on it (BC7 1 Ki blocks 4.32 -> 3.92 us, 2^16 5.16 -> 4.80, 2^18 5.70 -> 5.41; */          \
        /* ETC1 6.76 -> 6.47, 7.95 -> 7.64, 8.82 -> 8.54) */                                                                           \
        BU_GO(T, 1024, 1, 1, false, false, 0, (unsigned)((nb + 1023) / 1024), (unsigned)ctx->cu_count, 1024u);                                   \
    } else if (grid_cap == 0 && (many || BuBigCfg<T>::ALL_SIZES)) {                                                                    \
        using C = BuBigCfg<T>;                                                                                                          \
        const size_t tile_rt = bu_balanced_tile((size_t)C::WGS * C::BPT, nb, (size_t)ctx->cu_count * C::WG_PER_CU, C::DYN_TILE);        \
        const size_t btiles = (nb + tile_rt - 1) / tile_rt;                                                                             \
        const size_t bcap = bu_x_bcap((size_t)ctx->cu_count * C::WG_PER_CU);                                                            \
        /* generation priorities (kernel, `cus`) only when every workgroup walks the same number of tiles: with 1.25 tiles per */      \
        /* slot the one-tile generations run ahead of the two-tile ones (1.25 Mi blocks BC7 13.06 -> 11.57 us, ASTC 13.5 -> 11.0) */   \
        const unsigned pcus = (btiles <= bcap || btiles % bcap == 0) ? (unsigned)ctx->cu_count : 0u;                                    \
        if constexpr (C::NT > 1) {                                                                                                      \
            /* every load up front: exact grids of whole rectangular tiles only */                                                      \
            if (btiles % C::NT == 0 && btiles / C::NT <= bcap && rect_ok((size_t)C::WGS * C::BPT / RW)) {                               \
                hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, C::WGS, C::BPT, C::MINW, false, false, 0, BU_LAYOUT_RECT, C::NT>), dim3((unsigned)(btiles / C::NT)), \
                                   dim3(C::WGS), 0, stream, pin, pout, (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, pcus, rect_magic BU_STAMP_PASS); \
                break;                                                                                                                  \
            }                                                                                                                           \
        }                                                                                                                               \
        if constexpr (BU_GLDS_ON && ((T) == BU_TGT_BC7 || (T) == BU_TGT_ASTC) && C::WGS * C::BPT == 1024) {                            \
            /* one tile per workgroup: the tile goes straight into LDS (kernel, GLDS) */                                               \
            if (btiles <= bcap) {                                                                                                       \
                if (rect_ok((size_t)1024 / RW))                                                                                         \
                    hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, C::WGS, C::BPT, C::MINW, false, false, 0, BU_LAYOUT_RECT, 1, true>), dim3((unsigned)btiles), \
                                       dim3(C::WGS), 0, stream, pin, pout, (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, pcus, rect_magic BU_STAMP_PASS); \
                else                                                                                                                    \
                    hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, C::WGS, C::BPT, C::MINW, false, false, 0, BU_LAYOUT_STRIP, 1, true>), dim3((unsigned)btiles), \
                                       dim3(C::WGS), 0, stream, pin, pout, (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, pcus, 1024u BU_STAMP_PASS); \
                break;                                                                                                                  \
            }                                                                                                                           \
        }                                                                                                                               \
        BU_GO(T, C::WGS, C::BPT, C::MINW, C::PREFETCH, C::DIRECT, C::SKEW, (unsigned)(btiles < bcap ? btiles : bcap), pcus, (unsigned)tile_rt); \
    } else if (grid_cap == 0) {                                                                                                         \
        /* fewer than two 1024-block tiles per CU: 8 waves per tile, every tile resident (ETC1 at 2^16 blocks: 14.1 -> 11.3 us) */     \
        BU_GO(T, 512, 2, 1, false, false, 0, (unsigned)((nb + 1023) / 1024), (unsigned)ctx->cu_count, 1024u);                                    \
    } else                                                                                                                              \
        hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, BU_SORT_WGS, BU_SORT_BPT>), dim3(sgrid), dim3(BU_SORT_WGS), 0, stream, pin, pout,  \
                           (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)(BU_SORT_WGS * BU_SORT_BPT) BU_STAMP_PASS);
            // ETC1 / ETC2: three 1024-block workgroups (83 / 99 VGPRs) are resident per CU; up to there every tile of the small
            // shape runs at once and beats the 4096-block shape (2^19 blocks: 12.7 against 18.2 us, 786 432: 16.7 / 18.6),
            // beyond it the small shape needs a second round of workgroups (917 504 blocks: 21.5 against 18.9 us)
            const bool many = nb > (size_t)3 * 1024 * (size_t)ctx->cu_count;
            switch (target) {
            case BU_TARGET_ASTC: BU_LAUNCH_SORTED(BU_TGT_ASTC) break;
            case BU_TARGET_BC7: BU_LAUNCH_SORTED(BU_TGT_BC7) break;
            case BU_TARGET_ETC1: BU_LAUNCH_SORTED(BU_TGT_ETC1) break;
            case BU_TARGET_RGBA32: {
                // 64 B of output per block: results return through a 64 KiB LDS tile (1024 blocks x 4 rows, the input tile
                // aliased into row 0) so the image rows leave as coalesced 1 KiB stores; persistent workgroups walk their
                // tiles with prefetch, two per CU.  Up to 3 Mi blocks 1024 threads per tile (32 waves per CU: 2^18 blocks
                // 7.8 -> 7.2 us, 2^20 17.95 -> 16.9, 2^21 35.0 -> 33.75), above that 512 threads x 2 blocks (2^22 blocks 62.7
                // against 64.4 us, 2^24 252 against 265).  The zero-copy launches (grid_cap) keep the 512 x 2 shape.
                constexpr size_t rtile = 1024;
                const size_t rtiles = (nb + rtile - 1) / rtile;
                const size_t rcap = grid_cap ? (size_t)grid_cap : (size_t)ctx->cu_count * BU_RGBA_WG_PER_CU;
                const unsigned rgrid = (unsigned)(rtiles < rcap ? rtiles : rcap);
                // generation priorities only when every workgroup walks at least two tiles (2^19 blocks 10.7 -> 10.3 us and
                // 786 432 blocks 15.75 -> 14.24 without them, 2^20 blocks 16.7 against 18.7 with them)
                const unsigned rcus = rtiles >= 2 * (size_t)rgrid ? (unsigned)ctx->cu_count : 0u;
                if (grid_cap == 0 && nb <= ((size_t)3 << 20))
                    BU_GO(BU_TGT_RGBA, 1024, 1, 1, BU_RGBA_PREFETCH, false, BU_RGBA_SKEW, rgrid, rcus, (unsigned)rtile);
                else
                    BU_GO(BU_TGT_RGBA, 512, 2, 1, BU_RGBA_PREFETCH, false, BU_RGBA_SKEW, rgrid, rcus, (unsigned)rtile);
            } break;
            default: BU_LAUNCH_SORTED(BU_TGT_ETC2) break;
            }
#undef BU_LAUNCH_SORTED
#undef BU_GO
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

