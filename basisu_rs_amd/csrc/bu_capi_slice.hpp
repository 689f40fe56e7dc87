// C ABI, slice level: context, status strings, UASTC slice / per-block entry points (host and device pointers), page-locked
// buffers, the ETC1S back-end entry points.  Part of the single translation unit bu_hip.hip.
#pragma once
extern "C" {

size_t bu_target_block_bytes(bu_target target)
{
    switch (target) {
    case BU_TARGET_ASTC:
    case BU_TARGET_BC7:
    case BU_TARGET_ETC2: return 16;
    case BU_TARGET_ETC1: return 8;
    case BU_TARGET_RGBA32: return 64;
    default: return 0;
    }
}

const char* bu_status_string(bu_status st)
{
    switch (st) {
    case BU_OK: return "ok";
    case BU_ERR_INVALID_MODE: return "invalid mode index";                                        // uastc.rs:336
    case BU_ERR_INVALID_PATTERN: return "block pattern is not valid";                             // uastc.rs:364
    case BU_ERR_LENGTH: return "data length is not divisible by UASTC block size (16)";           // uastc.rs:56
    case BU_ERR_OUTPUT_SIZE: return "output buffer too small";
    case BU_ERR_ARGUMENT: return "invalid argument";
    case BU_ERR_INDEX_RANGE: return "ETC1S endpoint or selector index out of range";
    case BU_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
    case BU_ERR_HIP: return "HIP runtime error";
    case BU_ERR_SIG: return "Sig mismatch, not a Basis Universal file";                                  // basis.rs:309
    case BU_ERR_HEADER_TRUNCATED: return "Expected at least 77 byte header";                              // basis.rs:313
    case BU_ERR_HEADER_SIZE: return "File specified unexpected header size, expected 77";                 // basis.rs:323
    case BU_ERR_HEADER_CRC: return "Header CRC16 failed";                                                 // basis.rs:332
    case BU_ERR_DATA_CRC: return "Data CRC16 failed";                                                     // basis.rs:12
    case BU_ERR_TEX_FORMAT: return "Unknown texture format";                                              // basis.rs:404
    case BU_ERR_SLICE_DESC: return "Expected 23 byte slice desc";                                         // basis.rs:350
    case BU_ERR_ALPHA_SLICES: return "alpha slice layout is invalid (odd slice count, missing alpha flag or size mismatch)";  // basis.rs:19,29,34
    case BU_ERR_UNSUPPORTED: return "not implemented for this texture format";                            // unimplemented!()
    case BU_ERR_BASISLZ: return "BasisLZ stream is invalid";
    case BU_ERR_BOUNDS: return "offset outside the file or invalid stream state";
    default: return "unknown status";
    }
}

const char* bu_last_error(const bu_context* ctx) { return ctx ? ctx->err : "no context"; }

bu_status bu_context_create(int device, bu_context** out_ctx)
{
    if (!out_ctx) return BU_ERR_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return BU_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BU_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BU_ERR_NO_DEVICE;  // the code object is gfx950-only
    bu_context* ctx = new (std::nothrow) bu_context();
    if (!ctx) return BU_ERR_HIP;
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        // LDS a workgroup of this device may ask for (a partition mode or a runtime with a smaller limit than gfx950's 160 KiB simply
        // lowers the codebook size up to which the LDS-staged ETC1S kernels are used)
        int max_lds = 0, optin = 0;
        if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || max_lds <= 0) {
            (void)hipGetLastError();
            max_lds = 64 * 1024;
        }
        // (the figure a kernel can opt in to with hipFuncSetAttribute, where the runtime reports one)
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess && optin > max_lds) max_lds = optin;
        else (void)hipGetLastError();
        const size_t margin = 8 * 1024;  // the kernels' static LDS and allocation granularity
        const size_t lim = (size_t)max_lds > margin ? (size_t)max_lds - margin : 0;
        ctx->etc1s_lds_limit = lim < BU_ETC1S_LDS_MAX ? lim : BU_ETC1S_LDS_MAX;
    }
    {
        const char* e = getenv("BU_ENQUEUE_THREADS");
        ctx->single_thread_enqueue.store(e && e[0] == '0', std::memory_order_relaxed);
    }
    bu_status st = BU_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { st = BU_ERR_NO_DEVICE; break; }
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_tables), sizeof(BuTablesAll)) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_status), 64) != hipSuccess) { st = BU_ERR_HIP; break; }  // (eight words; [0] serves the host-pointer entry points)
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_tickets), (9 + BU_FOREIGN_TICKET_SETS) * BU_TICKET_WORDS * 4) != hipSuccess || hipMemset(ctx->d_tickets, 0, (9 + BU_FOREIGN_TICKET_SETS) * BU_TICKET_WORDS * 4) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_status), 64, hipHostMallocDefault) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->hd_status), ctx->h_status, 0) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { st = BU_ERR_HIP; break; }
        BuTablesAll* h = new (std::nothrow) BuTablesAll();
        if (!h) { st = BU_ERR_HIP; break; }
        bu_build_tables(h);
        hipError_t e = hipMemcpy(ctx->d_tables, h, sizeof(BuTablesAll), hipMemcpyHostToDevice);
        delete h;
        if (e != hipSuccess) { st = BU_ERR_HIP; break; }
        // CRC tables of the device-side data CRC: registers after (byte, k zero bytes), and x^(2048 k) for the in-piece fold
        BuCrcTables ct;
        for (int b = 0; b < 256; b++)
            for (int k = 0; k < 4; k++) {
                const uint8_t msg[4] = {(uint8_t)b, 0, 0, 0};
                ct.t[k][b] = bu_host::crc16_raw(msg, (size_t)k + 1, 0);
            }
        const uint16_t x2048 = bu_host::crc16_shift(1, 256);
        ct.pw[0] = 1;
        for (int k = 1; k < 256; k++) ct.pw[k] = bu_host::crc16_gf_mul(ct.pw[k - 1], x2048);
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_crc_tables), sizeof(BuCrcTables)) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipMemcpy(ctx->d_crc_tables, &ct, sizeof(ct), hipMemcpyHostToDevice) != hipSuccess) { st = BU_ERR_HIP; break; }
    } while (0);
    if (st != BU_OK) {
        bu_context_destroy(ctx);
        return st;
    }
    *out_ctx = ctx;
    return BU_OK;
}

void bu_context_destroy(bu_context* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_crc_tables) (void)hipFree(ctx->d_crc_tables);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
    if (ctx->d_in) (void)hipFree(ctx->d_in);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_aux) (void)hipFree(ctx->d_aux);
    if (ctx->h_idx) (void)hipHostFree(ctx->h_idx);
    free(ctx->lex_buf);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->ev_start)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_end)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->probe_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->probe_ev0) (void)hipEventDestroy(ctx->probe_ev0);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (hipStream_t es : ctx->extra_streams)
        if (es) (void)hipStreamDestroy(es);
    delete ctx;
}

bu_status bu_status_word_reset(bu_context* ctx, uint64_t* d_status, void* stream)
{
    if (!ctx || !d_status) return BU_ERR_ARGUMENT;
    hipLaunchKernelGGL(bu_status_reset_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long*>(d_status), 1u);
    BU_HIP(ctx, hipGetLastError());
    return BU_OK;
}

bu_status bu_status_word_decode(uint64_t word, uint64_t* first_bad_block)
{
    if (word == BU_STATUS_WORD_CLEAR) return BU_OK;
    // a report is (block << 8 | status) with status 1 or 2 (6 for ETC1S): anything else was never reset or was overwritten
    const unsigned st = (unsigned)(word & 0xFFu);
    if (st != BU_ERR_INVALID_MODE && st != BU_ERR_INVALID_PATTERN && st != BU_ERR_INDEX_RANGE) return BU_ERR_ARGUMENT;
    if (first_bad_block) *first_bad_block = word >> 8;
    return static_cast<bu_status>(st);
}

bu_status bu_host_alloc(bu_context* ctx, size_t bytes, void** out_ptr)
{
    if (!ctx || !out_ptr) return BU_ERR_ARGUMENT;
    *out_ptr = nullptr;
    if (bytes == 0) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BU_HIP(ctx, hipHostMalloc(out_ptr, bytes, hipHostMallocDefault));
    return BU_OK;
}

bu_status bu_host_free(bu_context* ctx, void* ptr)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    if (!ptr) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BU_HIP(ctx, hipHostFree(ptr));
    return BU_OK;
}

bu_status bu_uastc_transcode_device(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out,
                                    size_t blocks_per_row, uint64_t block_index_base, uint64_t* d_status, void* stream)
{
    if (!ctx || (n_blocks && (!d_in || !d_out))) return BU_ERR_ARGUMENT;
    if (bu_target_block_bytes(target) == 0) return BU_ERR_ARGUMENT;
    if (target == BU_TARGET_RGBA32 && blocks_per_row == 0) return BU_ERR_ARGUMENT;
    // the kernels store pixel rows 4*by+1..3 at the full image pitch: a ragged last block row would land past 64*n_blocks bytes
    if (target == BU_TARGET_RGBA32 && n_blocks % blocks_per_row != 0) return BU_ERR_ARGUMENT;
    return bu_launch_uastc(ctx, target, d_in, n_blocks, d_out, blocks_per_row, block_index_base, d_status, static_cast<hipStream_t>(stream), 0, -1);
}

}  // extern "C"
namespace {
// argument check of the batch entry points, then the slices merged into runs
bu_status bu_batch_runs(bu_context* ctx, bu_target target, size_t n_slices, const void* const* d_in, const size_t* n_blocks, void* const* d_out,
                        size_t blocks_per_row, const uint64_t* index_base, std::vector<BuRun>& runs)
{
    if (!ctx || (n_slices && (!d_in || !n_blocks || !d_out))) return BU_ERR_ARGUMENT;
    const size_t bb = bu_target_block_bytes(target);
    if (bb == 0 || (target == BU_TARGET_RGBA32 && blocks_per_row == 0)) return BU_ERR_ARGUMENT;
    for (size_t i = 0; i < n_slices; i++) {  // every argument is checked before the first launch
        if (n_blocks[i] && (!d_in[i] || !d_out[i])) return BU_ERR_ARGUMENT;
        if (target == BU_TARGET_RGBA32 && n_blocks[i] % blocks_per_row != 0) return BU_ERR_ARGUMENT;
    }
    bu_merge_runs(n_slices, d_in, n_blocks, d_out, bb, index_base, runs);
    return BU_OK;
}

// runs[0 .. n_runs) on ONE stream: one run is the plain launch; several runs at unrelated addresses are ONE launch per BU_MULTI_RUNS runs,
// the run table in the kernel arguments (kernel layout MULTI).  policy: BU_POLICY_* of the plain launches, -1 = the context's.
bu_status bu_launch_runs(bu_context* ctx, bu_target target, const BuRun* runs, size_t n_runs, size_t blocks_per_row, uint64_t* d_status, hipStream_t s, int policy)
{
    if (n_runs == 0) return BU_OK;
    if (n_runs == 1) return bu_launch_uastc(ctx, target, runs[0].in, runs[0].n, runs[0].out, blocks_per_row, runs[0].base, d_status, s, 0, policy);
    // Launching the runs one by one is bound by the ~4 us of host time per launch whatever the number of streams (64 slices
    // of 65 536 blocks: 290 us on one stream, 230-260 us on 2-8, profiles/r03_small_slices_streams_vs_one_launch.txt).
    // (a run too long for the table's 32-bit fields -- 2^32 blocks or more -- never enters it: it goes out as the plain launch
    // below, which cuts it into pieces of 2^26 blocks, exactly as it would on its own)
    BU_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long* stw = reinterpret_cast<unsigned long long*>(d_status);
    for (size_t r0 = 0; r0 < n_runs;) {
        BuRunTable tb;
        size_t k = 0, used = 0, n_tiles = 0;  // table entries, runs consumed, tiles
        bool all_whole = true;  // every run of this launch is tiled as whole rectangles: the kernel variant without per-lane validity tests
        const size_t bb = bu_target_block_bytes(target);
        // BC7 / ASTC / RGBA32: a run that is whole 64 x 16-block rectangles of a power-of-two grid is tiled that way -- the caller's blocks_per_row if it is one, else (block-
        // linear targets; RGBA32 is an image and has only its real pitch) a virtual pitch (bu_launch_uastc has the story: 16 segments of 1 KiB at >= 4 KiB pitch load faster than
        // 16 KiB in a row; multi-run launch over 32 slices of 2^20 blocks 6.0 -> 5.6 us per slice; RGBA32 14.8 -> 12.9: profiles/r06_ab_rgba_multi_run_rectangles.txt).
        // *whole_pitch = the pitch n blocks are whole rectangles of (0: none); returns the largest pitch of the list with at least `min_rows16` tile rows in n (ragged runs: their
        // whole PREFIX goes out as an entry of its own, the remainder as strips -- RGBA32 only: 64 ragged images of 1021 x 1024 blocks 15.0 -> 13.1 us per image, BC7 / ASTC unmoved:
        // profiles/r06_ab_rgba_multi_run_rectangles.txt)
        const bool rect_target = target == BU_TARGET_BC7 || target == BU_TARGET_ASTC || target == BU_TARGET_RGBA32;
        const size_t real = (blocks_per_row >= 128 && (blocks_per_row & (blocks_per_row - 1)) == 0 && blocks_per_row <= ((size_t)1 << 20)) ? blocks_per_row : 0;
        const size_t pitches[5] = {real, target == BU_TARGET_RGBA32 ? (size_t)0 : (size_t)1024, target == BU_TARGET_RGBA32 ? (size_t)0 : (size_t)2048,
                                   target == BU_TARGET_RGBA32 ? (size_t)0 : (size_t)512, target == BU_TARGET_RGBA32 ? (size_t)0 : (size_t)256};
        auto shift_of = [](size_t v) {
            uint32_t sh = 0;
            while (((size_t)BU_RECT_W << sh) < v) sh++;
            return sh;
        };
        // ETC1 / ETC2 batches of 2^20 blocks or more in runs long enough for them: 2048-block tiles, ONE tile per workgroup, dealt by the hardware dispatcher (512 x 4 under the
        // shared shape's launch bounds, two resident per CU) -- what the plain launch does from 2^20 blocks on (bu_launch_sorted): 64 slices of 2^20 blocks in separate
        // allocations 13.8 / 17.5 -> see profiles/r06_ab_etc_one_tile_workgroups.txt.  (Many short runs keep 1024-block tiles: a run's last tile is partly empty.)
        size_t tile = 1024;
        if (target == BU_TARGET_ETC1 || target == BU_TARGET_ETC2) {
            size_t total = 0, tiles2 = 0;
            for (size_t i = r0; i < n_runs && i < r0 + BU_MULTI_RUNS; i++) {
                total += runs[i].n;
                tiles2 += (runs[i].n + 2047) / 2048;
            }
            if (total >= ((size_t)1 << 20) && tiles2 * 2048 <= total + total / 8) tile = 2048;
        }
        auto emit = [&](const uint8_t* in, uint8_t* out, uint64_t base, size_t n, uint32_t vshift) {
            all_whole = all_whole && vshift != BU_RUN_STRIPS;
            tb.run[k] = BuRunDesc{reinterpret_cast<const uint4*>(in), out, base, (uint32_t)n, vshift};
            tb.first_tile[k] = (uint32_t)n_tiles;
            n_tiles += (n + tile - 1) / tile;
            k++;
        };
        for (; r0 + used < n_runs && k < BU_MULTI_RUNS; used++) {
            const BuRun& r = runs[r0 + used];
            const size_t t = (r.n + tile - 1) / tile;
            if (n_tiles + t + 1 >= (((size_t)1 << 32) / tile)) break;  // (tiles x tile size is the launch's 32-bit block count)
            size_t whole_pitch = 0, prefix_pitch = 0;
            if (rect_target) {
                for (const size_t v : pitches)
                    if (v && r.n % (16 * v) == 0) {
                        whole_pitch = v;
                        break;
                    }
                if (!whole_pitch && target == BU_TARGET_RGBA32 && k + 2 <= BU_MULTI_RUNS)  // (BC7 / ASTC gain nothing from the split: 5.95 / 6.3 us per slice either way)
                    for (const size_t v : pitches)
                        if (v && r.n >= 8 * 16 * v) {  // (at least eight tile rows of rectangles, or the split is not worth an entry)
                            prefix_pitch = v;
                            break;
                        }
            }
            if (whole_pitch) {
                emit(r.in, r.out, r.base, r.n, shift_of(whole_pitch));
            } else if (prefix_pitch) {
                const size_t prefix = r.n / (16 * prefix_pitch) * (16 * prefix_pitch);
                emit(r.in, r.out, r.base, prefix, shift_of(prefix_pitch));
                emit(r.in + prefix * 16, r.out + prefix * bb, r.base + prefix, r.n - prefix, BU_RUN_STRIPS);
            } else {
                emit(r.in, r.out, r.base, r.n, BU_RUN_STRIPS);
            }
        }
        if (used <= 1) {  // a run on its own (the last one of a long batch, or one of 2^32 blocks): the plain launch
            bu_status st = bu_launch_uastc(ctx, target, runs[r0].in, runs[r0].n, runs[r0].out, blocks_per_row, runs[r0].base, d_status, s, 0, policy);
            if (st) return st;
            r0 += 1;
            continue;
        }
        for (size_t i = k; i < BU_MULTI_RUNS + 32; i++) tb.first_tile[i] = 0xFFFFFFFFu;
        for (size_t i = k; i < BU_MULTI_RUNS; i++) tb.run[i] = BuRunDesc{nullptr, nullptr, 0, 0u, BU_RUN_STRIPS};
        // Shapes, as the plain launcher picks them by size (bu_context.hpp): at most one tile per CU 1024 threads on it; beyond that 512 x 2.
        // BC7 / ASTC / RGBA32 batches of more tiles than fit the chip at once run as a PERSISTENT grid (four / four / two workgroups per CU)
        // whose workgroups walk the tiles of all runs with the next tile's loads in flight -- a batch of large slices in separate
        // allocations is then one long launch that overlaps its own loads and compute (two 2^20-block slices 7.9 us each, eight 6.4, against
        // 8.4 for plain launches one after another and 9.2-10.2 through the round-4 table kernel without the prefetch).  ETC1 / ETC2 walk the same way with
        // TWO workgroups per CU (97 / 119 VGPRs: 16 waves are what fits): 64 slices of 2^20 blocks in separate allocations 15.4 / 19.4 -> 13.8 / 17.5 us per
        // slice against one-tile workgroups dealt by the dispatcher (tools/exp/etc_multi_persist.sh; their plain large shape sorts 4096-block tiles, the
        // table numbers 1024-block ones: 13.4 / 17.1 when the slices are adjacent and merge into one run).
        const bool one_per_cu = tile == 1024 && n_tiles <= (size_t)ctx->cu_count;
        // Launch policy of a grouped launch.  Under the shared policy (launches of other streams run beside this one: bu_uastc_transcode_batch_in_flight
        // with groups of small runs) the PERSISTENT grid is capped at about half of every CU -- two workgroups of 512 threads for BC7 / ASTC (16 of the 32 wave
        // slots, 56 of the 160 KiB; four of 256 in the whole-tile shape below), one for RGBA32 -- so that two such launches fit side by side (ETC1 / ETC2: one of the two that fit; 64 slices of
        // 65 536 blocks on four streams 66.3 / 80.2 -> 65.0 / 78.0 us, tools/exp/etc_small_slices.sh); the one-tile-per-CU shape is the same under both
        // policies (a tile's 1024 threads cannot be halved).
        int pol = policy < 0 ? ctx->launch_policy.load(std::memory_order_relaxed) : policy;
        if (pol == BU_POLICY_AUTO) pol = one_per_cu ? (int)BU_POLICY_EXCLUSIVE : bu_auto_policy(ctx, s);
        const bool half = pol == BU_POLICY_SHARED || pol == BU_POLICY_SHARED_FEW;
        auto go = [&](auto tgt) {
            constexpr int T = decltype(tgt)::value;
            constexpr bool PERSIST = true, ETC = T == BU_TGT_ETC1 || T == BU_TGT_ETC2;
            // BC7 / ASTC batches whose runs are all whole rectangular tiles (the variant without validity tests): 256 x 4, FIVE workgroups per CU (63 / 76 VGPRs, 31 / 27 KiB),
            // four under the shared policy.  ASTC's 512 x 2 form of that variant sits at exactly 64 VGPRs -- the compiler gets there by serialising -- and ran 64 atlases in
            // separate allocations at 5.95-6.0 us per atlas where the plain kernel does 5.5: 5.59-5.63 in this shape (64 / 512 slices of 65 536 blocks 32.8 / 243 -> 30.6 / 219 us);
            // BC7 5.57-5.75 -> 5.52-5.57, 512 small slices 224 -> 215 (in flight 202 -> 191): profiles/r06_ab_multi_run_256x4.txt.  Everything else 512 x 2, four / two per CU.
            constexpr bool WHOLE_T = T == BU_TGT_BC7 || T == BU_TGT_ASTC;
            const bool whole = WHOLE_T && all_whole;
            const size_t cap = (size_t)ctx->cu_count * (T == BU_TGT_RGBA ? (half ? 1 : 2) : ETC ? (half ? 1 : 2) : whole ? (half ? 4 : 5) : (half ? 2 : 4));
            const unsigned grid = (unsigned)(n_tiles < cap ? n_tiles : cap);
            // tile tickets for the long walks of a persistent grid that has the chip to itself, as bu_go_big (a batch of 64 slices of 2^20 blocks in
            // separate allocations: 64 tiles per workgroup)
            unsigned* const ticket = (PERSIST && !half && n_tiles >= BU_TICKET_MIN_WALK * (size_t)grid) ? bu_ticket_for(ctx, s) : nullptr;
            if (ETC && tile == 2048) {
                if constexpr (ETC)
                    hipLaunchKernelGGL((bu_uastc_multi_kernel<T, 512, 4>), dim3((unsigned)n_tiles), dim3(512), 0, s, tb, (unsigned)n_tiles, (unsigned)blocks_per_row, stw, ctx->d_tables,
                                       (unsigned*)nullptr);
            } else if (one_per_cu)
                hipLaunchKernelGGL((bu_uastc_multi_kernel<T, 1024, 1>), dim3(grid), dim3(1024), 0, s, tb, (unsigned)n_tiles, (unsigned)blocks_per_row, stw, ctx->d_tables, (unsigned*)nullptr);
            else if (whole) {
                if constexpr (WHOLE_T)
                    hipLaunchKernelGGL((bu_uastc_multi_kernel<T, 256, 4, PERSIST, true>), dim3(grid), dim3(256), 0, s, tb, (unsigned)n_tiles, (unsigned)blocks_per_row, stw, ctx->d_tables, ticket);
            } else
                hipLaunchKernelGGL((bu_uastc_multi_kernel<T, 512, 2, PERSIST>), dim3(grid), dim3(512), 0, s, tb, (unsigned)n_tiles, (unsigned)blocks_per_row, stw, ctx->d_tables, ticket);
        };
        switch (target) {
        case BU_TARGET_ASTC: go(std::integral_constant<int, BU_TGT_ASTC>()); break;
        case BU_TARGET_BC7: go(std::integral_constant<int, BU_TGT_BC7>()); break;
        case BU_TARGET_ETC1: go(std::integral_constant<int, BU_TGT_ETC1>()); break;
        case BU_TARGET_ETC2: go(std::integral_constant<int, BU_TGT_ETC2>()); break;
        default: go(std::integral_constant<int, BU_TGT_RGBA>()); break;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return bu_fail(ctx, e, "multi-run launch");
        r0 += used;
    }
    return BU_OK;
}
}  // namespace
extern "C" {

bu_status bu_uastc_transcode_batch_device(bu_context* ctx, bu_target target, size_t n_slices, const void* const* d_in,
                                          const size_t* n_blocks, void* const* d_out, size_t blocks_per_row,
                                          const uint64_t* index_base, uint64_t* d_status, void* stream)
{
    std::vector<BuRun> runs;
    const bu_status st = bu_batch_runs(ctx, target, n_slices, d_in, n_blocks, d_out, blocks_per_row, index_base, runs);
    if (st) return st;
    return bu_launch_runs(ctx, target, runs.data(), runs.size(), blocks_per_row, d_status, static_cast<hipStream_t>(stream), -1);
}

}  // extern "C"
namespace {
// launches groups[j] for j = first, first + step, ... on stream `s` (one stream's share of a pipelined batch, in order)
bu_status bu_enqueue_groups(bu_context* ctx, bu_target target, const std::vector<BuRun>& runs, const std::vector<BuRun>& extra, const std::vector<BuLaunchGroup>& groups,
                            size_t first, size_t step, size_t blocks_per_row, uint64_t* d_status, hipStream_t s, int policy)
{
    for (size_t j = first; j < groups.size(); j += step) {
        const BuLaunchGroup& g = groups[j];
        const BuRun* r = g.first >= runs.size() ? &extra[g.first - runs.size()] : &runs[g.first];
        const bu_status st = bu_launch_runs(ctx, target, r, g.count, blocks_per_row, d_status, s, policy);
        if (st) return st;
    }
    return BU_OK;
}

// From this many launches on, a pipelined batch is enqueued by one host thread per stream (the calling thread takes stream 0): one enqueue costs
// 4.6-4.8 us of host time against a period of 5.5-5.7 us per 2^20-block launch, so a single enqueueing thread is 20 % from setting the pace
// itself (and does set it under a profiler, where an enqueue costs 6-8 us).  Starting the threads costs ~0.1 ms per call (64 launches in one
// call: 7.4 us per atlas with them, 6.3 without): worth it from a millisecond and a half of GPU work.  The order inside every stream is the plan's; between streams it is whatever the threads make it, which the
// results cannot depend on (independent slices) and the status word does not (a minimum over failing blocks).
constexpr size_t BU_ENQUEUE_THREADS_MIN_LAUNCHES = 256;

// the launches of a planned batch on the context's streams 0..n_streams-1, launch j on stream j % n_streams
bu_status bu_issue_in_flight(bu_context* ctx, bu_target target, const std::vector<BuRun>& runs, const std::vector<BuRun>& extra, const std::vector<BuLaunchGroup>& groups,
                             size_t blocks_per_row, uint64_t* d_status, int n_streams, int policy)
{
    hipStream_t ss[8];
    for (int i = 0; i < n_streams; i++) ss[i] = ctx->extra_streams[i].load(std::memory_order_acquire);
    if (n_streams > 1 && groups.size() >= BU_ENQUEUE_THREADS_MIN_LAUNCHES && !ctx->single_thread_enqueue.load(std::memory_order_relaxed)) {
        bu_status sts[8] = {BU_OK, BU_OK, BU_OK, BU_OK, BU_OK, BU_OK, BU_OK, BU_OK};
        std::vector<std::thread> th;
        int started = 1;  // (stream 0 is the calling thread's)
        for (int si = 1; si < n_streams; si++) {
            try {
                th.emplace_back([&, si] {
                    if (hipSetDevice(ctx->device) != hipSuccess) {
                        sts[si] = bu_fail(ctx, hipGetLastError(), "hipSetDevice");
                        return;
                    }
                    sts[si] = bu_enqueue_groups(ctx, target, runs, extra, groups, (size_t)si, (size_t)n_streams, blocks_per_row, d_status, ss[si], policy);
                });
                started++;
            } catch (const std::exception&) {  // (no thread to be had: this stream's share is enqueued by the calling thread below)
                break;
            }
        }
        sts[0] = bu_enqueue_groups(ctx, target, runs, extra, groups, 0, (size_t)n_streams, blocks_per_row, d_status, ss[0], policy);
        for (int si = started; si < n_streams; si++)
            sts[si] = bu_enqueue_groups(ctx, target, runs, extra, groups, (size_t)si, (size_t)n_streams, blocks_per_row, d_status, ss[si], policy);
        for (auto& t : th) t.join();
        for (int si = 0; si < n_streams; si++)
            if (sts[si]) return sts[si];
        return BU_OK;
    }
    for (size_t j = 0; j < groups.size(); j++) {
        const bu_status st = bu_enqueue_groups(ctx, target, runs, extra, groups, j, groups.size(), blocks_per_row, d_status, ss[j % (size_t)n_streams], policy);
        if (st) return st;
    }
    return BU_OK;
}

// ---- the BLOCKING entry points over one contiguous device-resident range (bu_uastc_transcode_device_sync, bu_array_transcode_sharded) -------
// One launch on the context's internal stream, exclusive shape, and from 16 tiles per workgroup on with TILE TICKETS (bu_go_big): a 2^25-block
// array 174 us = 0.77 of the HBM roofline where the fixed walk took 188.5 (0.71).  Round 6 first cut such a range into four launches in flight
// on the context's streams, joined on the host (what the round-5 verdict asked for); measured, an ISOLATED call gains nothing from that -- the four
// pieces start in phase, the last of them finishes alone in the half-CU shape, and the call took 213 us where one launch took 206
// (profiles/r06_range_in_flight_threshold.txt); the pipeline's 0.76-0.78 belongs to a STREAM of arrays (bu_uastc_transcode_batch_in_flight:
// call, call, ..., one wait), the tickets give it to a single one.
// The status word is the context's page-locked one: the host resets it with a plain store BEFORE it enqueues (the submission orders it in front
// of the kernel), failing blocks report with a system-scope atomic min, and the host reads it once the stream is idle -- a reset launch in front
// and a copy behind the kernel cost 5 us per call.  begin() only enqueues (several contexts -- devices -- are started before the first is waited
// for); end() waits and reads.  Caller holds ctx->lock (the status word is the context's).
struct BuRangeJob {
    bool started = false;
};
bu_status bu_range_begin(bu_context* ctx, bu_target target, const void* in, size_t nb, void* out, size_t bpr, uint64_t base, BuRangeJob* job)
{
    *job = BuRangeJob();
    if (nb == 0) return BU_OK;
    ctx->h_status[0] = BU_STATUS_WORD_CLEAR;
    const bu_status st = bu_launch_uastc(ctx, target, in, nb, out, bpr, base, reinterpret_cast<uint64_t*>(ctx->hd_status), ctx->stream, 0, BU_POLICY_EXCLUSIVE);
    if (st) return st;
    job->started = true;
    return BU_OK;
}
// waits for the job's stream (polling: a blocking wait adds the 10-20 us a sleeping host thread needs to wake up to every call -- a tenth of a
// 2^25-block array; after BU_RANGE_SPIN_MS of polling the wait turns into a blocking one) and gives the status word
constexpr double BU_RANGE_SPIN_MS = 5.0;
bu_status bu_range_end(bu_context* ctx, const BuRangeJob& job, uint64_t* word)
{
    *word = BU_STATUS_WORD_CLEAR;
    if (!job.started) return BU_OK;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double, std::milli>(BU_RANGE_SPIN_MS);
    for (unsigned k = 0;; k++) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return bu_fail(ctx, q, "hipStreamQuery");
        (void)hipGetLastError();
        if ((k & 63u) == 63u && std::chrono::steady_clock::now() > deadline) {
            BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            break;
        }
    }
    *word = ctx->h_status[0];
    return BU_OK;
}
}  // namespace
extern "C" {

// The same loop at the rate of a PIPELINE of launches (include/basisu_hip.h): the runs are grouped into launches of about one 4096^2 atlas
// or more, a batch that makes fewer launches than streams has its largest runs cut into equal pieces, and launch j goes to context stream
// j % n_streams under the shared launch policy.  Only enqueues; bu_context_synchronize (or the streams) waits.
bu_status bu_uastc_transcode_batch_in_flight(bu_context* ctx, bu_target target, size_t n_slices, const void* const* d_in,
                                             const size_t* n_blocks, void* const* d_out, size_t blocks_per_row,
                                             const uint64_t* index_base, uint64_t* d_status, int n_streams)
{
    if (n_streams < 1 || n_streams > 8) return BU_ERR_ARGUMENT;
    std::vector<BuRun> runs;
    bu_status st = bu_batch_runs(ctx, target, n_slices, d_in, n_blocks, d_out, blocks_per_row, index_base, runs);
    if (st) return st;
    if (runs.empty()) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    // the streams, and how many launches they really keep in flight in this process (creation-time probe, bu_streams.hpp)
    int effective = n_streams;
    st = bu_ctx_in_flight_streams(ctx, n_streams, &effective, nullptr);
    if (st) return st;
    // Degraded: the streams share hardware queues whatever the library tried (plain streams AND CU-mask streams; or BU_STREAM_MODE=plain), so
    // "n launches in flight" would run as `effective`.  A pipeline two deep is 7.3 us per 2^20-block BC7 launch; ONE stream-ordered batch launch --
    // a persistent grid that walks all the runs with the next tile's loads in flight -- is 6.4 from eight such slices on, and a run that is one
    // allocation is a single launch at 0.70 of the roofline.  BC7 / ASTC / RGBA32 have that kernel; ETC1 / ETC2 keep the shallow pipeline (13.1 / 16.3
    // us with two in flight against 17.7 / 22.1 one at a time).  bu_context_query_in_flight tells the caller which of the two it got.
    const bool persist_target = target == BU_TARGET_BC7 || target == BU_TARGET_ASTC || target == BU_TARGET_RGBA32;
    if (n_streams > 1 && effective < n_streams && effective <= 2 && persist_target)
        return bu_launch_runs(ctx, target, runs.data(), runs.size(), blocks_per_row, d_status, ctx->extra_streams[0].load(std::memory_order_acquire), BU_POLICY_EXCLUSIVE);
    // launches of 2^20 .. 2^23 blocks (BC7 / ASTC: runs grouped up to 2^23 into multi-run launches; else up to 2^20); a batch with fewer launches than streams has
    // its largest runs cut (bu_batch_plan.hpp)
    std::vector<BuRun> extra;
    std::vector<BuLaunchGroup> groups;
    const size_t group_blocks = (target == BU_TARGET_BC7 || target == BU_TARGET_ASTC) ? (size_t)1 << 23 : (size_t)1 << 20;
    bu_plan_in_flight(runs, n_streams, blocks_per_row, bu_target_block_bytes(target), (size_t)BU_MULTI_RUNS, group_blocks, groups, extra);
    // launch j on stream j % n_streams, shared policy (a single launch gets the exclusive shape: nothing runs beside it)
    const int policy = groups.size() > 1 && n_streams > 1 ? BU_POLICY_SHARED : BU_POLICY_EXCLUSIVE;
    return bu_issue_in_flight(ctx, target, runs, extra, groups, blocks_per_row, d_status, n_streams, policy);
}

// What bu_uastc_transcode_batch_in_flight(..., n_streams) gets in THIS process (include/basisu_hip.h)
bu_status bu_context_query_in_flight(bu_context* ctx, int n_streams, int* out_effective_streams, int* out_stream_mode)
{
    if (!ctx || n_streams < 1 || n_streams > 8) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    int eff = n_streams, mode = 0;
    const bu_status st = bu_ctx_in_flight_streams(ctx, n_streams, &eff, &mode);
    if (st) return st;
    if (out_effective_streams) *out_effective_streams = eff;
    if (out_stream_mode) *out_stream_mode = mode == BU_STREAMS_CU_MASK ? BU_STREAM_QUEUE_CU_MASK : BU_STREAM_QUEUE_POOL;
    return BU_OK;
}

// bu_uastc_transcode_device that WAITS (include/basisu_hip.h): one exclusive launch (tile tickets on long walks), page-locked status word
bu_status bu_uastc_transcode_device_sync(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out, size_t blocks_per_row,
                                         uint64_t block_index_base, uint64_t* out_status_word)
{
    if (!ctx || (n_blocks && (!d_in || !d_out))) return BU_ERR_ARGUMENT;
    if (bu_target_block_bytes(target) == 0) return BU_ERR_ARGUMENT;
    if (target == BU_TARGET_RGBA32 && (blocks_per_row == 0 || n_blocks % blocks_per_row != 0)) return BU_ERR_ARGUMENT;
    if (out_status_word) *out_status_word = BU_STATUS_WORD_CLEAR;
    if (n_blocks == 0) return BU_OK;
    std::lock_guard<std::mutex> g(ctx->lock);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BuDrain drain(ctx);
    BuRangeJob job;
    bu_status st = bu_range_begin(ctx, target, d_in, n_blocks, d_out, blocks_per_row, block_index_base, &job);
    if (st) return st;
    uint64_t word = BU_STATUS_WORD_CLEAR;
    st = bu_range_end(ctx, job, &word);
    if (st) return st;
    drain.armed = false;
    if (out_status_word) *out_status_word = word;
    return BU_OK;
}

bu_status bu_uastc_transcode(bu_context* ctx, bu_target target, const uint8_t* in, size_t in_bytes, uint8_t* out,
                             size_t out_bytes, uint64_t* first_bad_block)
{
    if (target == BU_TARGET_RGBA32) return BU_ERR_ARGUMENT;  // uastc.rs:41-47 has no RGBA member; use bu_uastc_decode_to_rgba
    return bu_uastc_host(ctx, target, in, in_bytes, 1, out, out_bytes, first_bad_block);
}

bu_status bu_uastc_decode_to_rgba(bu_context* ctx, const uint8_t* in, size_t in_bytes, size_t blocks_per_row, uint8_t* out,
                                  size_t out_bytes, uint64_t* first_bad_block)
{
    if (blocks_per_row == 0) return BU_ERR_ARGUMENT;
    // the reference's image has exactly 64*n bytes (uastc.rs:95); a ragged last row would index past it
    // (Rust panics there), so require whole block rows
    if (in_bytes % 16 == 0 && (in_bytes / 16) % blocks_per_row != 0) return BU_ERR_ARGUMENT;
    return bu_uastc_host(ctx, BU_TARGET_RGBA32, in, in_bytes, blocks_per_row, out, out_bytes, first_bad_block);
}

// ---- launch policy of the slice-level device entry points (include/basisu_hip.h; shapes: bu_context.hpp, BuBigShape) ----
bu_status bu_context_set_launch_policy(bu_context* ctx, bu_launch_policy policy)
{
    if (!ctx || (policy != BU_LAUNCH_EXCLUSIVE && policy != BU_LAUNCH_SHARED && policy != BU_LAUNCH_AUTO)) return BU_ERR_ARGUMENT;
    static_assert((int)BU_LAUNCH_EXCLUSIVE == BU_POLICY_EXCLUSIVE && (int)BU_LAUNCH_SHARED == BU_POLICY_SHARED && (int)BU_LAUNCH_AUTO == BU_POLICY_AUTO, "one numbering");
    ctx->launch_policy.store((int)policy, std::memory_order_relaxed);
    return BU_OK;
}
bu_status bu_context_get_launch_policy(const bu_context* ctx, bu_launch_policy* out_policy)
{
    if (!ctx || !out_policy) return BU_ERR_ARGUMENT;
    *out_policy = static_cast<bu_launch_policy>(ctx->launch_policy.load(std::memory_order_relaxed));
    return BU_OK;
}

bu_status bu_context_stream(bu_context* ctx, int index, void** out_stream)
{
    if (!ctx || !out_stream || index < 0 || index >= 8) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    const bu_status st = bu_ctx_streams(ctx, index + 1);
    if (st) return st;
    *out_stream = ctx->extra_streams[index];
    return BU_OK;
}

bu_status bu_context_synchronize(bu_context* ctx)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->stream) BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; i++) {  // (published streams are final: bu_streams.hpp)
        hipStream_t e = ctx->extra_streams[i].load(std::memory_order_acquire);
        if (e) BU_HIP(ctx, hipStreamSynchronize(e));
    }
    return BU_OK;
}

// Do the context's streams 0..n_streams-1 really run side by side -- now, as opposed to when they were created (bu_streams.hpp: a sleeping wave
// on every stream behind a common event)?  *out_max_sharing = the largest number of the probed streams on one hardware queue.  Waits; ~0.4 ms.
bu_status bu_context_probe_streams(bu_context* ctx, int n_streams, int* out_max_sharing)
{
    if (!ctx || !out_max_sharing || n_streams < 1 || n_streams > 8) return BU_ERR_ARGUMENT;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    {
        const bu_status st = bu_ctx_streams(ctx, n_streams);
        if (st) return st;
    }
    std::lock_guard<std::mutex> g(ctx->stream_lock);  // (the probe events are the context's)
    hipStream_t ss[8];
    for (int i = 0; i < n_streams; i++) ss[i] = ctx->extra_streams[i].load(std::memory_order_acquire);
    return bu_probe_streams_locked(ctx, ss, n_streams, out_max_sharing);
}

// ---- per-block API (lib.rs:29-53) -----------------------------------------------------------------------------------------------
// One 16-byte block is not worth a kernel launch (upload + 1-block launch + download: tens of microseconds; the reference's own
// benchmark calls these 32 000 times, benches/benchmark.rs:66-98).  They run the product's OWN block code -- the mode-templated
// front-end and packers of bu_uastc_*.hpp, which this translation unit compiles for the host as well as for gfx950 (BU_DEV) --
// on the calling thread, against a host copy of the table blob the kernels stage in LDS.  A context is still required (the
// library has no CPU mode: bu_context_create fails without a gfx950 device); bu_block_api_on_device(ctx, 1) routes the calls
// through the one-lane-per-block kernel instead (what rounds 1-3 shipped; tests compare both with the reference's vectors).
bu_status bu_block_api_on_device(bu_context* ctx, int enable)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    ctx->block_api_on_device.store(enable != 0, std::memory_order_relaxed);
    return BU_OK;
}

#define BU_BLOCK_API(TGT, target_enum, out_bytes)                                                                    \
    if (!ctx || !in || !out) return BU_ERR_ARGUMENT;                                                                 \
    if (ctx->block_api_on_device.load(std::memory_order_relaxed))                                                    \
        return bu_uastc_host(ctx, target_enum, in, 16, 1, reinterpret_cast<uint8_t*>(out), out_bytes, nullptr);      \
    return bu_block_on_host<TGT>(in, out)

bu_status bu_unpack_uastc_block_to_rgba(bu_context* ctx, const uint8_t in[16], uint32_t out[16]) { BU_BLOCK_API(BU_TGT_RGBA, BU_TARGET_RGBA32, 64); }
bu_status bu_transcode_uastc_block_to_astc(bu_context* ctx, const uint8_t in[16], uint8_t out[16]) { BU_BLOCK_API(BU_TGT_ASTC, BU_TARGET_ASTC, 16); }
bu_status bu_transcode_uastc_block_to_bc7(bu_context* ctx, const uint8_t in[16], uint8_t out[16]) { BU_BLOCK_API(BU_TGT_BC7, BU_TARGET_BC7, 16); }
bu_status bu_transcode_uastc_block_to_etc1(bu_context* ctx, const uint8_t in[16], uint8_t out[8]) { BU_BLOCK_API(BU_TGT_ETC1, BU_TARGET_ETC1, 8); }
bu_status bu_transcode_uastc_block_to_etc2(bu_context* ctx, const uint8_t in[16], uint8_t out[16]) { BU_BLOCK_API(BU_TGT_ETC2, BU_TARGET_ETC2, 16); }
#undef BU_BLOCK_API

// ---- ETC1S ---------------------------------------------------------------------------------------
void bu_etc1s_selector_from_rows(const uint8_t rows[4], uint8_t out_entry[8]) { bu_host::selector_from_rows(rows, out_entry); }

// Dynamic LDS above 64 KiB has to be allowed per kernel, once (hipFuncSetAttribute).  May the LDS-staged kernel be used for `lds`
// bytes of codebooks on this context's device?  The first question per kernel asks the runtime for the most the kernels were
// measured with (BU_ETC1S_LDS_MAX); a device, partition mode or runtime that refuses is asked again for what the device itself
// reports (bu_context_create: etc1s_lds_limit); if that fails too the L2-gather kernels serve every size.  The granted size is
// remembered per kernel in an atomic (the device entry points may be called from several threads; asking twice is harmless):
// 0 = not asked yet, 1 = refused, otherwise the bytes allowed.
static bool bu_etc1s_staged_ok(bu_context* ctx, bool rgba, size_t lds)
{
    std::atomic<size_t>& state = ctx->etc1s_lds_state[rgba ? 1 : 0];
    size_t s = state.load(std::memory_order_acquire);
    if (s == 0) {
        const void* fn = rgba ? reinterpret_cast<const void*>(&bu_etc1s_staged_kernel<true>) : reinterpret_cast<const void*>(&bu_etc1s_staged_kernel<false>);
        s = 1;
        for (const size_t want : {(size_t)BU_ETC1S_LDS_MAX, ctx->etc1s_lds_limit}) {
            if (want < 4096 || want > BU_ETC1S_LDS_MAX) continue;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess) {
                s = want;
                break;
            }
            (void)hipGetLastError();  // (not sticky)
        }
        state.store(s, std::memory_order_release);
    }
    return s > 1 && lds <= s;
}

bu_status bu_etc1s_transcode_etc1_device(bu_context* ctx, const uint32_t* d_idx, size_t n_blocks, const uint32_t* d_endpoints,
                                         uint32_t n_endpoints, const void* d_selectors, uint32_t n_selectors, void* d_out,
                                         uint64_t* d_status, void* stream)
{
    if (!ctx || (n_blocks && (!d_idx || !d_endpoints || !d_selectors || !d_out))) return BU_ERR_ARGUMENT;
    if (n_blocks == 0) return BU_OK;
    const size_t lds = ((size_t)n_endpoints + n_selectors) * 4;
    if (n_blocks >= BU_ETC1S_STAGED_MIN && bu_etc1s_staged_ok(ctx, false, lds)) {
        // codebooks in LDS, one persistent workgroup per CU (with four blocks per lane and step in flight a second workgroup only
        // doubles the staging: 2^21 blocks 8.0 against 9.3 us, 2^22 12.95 / 13.2, 2^24 37.4 / 39.3)
        const unsigned per_cu = 1u;
        hipLaunchKernelGGL(bu_etc1s_staged_kernel<false>, dim3((unsigned)ctx->cu_count * per_cu), dim3(1024), lds, static_cast<hipStream_t>(stream), d_idx,
                           nullptr, 1u, n_blocks, d_endpoints, n_endpoints, static_cast<const uint2*>(d_selectors), n_selectors,
                           static_cast<uint8_t*>(d_out), reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
        BU_HIP(ctx, hipGetLastError());
        return BU_OK;
    }
    hipLaunchKernelGGL(bu_etc1s_etc1_kernel, dim3(bu_grid_for(n_blocks, ctx->cu_count)), dim3(BU_WG), 0, static_cast<hipStream_t>(stream), d_idx,
                       n_blocks, d_endpoints, n_endpoints, static_cast<const uint2*>(d_selectors), n_selectors, static_cast<uint2*>(d_out),
                       reinterpret_cast<unsigned long long*>(d_status));
    BU_HIP(ctx, hipGetLastError());
    return BU_OK;
}

bu_status bu_etc1s_decode_rgba_device(bu_context* ctx, const uint32_t* d_idx, const uint32_t* d_alpha_idx, size_t nbx, size_t nby,
                                      const uint32_t* d_endpoints, uint32_t n_endpoints, const void* d_selectors,
                                      uint32_t n_selectors, void* d_out, uint64_t* d_status, void* stream)
{
    const size_t n_blocks = nbx * nby;
    if (!ctx || (n_blocks && (!d_idx || !d_endpoints || !d_selectors || !d_out))) return BU_ERR_ARGUMENT;
    if (n_blocks == 0) return BU_OK;
    const size_t lds = ((size_t)n_endpoints + n_selectors + 256) * 4;
    if (n_blocks >= BU_ETC1S_STAGED_MIN && bu_etc1s_staged_ok(ctx, true, lds)) {
        hipLaunchKernelGGL(bu_etc1s_staged_kernel<true>, dim3((unsigned)ctx->cu_count), dim3(1024), lds, static_cast<hipStream_t>(stream), d_idx, d_alpha_idx,
                           (unsigned)nbx, n_blocks, d_endpoints, n_endpoints, static_cast<const uint2*>(d_selectors), n_selectors,
                           static_cast<uint8_t*>(d_out), reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
        BU_HIP(ctx, hipGetLastError());
        return BU_OK;
    }
    hipLaunchKernelGGL(bu_etc1s_rgba_kernel, dim3(bu_grid_for(n_blocks, ctx->cu_count)), dim3(BU_WG), 0, static_cast<hipStream_t>(stream), d_idx,
                       d_alpha_idx, (unsigned)nbx, n_blocks, d_endpoints, n_endpoints, static_cast<const uint2*>(d_selectors), n_selectors,
                       static_cast<uint4*>(d_out), reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
    BU_HIP(ctx, hipGetLastError());
    return BU_OK;
}

static bu_status bu_etc1s_host(bu_context* ctx, bool rgba, const uint32_t* idx, const uint32_t* alpha_idx, size_t nbx, size_t nby,
                               const uint32_t* endpoints, uint32_t n_ep, const uint8_t* selectors, uint32_t n_sel, uint8_t* out,
                               size_t out_bytes, uint64_t* first_bad)
{
    const size_t n = nbx * nby;
    if (!ctx || !out || (n && (!idx || !endpoints || !selectors))) return BU_ERR_ARGUMENT;
    const size_t bb = rgba ? 64 : 8;
    if (out_bytes < n * bb) return BU_ERR_OUTPUT_SIZE;
    if (n == 0) return BU_OK;
    std::lock_guard<std::mutex> g(ctx->lock);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    bu_status st = bu_reserve(ctx, &ctx->d_in, &ctx->in_cap, n * 4);
    if (st) return st;
    st = bu_reserve(ctx, &ctx->d_out, &ctx->out_cap, n * bb);
    if (st) return st;
    const size_t ep_bytes = ((size_t)n_ep * 4 + 15) & ~(size_t)15, sel_bytes = ((size_t)n_sel * 8 + 15) & ~(size_t)15;
    const size_t a_bytes = alpha_idx ? n * 4 : 0;
    st = bu_reserve(ctx, &ctx->d_aux, &ctx->aux_cap, ep_bytes + sel_bytes + a_bytes);
    if (st) return st;
    uint8_t* aux = static_cast<uint8_t*>(ctx->d_aux);
    uint64_t word = 0;
    BuDrain drain(ctx);
    BU_HIP(ctx, hipMemcpyAsync(ctx->d_in, idx, n * 4, hipMemcpyHostToDevice, ctx->stream));
    BU_HIP(ctx, hipMemcpyAsync(aux, endpoints, (size_t)n_ep * 4, hipMemcpyHostToDevice, ctx->stream));
    BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes, selectors, (size_t)n_sel * 8, hipMemcpyHostToDevice, ctx->stream));
    if (alpha_idx) BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes + sel_bytes, alpha_idx, n * 4, hipMemcpyHostToDevice, ctx->stream));
    BU_HIP(ctx, hipMemsetAsync(ctx->d_status, 0xFF, sizeof(uint64_t), ctx->stream));
    uint64_t* ds = reinterpret_cast<uint64_t*>(ctx->d_status);
    const uint32_t* d_ep = reinterpret_cast<const uint32_t*>(aux);
    const void* d_sel = aux + ep_bytes;
    if (rgba)
        st = bu_etc1s_decode_rgba_device(ctx, static_cast<const uint32_t*>(ctx->d_in),
                                         alpha_idx ? reinterpret_cast<const uint32_t*>(aux + ep_bytes + sel_bytes) : nullptr, nbx, nby, d_ep,
                                         n_ep, d_sel, n_sel, ctx->d_out, ds, ctx->stream);
    else
        st = bu_etc1s_transcode_etc1_device(ctx, static_cast<const uint32_t*>(ctx->d_in), n, d_ep, n_ep, d_sel, n_sel, ctx->d_out, ds, ctx->stream);
    if (st) return st;
    BU_HIP(ctx, hipMemcpyAsync(&word, ctx->d_status, sizeof(word), hipMemcpyDeviceToHost, ctx->stream));
    BU_HIP(ctx, hipMemcpyAsync(out, ctx->d_out, n * bb, hipMemcpyDeviceToHost, ctx->stream));
    BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain.armed = false;
    return bu_status_word_decode(word, first_bad);
}

bu_status bu_etc1s_transcode_etc1(bu_context* ctx, const uint32_t* idx, size_t n_blocks, const uint32_t* endpoints, uint32_t n_endpoints,
                                  const uint8_t* selectors, uint32_t n_selectors, uint8_t* out, size_t out_bytes, uint64_t* first_bad_block)
{
    return bu_etc1s_host(ctx, false, idx, nullptr, n_blocks, 1, endpoints, n_endpoints, selectors, n_selectors, out, out_bytes, first_bad_block);
}

bu_status bu_etc1s_decode_rgba(bu_context* ctx, const uint32_t* idx, const uint32_t* alpha_idx, size_t nbx, size_t nby,
                               const uint32_t* endpoints, uint32_t n_endpoints, const uint8_t* selectors, uint32_t n_selectors, uint8_t* out,
                               size_t out_bytes, uint64_t* first_bad_block)
{
    if (nbx == 0 && nby != 0) return BU_ERR_ARGUMENT;
    return bu_etc1s_host(ctx, true, idx, alpha_idx, nbx, nby, endpoints, n_endpoints, selectors, n_selectors, out, out_bytes, first_bad_block);
}


}  // extern "C"
