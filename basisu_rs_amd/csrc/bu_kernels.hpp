// gfx950 kernels of the block-transcode path: the one-lane-per-block kernel, the mode-sorted kernel (every UASTC target), the
// ETC1S codebook-lookup kernels, the status-word reset and the copy kernel of the measurement harness.
// Part of the single translation unit bu_hip.hip (included there; not a stand-alone header).
#pragma once
namespace {

constexpr int BU_WG = 256;            // 4 waves
// Below this the plain one-lane-per-block kernel is used.  It runs one mode path per DISTINCT mode present, so it only
// wins for a handful of blocks (BC7: 1 block 2.3 vs 3.3 us, 8 blocks 4.2 vs 3.7 us, 64 blocks 7.8 vs 4.4 us,
// 1024 blocks 16.1 vs 4.8 us; ETC1 at 128 blocks 38.6 vs 15.0 us).
constexpr int BU_SORT_MIN_BLOCKS = 8;
// shapes of the mode-sorted kernel whose tile size is a run-time argument (bu_balanced_tile): ETC1 / ETC2 on 4096-block tiles
constexpr bool bu_dyn_tile(int target, int tile) { return (target == BU_TGT_ETC1 || target == BU_TGT_ETC2) && tile == 4096; }

// ------------------------------------------------------------------------------------------------
// LDS image of the tables of TARGET: [BuBc7Tables (BC7 only)][BuTables up to the end of the target's ranges]
constexpr unsigned bu_lds_front(int target) { return target == 1 ? (unsigned)sizeof(BuBc7Tables) : 0u; }
constexpr unsigned bu_lds_table_bytes(int target) { return bu_lds_front(target) + bu_table_bytes(target); }
// stage the parts of the table blob TARGET reads (bu_table_range; BC7: its own tables in front), 16 bytes per thread per step
template <int WGS, int TARGET>
__device__ __forceinline__ void bu_stage_tables_n(uint4* dst, const BuTablesAll* __restrict__ src)
{
    constexpr BuTableRange R = bu_table_range(TARGET);
    constexpr int F = (int)bu_lds_front(TARGET) / 16;
    // (for BC7 the image is the blob from its first byte; for the others it starts at BuTablesAll::t)
    const uint4* s = reinterpret_cast<const uint4*>(TARGET == 1 ? reinterpret_cast<const void*>(src) : reinterpret_cast<const void*>(&src->t));
    uint4* d = dst;
    for (int i = threadIdx.x; i < F + (int)(R.hi / 16); i += WGS)
        if (i < F || i >= F + (int)(R.lo / 16)) d[i] = s[i];
    if constexpr (R.lo2 < R.hi2) {
        for (int i = F + R.lo2 / 16 + threadIdx.x; i < F + (int)(R.hi2 / 16); i += WGS) d[i] = s[i];
    }
}

// System scope: the status word may live in page-locked HOST memory (the blocking entry points hand the kernels a word the host reads
// directly once the streams are idle: no reset launch in front, no copy behind -- bu_range_begin); for a word in device memory the
// scope changes nothing.  Only a failing block ever gets here.
__device__ __forceinline__ void bu_report(unsigned long long* status, unsigned long long block, int st)
{
    if (status) (void)__hip_atomic_fetch_min(status, (block << 8) | (unsigned long long)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Every block is read once and every result written once: non-temporal (streaming) accesses keep the 32 MiB of a 4096^2
// atlas from being allocated in L2 / Infinity Cache with normal retention.  Measured on the BC7 headline: 14.4 -> 13.7 us.
typedef unsigned int bu_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int bu_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 bu_ld_stream(const uint4* p)
{
    const bu_v4u r = __builtin_nontemporal_load(reinterpret_cast<const bu_v4u*>(p));
    return make_uint4(r.x, r.y, r.z, r.w);
}
// Output stores.  BU_ST_MODE selects the cache policy (experiment knob).  A/B inside one run (tools/exp/ab.sh), 2^20 blocks:
//   0 nontemporal (nt)            copy 7.15  BC7 10.93  ETC1 25.1  RGBA32 21.2 us
//   1 plain                            7.07      12.70       25.9         23.1     (results linger dirty in L2)
//   2 write-through (sc1)              7.16      10.88       24.9         20.6
//   3 sc0 sc1                          7.13      10.86       24.8         20.7
//   4 sc1 nt  <- shipped               6.99      10.70       24.75        20.6
#ifndef BU_ST_MODE
#define BU_ST_MODE 4
#endif
#if BU_ST_MODE == 2
#define BU_ST_BITS " sc1"
#elif BU_ST_MODE == 3
#define BU_ST_BITS " sc0 sc1"
#elif BU_ST_MODE == 4
#define BU_ST_BITS " sc1 nt"
#endif
__device__ __forceinline__ void bu_st_stream(uint4* p, const uint4 v)
{
    bu_v4u r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
#if BU_ST_MODE == 0
    __builtin_nontemporal_store(r, reinterpret_cast<bu_v4u*>(p));
#elif BU_ST_MODE == 1
    *reinterpret_cast<bu_v4u*>(p) = r;
#else
    // hipcc does not model an asm store: the s_nop 1 keeps its next instruction from overwriting the data registers before
    // the store has read them (two wait states behind a store of more than 8 bytes on gfx940+)
    asm volatile("global_store_dwordx4 %0, %1, off" BU_ST_BITS "\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
#endif
}
__device__ __forceinline__ void bu_st_stream(uint2* p, const uint2 v)
{
    bu_v2u r;
    r.x = v.x; r.y = v.y;
    // 8-byte stores stay nontemporal: an sc1 store narrower than 16 bytes is one fabric write per lane
    // (ETC1S -> ETC1 at 2^18 blocks: 4.7 -> 6.2 us with sc1 nt)
#if BU_ST_MODE == 1
    *reinterpret_cast<bu_v2u*>(p) = r;
#else
    __builtin_nontemporal_store(r, reinterpret_cast<bu_v2u*>(p));
#endif
}

// UASTC -> {ASTC, BC7, ETC1, ETC2, RGBA32}: replaces the loop of uastc.rs:157-165 / 96-107
template <int TARGET>
__global__ __launch_bounds__(BU_WG) void bu_uastc_kernel(const uint4* __restrict__ in, void* __restrict__ out, size_t n_blocks,
                                                         unsigned bpr, unsigned long long base, unsigned long long* status,
                                                         const BuTablesAll* __restrict__ tables)
{
    __shared__ uint4 t_store[bu_lds_table_bytes(TARGET) / 16];  // the blob as far as TARGET reads it (BC7: its own tables in front)
    BuTables& T = *reinterpret_cast<BuTables*>(t_store + bu_lds_front(TARGET) / 16);
    const size_t stride = (size_t)gridDim.x * BU_WG;
    size_t idx = (size_t)blockIdx.x * BU_WG + threadIdx.x;
    // first block load is issued before the table copy so both are in flight together
    uint4 v = idx < n_blocks ? bu_ld_stream(in + idx) : make_uint4(0, 0, 0, 0);
    bu_stage_tables_n<BU_WG, TARGET>(t_store, tables);
    __syncthreads();
    while (idx < n_blocks) {
        const size_t next = idx + stride;
        const uint4 vn = next < n_blocks ? in[next] : make_uint4(0, 0, 0, 0);
        BuBlk b;
        b.w[0] = v.x;
        b.w[1] = v.y;
        b.w[2] = v.z;
        b.w[3] = v.w;
        const uint32_t mode = T.mode_lut[v.x & 127u];
        uint32_t o[TARGET == BU_TGT_RGBA ? 16 : 4];
#pragma unroll
        for (int i = 0; i < (TARGET == BU_TGT_RGBA ? 16 : 4); i++) o[i] = 0;
        const int st = bu_block_any<TARGET>(T, mode, b, o);
        if (st) {
            bu_report(status, base + idx, st);
#pragma unroll
            for (int i = 0; i < (TARGET == BU_TGT_RGBA ? 16 : 4); i++) o[i] = 0;
        }
        if constexpr (TARGET == BU_TGT_ETC1) {
            reinterpret_cast<uint2*>(out)[idx] = make_uint2(o[0], o[1]);
        } else if constexpr (TARGET == BU_TGT_RGBA) {
            const size_t by = idx / bpr, bx = idx - by * bpr;
            uint4* img = reinterpret_cast<uint4*>(out);
#pragma unroll
            for (int r = 0; r < 4; r++) img[(4 * by + r) * (size_t)bpr + bx] = make_uint4(o[4 * r], o[4 * r + 1], o[4 * r + 2], o[4 * r + 3]);
        } else {
            reinterpret_cast<uint4*>(out)[idx] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        v = vn;
        idx = next;
    }
}

// ------------------------------------------------------------------------------------------------
// Mode-sorted kernel (every target; RGBA32 returns its 64 B per block through an LDS row tile).
//
// The per-mode code paths are straight-line and short (100-260 VALU each for BC7) but there are 19 of
// them: a wave whose 64 lanes hold a random mix of modes executes all 19 serially (measured: 69 us
// per 4096x4096 atlas vs a 7 us copy).  So each workgroup first sorts its tile of BU_TILE blocks by
// mode through LDS (counting sort: one LDS atomic per block), cuts every mode's run into chunks of
// <= 64 blocks, and each wave then transcodes whole chunks with a wave-uniform mode (scalar branch,
// no exec-mask divergence).  Results go back to LDS at the sorted slot and leave in original order,
// so global loads and stores stay fully coalesced (1 KiB per wave instruction).
//   LDS per workgroup: tile 16 B x BU_TILE + the target's tables + (ETC, RGBA32) 1 B x BU_TILE status + counters.
// Environment knobs read by the host code (diagnostics, not configuration): BU_TRACE (phase times of bu_read_to on stderr, and
// for the streamed ETC1S front door when every work item started and ended), BU_RUN_PIECE_MIB (piece size of the two-stream upload
// pipeline, 0 = off), BU_ETC1S_ONE_LAUNCH (ETC1S files: decode everything, then one launch -- the round-3 path),
// BU_ETC1S_ONE_THREAD (streamed ETC1S front door: every slice's symbol loop on one thread); round 6: BU_STREAM_MODE=plain | cumask (how the context's own
// streams are created: never / always with CU masks instead of by the hardware-queue check, bu_streams.hpp), BU_ENQUEUE_THREADS=0 (the pipelined batch call
// enqueues from the calling thread alone), BU_TILE_TICKETS=0 (every persistent launch walks fixed shares of the tiles).
// inclusive add-scan over lanes 0..31 (and 32..63) with DPP row shifts: 5 VALU, no LDS round trips
__device__ __forceinline__ uint32_t bu_scan32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    return v;
}

// RECT: a tile is a BU_RECT_W x (tile / BU_RECT_W) RECTANGLE of the block grid (64 x 16 for the 1024-block tiles) instead of a strip
// of consecutive blocks (the launcher picks it when blocks_per_row is a multiple of 64 and the slice is whole rows of such tiles,
// bu_launch_uastc).  Texture content is coherent in two dimensions: a 1024-block strip of a 4096-px-wide image cuts an 8 x 8-block
// region of one UASTC mode into eight runs of eight, a rectangle keeps it whole -- fewer, fuller runs per tile, fewer partly
// filled chunks (mode-coherent atlas: BC7 -5...-10 %).  64 blocks wide: a wave's load / store instruction is still one contiguous
// KiB (32-wide tiles, two 512-byte segments per instruction, cost the uniform-random atlas +3 % on BC7).  Nothing else changes (the
// sort works on the index inside the tile).  A compile-time variant: as a run-time switch the extra index arithmetic cost the
// strip path 4 % (round 2).
constexpr unsigned BU_RECT_W = 64;
// MULTI (LAYOUT 2): the launch covers SEVERAL runs of blocks at unrelated addresses (slices of a texture array in separate
// allocations); tiles never straddle runs.  The run table travels IN THE KERNEL ARGUMENTS (bu_uastc_multi_kernel: no device
// buffer, no copy, nothing to free behind the launch): per run its input, output, block-index base, size and the number of its
// first tile.  A workgroup finds the run of tile t by comparing t with 64 first-tile numbers at a time (one vector load and a
// ballot per wave) and reads that run's record with scalar loads.  A loop over small slices becomes one launch
// (bu_uastc_transcode_batch_device); more than BU_MULTI_RUNS runs go out as several launches.
struct BuRunDesc {
    const uint4* in;  // the run's first block
    void* out;        // its output
    uint64_t base;    // block-index base of the run (status words)
    uint32_t n;       // blocks in the run
    uint32_t vshift;  // BU_RUN_STRIPS: tiles are strips of 1024 consecutive blocks; else the run is whole 64 x 16-block rectangles of a (virtual) grid
                      // 64 << vshift blocks wide (bu_launch_runs: a power of two, the run a multiple of 16 rows of it)
};
constexpr unsigned BU_MULTI_RUNS = 96;
constexpr uint32_t BU_RUN_STRIPS = 0xFFFFFFFFu;
struct BuRunTable {
    BuRunDesc run[BU_MULTI_RUNS];
    uint32_t first_tile[BU_MULTI_RUNS + 32];  // ascending; entries past the last run hold 0xFFFFFFFF (128 entries: two 64-lane loads)
};
static_assert(sizeof(BuRunDesc) == 32 && sizeof(BuRunTable) <= 3968, "the run table must fit the 4 KiB of kernel arguments beside the other parameters");
// the tile the kernel is working on: its run's addresses, where it starts inside the run, how many blocks it holds
struct BuTileDesc {
    const uint4* in;
    void* out;
    uint32_t first;   // strips: the tile's first block inside its run; rectangles: (16 * tile row) * width + 64 * tile column, the block index of the tile's corner
    uint32_t n;
    uint64_t base;
    uint32_t width;   // rectangles: blocks per row of the run's (virtual) grid; 0: strips
};
enum { BU_LAYOUT_STRIP = 0, BU_LAYOUT_RECT = 1, BU_LAYOUT_MULTI = 2,
       BU_LAYOUT_MULTI_WHOLE = 3 };  // MULTI with every run tiled as whole rectangles (BuRunDesc::vshift): no lane ever lacks a block, and the validity tests fold away as in RECT
// one set of tile tickets (kernel, `ticket`): eight counters BU_TICKET_STRIDE words apart, then the count of workgroups that have left
constexpr unsigned BU_TICKET_STRIDE = 32, BU_TICKET_DONE = 8 * BU_TICKET_STRIDE, BU_TICKET_WORDS = 9 * BU_TICKET_STRIDE;
// WGS threads per workgroup, BPT blocks per thread: tile = WGS * BPT blocks.  PREFETCH: a workgroup that walks several tiles
// loads tile k+1 while it transcodes tile k (BPT more uint4 registers).
template <int TARGET, int WGS, int BPT, bool PREFETCH, int LAYOUT>
__device__ __forceinline__ void bu_uastc_sorted_body(const uint4* __restrict__ in, void* __restrict__ out, unsigned n_blocks,
                                                     unsigned bpr, unsigned long long base, unsigned long long* status,
                                                     const BuTablesAll* __restrict__ tables, unsigned cus, unsigned tile_rt,
                                                     const BuRunTable* __restrict__ runs, unsigned* __restrict__ ticket = nullptr)
{
    // Static priority by residency generation.  Workgroups are dealt breadth-first (b, b + CUs, b + 2 CUs, ... share a CU:
    // tools/exp/census.hip), and the instruction arbiter serves the OLDEST wave first, so the four tiles of a CU finish
    // 1.5 us apart and the last one runs its latency-bound chain with the vector units nearly idle (phase stamps,
    // profiles/r02_*stamps*).  Raising the later generations' priority makes them catch up while the older ones fill
    // the gaps: BC7 10.51 -> 10.2 us, ASTC 9.74 -> 9.47, RGBA32 20.4 -> 19.95 in an A/B run.  Speed only: any placement is correct.
    // cus = 0 switches it off: RGBA32 launches in which the workgroups do not all walk the same number of tiles (786 432
    // blocks: the one-tile workgroups of the second generation would run ahead of the two-tile ones, 15.75 against 14.24 us).
    // Round 3, on the leaner kernels: generation priorities 0, 3, 2, 1 beat 0, 1, 2, 3 for BC7 (9.13 -> 8.96 us) and ASTC (9.0 -> 8.8),
    // not for RGBA32 (15.85 -> 16.0); no priorities 9.14 / 9.04 / 17.4 (profiles/r03_ab_wave_priorities.txt).
    if (cus != 0) {  // (comparisons, not blockIdx / cus: a scalar division is ~25 instructions in front of the first load)
        constexpr bool ROT = TARGET == BU_TGT_BC7 || TARGET == BU_TGT_ASTC;
        if (blockIdx.x >= 3 * cus) __builtin_amdgcn_s_setprio(ROT ? 1 : 3);
        else if (blockIdx.x >= 2 * cus) __builtin_amdgcn_s_setprio(2);
        else if (blockIdx.x >= cus) __builtin_amdgcn_s_setprio(ROT ? 3 : 1);
    }
    constexpr int BU_WG = WGS, BU_BPT = BPT, BU_TILE = WGS * BPT;
    __shared__ uint4 t_store[bu_lds_table_bytes(TARGET) / 16];  // the blob as far as TARGET reads it (BC7: its own tables in front)
    BuTables& T = *reinterpret_cast<BuTables*>(t_store + bu_lds_front(TARGET) / 16);
    // RGBA32 through LDS: four pixel rows of 16 B per block, stored row-major by row index so that both the
    // sorted-order writes and the original-order reads are 16-byte strided (no bank conflicts).  The sorted input tile
    // lives IN row 0 of that output tile: a lane reads its block from slot s and later overwrites exactly slot s with
    // the block's first pixel row, so no other lane's input is ever clobbered -- 64 KiB instead of 80 per 1024 blocks,
    // which is what lets two workgroups share a CU.
    constexpr bool RECT = LAYOUT == BU_LAYOUT_RECT, MULTI = LAYOUT == BU_LAYOUT_MULTI || LAYOUT == BU_LAYOUT_MULTI_WHOLE;
    constexpr bool WHOLE = RECT || LAYOUT == BU_LAYOUT_MULTI_WHOLE;  // every tile of the launch holds BU_TILE blocks
    constexpr bool BU_ALIAS = TARGET == BU_TGT_RGBA;
    // two RGBA32 workgroups must fit the 160 KiB of a CU: output tile + table blob + status bytes + counters / chunk list
    static_assert(!BU_ALIAS || bu_lds_table_bytes(TARGET) + 4 * BU_TILE * 16 + BU_TILE + 1536 <= 80 * 1024,
                  "the RGBA32 workgroup no longer fits twice per CU: shrink BuTables or stage it per target in LDS too");
    __shared__ uint4 sblk_store[BU_ALIAS ? 1 : BU_TILE];
    __shared__ uint4 sout[BU_ALIAS ? 4 * BU_TILE : 1];
    uint4* const sblk = BU_ALIAS ? sout : sblk_store;
    // BC7 and ASTC carry a failing block's status inside its result slot (a valid block of either format has a non-zero first
    // byte: BC7's unary mode prefix, ASTC's block mode / void-extent marker), the other targets in a byte per block
    constexpr bool INBLOCK = TARGET == BU_TGT_BC7 || TARGET == BU_TGT_ASTC;
    __shared__ uint8_t sst[INBLOCK ? 16 : BU_TILE];
    // counters and the chunk ticket are double-buffered by tile parity: the buffer of tile t+1 is cleared during tile t,
    // after everyone has finished with its previous use (tile t-1), so no barrier is spent on the reset
    __shared__ uint32_t cnt[2][32], next_chunk[2];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    // Blocks per tile.  The exclusive ETC shape is one workgroup per CU on tiles of up to 4096 blocks; with a fixed tile a slice of
    // 1.5 tiles per CU takes as long as one of 2 (1.5 Mi blocks 33.4 us, 2 Mi 34.2).  There the launcher sizes the tile so
    // that every CU gets the same number of equal tiles (`tile_rt` <= BU_TILE, a multiple of 64); threads past the end of
    // a shorter tile sit out like threads past the end of the slice.  Every other shape passes tile_rt = BU_TILE and
    // compiles to what it was.
    constexpr bool DYN_TILE = bu_dyn_tile(TARGET, BU_TILE) && LAYOUT != BU_LAYOUT_RECT;
    const unsigned tile_blocks = DYN_TILE ? tile_rt : (unsigned)BU_TILE;
    const unsigned n_tiles = (n_blocks + tile_blocks - 1) / tile_blocks;  // 32-bit indices: the host splits launches above 2^26 blocks
    auto in_tile = [&](unsigned l) { return !DYN_TILE || l < tile_blocks; };
    static_assert(!RECT || BU_TILE % BU_RECT_W == 0, "rectangular tiles are BU_RECT_W blocks wide and of fixed size");
    // slice index of block l of tile t.  RECT: `tile_rt` carries ceil(2^32 / tiles per row) + 0 (bu_launch_uastc), which makes
    // t / tpr one s_mul_hi_u32 (exact for t, tpr < 2^16: at most 2^26 blocks per launch, rows below 2^21 blocks)
    const unsigned tpr = RECT ? bpr / BU_RECT_W : 1u;  // tiles per row of tiles
    auto tile_xy = [&](unsigned t, unsigned& ty, unsigned& tx) {
        ty = (unsigned)(((unsigned long long)t * tile_rt) >> 32);
        tx = t - ty * tpr;
    };
    auto gidx = [&](unsigned t, unsigned l) {
        if constexpr (RECT) {
            unsigned ty, tx;
            tile_xy(t, ty, tx);  // (wave-uniform: scalar)
            return ((unsigned)(BU_TILE / BU_RECT_W) * ty + l / BU_RECT_W) * bpr + BU_RECT_W * tx + (l % BU_RECT_W);
        } else {
            return t * tile_blocks + l;
        }
    };
    // RECT launches hold whole tiles only: every lane of every tile has a block (keys are 0..19, never the "no block" 31), and the
    // validity tests below fold away
    auto has_block = [&](uint32_t k) { return WHOLE || k < 20u; };
    unsigned tile = blockIdx.x;
    // Tile tickets (`ticket` != nullptr): a persistent workgroup's FIRST tile is its block index; every further tile is the next number off a
    // counter in device memory instead of tile + gridDim.x -- workgroups that run ahead (a cheaper mode mix, a luckier HBM channel, a CU they
    // share with fewer others) take more tiles, and the launch ends when the tiles do, not when the slowest fixed share does.
    // EIGHT counters, 128 bytes apart, workgroup b draws from counter b % 8 and its k-th ticket is tile gridDim.x + 8 k + b % 8 (every tile >=
    // gridDim.x exactly once): a device-scope atomic on one address completes once per ~12 ns on this chip (eight XCDs, one coherence point:
    // one counter for a 2^25-block launch = 32 768 tickets = 390 us, profiles/r06_ab_tile_tickets.txt), eight counters are not a bottleneck.
    // Thread 0 draws the ticket one whole tile ahead of its use (the atomic's round trip lies under a tile's work) and hands it to the workgroup
    // through LDS at barrier (1).  ticket[BU_TICKET_DONE] counts the workgroups that have left; the last one zeroes all nine words for the next
    // launch on the same stream (the host hands every stream its own set: launches of one stream never overlap).
    __shared__ uint32_t s_next_tile[2];
    // thread 0: the counter value drawn for the tile behind the next one.  It stays RAW until it is handed over: arithmetic on it right behind the
    // atomic would make the wave wait for the atomic's round trip on the spot, with the whole workgroup behind it at the next barrier
    uint32_t my_draw = 0;
    unsigned* const my_counter = ticket ? ticket + (blockIdx.x & 7u) * BU_TICKET_STRIDE : nullptr;
    auto tile_of_draw = [&](uint32_t d) { return gridDim.x + 8u * d + (blockIdx.x & 7u); };
    if (ticket && tid == 0) my_draw = atomicAdd(my_counter, 1u);
    // MULTI: the current tile's descriptor (wave-uniform: scalar loads); every other layout addresses the one slice of the launch
    // `td` = the descriptor of the tile being sorted / transcoded / written back; `tl` = the descriptor of the tile whose blocks are being
    // LOADED (the same tile, or with PREFETCH the next one: its loads are in flight while `td`'s tile is transcoded)
    BuTileDesc td = {in, out, 0u, 0u, base, 0u}, tl = td;
    // MULTI, persistent grid: the run table is copied to LDS once per workgroup (3.9 KiB) and every later look-up reads it there.  From the kernel
    // arguments a look-up is two vector loads, a ballot and dependent scalar loads -- a microsecond of round trips between the scatter and the prefetch
    // loads of EVERY tile, with the workgroup waiting at barrier (2) behind it (8 % of a multi-run launch over 2^20-block slices)
    constexpr bool RUNS_IN_LDS = MULTI && PREFETCH;
    __shared__ uint4 s_runs[RUNS_IN_LDS ? sizeof(BuRunTable) / 16 : 1];  // (16 bytes in every other instantiation)
    const BuRunTable* R = runs;  // the table the look-ups read: the kernel arguments until the copy is complete (the first barrier), LDS from then on
    if constexpr (RUNS_IN_LDS) {
        static_assert(sizeof(BuRunTable) % 16 == 0, "copied in 16-byte pieces");
        for (unsigned i = tid; i < sizeof(BuRunTable) / 16; i += WGS) s_runs[i] = reinterpret_cast<const uint4*>(runs)[i];
    }
    auto uni32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    auto uni64 = [&](uint64_t x) { return (uint64_t)uni32((uint32_t)x) | ((uint64_t)uni32((uint32_t)(x >> 32)) << 32); };
    auto desc_of = [&](unsigned t, BuTileDesc& d) {
        if constexpr (MULTI) {
            if (t < n_tiles) {
                // runs 0..r start at or before tile t: r = (number of first-tile entries <= t) - 1; the unused entries are ~0
                uint32_t r = 0;
#pragma unroll
                for (unsigned c = 0; c < BU_MULTI_RUNS + 32; c += 64) {
                    const uint32_t cnt = (uint32_t)__popcll(__ballot(R->first_tile[c + lane] <= t));
                    if (cnt) r = c + cnt - 1u;
                }
                r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
                // (the record comes back in vector registers -- an LDS read --, but every lane read the same one: made scalar here, so that the tile's addresses
                //  are an SGPR base and a VGPR offset like the one-slice kernel's, not 64-bit vector arithmetic per load and store)
                const BuRunDesc rv = R->run[r];
                const BuRunDesc rd = {reinterpret_cast<const uint4*>(uni64(reinterpret_cast<uint64_t>(rv.in))), reinterpret_cast<void*>(uni64(reinterpret_cast<uint64_t>(rv.out))),
                                      uni64(rv.base), uni32(rv.n), uni32(rv.vshift)};
                const uint32_t lt = t - uni32(R->first_tile[r]);  // the tile's number inside its run
                if (rd.vshift != BU_RUN_STRIPS) {
                    // a 64 x 16-block rectangle of the run's grid (tiles per row a power of two): 16 segments of 1 KiB at the grid's pitch instead of 16 KiB in a row
                    const uint32_t ty = lt >> rd.vshift, tx = lt & ((1u << rd.vshift) - 1u), width = (uint32_t)BU_RECT_W << rd.vshift;
                    d = BuTileDesc{rd.in, rd.out, (uint32_t)(BU_TILE / BU_RECT_W) * ty * width + (uint32_t)BU_RECT_W * tx, (uint32_t)BU_TILE, rd.base, width};
                } else {
                    const uint32_t first = lt * (uint32_t)BU_TILE, left = rd.n - first;
                    d = BuTileDesc{rd.in, rd.out, first, left < (uint32_t)BU_TILE ? left : (uint32_t)BU_TILE, rd.base, 0u};
                }
            }
        }
    };
    desc_of(tile, td);
    tl = td;
    // block l of tile t (whose descriptor is `tl`): where it is loaded from, whether it exists
    // MULTI: block l of the tile described by d, as an index inside d's run (strips: consecutive; rectangles: row l / 64, column l % 64 of the tile)
    auto run_idx = [&](const BuTileDesc& d, unsigned l) { return d.first + ((WHOLE || d.width) ? (l / BU_RECT_W) * d.width + (l % BU_RECT_W) : l); };
    auto blk_src = [&](unsigned t, unsigned l) { return MULTI ? tl.in + run_idx(tl, l) : in + gidx(t, l); };
    auto blk_valid = [&](unsigned t, unsigned l) {
        if constexpr (MULTI) return t < n_tiles && (WHOLE || l < tl.n);
        else return RECT ? t < n_tiles : (t < n_tiles && gidx(t, l) < n_blocks && in_tile(l));
    };
    // Table staging.  The staged 16-byte pieces of the LDS image are numbered 0..TVT-1: BC7's own tables, then the target's one or
    // two ranges of the common blob; piece i sits at t_store[tdst(i)] and comes from the same index of the image's source in device
    // memory.  Every thread issues ALL its table loads back to back (round 2 ran a load / wait / store loop: a second L2 round
    // trip per 8 KiB of tables in front of the first barrier) and, where the image is small, BEFORE the tile's block loads.
    // Images above 16 KiB (ETC: 26.6 KiB) go tile loads first, only key_lut (128 B: all the sort phases read) staged up front, the
    // rest held in registers until the first tile's rank atomics are out (A/B: ETC1 19.45 -> 19.0 us in round 2; tables first
    // 19.2 -> 19.8 in round 3).
    constexpr BuTableRange TR = bu_table_range(TARGET);
    constexpr bool SPLIT = bu_lds_table_bytes(TARGET) > 16384;
    constexpr int TF = (int)bu_lds_front(TARGET) / 16, TV1 = (int)(TR.hi - TR.lo) / 16, TV2 = TR.lo2 < TR.hi2 ? (int)(TR.hi2 - TR.lo2) / 16 : 0;
    constexpr int TVT = TF + TV1 + TV2, TVN = (TVT + WGS - 1) / WGS;
    const uint4* const tsrc = reinterpret_cast<const uint4*>(TARGET == BU_TGT_BC7 ? reinterpret_cast<const void*>(tables) : reinterpret_cast<const void*>(&tables->t));
    auto tdst = [&](int i) { return i < TF ? i : (i < TF + TV1 ? (int)(TR.lo / 16) + i : TF + (int)(TR.lo2 / 16) + (i - TF - TV1)); };
    uint4 tv[TVN];
    if constexpr (!SPLIT) {
#pragma unroll
        for (int k = 0; k < TVN; k++) {
            const int i = k * WGS + (int)tid;
            tv[k] = i < TVT ? tsrc[tdst(i)] : make_uint4(0, 0, 0, 0);
        }
    }
    uint4 v[BU_BPT];
#pragma unroll
    for (int j = 0; j < BU_BPT; j++)
        v[j] = (RECT || blk_valid(tile, j * BU_WG + tid)) ? bu_ld_stream(blk_src(tile, j * BU_WG + tid)) : make_uint4(0, 0, 0, 0);  // (RECT: the grid is at most n_tiles)
    if constexpr (!SPLIT) {
#pragma unroll
        for (int k = 0; k < TVN; k++) {
            const int i = k * WGS + (int)tid;
            if (i < TVT) t_store[tdst(i)] = tv[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < TVN; k++) {
            const int i = k * WGS + (int)tid;
            tv[k] = i < TVT ? tsrc[tdst(i)] : make_uint4(0, 0, 0, 0);
        }
        static_assert(offsetof(BuTables, key_lut) % 4 == 0 && sizeof(T.key_lut[0]) == 128, "the target's key_lut is staged as 32 dwords");
        if (tid < 32) reinterpret_cast<uint32_t*>(T.key_lut[TARGET])[tid] = reinterpret_cast<const uint32_t*>(tables->t.key_lut[TARGET])[tid];
    }
    bool tables_staged = !SPLIT;
    if (tid < 64) (&cnt[0][0])[tid] = 0;
    if (tid < 2) next_chunk[tid] = 0;
    __syncthreads();
    if constexpr (RUNS_IN_LDS) R = reinterpret_cast<const BuRunTable*>(s_runs);
    unsigned par = 0;
    unsigned next_of_loop = 0;
    for (; tile < n_tiles; tile = next_of_loop, par ^= 1u) {
        const unsigned tbase = tile * tile_blocks;
        // ---- A: sort key + rank within the key (counting sort, pass 1) ----
        // key = position of the block's mode in BU_COST_ORDER (runs are laid out heaviest code path first).
        // Rank within the key = one LDS atomic per block.  64 lanes adding to ONE counter serialise, though, and that is
        // exactly what coherent textures produce (flat regions: long runs of one mode).  A wave whose loads are each of a
        // single mode therefore takes an aggregated path -- one atomic of 64 by lane 0 per load, rank = lane id -- chosen
        // by a wave-uniform branch; every other wave runs the plain per-lane atomics unchanged.
        uint32_t key[BU_BPT], pos[BU_BPT];
        bool uniform = true;
#pragma unroll
        for (int j = 0; j < BU_BPT; j++) {
            const bool valid = WHOLE || (MULTI ? (unsigned)(j * BU_WG) + tid < td.n : (tbase + j * BU_WG + tid < n_blocks && in_tile(j * BU_WG + tid)));  // (RECT: whole tiles only)
            key[j] = valid ? T.key_lut[TARGET][v[j].x & 127u] : 31u;
            uniform = uniform && (__ballot(key[j] == (uint32_t)__builtin_amdgcn_readfirstlane(key[j])) == ~0ull) && has_block(key[j]);
        }
        if (uniform) {
            uint32_t lead[BU_BPT];
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) lead[j] = lane == 0 ? atomicAdd(&cnt[par][key[j]], 64u) : 0u;
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) pos[j] = (uint32_t)__builtin_amdgcn_readfirstlane(lead[j]) + lane;
        } else {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) pos[j] = atomicAdd(&cnt[par][key[j]], 1u);  // lanes past the end hit the dummy counter 31: no exec-mask region, the atomics issue back to back
        }
        if (ticket && tid == 0) s_next_tile[par] = tile_of_draw(my_draw);
        if (!tables_staged) {  // (key_lut is rewritten with the bytes it already holds)
#pragma unroll
            for (int k = 0; k < TVN; k++) {
                const int i = k * WGS + (int)tid;
                if (i < TVT) t_store[tdst(i)] = tv[k];
            }
            tables_staged = true;
        }
        __syncthreads();  // (1) every rank is final; the tables are in LDS
        // ---- B: run starts and the chunk map, derived by EVERY wave for itself ----
        // Lane k < 20 holds run k: blocks in the low half, 64-block chunks in the high half of one word; a DPP scan gives
        // every run's first slot and first chunk number.  No wave waits for another one here (the round-1 kernel had one
        // wave build a chunk list in LDS while seven stood at a barrier).
        const uint32_t run_c = lane < 20u ? cnt[par][lane] : 0u;
        const uint32_t run_pk = run_c | (((run_c + 63u) >> 6) << 16);
        const uint32_t run_incl = bu_scan32(run_pk), run_excl = run_incl - run_pk;  // lanes 20..31 carry the totals
        const uint32_t nc = (uint32_t)__builtin_amdgcn_readlane((int)run_incl, 31) >> 16;
        if (tid < 32) cnt[par ^ 1u][tid] = 0;  // the other parity: last read in B of the previous tile, next written in A of the next one
        if (tid == 0) next_chunk[par ^ 1u] = 0;
        // ---- scatter into sorted order (counting sort, pass 2) ----
        uint32_t dest[BU_BPT];
#pragma unroll
        for (int j = 0; j < BU_BPT; j++) {
            const uint32_t st = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(key[j] << 2), (int)run_excl) & 0xFFFFu;
            dest[j] = has_block(key[j]) ? st + pos[j] : 0u;
            if (has_block(key[j])) sblk[dest[j]] = v[j];
        }
        // prefetch the next tile while this one is transcoded
        const unsigned ntile = ticket ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_next_tile[par]) : tile + gridDim.x;
        next_of_loop = ntile;
        uint4 vn[BU_BPT];
        if constexpr (PREFETCH) {
            desc_of(ntile, tl);
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) {
                vn[j] = blk_valid(ntile, j * BU_WG + tid) ? bu_ld_stream(blk_src(ntile, j * BU_WG + tid)) : make_uint4(0, 0, 0, 0);
            }
        }
        if (ticket && tid == 0 && ntile < n_tiles) my_draw = atomicAdd(my_counter, 1u);  // (for the tile after the next: needed one tile from now)
        __syncthreads();  // (2) the sorted tile is complete
        // ---- C: whole chunks, wave-uniform mode ----
        // dynamic chunk scheduling: waves take the next chunk as they free up (one LDS atomic per chunk).  The claim for the
        // FOLLOWING chunk is issued before the current one is transcoded, so its LDS round trip hides under the transcode.
        uint32_t c_next = 0;
        if (lane == 0) c_next = atomicAdd(&next_chunk[par], 1u);
        for (;;) {
            const uint32_t c = __builtin_amdgcn_readfirstlane(c_next);
            if (c >= nc) break;
            if (lane == 0) c_next = atomicAdd(&next_chunk[par], 1u);
            // chunk c belongs to the first run whose inclusive chunk count exceeds c
            const uint32_t r = (uint32_t)__builtin_ctzll(__ballot((run_incl >> 16) > c));
            const uint32_t r_pk = (uint32_t)__builtin_amdgcn_readlane((int)run_pk, (int)r), r_ex = (uint32_t)__builtin_amdgcn_readlane((int)run_excl, (int)r);
            const uint32_t k64 = (c - (r_ex >> 16)) << 6;
            const uint32_t s0 = (r_ex & 0xFFFFu) + k64, left = (r_pk & 0xFFFFu) - k64, count = left < 64u ? left : 64u;
            const bool active = lane < count;
            const uint32_t slot = s0 + (active ? lane : 0u);
            const uint4 bv = sblk[slot];
            BuBlk b;
            b.w[0] = bv.x;
            b.w[1] = bv.y;
            b.w[2] = bv.z;
            b.w[3] = bv.w;
            constexpr int NO = TARGET == BU_TGT_RGBA ? 16 : 4;
            uint32_t o[NO];
#pragma unroll
            for (int i = 0; i < NO; i++) o[i] = 0;
            int st = BU_ST_BAD_MODE;
            if (active) {
                switch (r) {  // the run number IS the sort key: run k holds mode BU_COST_ORDER[k] (run 19: invalid mode codes)
#define BU_CASE(k) \
    case k: st = bu_block_mode<TARGET, BU_COST_ORDER[TARGET][k]>(T, b, o); break;
                    BU_CASE(0) BU_CASE(1) BU_CASE(2) BU_CASE(3) BU_CASE(4) BU_CASE(5) BU_CASE(6) BU_CASE(7) BU_CASE(8) BU_CASE(9)
                    BU_CASE(10) BU_CASE(11) BU_CASE(12) BU_CASE(13) BU_CASE(14) BU_CASE(15) BU_CASE(16) BU_CASE(17) BU_CASE(18)
#undef BU_CASE
                default: break;
                }
                // (a failing block leaves o[] at the zeros it was initialised with: every path checks before it writes)
                if constexpr (TARGET == BU_TGT_RGBA) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) sout[r2 * BU_TILE + slot] = make_uint4(o[4 * r2], o[4 * r2 + 1], o[4 * r2 + 2], o[4 * r2 + 3]);
                    sst[slot] = (uint8_t)st;
                } else if constexpr (INBLOCK) {
                    sblk[slot] = make_uint4(o[0], o[1], o[2], o[3] | (uint32_t)st);  // (a failing block's o[] is all zeros)
                } else {
                    sblk[slot] = make_uint4(o[0], o[1], o[2], o[3]);
                    sst[slot] = (uint8_t)st;
                }
            }
        }
        __syncthreads();  // (3) every result is in LDS
        // ---- D: results leave in original order ----
#pragma unroll
        for (int j = 0; j < BU_BPT; j++) {
            if (has_block(key[j])) {
                const unsigned idx = MULTI ? run_idx(td, j * BU_WG + tid) : gidx(tile, j * BU_WG + tid);  // (MULTI: inside the tile's slice)
                void* const out = td.out;                  // (the launch's `out` unless MULTI)
                const unsigned long long base = td.base;
                if constexpr (INBLOCK) {
                    uint4 r = sblk[dest[j]];
                    if ((r.x & 0xFFu) == 0u) {  // no valid block of these formats starts with a zero byte: word 3 is the status
                        bu_report(status, base + idx, (int)r.w);
                        r.w = 0;
                    }
                    bu_st_stream(reinterpret_cast<uint4*>(out) + idx, r);
                    continue;
                }
                const uint32_t st = sst[dest[j]];
                if (st) bu_report(status, base + idx, (int)st);
                if constexpr (TARGET == BU_TGT_RGBA) {
                    unsigned by, bx;
                    if constexpr (RECT) {  // block row and column straight from the tile coordinates: no division
                        const unsigned l = j * BU_WG + tid;
                        unsigned ty, tx;
                        tile_xy(tile, ty, tx);
                        by = (unsigned)(BU_TILE / BU_RECT_W) * ty + l / BU_RECT_W;
                        bx = BU_RECT_W * tx + (l % BU_RECT_W);
                    } else {
                        by = idx / bpr;
                        bx = idx - by * bpr;
                    }
                    uint4* img = reinterpret_cast<uint4*>(out);
#pragma unroll
                    for (int r = 0; r < 4; r++) bu_st_stream(img + (size_t)((4 * by + r) * bpr + bx), sout[r * BU_TILE + dest[j]]);
                } else {
                    const uint4 r = sblk[dest[j]];
                    if constexpr (TARGET == BU_TGT_ETC1) bu_st_stream(reinterpret_cast<uint2*>(out) + idx, make_uint2(r.x, r.y));
                    else bu_st_stream(reinterpret_cast<uint4*>(out) + idx, r);
                }
            }
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) v[j] = vn[j];
            td = tl;
        } else {
            desc_of(ntile, tl);
            td = tl;
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) v[j] = blk_valid(ntile, j * BU_WG + tid) ? bu_ld_stream(blk_src(ntile, j * BU_WG + tid)) : make_uint4(0, 0, 0, 0);
        }
        // no barrier here: the next tile's scatter into `sblk` sits behind its barrier (1), which every wave reaches only
        // after its reads of this tile's results have completed
    }
    if (ticket && tid == 0) {  // the last workgroup out resets the set (every other one has drawn its last ticket by then)
        if (atomicAdd(ticket + BU_TICKET_DONE, 1u) == gridDim.x - 1u) {
            for (unsigned k = 0; k < 8; k++) atomicExch(ticket + k * BU_TICKET_STRIDE, 0u);
            atomicExch(ticket + BU_TICKET_DONE, 0u);
        }
    }
}

// MINW: minimum waves per SIMD the register allocation must leave room for (the second __launch_bounds__ argument)
template <int TARGET, int WGS, int BPT, int MINW, bool PREFETCH, int LAYOUT>
__global__ __launch_bounds__(WGS, MINW) void bu_uastc_sorted_kernel(const uint4* __restrict__ in, void* __restrict__ out, unsigned n_blocks,
                                                                unsigned bpr, unsigned long long base, unsigned long long* status,
                                                                const BuTablesAll* __restrict__ tables, unsigned cus, unsigned tile_rt,
                                                                unsigned* __restrict__ ticket)
{
    static_assert(LAYOUT != BU_LAYOUT_MULTI && LAYOUT != BU_LAYOUT_MULTI_WHOLE, "several runs per launch: bu_uastc_multi_kernel");
    bu_uastc_sorted_body<TARGET, WGS, BPT, PREFETCH, LAYOUT>(in, out, n_blocks, bpr, base, status, tables, cus, tile_rt, nullptr, ticket);
}

// several runs in one launch (layout MULTI): n_tiles 1024-block tiles over the runs of `table` (a kernel argument, by value).
// PREFETCH: the launch is a persistent grid whose workgroups walk many tiles (batches of large slices): the next tile's blocks -- of
// whatever run it belongs to -- are loaded while the current tile is transcoded, as in the one-slice kernel.
// WHOLE: every run of the table is tiled as whole rectangles (the host checked: BuRunDesc::vshift of all of them)
// (ASTC, 512 threads: the register allocation is held to the 64 VGPRs at which FOUR such workgroups fit a CU -- left alone it takes 66, three fit, and a persistent grid of four
//  per CU runs its fourth behind the others: 64 ragged slices of ~2^20 blocks 6.7 -> see profiles/r06_ab_astc_64_vgprs.txt)
// (ETC1 / ETC2 also as 512 x 4 on 2048-block tiles -- the host then numbers 2048-block tiles -- under the shared shape's __launch_bounds__(512, 4): two workgroups per CU)
constexpr bool bu_multi_etc_2048(int target, int tile) { return (target == BU_TGT_ETC1 || target == BU_TGT_ETC2) && tile == 2048; }
template <int TARGET, int WGS, int BPT, bool PREFETCH = false, bool WHOLE = false>
__global__ __launch_bounds__(WGS, (TARGET == BU_TGT_ASTC && WGS == 512) ? 8 : bu_multi_etc_2048(TARGET, WGS * BPT) ? 4 : 1) void bu_uastc_multi_kernel(
    const BuRunTable table, unsigned n_tiles, unsigned bpr, unsigned long long* status, const BuTablesAll* __restrict__ tables, unsigned* __restrict__ ticket)
{
    static_assert(WGS * BPT == 1024 || bu_multi_etc_2048(TARGET, WGS * BPT), "the host numbers 1024-block tiles (ETC1 / ETC2 large batches: 2048-block ones)");
    static_assert(sizeof(BuRunTable) + 40 <= 4096, "the run table and the other arguments share the 4 KiB of kernel arguments");
    bu_uastc_sorted_body<TARGET, WGS, BPT, PREFETCH, WHOLE ? BU_LAYOUT_MULTI_WHOLE : BU_LAYOUT_MULTI>(nullptr, nullptr, n_tiles * (unsigned)(WGS * BPT), bpr, 0ull, status, tables, 0u,
                                                                                                        (unsigned)(WGS * BPT), &table, ticket);
}

// status words back to "no failing block".  A kernel, not hipMemsetAsync: the reset is part of what callers capture into
// hipGraphs, and a captured 8-byte memset node replayed as zeros on ROCm 7.2 (tests/test_gpu_round2.py, graph test).
__global__ void bu_status_reset_kernel(unsigned long long* words, unsigned n)
{
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) words[i] = ~0ull;
}

// ---- CRC-16/GENIBUS of an uploaded file range (basis.rs:364-372; the data CRC of basis.rs:338-341) ----------------------------
// The CRC is linear over GF(2): register(A || B) = register(A) * x^(8|B|) + register(B).  One workgroup covers a 64 KiB piece:
// every thread runs the table-driven register over its own 256 bytes (slicing by four, tables in LDS), multiplies it by
// x^(8 * bytes behind it inside the piece) and the workgroup XORs the 256 contributions.  The host folds the pieces' registers
// (and the few bytes the device never sees: slice table, gaps, tails) in file order -- bu_read_to.
struct BuCrcTables {
    uint16_t t[4][256];  // t[k][b]: register after byte b followed by k zero bytes
    uint16_t pw[256];    // pw[k] = x^(8 * 256 * k) mod P
};
constexpr unsigned BU_CRC_PIECE = 65536;

__device__ __forceinline__ uint32_t bu_crc_gf_mul(uint32_t a, uint32_t b)  // a * b mod x^16 + x^12 + x^5 + 1
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        r ^= (r & 0x10000u) ? 0x11021u : 0u;
        r ^= ((b >> i) & 1u) ? a : 0u;
    }
    return r & 0xFFFFu;
}

__global__ __launch_bounds__(256) void bu_crc16_pieces_kernel(const uint4* __restrict__ data, uint16_t* __restrict__ partial,
                                                              const BuCrcTables* __restrict__ tables)
{
    __shared__ BuCrcTables C;
    __shared__ uint32_t red[4];
    static_assert(sizeof(BuCrcTables) % 16 == 0, "copied in 16-byte pieces");
    for (unsigned i = threadIdx.x; i < sizeof(BuCrcTables) / 16; i += 256) reinterpret_cast<uint4*>(&C)[i] = reinterpret_cast<const uint4*>(tables)[i];
    __syncthreads();
    const uint4* p = data + (size_t)blockIdx.x * (BU_CRC_PIECE / 16) + threadIdx.x * 16;
    uint32_t s = 0;
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const uint4 v = p[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {  // bytes in memory order: b0 = low byte of the little-endian word
            const uint32_t x = w[k];
            s = (uint32_t)C.t[3][((s >> 8) ^ x) & 255u] ^ (uint32_t)C.t[2][(s ^ (x >> 8)) & 255u] ^ (uint32_t)C.t[1][(x >> 16) & 255u] ^ (uint32_t)C.t[0][x >> 24];
        }
    }
    uint32_t c = bu_crc_gf_mul(s, C.pw[255u - threadIdx.x]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (uint16_t)(red[0] ^ red[1] ^ red[2] ^ red[3]);
}

// one wave that does nothing for `ticks` ticks of the constant 100 MHz clock (bu_context_probe_streams: kernels of different streams that
// sit on different hardware queues sleep side by side, kernels of streams that share a queue one after the other)
__global__ __launch_bounds__(64) void bu_sleep_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// uint4 -> uint4 copy (measurement only): the practical ceiling any 16 B in / 16 B out kernel is compared with -- one pass per thread, every load in flight before
// the first store, nontemporal both ways.  The shape is the fastest of tools/exp/copy_big.hip at the size (profiles/r06_copy_ceiling_by_size.txt): 1024 threads x one
// element from 2^22 elements on (2^25: 164.5 us = 6.5 TB/s = 0.82 of the data sheet; 256 x 4: 166, 512 x 4 -- this kernel until round 6 -- 173.6, plain loads and
// stores 181), 256 x 4 below (2^20: 6.7 us against 7.0 / 7.1).
template <int WG, int EPT>
__global__ __launch_bounds__(WG) void bu_copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (WG * EPT) + threadIdx.x;
    uint4 v[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const size_t i = base + (size_t)k * WG;
        if (i < n) v[k] = bu_ld_stream(in + i);
    }
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const size_t i = base + (size_t)k * WG;
        if (i < n) {
            bu_v4u r;
            r.x = v[k].x; r.y = v[k].y; r.z = v[k].z; r.w = v[k].w;
            __builtin_nontemporal_store(r, reinterpret_cast<bu_v4u*>(out + i));
        }
    }
}

// ---- ETC1S back-end ----------------------------------------------------------------------------
// basis_lz/mod.rs:122-146 for one block: 16 texels = colours[selector] of the colour endpoint, alpha = colours[selector].g of the
// alpha slice's endpoint (:139-143).  Byte palettes: a channel's four colours are one etc1s_pal word (etc.rs:396-431 tabulated),
// and the selectors of a block COLUMN are byte-aligned -- texel (x, y) sits at bits 8y + 2x of `rows` (etc.rs:354-361), so
// (rows >> 2x) & 0x03030303 is the column's four selectors, one per byte: exactly a v_perm_b32 selector.  One v_perm per channel
// and column looks the four texels up, two levels of byte permutes turn the channel columns into texel words (round 2: sixteen
// four-way select chains per plane).
__device__ __forceinline__ void bu_etc1s_block_rgba(const uint32_t* pal_lut, uint32_t ep, uint32_t rows, bool has_a, uint32_t aep, uint32_t arows,
                                                    uint32_t px[16])
{
    const uint32_t it = (ep >> 19) & 0xE0u;  // inten << 5
    const uint32_t pr = pal_lut[it | (ep & 31u)], pg = pal_lut[it | ((ep >> 8) & 31u)], pb = pal_lut[it | ((ep >> 16) & 31u)];
    uint32_t pa = 0;
    if (has_a) pa = pal_lut[((aep >> 19) & 0xE0u) | ((aep >> 8) & 31u)];  // .a = colors[sel].g of the alpha endpoint
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const uint32_t sel = (rows >> (2 * x)) & 0x03030303u;
        const uint32_t r = bu_perm(0u, pr, sel), g = bu_perm(0u, pg, sel), b = bu_perm(0u, pb, sel);
        const uint32_t t01 = bu_perm(g, r, 0x05010400u), t23 = bu_perm(g, r, 0x07030602u);  // R0 G0 R1 G1 / R2 G2 R3 G3
        if (has_a) {
            const uint32_t a = bu_perm(0u, pa, (arows >> (2 * x)) & 0x03030303u);
            const uint32_t u01 = bu_perm(a, b, 0x05010400u), u23 = bu_perm(a, b, 0x07030602u);
            px[x] = bu_perm(u01, t01, 0x05040100u);
            px[4 + x] = bu_perm(u01, t01, 0x07060302u);
            px[8 + x] = bu_perm(u23, t23, 0x05040100u);
            px[12 + x] = bu_perm(u23, t23, 0x07060302u);
        } else {
            px[x] = bu_perm(b, t01, 0x0D040100u);  // R G B 255
            px[4 + x] = bu_perm(b, t01, 0x0D050302u);
            px[8 + x] = bu_perm(b, t23, 0x0D060100u);
            px[12 + x] = bu_perm(b, t23, 0x0D070302u);
        }
    }
}

// basis_lz/mod.rs:163-181
__global__ __launch_bounds__(BU_WG) void bu_etc1s_etc1_kernel(const uint32_t* __restrict__ idx, size_t n_blocks,
                                                              const uint32_t* __restrict__ endpoints, uint32_t n_ep,
                                                              const uint2* __restrict__ selectors, uint32_t n_sel,
                                                              uint2* __restrict__ out, unsigned long long* status)
{
    const size_t stride = (size_t)gridDim.x * BU_WG;
    for (size_t i = (size_t)blockIdx.x * BU_WG + threadIdx.x; i < n_blocks; i += stride) {
        const uint32_t ix = __builtin_nontemporal_load(idx + i);  // streamed once; the codebook gathers below stay cached
        const uint32_t e = ix & 0xFFFFu, s = ix >> 16;
        uint2 o = make_uint2(0, 0);
        if (e >= n_ep || s >= n_sel) {
            bu_report(status, i, BU_ERR_INDEX_RANGE);
        } else {
            const uint32_t ep = endpoints[e];
            const uint32_t inten = ep >> 24;
            // bytes: r5<<3, g5<<3, b5<<3, inten<<5 | inten<<2 | 0b11 (u8 arithmetic)
            o.x = ((ep << 3) & 0x00F8F8F8u) | ((((inten << 5) | (inten << 2) | 3u) & 0xFFu) << 24);
            o.y = selectors[s].y;
        }
        bu_st_stream(out + i, o);
    }
}

// basis_lz/mod.rs:122-146 (+ the alpha pass :139-143 fused)
__global__ __launch_bounds__(BU_WG) void bu_etc1s_rgba_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ aidx,
                                                              unsigned nbx, size_t n_blocks, const uint32_t* __restrict__ endpoints,
                                                              uint32_t n_ep, const uint2* __restrict__ selectors, uint32_t n_sel,
                                                              uint4* __restrict__ out, unsigned long long* status,
                                                              const BuTablesAll* __restrict__ tables)
{
    __shared__ uint32_t pal_lut[256];
    pal_lut[threadIdx.x] = tables->t.etc1s_pal[threadIdx.x];
    static_assert(BU_WG == 256, "one palette word per thread");
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * BU_WG;
    for (size_t i = (size_t)blockIdx.x * BU_WG + threadIdx.x; i < n_blocks; i += stride) {
        const uint32_t ix = __builtin_nontemporal_load(idx + i);
        const uint32_t e = ix & 0xFFFFu, s = ix >> 16;
        uint32_t ae = 0, as = 0;
        bool bad = e >= n_ep || s >= n_sel;
        if (aidx) {
            const uint32_t ax = __builtin_nontemporal_load(aidx + i);
            ae = ax & 0xFFFFu;
            as = ax >> 16;
            bad = bad || ae >= n_ep || as >= n_sel;
        }
        uint32_t px[16];
#pragma unroll
        for (int k = 0; k < 16; k++) px[k] = 0;
        if (bad) {
            bu_report(status, i, BU_ERR_INDEX_RANGE);
        } else {
            const uint32_t ep = endpoints[e], rows = selectors[s].x;
            uint32_t aep = 0, arows = 0;
            if (aidx) {
                aep = endpoints[ae];
                arows = selectors[as].x;
            }
            bu_etc1s_block_rgba(pal_lut, ep, rows, aidx != nullptr, aep, arows, px);
        }
        const size_t by = i / nbx, bx = i - by * nbx;
#pragma unroll
        for (int r = 0; r < 4; r++) bu_st_stream(out + (4 * by + r) * (size_t)nbx + bx, make_uint4(px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]));
    }
}


// ---- large slices: both codebooks staged in LDS -------------------------------------------------------------------------------
// One persistent 1024-thread workgroup per CU (two where they fit) copies the endpoint codebook (4 B per entry) and the half of the
// selector codebook its target reads (4 B per entry: texel rows for RGBA32, ETC1 selector bytes for ETC1) into dynamic LDS and
// walks the slice with LDS lookups.  Against the L2 gather above (tools/exp/etc1s_sweep.py, 4096 + 8192 entries, cold rotation):
// 2^18 blocks 4.3 / 5.4 us against 4.3 / 5.0 (ETC1 / RGBA32: launch-bound either way), 2^20 6.7 / 16.0 against 10.3 / 23.3,
// 2^22 15.8 / 56.1 against 38.7 / 91.8, 2^24 41.6 / 218 against 147 / 366 us (ETC1 at 4.8 TB/s, RGBA32 at 5.2 TB/s): the gather
// is bound by the L2's random 4- and 8-byte reads, not by HBM.  The launcher takes this kernel from 2^19 blocks up.
template <bool RGBA>
__global__ __launch_bounds__(1024) void bu_etc1s_staged_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ aidx, unsigned nbx,
                                                               size_t n_blocks, const uint32_t* __restrict__ endpoints, uint32_t n_ep,
                                                               const uint2* __restrict__ selectors, uint32_t n_sel, uint8_t* __restrict__ out,
                                                               unsigned long long* status, const BuTablesAll* __restrict__ tables)
{
    extern __shared__ uint32_t bu_etc1s_lds[];
    uint32_t* s_ep = bu_etc1s_lds;
    uint32_t* s_sel = bu_etc1s_lds + n_ep;
    uint32_t* pal_lut = s_sel + n_sel;
    const size_t stride = (size_t)gridDim.x * 1024, first = (size_t)blockIdx.x * 1024 + threadIdx.x;
    // ETC1, index array 8-byte and output 16-byte aligned: FOUR blocks per lane and step, a wave on 256 consecutive blocks -- the
    // lane's blocks 2L, 2L+1 and 128+2L, 128+2L+1, so that both of its 8-byte index loads and both of its 16-byte result stores
    // are contiguous across the wave (512 B / 1 KiB per instruction; four CONSECUTIVE blocks per lane make every store
    // instruction write half of each cache line: 2^22 blocks 16 -> 31 us) -- with the next step's indices already in flight.
    // The first loads are issued BEFORE the codebooks are staged, so their latency hides behind the staging.
    const bool vec4 = !RGBA && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    const size_t n256 = vec4 ? n_blocks / 256 : 0, wstride = stride / 64, wfirst = first / 64;  // 256-block chunks; waves
    const uint2* idx2 = reinterpret_cast<const uint2*>(idx);
    const unsigned lane = threadIdx.x & 63u;
    bu_v2u curA = {0, 0}, curB = {0, 0};
    uint32_t cur = 0, acur = 0;
    if (vec4) {
        if (wfirst < n256) {
            curA = __builtin_nontemporal_load(reinterpret_cast<const bu_v2u*>(idx2 + wfirst * 128 + lane));
            curB = __builtin_nontemporal_load(reinterpret_cast<const bu_v2u*>(idx2 + wfirst * 128 + 64 + lane));
        }
    } else if (first < n_blocks) {
        cur = __builtin_nontemporal_load(idx + first);
        if (RGBA && aidx) acur = __builtin_nontemporal_load(aidx + first);
    }
    // staging, 16 bytes per load where the source allows (selectors: two 8-byte entries, of which the target keeps 4 bytes each)
    if ((reinterpret_cast<uintptr_t>(endpoints) & 15u) == 0) {
        for (uint32_t i = threadIdx.x; i < n_ep / 4; i += 1024) reinterpret_cast<uint4*>(s_ep)[i] = reinterpret_cast<const uint4*>(endpoints)[i];
        for (uint32_t i = (n_ep & ~3u) + threadIdx.x; i < n_ep; i += 1024) s_ep[i] = endpoints[i];
    } else {
        for (uint32_t i = threadIdx.x; i < n_ep; i += 1024) s_ep[i] = endpoints[i];
    }
    if ((reinterpret_cast<uintptr_t>(selectors) & 15u) == 0 && (n_ep & 1u) == 0) {
        for (uint32_t i = threadIdx.x; i < n_sel / 2; i += 1024) {
            const uint4 two = reinterpret_cast<const uint4*>(selectors)[i];
            reinterpret_cast<uint2*>(s_sel)[i] = RGBA ? make_uint2(two.x, two.z) : make_uint2(two.y, two.w);
        }
        if ((n_sel & 1u) && threadIdx.x == 0) s_sel[n_sel - 1] = RGBA ? selectors[n_sel - 1].x : selectors[n_sel - 1].y;
    } else {
        for (uint32_t i = threadIdx.x; i < n_sel; i += 1024) s_sel[i] = RGBA ? selectors[i].x : selectors[i].y;
    }
    if (RGBA && threadIdx.x < 256) pal_lut[threadIdx.x] = tables->t.etc1s_pal[threadIdx.x];
    __syncthreads();
    // basis_lz/mod.rs:163-181 for one block
    auto etc1_block = [&](uint32_t ix, size_t i) {
        const uint32_t e = ix & 0xFFFFu, sl = ix >> 16;
        uint2 o = make_uint2(0, 0);
        if (e >= n_ep || sl >= n_sel) {
            bu_report(status, i, BU_ERR_INDEX_RANGE);
        } else {
            const uint32_t ep = s_ep[e], inten = ep >> 24;
            o.x = ((ep << 3) & 0x00F8F8F8u) | ((((inten << 5) | (inten << 2) | 3u) & 0xFFu) << 24);
            o.y = s_sel[sl];
        }
        return o;
    };
    if (vec4) {
        for (size_t w = wfirst; w < n256; w += wstride) {
            const size_t wn = w + wstride;
            bu_v2u nxtA = {0, 0}, nxtB = {0, 0};
            if (wn < n256) {
                nxtA = __builtin_nontemporal_load(reinterpret_cast<const bu_v2u*>(idx2 + wn * 128 + lane));
                nxtB = __builtin_nontemporal_load(reinterpret_cast<const bu_v2u*>(idx2 + wn * 128 + 64 + lane));
            }
            const size_t i0 = w * 256 + 2 * lane;
            const uint2 a = etc1_block(curA.x, i0), b = etc1_block(curA.y, i0 + 1), c = etc1_block(curB.x, i0 + 128), d = etc1_block(curB.y, i0 + 129);
            uint4* o4 = reinterpret_cast<uint4*>(out) + w * 128 + lane;
            bu_st_stream(o4, make_uint4(a.x, a.y, b.x, b.y));
            bu_st_stream(o4 + 64, make_uint4(c.x, c.y, d.x, d.y));
            curA = nxtA;
            curB = nxtB;
        }
        // the last n_blocks % 256 blocks
        const size_t t = 256 * n256 + first;
        if (t < n_blocks) bu_st_stream(reinterpret_cast<uint2*>(out) + t, etc1_block(__builtin_nontemporal_load(idx + t), t));
        return;
    }
    for (size_t i = first; i < n_blocks; i += stride) {
        const size_t in = i + stride;
        uint32_t nxt = 0, anxt = 0;
        if (in < n_blocks) {
            nxt = __builtin_nontemporal_load(idx + in);
            if (RGBA && aidx) anxt = __builtin_nontemporal_load(aidx + in);
        }
        if constexpr (!RGBA) {
            bu_st_stream(reinterpret_cast<uint2*>(out) + i, etc1_block(cur, i));
        } else {  // basis_lz/mod.rs:122-146
            const uint32_t e = cur & 0xFFFFu, sl = cur >> 16, ae = acur & 0xFFFFu, as = acur >> 16;
            const bool bad = e >= n_ep || sl >= n_sel || (aidx && (ae >= n_ep || as >= n_sel));
            uint32_t px[16];
#pragma unroll
            for (int k = 0; k < 16; k++) px[k] = 0;
            if (bad) bu_report(status, i, BU_ERR_INDEX_RANGE);
            else bu_etc1s_block_rgba(pal_lut, s_ep[e], s_sel[sl], aidx != nullptr, aidx ? s_ep[ae] : 0u, aidx ? s_sel[as] : 0u, px);
            const size_t by = i / nbx, bx = i - by * nbx;
            uint4* img = reinterpret_cast<uint4*>(out);
#pragma unroll
            for (int r = 0; r < 4; r++) bu_st_stream(img + (4 * by + r) * (size_t)nbx + bx, make_uint4(px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]));
        }
        cur = nxt;
        acur = anxt;
    }
}
// blocks from which the staged kernel is launched, and the LDS one CU can give a workgroup (160 KiB less a margin)
constexpr size_t BU_ETC1S_STAGED_MIN = (size_t)1 << 19, BU_ETC1S_LDS_MAX = 152 * 1024;

// ---- whole-file ETC1S launches (bu_read_to): every slice of the file in ONE launch -----------------------------------
// The host concatenates the per-slice index arrays (each padded to a multiple of 64 words) and describes the slices in a
// small table; a wave owns one 64-block unit, finds its slice by a scalar binary search over the units' prefix and then
// does exactly what the per-slice kernels do.  One status word per image, as the sequential drivers report.
struct BuEtc1sSlice {
    uint32_t unit0;     // first 64-block unit of this slice (the table ends with a sentinel holding the total)
    uint32_t n_blocks;  // nbx * nby
    uint32_t nbx;       // blocks per row (RGBA addressing)
    uint32_t idx_ofs;   // colour indices, in words from the start of the staged index buffer
    uint32_t aidx_ofs;  // alpha indices (RGBA with alpha pairs), 0xFFFFFFFF = none
    uint32_t image;     // status word / image number
    uint64_t out_ofs;   // byte offset of the image in the output buffer
};
static_assert(sizeof(BuEtc1sSlice) == 32, "descriptor layout is shared with the host code");

template <bool RGBA>
__global__ __launch_bounds__(BU_WG) void bu_etc1s_file_kernel(const uint32_t* __restrict__ idx, const BuEtc1sSlice* __restrict__ slices, uint32_t n_slices,
                                                              uint32_t unit_begin, uint32_t n_units, const uint32_t* __restrict__ endpoints, uint32_t n_ep,
                                                              const uint2* __restrict__ selectors, uint32_t n_sel, uint8_t* __restrict__ out,
                                                              unsigned long long* status, const BuTablesAll* __restrict__ tables)
{
    __shared__ uint32_t pal_lut[RGBA ? 256 : 1];
    if constexpr (RGBA) {
        pal_lut[threadIdx.x] = tables->t.etc1s_pal[threadIdx.x];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, wpg = BU_WG / 64;
    // units [unit_begin, n_units) of the file: the streamed front door launches the bands of a slice as their rows are decoded
    for (uint32_t unit = unit_begin + blockIdx.x * wpg + (threadIdx.x >> 6); unit < n_units; unit += gridDim.x * wpg) {
        // largest s with slices[s].unit0 <= unit (unit is wave-uniform: the search runs on the scalar unit)
        uint32_t lo = 0, hi = n_slices;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)slices[mid].unit0) <= unit) lo = mid;
            else hi = mid;
        }
        const BuEtc1sSlice sd = slices[lo];
        const uint32_t i = (unit - sd.unit0) * 64u + lane;
        if (i >= sd.n_blocks) continue;
        const uint32_t ix = __builtin_nontemporal_load(idx + sd.idx_ofs + i);
        const uint32_t e = ix & 0xFFFFu, sl = ix >> 16;
        bool bad = e >= n_ep || sl >= n_sel;
        if constexpr (!RGBA) {
            uint2 o = make_uint2(0, 0);
            if (bad) {
                bu_report(status + sd.image, i, BU_ERR_INDEX_RANGE);
            } else {  // basis_lz/mod.rs:163-181
                const uint32_t ep = endpoints[e];
                const uint32_t inten = ep >> 24;
                o.x = ((ep << 3) & 0x00F8F8F8u) | ((((inten << 5) | (inten << 2) | 3u) & 0xFFu) << 24);
                o.y = selectors[sl].y;
            }
            bu_st_stream(reinterpret_cast<uint2*>(out + sd.out_ofs) + i, o);
        } else {  // basis_lz/mod.rs:122-146
            const bool has_a = sd.aidx_ofs != 0xFFFFFFFFu;
            uint32_t ae = 0, as = 0;
            if (has_a) {
                const uint32_t ax = __builtin_nontemporal_load(idx + sd.aidx_ofs + i);
                ae = ax & 0xFFFFu;
                as = ax >> 16;
                bad = bad || ae >= n_ep || as >= n_sel;
            }
            uint32_t px[16];
#pragma unroll
            for (int k = 0; k < 16; k++) px[k] = 0;
            if (bad) {
                bu_report(status + sd.image, i, BU_ERR_INDEX_RANGE);
            } else {
                const uint32_t ep = endpoints[e], rows = selectors[sl].x;
                uint32_t aep = 0, arows = 0;
                if (has_a) {
                    aep = endpoints[ae];
                    arows = selectors[as].x;
                }
                bu_etc1s_block_rgba(pal_lut, ep, rows, has_a, aep, arows, px);
            }
            const uint32_t by = i / sd.nbx, bx = i - by * sd.nbx;
            uint4* img = reinterpret_cast<uint4*>(out + sd.out_ofs);
#pragma unroll
            for (int r = 0; r < 4; r++) bu_st_stream(img + (size_t)(4 * by + r) * sd.nbx + bx, make_uint4(px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]));
        }
    }
}

// ------------------------------------------------------------------------------------------------
unsigned bu_grid_for(size_t n_blocks, int cu_count)
{
    // enough workgroups to fill the chip several times over, capped so every workgroup amortises its
    // table copy over >= 2 batches on large inputs (guide: grid ~ CUs x 8 for memory-bound kernels)
    size_t wgs = (n_blocks + BU_WG - 1) / BU_WG;
    const size_t cap = (size_t)cu_count * 8;
    if (wgs > cap) wgs = cap;
    if (wgs == 0) wgs = 1;
    return (unsigned)wgs;
}

}  // namespace
