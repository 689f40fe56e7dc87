// gfx950 kernels and the C ABI (include/basisu_hip.h) of the UASTC / ETC1S block-transcode path.
//
// Data layout in HBM
//   UASTC slice      n x 16-byte blocks, raster order (uastc.rs:49-75)            -> read once as uint4
//   ASTC/BC7/ETC2    n x 16-byte blocks, same order                                -> written once as uint4
//   ETC1             n x  8-byte blocks                                            -> uint2
//   RGBA32           row-major image, pitch 16*blocks_per_row bytes (uastc.rs:96) -> 4 x uint4 per block
//   tables           one BuTables blob (5.9 KiB), copied to LDS by every workgroup
// Mapping: one lane = one block.  A wave reads 64 x 16 B = 1 KiB contiguous and writes 1 KiB (512 B
// for ETC1; for RGBA32 four 1 KiB row segments when the row has >= 64 blocks).  No MFMA: the work is
// bit-field surgery on 128-bit values; the bound is HBM (see DESIGN.md for bytes/block).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include <chrono>
#include <future>
#include <mutex>
#include <new>

#include "../../include/basisu_hip.h"
#include "bu_basis.hpp"
#include "bu_uastc_dispatch.hpp"

namespace {

constexpr int BU_WG = 256;            // 4 waves
constexpr int BU_TABLE_VEC = (int)(sizeof(BuTables) / 16);
// Below this the plain one-lane-per-block kernel is used.  It runs one mode path per DISTINCT mode present, so it only
// wins for a handful of blocks (BC7: 1 block 2.3 vs 3.3 us, 8 blocks 4.2 vs 3.7 us, 64 blocks 7.8 vs 4.4 us,
// 1024 blocks 16.1 vs 4.8 us; ETC1 at 128 blocks 38.6 vs 15.0 us).
constexpr int BU_SORT_MIN_BLOCKS = 8;

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bu_stage_tables(BuTables& dst, const BuTables* __restrict__ src)
{
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(&dst);
    for (int i = threadIdx.x; i < BU_TABLE_VEC; i += BU_WG) d[i] = s[i];
}

__device__ __forceinline__ void bu_report(unsigned long long* status, unsigned long long block, int st)
{
    if (status) atomicMin(status, (block << 8) | (unsigned long long)st);
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
                                                         const BuTables* __restrict__ tables)
{
    __shared__ BuTables T;
    const size_t stride = (size_t)gridDim.x * BU_WG;
    size_t idx = (size_t)blockIdx.x * BU_WG + threadIdx.x;
    // first block load is issued before the table copy so both are in flight together
    uint4 v = idx < n_blocks ? bu_ld_stream(in + idx) : make_uint4(0, 0, 0, 0);
    bu_stage_tables(T, tables);
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
//   LDS per workgroup: tile 16 B x BU_TILE + tables 5.9 KiB + 1 B x BU_TILE status + counters and the chunk list.
// Environment knobs read by the host code (diagnostics, not configuration): BU_TRACE (phase times of bu_read_to on stderr),
// BU_RUN_PIECE_MIB (piece size of the two-stream upload pipeline, 0 = off).
// modes by descending code-path length (BC7 VALU counts), 5 bits each: entries 0-11 / 12-19
constexpr unsigned long long bu_cost_pack(int from, int n)
{
    unsigned long long v = 0;
    for (int i = 0; i < n; i++) v |= (unsigned long long)BU_COST_ORDER[from + i] << (5 * i);
    return v;
}
constexpr unsigned long long BU_COST_ORDER_LO = bu_cost_pack(0, 12), BU_COST_ORDER_HI = bu_cost_pack(12, 8);
static_assert(BU_COST_ORDER_LO == 0x2c8cb0b0e281123ull && BU_COST_ORDER_HI == 0x9bdb1401caull, "cost order moved");
// mode of sort key k (scalar)
__device__ __forceinline__ uint32_t bu_mode_of_key(uint32_t k)
{
    return (uint32_t)((k < 12 ? (BU_COST_ORDER_LO >> (5 * k)) : (BU_COST_ORDER_HI >> (5 * (k - 12)))) & 31u);
}
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

// WGS threads per workgroup, BPT blocks per thread: tile = WGS*BPT blocks
constexpr int BU_SORT_WGS = 256, BU_SORT_BPT = 4;
// The large-input configurations (>= 512 Ki blocks), per target; all A/B'd inside one run (tools/exp/ab.sh) on the
// BC7 headline, a 4096^2 atlas = 4096 blocks per CU:
//   1024 x 4 (4096-block tile), one workgroup per CU                          13.65 us
//   512 x 4 (2048), two per CU (16 waves)                                     12.80    -- half-size workgroups overlap each
//   + second half of the grid started ~1 us late (s_sleep 40)                 12.65       other's barrier-bound sort phases
//   256x4 13.8, 512x8 16.2, 1024x2 15.9, 256x8 15.7 at the same register count
// The kernels are built with machine-LICM off (basisu_rs_amd/build.py): hoisting every mode path's constants out of the
// chunk loop cost ~30 VGPRs.  BC7 then needs 62 instead of 93, which allows 32 waves per CU:
//   1024 x 2 (2048), two per CU (32 waves)                                    11.6
//   512 x 2 (1024), four per CU (32 waves)                                    11.4     <- BC7, ASTC
// ASTC reaches 63 VGPRs with its modes 3, 4 and 7 rewritten on packed digit strings (bu_uastc_astc.hpp); ETC1/ETC2 (81)
// do not reach 8 waves per SIMD and keep 512 x 4, two per CU.
// ETC1 / ETC2 (72-78 VGPRs): 2048-block tiles, two workgroups per CU.  Tried in round 2 (tools/exp/ab.sh): 1024-block
// tiles with three workgroups per CU 25.7 us, with four (64 VGPRs, 7 spilled) 27.4, against 24.9 -- the ALUs are saturated
// at 16 waves per CU, more waves only add sort overhead.
template <int TARGET>
struct BuBigCfg {
    static constexpr bool PREFETCH = false, DIRECT = false;
    static constexpr int WGS = 512, BPT = 4, WG_PER_CU = 2, SKEW = 40, MINW = 1;
    static constexpr bool ALL_SIZES = false;  // below 2 Ki blocks per CU the launcher switches to 512 x 2 (1024-block tiles)
};
template <>
struct BuBigCfg<BU_TGT_BC7> {
    static constexpr int WGS = 512, BPT = 2, WG_PER_CU = 4, SKEW = 0, MINW = 1;
    static constexpr bool ALL_SIZES = true;  // 8 waves on a 1024-block tile beat 4: 2^16 blocks 7.5 -> 5.7 us, 2^18 8.1 -> 6.3 us
    static constexpr bool PREFETCH = false, DIRECT = false;
};
template <>
struct BuBigCfg<BU_TGT_ASTC> {
    static constexpr bool PREFETCH = false, DIRECT = false;
    static constexpr int WGS = 512, BPT = 2, WG_PER_CU = 4, SKEW = 0, MINW = 1;
    static constexpr bool ALL_SIZES = true;
};
// RGBA32 configuration (tile = 1024 blocks either way)
#ifndef BU_RGBA_WGS
#define BU_RGBA_WGS 512
#define BU_RGBA_BPT 2
#define BU_RGBA_WG_PER_CU 2
#endif
#ifndef BU_RGBA_PREFETCH
#define BU_RGBA_PREFETCH true
#endif
#ifndef BU_RGBA_SKEW
#define BU_RGBA_SKEW 20
#endif


// stage the parts of the table blob TARGET reads (bu_table_range), 16 bytes per thread per step
template <int WGS, int TARGET>
__device__ __forceinline__ void bu_stage_tables_n(BuTables& dst, const BuTables* __restrict__ src)
{
    constexpr BuTableRange R = bu_table_range(TARGET);
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(&dst);
    for (int i = R.lo / 16 + threadIdx.x; i < (int)(R.hi / 16); i += WGS) d[i] = s[i];
    if constexpr (R.lo2 < R.hi2) {
        for (int i = R.lo2 / 16 + threadIdx.x; i < (int)(R.hi2 / 16); i += WGS) d[i] = s[i];
    }
}

#ifndef BU_STAMP
#define BU_STAMP(k)
#define BU_STAMP_ARG
#define BU_STAMP_PASS
#endif
// DIRECT: results are stored to global memory straight from the chunk loop at the block's original
// index (16-byte pieces, not coalesced across lanes) instead of returning through LDS.  Always used for
// RGBA32 (64 B per block do not fit a second LDS tile; `bpr` = blocks per image row).
template <int TARGET, int WGS, int BPT, int MINW = 1, bool PREFETCH = true, bool DIRECT = (TARGET == BU_TGT_RGBA), int SKEW = 0>
__global__ __launch_bounds__(WGS, MINW) void bu_uastc_sorted_kernel(const uint4* __restrict__ in, void* __restrict__ out, unsigned n_blocks,
                                                                unsigned bpr, unsigned long long base, unsigned long long* status,
                                                                const BuTables* __restrict__ tables, unsigned cus BU_STAMP_ARG)
{
    BU_STAMP(0)
    if constexpr (SKEW > 0) {
        if (blockIdx.x >= gridDim.x / 2 && gridDim.x > 1) __builtin_amdgcn_s_sleep(SKEW);
    }
    // Static priority by residency generation.  Workgroups are dealt breadth-first (b, b + CUs, b + 2 CUs, ... share a CU:
    // tools/exp/census.hip), and the instruction arbiter serves the OLDEST wave first, so the four tiles of a CU finish
    // 1.5 us apart and the last one runs its latency-bound chain with the vector units nearly idle (phase stamps,
    // profiles/r02_*stamps*).  Raising the later generations' priority makes them catch up while the older ones fill
    // the gaps: BC7 10.51 -> 10.2 us, ASTC 9.74 -> 9.47, RGBA32 20.4 -> 19.95 in an A/B run.  Speed only: any placement is correct.
    {
        const unsigned gen = blockIdx.x / (cus ? cus : 1u);
        if (gen == 1) __builtin_amdgcn_s_setprio(1);
        if (gen == 2) __builtin_amdgcn_s_setprio(2);
        if (gen >= 3) __builtin_amdgcn_s_setprio(3);
    }
    constexpr int BU_WG = WGS, BU_BPT = BPT, BU_TILE = WGS * BPT;
    __shared__ BuTables T;
    // RGBA32 through LDS: four pixel rows of 16 B per block, stored row-major by row index so that both the
    // sorted-order writes and the original-order reads are 16-byte strided (no bank conflicts).  The sorted input tile
    // lives IN row 0 of that output tile: a lane reads its block from slot s and later overwrites exactly slot s with
    // the block's first pixel row, so no other lane's input is ever clobbered -- 64 KiB instead of 80 per 1024 blocks,
    // which is what lets two workgroups share a CU.
    constexpr bool BU_ALIAS = (TARGET == BU_TGT_RGBA && !DIRECT);
    // two RGBA32 workgroups must fit the 160 KiB of a CU: output tile + table blob + status bytes + counters / chunk list
    static_assert(!BU_ALIAS || sizeof(BuTables) + 4 * BU_TILE * 16 + BU_TILE + 1536 <= 80 * 1024,
                  "the RGBA32 workgroup no longer fits twice per CU: shrink BuTables or stage it per target in LDS too");
    __shared__ uint4 sblk_store[BU_ALIAS ? 1 : BU_TILE];
    __shared__ uint4 sout[BU_ALIAS ? 4 * BU_TILE : 1];
    uint4* const sblk = BU_ALIAS ? sout : sblk_store;
    __shared__ uint8_t sst[DIRECT ? 16 : BU_TILE];
    __shared__ uint16_t sorig[DIRECT ? BU_TILE : 16];
    // counters and the chunk ticket are double-buffered by tile parity: the buffer of tile t+1 is cleared during tile t,
    // after everyone has finished with its previous use (tile t-1), so no barrier is spent on the reset
    __shared__ uint32_t cnt[2][32], next_chunk[2];
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned n_tiles = (n_blocks + BU_TILE - 1) / BU_TILE;  // 32-bit indices: the host splits launches above 2^26 blocks
    unsigned tile = blockIdx.x;
    uint4 v[BU_BPT];
#pragma unroll
    for (int j = 0; j < BU_BPT; j++) {
        const unsigned idx = tile * BU_TILE + j * BU_WG + tid;
        v[j] = (tile < n_tiles && idx < n_blocks) ? bu_ld_stream(in + idx) : make_uint4(0, 0, 0, 0);
    }
    bu_stage_tables_n<WGS, TARGET>(T, tables);
    if (tid < 64) (&cnt[0][0])[tid] = 0;
    if (tid < 2) next_chunk[tid] = 0;
    __syncthreads();
    BU_STAMP(1)
    unsigned par = 0;
    for (; tile < n_tiles; tile += gridDim.x, par ^= 1u) {
        const unsigned tbase = tile * BU_TILE;
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
            const bool valid = tbase + j * BU_WG + tid < n_blocks;
            key[j] = valid ? T.key_lut[v[j].x & 127u] : 31u;
            uniform = uniform && (__ballot(key[j] == (uint32_t)__builtin_amdgcn_readfirstlane(key[j])) == ~0ull) && key[j] < 20u;
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
        BU_STAMP(2)
        __syncthreads();  // (1) every rank is final
        BU_STAMP(3)
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
            dest[j] = key[j] < 20u ? st + pos[j] : 0u;
            if (key[j] < 20u) {
                sblk[dest[j]] = v[j];
                if constexpr (DIRECT) sorig[dest[j]] = (uint16_t)(j * BU_WG + tid);
            }
        }
        // prefetch the next tile while this one is transcoded
        const unsigned ntile = tile + gridDim.x;
        uint4 vn[BU_BPT];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) {
                const unsigned idx = ntile * BU_TILE + j * BU_WG + tid;
                vn[j] = (ntile < n_tiles && idx < n_blocks) ? bu_ld_stream(in + idx) : make_uint4(0, 0, 0, 0);
            }
        }
        BU_STAMP(4)
        __syncthreads();  // (2) the sorted tile is complete
        BU_STAMP(5)
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
            const uint32_t m = bu_mode_of_key(r), s0 = (r_ex & 0xFFFFu) + k64, left = (r_pk & 0xFFFFu) - k64, count = left < 64u ? left : 64u;
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
                switch (m) {
#define BU_CASE(k) \
    case k: st = bu_block_mode<TARGET, k>(T, b, o); break;
                    BU_CASE(0) BU_CASE(1) BU_CASE(2) BU_CASE(3) BU_CASE(4) BU_CASE(5) BU_CASE(6) BU_CASE(7) BU_CASE(8) BU_CASE(9)
                    BU_CASE(10) BU_CASE(11) BU_CASE(12) BU_CASE(13) BU_CASE(14) BU_CASE(15) BU_CASE(16) BU_CASE(17) BU_CASE(18)
#undef BU_CASE
                default: break;
                }
                // (a failing block leaves o[] at the zeros it was initialised with: every path checks before it writes)
                if constexpr (DIRECT) {
                    const unsigned idx = tbase + sorig[slot];
                    if (st) bu_report(status, base + idx, st);
                    if constexpr (TARGET == BU_TGT_RGBA) {
                        const unsigned by = idx / bpr, bx = idx - by * bpr;
                        uint4* img = reinterpret_cast<uint4*>(out);
#pragma unroll
                        for (int r2 = 0; r2 < 4; r2++) img[(size_t)((4 * by + r2) * bpr + bx)] = make_uint4(o[4 * r2], o[4 * r2 + 1], o[4 * r2 + 2], o[4 * r2 + 3]);
                    } else if constexpr (TARGET == BU_TGT_ETC1) {
                        reinterpret_cast<uint2*>(out)[idx] = make_uint2(o[0], o[1]);
                    } else {
                        reinterpret_cast<uint4*>(out)[idx] = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                } else if constexpr (TARGET == BU_TGT_RGBA) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) sout[r2 * BU_TILE + slot] = make_uint4(o[4 * r2], o[4 * r2 + 1], o[4 * r2 + 2], o[4 * r2 + 3]);
                    sst[slot] = (uint8_t)st;
                } else {
                    sblk[slot] = make_uint4(o[0], o[1], o[2], o[3]);
                    sst[slot] = (uint8_t)st;
                }
            }
        }
        BU_STAMP(6)
        __syncthreads();  // (3) every result is in LDS
        BU_STAMP(7)
        // ---- D: results leave in original order ----
        if constexpr (!DIRECT) {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) {
                if (key[j] < 20u) {
                    const unsigned idx = tbase + j * BU_WG + tid;
                    const uint32_t st = sst[dest[j]];
                    if (st) bu_report(status, base + idx, (int)st);
                    if constexpr (TARGET == BU_TGT_RGBA) {
                        const unsigned by = idx / bpr, bx = idx - by * bpr;
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
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) v[j] = vn[j];
        } else {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) {
                const unsigned idx = ntile * BU_TILE + j * BU_WG + tid;
                v[j] = (ntile < n_tiles && idx < n_blocks) ? bu_ld_stream(in + idx) : make_uint4(0, 0, 0, 0);
            }
        }
        // no barrier here: the next tile's scatter into `sblk` sits behind its barrier (1), which every wave reaches only
        // after its reads of this tile's results have completed
    }
    BU_STAMP(8)
}

// status words back to "no failing block".  A kernel, not hipMemsetAsync: the reset is part of what callers capture into
// hipGraphs, and a captured 8-byte memset node replayed as zeros on ROCm 7.2 (tests/test_gpu_round2.py, graph test).
__global__ void bu_status_reset_kernel(unsigned long long* words, unsigned n)
{
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) words[i] = ~0ull;
}

// uint4 -> uint4 copy with the transcoders' launch shape (measurement only)
__global__ __launch_bounds__(BU_WG) void bu_copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * BU_WG;
    for (size_t idx = (size_t)blockIdx.x * BU_WG + threadIdx.x; idx < n; idx += stride) bu_st_stream(out + idx, bu_ld_stream(in + idx));  // same streaming hints as the transcoders

}

// ---- ETC1S back-end ----------------------------------------------------------------------------
// etc.rs:396-431 for one base colour: colour k = clamp(extend5(c5) + modifier[inten][k])
__device__ __forceinline__ uint32_t bu_etc1s_color(const int16_t* mods, uint32_t ep, int k)
{
    const int md = mods[((ep >> 24) & 7u) * 4 + k];
    uint32_t c = 0xFF000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const uint32_t c5 = (ep >> (8 * ch)) & 0xFFu;
        const int base = (int)(((c5 << 3) | (c5 >> 2)) & 0xFFu);
        const int v = base + md;
        c |= (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * ch);
    }
    return c;
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
                                                              const BuTables* __restrict__ tables)
{
    __shared__ int16_t mods[32];
    if (threadIdx.x < 32) mods[threadIdx.x] = tables->etc1_mod[threadIdx.x];
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
            const uint32_t ep = endpoints[e];
            const uint32_t rows = selectors[s].x;
            uint32_t col[4];
#pragma unroll
            for (int k = 0; k < 4; k++) col[k] = bu_etc1s_color(mods, ep, k);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint32_t sel = (rows >> (2 * t)) & 3u;  // row y in byte y, x = 0 in the low bits (etc.rs:354-361)
                px[t] = sel == 0 ? col[0] : sel == 1 ? col[1] : sel == 2 ? col[2] : col[3];
            }
            if (aidx) {
                const uint32_t aep = endpoints[ae];
                const uint32_t arows = selectors[as].x;
                uint32_t ag[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ag[k] = (bu_etc1s_color(mods, aep, k) >> 8) & 0xFFu;  // .a = colors[sel].g
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const uint32_t sel = (arows >> (2 * t)) & 3u;
                    const uint32_t a = sel == 0 ? ag[0] : sel == 1 ? ag[1] : sel == 2 ? ag[2] : ag[3];
                    px[t] = (px[t] & 0x00FFFFFFu) | (a << 24);
                }
            }
        }
        const size_t by = i / nbx, bx = i - by * nbx;
#pragma unroll
        for (int r = 0; r < 4; r++) bu_st_stream(out + (4 * by + r) * (size_t)nbx + bx, make_uint4(px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]));
    }
}


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
                                                              uint32_t n_units, const uint32_t* __restrict__ endpoints, uint32_t n_ep,
                                                              const uint2* __restrict__ selectors, uint32_t n_sel, uint8_t* __restrict__ out,
                                                              unsigned long long* status, const BuTables* __restrict__ tables)
{
    __shared__ int16_t mods[32];
    if constexpr (RGBA) {
        if (threadIdx.x < 32) mods[threadIdx.x] = tables->etc1_mod[threadIdx.x];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, wpg = BU_WG / 64;
    for (uint32_t unit = blockIdx.x * wpg + (threadIdx.x >> 6); unit < n_units; unit += gridDim.x * wpg) {
        // largest s with slices[s].unit0 <= unit (unit is wave-uniform: the search runs on the scalar unit)
        uint32_t lo = 0, hi = n_slices;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (__builtin_amdgcn_readfirstlane(slices[mid].unit0) <= unit) lo = mid;
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
                const uint32_t ep = endpoints[e];
                const uint32_t rows = selectors[sl].x;
                uint32_t col[4];
#pragma unroll
                for (int k = 0; k < 4; k++) col[k] = bu_etc1s_color(mods, ep, k);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const uint32_t sel = (rows >> (2 * t)) & 3u;
                    px[t] = sel == 0 ? col[0] : sel == 1 ? col[1] : sel == 2 ? col[2] : col[3];
                }
                if (has_a) {
                    const uint32_t aep = endpoints[ae];
                    const uint32_t arows = selectors[as].x;
                    uint32_t ag[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) ag[k] = (bu_etc1s_color(mods, aep, k) >> 8) & 0xFFu;
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const uint32_t sel = (arows >> (2 * t)) & 3u;
                        const uint32_t a = sel == 0 ? ag[0] : sel == 1 ? ag[1] : sel == 2 ? ag[2] : ag[3];
                        px[t] = (px[t] & 0x00FFFFFFu) | (a << 24);
                    }
                }
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

// ================================================================================================
struct bu_context {
    int device = -1;
    int cu_count = 256;
    hipStream_t stream = nullptr;
    BuTables* d_tables = nullptr;
    void* d_in = nullptr;
    size_t in_cap = 0;
    void* d_out = nullptr;
    size_t out_cap = 0;
    void* d_aux = nullptr;  // codebooks / alpha indices of the host-pointer ETC1S calls
    size_t aux_cap = 0;
    unsigned long long* d_status = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t extra_streams[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
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
            // large inputs: the per-target BuBigCfg configuration, see its definition
#define BU_LAUNCH_SORTED(T)                                                                                                             \
    if (grid_cap == 0 && (many || BuBigCfg<T>::ALL_SIZES)) {                                                                           \
        using C = BuBigCfg<T>;                                                                                                          \
        const size_t btiles = (nb + (size_t)C::WGS * C::BPT - 1) / ((size_t)C::WGS * C::BPT);                                           \
        const size_t bcap = (size_t)ctx->cu_count * C::WG_PER_CU;                                                                       \
        hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, C::WGS, C::BPT, C::MINW, C::PREFETCH, C::DIRECT, C::SKEW>), dim3((unsigned)(btiles < bcap ? btiles : bcap)), \
                           dim3(C::WGS), 0, stream, pin, pout, (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, (unsigned)ctx->cu_count BU_STAMP_PASS);      \
    } else if (grid_cap == 0) {                                                                                                         \
        /* fewer than two 1024-block tiles per CU: 8 waves per tile, every tile resident (ETC1 at 2^16 blocks: 14.1 -> 11.3 us) */     \
        hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, 512, 2, 1, false, false, 0>), dim3((unsigned)((nb + 1023) / 1024)), dim3(512), 0, stream, pin, pout, \
                           (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, (unsigned)ctx->cu_count BU_STAMP_PASS);                                         \
    } else                                                                                                                              \
        hipLaunchKernelGGL((bu_uastc_sorted_kernel<T, BU_SORT_WGS, BU_SORT_BPT>), dim3(sgrid), dim3(BU_SORT_WGS), 0, stream, pin, pout,  \
                           (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, (unsigned)ctx->cu_count BU_STAMP_PASS);
            const bool many = nb >= (size_t)4096 * (size_t)ctx->cu_count / 2;
            switch (target) {
            case BU_TARGET_ASTC: BU_LAUNCH_SORTED(BU_TGT_ASTC) break;
            case BU_TARGET_BC7: BU_LAUNCH_SORTED(BU_TGT_BC7) break;
            case BU_TARGET_ETC1: BU_LAUNCH_SORTED(BU_TGT_ETC1) break;
            case BU_TARGET_RGBA32: {
                // 64 B of output per block: results return through a 64 KiB LDS tile (1024 blocks x 4 rows, the input tile
                // aliased into row 0) so the image rows leave as coalesced 1 KiB stores; persistent workgroups walk their
                // tiles with prefetch.  BU_RGBA_WGS threads x BU_RGBA_BPT blocks, BU_RGBA_WG_PER_CU resident per CU.
                const size_t rtiles = (nb + (BU_RGBA_WGS * BU_RGBA_BPT) - 1) / (BU_RGBA_WGS * BU_RGBA_BPT);
                const size_t rcap = grid_cap ? (size_t)grid_cap : (size_t)ctx->cu_count * BU_RGBA_WG_PER_CU;
                const unsigned rgrid = (unsigned)(rtiles < rcap ? rtiles : rcap);
                hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_RGBA, BU_RGBA_WGS, BU_RGBA_BPT, 1, BU_RGBA_PREFETCH, false, BU_RGBA_SKEW>), dim3(rgrid), dim3(BU_RGBA_WGS), 0, stream, pin,
                                   pout, (unsigned)nb, (unsigned)bpr, pbase, st, ctx->d_tables, (unsigned)ctx->cu_count BU_STAMP_PASS);
            } break;
            default: BU_LAUNCH_SORTED(BU_TGT_ETC2) break;
            }
#undef BU_LAUNCH_SORTED
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

}  // namespace

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
    bu_status st = BU_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { st = BU_ERR_NO_DEVICE; break; }
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_tables), sizeof(BuTables)) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_status), 64) != hipSuccess) { st = BU_ERR_HIP; break; }
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { st = BU_ERR_HIP; break; }
        BuTables* h = new (std::nothrow) BuTables();
        if (!h) { st = BU_ERR_HIP; break; }
        bu_build_tables(h);
        hipError_t e = hipMemcpy(ctx->d_tables, h, sizeof(BuTables), hipMemcpyHostToDevice);
        delete h;
        if (e != hipSuccess) { st = BU_ERR_HIP; break; }
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
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_in) (void)hipFree(ctx->d_in);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_aux) (void)hipFree(ctx->d_aux);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
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
    return bu_launch_uastc(ctx, target, d_in, n_blocks, d_out, blocks_per_row, block_index_base, d_status, static_cast<hipStream_t>(stream));
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

bu_status bu_unpack_uastc_block_to_rgba(bu_context* ctx, const uint8_t in[16], uint32_t out[16])
{
    return bu_uastc_host(ctx, BU_TARGET_RGBA32, in, 16, 1, reinterpret_cast<uint8_t*>(out), 64, nullptr);
}
bu_status bu_transcode_uastc_block_to_astc(bu_context* ctx, const uint8_t in[16], uint8_t out[16])
{
    return bu_uastc_host(ctx, BU_TARGET_ASTC, in, 16, 1, out, 16, nullptr);
}
bu_status bu_transcode_uastc_block_to_bc7(bu_context* ctx, const uint8_t in[16], uint8_t out[16])
{
    return bu_uastc_host(ctx, BU_TARGET_BC7, in, 16, 1, out, 16, nullptr);
}
bu_status bu_transcode_uastc_block_to_etc1(bu_context* ctx, const uint8_t in[16], uint8_t out[8])
{
    return bu_uastc_host(ctx, BU_TARGET_ETC1, in, 16, 1, out, 8, nullptr);
}
bu_status bu_transcode_uastc_block_to_etc2(bu_context* ctx, const uint8_t in[16], uint8_t out[16])
{
    return bu_uastc_host(ctx, BU_TARGET_ETC2, in, 16, 1, out, 16, nullptr);
}

// ---- ETC1S ---------------------------------------------------------------------------------------
void bu_etc1s_selector_from_rows(const uint8_t rows[4], uint8_t out_entry[8]) { bu_host::selector_from_rows(rows, out_entry); }

bu_status bu_etc1s_transcode_etc1_device(bu_context* ctx, const uint32_t* d_idx, size_t n_blocks, const uint32_t* d_endpoints,
                                         uint32_t n_endpoints, const void* d_selectors, uint32_t n_selectors, void* d_out,
                                         uint64_t* d_status, void* stream)
{
    if (!ctx || (n_blocks && (!d_idx || !d_endpoints || !d_selectors || !d_out))) return BU_ERR_ARGUMENT;
    if (n_blocks == 0) return BU_OK;
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

// ---- whole-file level (basis.rs) --------------------------------------------------------------------
bu_status bu_basis_read_header(const uint8_t* file, size_t len, bu_basis_header* out)
{
    if (!file || !out) return BU_ERR_ARGUMENT;
    return bu_host::read_header(file, len, out);
}

static bu_status bu_basis_read_slice_descs_impl(const uint8_t* file, size_t len, const bu_basis_header* header, bu_slice_desc* out, size_t max_descs,
                                              size_t* n_descs)
{
    if (!file || !header) return BU_ERR_ARGUMENT;
    std::vector<bu_slice_desc> v;
    bu_status st = bu_host::read_slice_descs(file, len, header, v);
    if (st) return st;
    if (n_descs) *n_descs = v.size();
    if (out) {
        if (v.size() > max_descs) return BU_ERR_OUTPUT_SIZE;
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    }
    return BU_OK;
}

uint16_t bu_basis_crc16(const uint8_t* data, size_t len, uint16_t crc) { return bu_host::crc16(data, len, crc); }

using bu_host::BuFilePlan;
using bu_host::bu_plan_file;
using bu_host::bu_make_lz;

static bu_status bu_read_query_impl(bu_read_target target, const uint8_t* file, size_t len, size_t* n_images, size_t* out_bytes)
{
    BuFilePlan p;
    bu_status st = bu_plan_file(target, file, len, p);
    if (st) return st;
    if (n_images) *n_images = p.images.size();
    if (out_bytes) *out_bytes = p.out_bytes;
    return BU_OK;
}

static bu_status bu_basislz_decode_impl(const uint8_t* file, size_t len, uint32_t slice_index, uint32_t* endpoints_out, uint8_t* selectors_out,
                                       uint32_t* idx_out)
{
    if (!file) return BU_ERR_ARGUMENT;
    bu_basis_header h;
    bu_status st = bu_host::read_header(file, len, &h);
    if (st) return st;
    if (h.tex_format != 0) return BU_ERR_UNSUPPORTED;
    std::vector<bu_slice_desc> slices;
    st = bu_host::read_slice_descs(file, len, &h, slices);
    if (st) return st;
    bu_host::BasisLz lz;
    st = bu_make_lz(file, len, h, lz);
    if (st) return st;
    if (endpoints_out) memcpy(endpoints_out, lz.endpoints.data(), lz.endpoints.size() * 4);
    if (selectors_out) memcpy(selectors_out, lz.selectors.data(), lz.selectors.size());
    if (idx_out) {
        if (slice_index >= slices.size()) return BU_ERR_ARGUMENT;
        const bu_slice_desc& s = slices[slice_index];
        if (!bu_host::in_file(len, s.file_ofs, s.file_size)) return BU_ERR_BOUNDS;
        st = lz.decode_slice(s.num_blocks_x, s.num_blocks_y, file + s.file_ofs, s.file_size, idx_out);
    }
    return st;
}

static bu_status bu_read_to_impl(bu_context* ctx, bu_read_target target, const uint8_t* file, size_t len, bu_basis_header* header_out, bu_image* images,
                                size_t max_images, size_t* n_images, uint8_t* out, size_t out_bytes)
{
    if (!ctx || !out) return BU_ERR_ARGUMENT;
    const bool trace = getenv("BU_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t = now();
        fprintf(stderr, "[bu_read_to] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    BuFilePlan p;
    // Large UASTC files: the payload CRC (0.3 ms per 16 MiB on the host cores) runs beside the upload instead of before
    // it.  The reference checks it before anything else behind the header (basis.rs:338-341), so a CRC failure takes
    // precedence over every later error, and nothing is reported as success before it is known.
    std::future<bool> crc_later;
    bool crc_deferred = false;
    {
        bu_basis_header h0;
        if (file && len >= ((size_t)1 << 20) && bu_host::read_header(file, len, &h0) == BU_OK && h0.tex_format == 1 && target != BU_READ_UASTC) {
            const uint16_t want = h0.data_crc16;
            try {
                crc_later = std::async(std::launch::async, [file, len, want] { return bu_host::crc16(file + 77, len - 77, 0) == want; });
                crc_deferred = true;
            } catch (...) {  // no thread to be had: the plan below checks the CRC inline, in the reference's order
                crc_deferred = false;
            }
        }
    }
    auto settle = [&](bu_status s) {  // the status to report once the deferred CRC is known
        if (crc_deferred) {
            crc_deferred = false;
            if (!crc_later.get()) return BU_ERR_DATA_CRC;
        }
        return s;
    };
    bu_status st = bu_plan_file(target, file, len, p, !crc_deferred);
    lap("plan (parse + CRCs)");
    if (st) return settle(st);
    const auto rest = [&]() -> bu_status {
    if (header_out) *header_out = p.h;
    if (n_images) *n_images = p.images.size();
    if (out_bytes < p.out_bytes) return BU_ERR_OUTPUT_SIZE;
    if (images) {
        if (p.images.size() > max_images) return BU_ERR_OUTPUT_SIZE;
        for (size_t i = 0; i < p.images.size(); i++) images[i] = p.images[i];
    }
    if (p.images.empty()) return BU_OK;
    if (!p.etc1s && target == BU_READ_UASTC) {  // uastc.rs:85-87: plain copies, no device work
        for (size_t k = 0; k < p.images.size(); k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            if (s.file_size) memcpy(out + p.images[k].offset, file + s.file_ofs, s.file_size);
        }
        return BU_OK;
    }
    // Batched front door: every slice's input is staged at an aligned offset of one device buffer, the device output
    // buffer mirrors `out`, all launches go to the context stream back to back (one status word per image) and a
    // single synchronisation ends the call.  The host-side BasisLZ decode of all slices happens before any upload.
    bu_host::BasisLz lz;
    const size_t n_img = p.images.size();
    std::vector<size_t> in_off(n_img, 0), ain_off(n_img, 0), run_of(n_img, 0);  // run_of[k]: first image of k's run
    std::vector<uint32_t> idx_all;
    size_t total_in = 0;
    auto align_up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    if (p.etc1s) {
        st = bu_make_lz(file, len, p.h, lz);
        if (st) return st;
        lap("codebooks + tables");
        size_t words = 0;
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            const size_t nblk = (size_t)s.num_blocks_x * s.num_blocks_y;
            in_off[k] = words * 4;
            words += (nblk + 63) & ~(size_t)63;
            if (p.alpha_pairs) {
                ain_off[k] = words * 4;
                words += (nblk + 63) & ~(size_t)63;
            }
        }
        idx_all.assign(words ? words : 1, 0);
        std::vector<bu_host::SliceJob> jobs;  // file order: colour slice, then its alpha slice
        jobs.reserve(n_img * (p.alpha_pairs ? 2 : 1));
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            jobs.push_back({s.num_blocks_x, s.num_blocks_y, file + s.file_ofs, s.file_size, idx_all.data() + in_off[k] / 4, BU_OK});
            if (p.alpha_pairs) {
                const bu_slice_desc& a = p.slices[p.first_slice[k] + 1];
                jobs.push_back({a.num_blocks_x, a.num_blocks_y, file + a.file_ofs, a.file_size, idx_all.data() + ain_off[k] / 4, BU_OK});
            }
        }
        st = bu_host::decode_slices(lz, jobs);  // host cores in parallel; the symbol stream is serial only within a slice
        if (st) return st;
        lap("slice symbol streams");
        total_in = words * 4;
    } else {
        // Runs: consecutive slices that sit back to back in the file (the usual layout of a mip chain or a texture array)
        // are staged back to back with ONE upload, and -- for the block-linear targets, whose outputs are then contiguous
        // too -- transcoded with ONE launch over the whole run: 512 slices of 65 536 blocks are one 33 M-block launch
        // (0.3 ms) instead of 512 latency-bound ones (3.9 ms).  The lowest failing block of a run lies in its first
        // failing slice, so the reported error is the sequential loop's.
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            const bool joins = k > 0 && run_of[k - 1] != SIZE_MAX && s.file_size % 16 == 0 && s.file_size != 0 &&
                               p.slices[p.first_slice[k - 1]].file_size % 16 == 0 && p.slices[p.first_slice[k - 1]].file_size != 0 &&
                               (size_t)p.slices[p.first_slice[k - 1]].file_ofs + p.slices[p.first_slice[k - 1]].file_size == s.file_ofs;
            if (joins) {
                run_of[k] = run_of[k - 1];
                in_off[k] = in_off[k - 1] + p.slices[p.first_slice[k - 1]].file_size;
                total_in = in_off[k] + s.file_size;
            } else {
                total_in = align_up(total_in);
                run_of[k] = k;
                in_off[k] = total_in;
                total_in += s.file_size;
            }
        }
        total_in = align_up(total_in);
    }
    std::lock_guard<std::mutex> g(ctx->lock);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> words(n_img, 0);  // status landing area: declared before anything is queued, outlives the drain
    std::vector<BuEtc1sSlice> descs;        // (likewise: source of an upload)
    BuDrain drain(ctx);
    if ((st = bu_reserve(ctx, &ctx->d_in, &ctx->in_cap, total_in ? total_in : 16))) return st;
    // a page-locked `out` (bu_host_alloc) receives the kernels' stores directly over PCIe: no device output buffer, no download
    void* zout = nullptr;
    const bool direct_out = bu_device_view(out, &zout);
    if (!direct_out && (st = bu_reserve(ctx, &ctx->d_out, &ctx->out_cap, p.out_bytes ? p.out_bytes : 16))) return st;
    const size_t ep_bytes = p.etc1s ? align_up(lz.endpoints.size() * 4) : 0, sel_bytes = p.etc1s ? align_up(lz.selectors.size()) : 0;
    // ETC1S: one descriptor per image (+ sentinel) behind the codebooks and the status words
    uint32_t n_units = 0;
    if (p.etc1s) {
        descs.reserve(n_img + 1);
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& sl = p.slices[p.first_slice[k]];
            const size_t nblk = (size_t)sl.num_blocks_x * sl.num_blocks_y;
            if (p.images[k].size == 0 || nblk == 0) continue;
            BuEtc1sSlice d;
            d.unit0 = n_units;
            d.n_blocks = (uint32_t)nblk;
            d.nbx = sl.num_blocks_x;
            d.idx_ofs = (uint32_t)(in_off[k] / 4);
            d.aidx_ofs = (p.alpha_pairs && target == BU_READ_RGBA) ? (uint32_t)(ain_off[k] / 4) : 0xFFFFFFFFu;
            d.image = (uint32_t)k;
            d.out_ofs = p.images[k].offset;
            descs.push_back(d);
            n_units += (uint32_t)((nblk + 63) / 64);
        }
        BuEtc1sSlice end = {};
        end.unit0 = n_units;
        descs.push_back(end);
    }
    const size_t desc_bytes = align_up(descs.size() * sizeof(BuEtc1sSlice)), status_bytes = align_up(8 * n_img);
    if ((st = bu_reserve(ctx, &ctx->d_aux, &ctx->aux_cap, ep_bytes + sel_bytes + status_bytes + desc_bytes + 256))) return st;
    uint8_t* d_in = static_cast<uint8_t*>(ctx->d_in);
    uint8_t* d_out = direct_out ? static_cast<uint8_t*>(zout) : static_cast<uint8_t*>(ctx->d_out);
    uint8_t* aux = static_cast<uint8_t*>(ctx->d_aux);
    uint64_t* d_status = reinterpret_cast<uint64_t*>(aux + ep_bytes + sel_bytes);
    BU_HIP(ctx, hipMemsetAsync(d_status, 0xFF, 8 * n_img, ctx->stream));
    if (p.etc1s) {
        BU_HIP(ctx, hipMemcpyAsync(d_in, idx_all.data(), total_in, hipMemcpyHostToDevice, ctx->stream));
        if (!lz.endpoints.empty()) BU_HIP(ctx, hipMemcpyAsync(aux, lz.endpoints.data(), lz.endpoints.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!lz.selectors.empty()) BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes, lz.selectors.data(), lz.selectors.size(), hipMemcpyHostToDevice, ctx->stream));
        BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes + sel_bytes + status_bytes, descs.data(), descs.size() * sizeof(BuEtc1sSlice), hipMemcpyHostToDevice, ctx->stream));
        // ONE launch for the whole file (basis.rs:42-58 / 103-123 walk the slices one by one)
        if (n_units) {
            const uint32_t n_cb0 = (uint32_t)lz.endpoints.size();
            const unsigned grid = bu_grid_for((size_t)n_units * 64, ctx->cu_count);
            const BuEtc1sSlice* d_descs = reinterpret_cast<const BuEtc1sSlice*>(aux + ep_bytes + sel_bytes + status_bytes);
            if (target == BU_READ_RGBA)
                hipLaunchKernelGGL(bu_etc1s_file_kernel<true>, dim3(grid), dim3(BU_WG), 0, ctx->stream, reinterpret_cast<const uint32_t*>(d_in), d_descs,
                                   (uint32_t)(descs.size() - 1), n_units, reinterpret_cast<const uint32_t*>(aux), n_cb0,
                                   reinterpret_cast<const uint2*>(aux + ep_bytes), n_cb0, d_out, reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
            else
                hipLaunchKernelGGL(bu_etc1s_file_kernel<false>, dim3(grid), dim3(BU_WG), 0, ctx->stream, reinterpret_cast<const uint32_t*>(d_in), d_descs,
                                   (uint32_t)(descs.size() - 1), n_units, reinterpret_cast<const uint32_t*>(aux), n_cb0,
                                   reinterpret_cast<const uint2*>(aux + ep_bytes), n_cb0, d_out, reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
            BU_HIP(ctx, hipGetLastError());
        }
    }
    bool used_extra = false;
    size_t run_piece_bytes = (size_t)16 << 20;
    if (const char* e = getenv("BU_RUN_PIECE_MIB")) run_piece_bytes = (size_t)atoll(e) << 20;  // 0 disables the pieced pipeline
    for (size_t k = 0; k < n_img; k++) {
        const bu_slice_desc& s = p.slices[p.first_slice[k]];
        const bu_image& im = p.images[k];
        if (im.size == 0) continue;
        if (p.etc1s) {
            // (launched once for the whole file above)
        } else {
            size_t run_end = k;  // last image of the run starting at k (only evaluated for run leaders)
            bool pieced = false;
            if (run_of[k] == k) {
                while (run_end + 1 < n_img && run_of[run_end + 1] == k) run_end++;
                const size_t run_bytes = in_off[run_end] + p.slices[p.first_slice[run_end]].file_size - in_off[k];
                // A large block-linear run with a mapped (page-locked) output: upload and transcode in pieces on two
                // streams, so that piece i's results cross PCIe upstream while piece i+1 comes down.
                const size_t piece_bytes = run_piece_bytes;
                if (target != BU_READ_RGBA && direct_out && piece_bytes && run_bytes >= 2 * piece_bytes) {
                    pieced = true;
                    if (!ctx->extra_streams[0]) BU_HIP(ctx, hipStreamCreateWithFlags(&ctx->extra_streams[0], hipStreamNonBlocking));
                    if (!used_extra) {
                        BU_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));  // the status words are reset on the context stream
                        BU_HIP(ctx, hipStreamWaitEvent(ctx->extra_streams[0], ctx->ev0, 0));
                        used_extra = true;
                    }
                    const bu_target pbt = target == BU_READ_ASTC ? BU_TARGET_ASTC : target == BU_READ_BC7 ? BU_TARGET_BC7
                                          : target == BU_READ_ETC1 ? BU_TARGET_ETC1 : BU_TARGET_ETC2;
                    const size_t obytes = bu_target_block_bytes(pbt);
                    size_t piece_no = 0;
                    for (size_t done = 0; done < run_bytes; done += piece_bytes, piece_no++) {
                        const size_t nbytes = run_bytes - done < piece_bytes ? run_bytes - done : piece_bytes;
                        hipStream_t ps = (piece_no & 1) ? ctx->extra_streams[0] : ctx->stream;
                        BU_HIP(ctx, hipMemcpyAsync(d_in + in_off[k] + done, file + s.file_ofs + done, nbytes, hipMemcpyHostToDevice, ps));
                        st = bu_launch_uastc(ctx, pbt, d_in + in_off[k] + done, nbytes / 16, d_out + im.offset + (done / 16) * obytes, 1, done / 16, d_status + k, ps,
                                             BU_ZEROCOPY_GRID);
                        if (st) return st;
                    }
                } else {
                    BU_HIP(ctx, hipMemcpyAsync(d_in + in_off[k], file + s.file_ofs, run_bytes, hipMemcpyHostToDevice, ctx->stream));
                }
            }
            const bu_target bt = target == BU_READ_RGBA ? BU_TARGET_RGBA32
                                 : target == BU_READ_ASTC ? BU_TARGET_ASTC
                                 : target == BU_READ_BC7  ? BU_TARGET_BC7
                                 : target == BU_READ_ETC1 ? BU_TARGET_ETC1
                                                          : BU_TARGET_ETC2;
            if (target == BU_READ_RGBA) {  // image geometry differs per slice: one launch each
                st = bu_launch_uastc(ctx, bt, d_in + in_off[k], s.file_size / 16, d_out + im.offset, s.num_blocks_x ? s.num_blocks_x : 1, 0, d_status + k,
                                     ctx->stream, direct_out ? BU_ZEROCOPY_GRID : 0);
            } else if (run_of[k] == k && !pieced) {  // block-linear: the run's outputs are contiguous from im.offset on
                const size_t run_bytes = in_off[run_end] + p.slices[p.first_slice[run_end]].file_size - in_off[k];
                st = bu_launch_uastc(ctx, bt, d_in + in_off[k], run_bytes / 16, d_out + im.offset, 1, 0, d_status + k, ctx->stream,
                                     direct_out ? BU_ZEROCOPY_GRID : 0);
            }
        }
        if (st) return st;
    }
    lap("reserve + enqueue");
    if (used_extra) {  // the status words are read on the context stream: it must see the second stream's kernels
        BU_HIP(ctx, hipEventRecord(ctx->ev1, ctx->extra_streams[0]));
        BU_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev1, 0));
    }
    BU_HIP(ctx, hipMemcpyAsync(words.data(), d_status, 8 * n_img, hipMemcpyDeviceToHost, ctx->stream));
    if (p.out_bytes && !direct_out) BU_HIP(ctx, hipMemcpyAsync(out, d_out, p.out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (used_extra) BU_HIP(ctx, hipStreamSynchronize(ctx->extra_streams[0]));
    BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain.armed = false;
    lap("download + synchronise");
    for (size_t k = 0; k < n_img; k++) {  // first Err (in slice order) aborts the whole call, like the `?` in the reference drivers
        st = bu_status_word_decode(words[k], nullptr);
        if (st) return st;
    }
    return BU_OK;
    };
    return settle(rest());
}


// C++ exceptions must not cross the C ABI (a ctypes or Rust caller would be terminated): vectors sized from untrusted
// file fields can throw std::bad_alloc, thread creation std::system_error.  A file that asks for more memory than
// exists is reported like any other out-of-bounds field.
#define BU_GUARDED(call)                 \
    try {                                \
        return call;                     \
    } catch (const std::bad_alloc&) {    \
        return BU_ERR_BOUNDS;            \
    } catch (...) {                      \
        return BU_ERR_HIP;               \
    }
bu_status bu_basis_read_slice_descs(const uint8_t* file, size_t len, const bu_basis_header* header, bu_slice_desc* out, size_t max_descs,
                                    size_t* n_descs)
{
    BU_GUARDED(bu_basis_read_slice_descs_impl(file, len, header, out, max_descs, n_descs))
}
bu_status bu_read_query(bu_read_target target, const uint8_t* file, size_t len, size_t* n_images, size_t* out_bytes)
{
    BU_GUARDED(bu_read_query_impl(target, file, len, n_images, out_bytes))
}
bu_status bu_basislz_decode(const uint8_t* file, size_t len, uint32_t slice_index, uint32_t* endpoints_out, uint8_t* selectors_out,
                            uint32_t* idx_out)
{
    BU_GUARDED(bu_basislz_decode_impl(file, len, slice_index, endpoints_out, selectors_out, idx_out))
}
bu_status bu_read_to(bu_context* ctx, bu_read_target target, const uint8_t* file, size_t len, bu_basis_header* header_out, bu_image* images,
                     size_t max_images, size_t* n_images, uint8_t* out, size_t out_bytes)
{
    BU_GUARDED(bu_read_to_impl(ctx, target, file, len, header_out, images, max_images, n_images, out, out_bytes))
}
#undef BU_GUARDED

bu_status bu_basis_write_uastc(const bu_slice_desc* descs, const uint8_t* const* slice_data, const size_t* slice_bytes, size_t n_slices,
                               uint16_t header_flags, uint8_t tex_type, uint8_t* out, size_t out_cap, size_t* out_len)
{
    if ((n_slices && (!descs || !slice_data || !slice_bytes)) || n_slices >= (1u << 24)) return BU_ERR_ARGUMENT;
    size_t total = 77 + 23 * n_slices;
    for (size_t i = 0; i < n_slices; i++) total += slice_bytes[i];
    if (out_len) *out_len = total;
    if (!out) return BU_OK;
    if (out_cap < total || total > 0xFFFFFFFFull) return BU_ERR_OUTPUT_SIZE;
    auto put = [&](size_t pos, uint32_t v, int n) { for (int k = 0; k < n; k++) out[pos + k] = (uint8_t)(v >> (8 * k)); };
    memset(out, 0, 77 + 23 * n_slices);
    size_t ofs = 77 + 23 * n_slices;
    uint32_t n_images = 0;
    for (size_t i = 0; i < n_slices; i++) {
        const size_t d = 77 + 23 * i;
        put(d, descs[i].image_index, 3);
        out[d + 3] = descs[i].level_index;
        out[d + 4] = descs[i].flags;
        put(d + 5, descs[i].orig_width, 2);
        put(d + 7, descs[i].orig_height, 2);
        put(d + 9, descs[i].num_blocks_x, 2);
        put(d + 11, descs[i].num_blocks_y, 2);
        put(d + 13, (uint32_t)ofs, 4);
        put(d + 17, (uint32_t)slice_bytes[i], 4);
        put(d + 21, bu_host::crc16(slice_data[i], slice_bytes[i], 0), 2);
        if (slice_bytes[i]) memcpy(out + ofs, slice_data[i], slice_bytes[i]);
        ofs += slice_bytes[i];
        if (descs[i].image_index + 1 > n_images) n_images = descs[i].image_index + 1;
    }
    put(0, 0x4273, 2);   // sig
    put(2, 0x13, 2);     // ver
    put(4, 77, 2);       // header_size
    put(8, (uint32_t)(total - 77), 4);
    put(12, bu_host::crc16(out + 77, total - 77, 0), 2);
    put(14, (uint32_t)n_slices, 3);
    put(17, n_images, 3);
    out[20] = 1;         // UASTC4x4
    put(21, header_flags, 2);
    out[23] = tex_type;
    put(65, 77, 4);      // slice_desc_file_ofs
    put(6, bu_host::crc16(out + 8, 77 - 8, 0), 2);
    return BU_OK;
}

// ---- measurement helpers ---------------------------------------------------------------------------
bu_status bu_copy_ceiling_device(bu_context* ctx, const void* d_in, size_t n_blocks, void* d_out, void* stream)
{
    if (!ctx || (n_blocks && (!d_in || !d_out))) return BU_ERR_ARGUMENT;
    if (n_blocks == 0) return BU_OK;
    hipLaunchKernelGGL(bu_copy_kernel, dim3(bu_grid_for(n_blocks, ctx->cu_count)), dim3(BU_WG), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4*>(d_in), static_cast<uint4*>(d_out), n_blocks);
    BU_HIP(ctx, hipGetLastError());
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
    // poll instead of a blocking wait: the caller's wall clock around this call (bench.py's `value`) should not carry the
    // tens of microseconds a sleeping host thread needs to be woken up -- they are as long as several steps
    for (;;) {
        const hipError_t q = hipEventQuery(ctx->ev1);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return bu_fail(ctx, q, "hipEventQuery");
    }
    (void)hipGetLastError();
    BU_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
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
    for (int i = 0; i < n_streams; i++)
        if (!ctx->extra_streams[i]) BU_HIP(ctx, hipStreamCreateWithFlags(&ctx->extra_streams[i], hipStreamNonBlocking));
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

#include "bu_multi.hpp"
