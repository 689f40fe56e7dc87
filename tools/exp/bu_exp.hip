// EXPERIMENTS ONLY (never shipped): time-attribution variants of the mode-sorted kernel.
#include <hip/hip_runtime.h>
// in-kernel stamps (diagnostic build only): s_memrealtime (100 MHz) is too coarse; s_memtime = shader clock
#define BU_STAMP_ARG , unsigned long long* __restrict__ stamps
#define BU_STAMP_PASS , (unsigned long long*)nullptr
#define BU_STAMP_FWD , stamps
#define BU_STAMP(k)                                                                                   \
    if (stamps && (threadIdx.x & 63u) == 0) {                                                         \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + ((k) >= 2 ? 16 * bu_stamp_it_ : 0) + (k)] = t_;        \
        if ((k) == 0 || (k) == 8) {                                                                   \
            unsigned long long r_;                                                                    \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_)::"memory");            \
            stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + ((k) >= 2 ? 16 * bu_stamp_it_ : 0) + 9 + (k) / 8] = r_; \
        }                                                                                             \
    }
#define BU_CHUNK_DECL unsigned chunk_no_ = 0;
#define BU_STAMP_DECL unsigned bu_stamp_it_ = 0;
#define BU_STAMP_NEXT bu_stamp_it_ = 1;
// per-chunk stamps: slot k of chunk record n of this wave: [time, value]
#define BU_CHUNK_STAMP(k, val)                                                                                     \
    if (stamps && (threadIdx.x & 63u) == 0) {                                                                      \
        unsigned long long t_;                                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                  \
        unsigned long long* rec_ = stamps + ((size_t)gridDim.x * (blockDim.x >> 6)) * 16 +                          \
                                   (((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12 + (chunk_no_ % 12)) * 8; \
        rec_[2 * (k)] = t_;                                                                                         \
        rec_[2 * (k) + 1] = (unsigned long long)(val);                                                              \
        if ((k) == 2) chunk_no_++;                                                                                  \
    }
#include "../../basisu_rs_amd/csrc/bu_hip.hip"

namespace {
constexpr int BU_BPT = 4, BU_TILE = 1024, BU_MAX_CHUNKS = BU_TILE / 64 + 20;
// V=1 no transcode (identity through the sort); V=2 additionally no atomics/sort; V=3 load->LDS->store
template <int V>
__global__ __launch_bounds__(BU_WG) void bu_exp_kernel(const uint4* __restrict__ in, void* __restrict__ out, size_t n_blocks,
                                                       const BuTablesAll* __restrict__ tables)
{
    __shared__ uint4 t_all[sizeof(BuTablesAll) / 16];
    BuTables& T = reinterpret_cast<BuTablesAll*>(t_all)->t;
    __shared__ uint4 sblk[BU_TILE];
    __shared__ uint8_t sst[BU_TILE];
    __shared__ uint32_t cnt[32], start[32], chunk[BU_MAX_CHUNKS + 4], n_chunks;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const size_t n_tiles = (n_blocks + BU_TILE - 1) / BU_TILE;
    size_t tile = blockIdx.x;
    uint4 v[BU_BPT];
#pragma unroll
    for (int j = 0; j < BU_BPT; j++) {
        const size_t idx = tile * BU_TILE + (size_t)j * BU_WG + tid;
        v[j] = (tile < n_tiles && idx < n_blocks) ? in[idx] : make_uint4(0, 0, 0, 0);
    }
    if (V != 3) bu_stage_tables_n<BU_WG, BU_TGT_BC7>(t_all, tables);
    if (tid < 32) cnt[tid] = 0;
    __syncthreads();
    for (; tile < n_tiles; tile += gridDim.x) {
        const size_t tbase = tile * BU_TILE;
        uint32_t mode[BU_BPT], pos[BU_BPT], dest[BU_BPT];
        if constexpr (V == 3) {
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) sblk[j * BU_WG + tid] = v[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < BU_BPT; j++) reinterpret_cast<uint4*>(out)[tbase + (size_t)j * BU_WG + tid] = sblk[j * BU_WG + tid];
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int j = 0; j < BU_BPT; j++) {
            mode[j] = T.mode_lut[v[j].x & 127u];
            if constexpr (V == 1) pos[j] = atomicAdd(&cnt[mode[j]], 1u);
            else pos[j] = 0;
        }
        __syncthreads();
        if (wave == 0 && V == 1) {
            const uint32_t c = lane < 20 ? cnt[lane] : 0u;
            uint32_t incl = c, nch = (c + 63u) >> 6, cincl = nch;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const uint32_t a = __shfl_up(incl, d), b2 = __shfl_up(cincl, d);
                if (lane >= (unsigned)d) { incl += a; cincl += b2; }
            }
            const uint32_t st = incl - c, cst = cincl - nch;
            if (lane < 20) { start[lane] = st; cnt[lane] = 0; }
            for (uint32_t k = 0; k < nch; k++) {
                const uint32_t left = c - 64u * k;
                chunk[cst + k] = lane | ((st + 64u * k) << 8) | ((left < 64u ? left : 64u) << 24);
            }
            if (lane == 19) n_chunks = cincl;
        }
        if (V == 2 && tid < 16) { chunk[tid] = 0 | ((64u * tid) << 8) | (64u << 24); n_chunks = 16; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BU_BPT; j++) {
            dest[j] = V == 1 ? start[mode[j]] + pos[j] : j * BU_WG + tid;
            sblk[dest[j]] = v[j];
        }
        __syncthreads();
        const uint32_t nc = n_chunks;
        for (uint32_t c = wave; c < nc; c += BU_WG / 64) {
            const uint32_t desc = __builtin_amdgcn_readfirstlane(chunk[c]);
            const uint32_t s0 = (desc >> 8) & 0xFFFFu, count = desc >> 24;
            const bool active = lane < count;
            const uint32_t slot = s0 + (active ? lane : 0u);
            uint4 bv = sblk[slot];
            bv.x ^= 1u;
            if (active) { sblk[slot] = bv; sst[slot] = 0; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BU_BPT; j++) {
            const size_t idx = tbase + (size_t)j * BU_WG + tid;
            uint4 r = sblk[dest[j]];
            r.y += sst[dest[j]];
            reinterpret_cast<uint4*>(out)[idx] = r;
        }
        __syncthreads();
    }
}
}  // namespace

static unsigned long long* g_stamps = nullptr;
extern "C" void bu_exp_set_stamps(unsigned long long* p) { g_stamps = p; }
extern "C" bu_status bu_exp_time(bu_context* ctx, int variant, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                 size_t n_blocks, int launches, void* stream, float* out_ms)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    BU_HIP(ctx, hipEventRecord(ctx->ev0, s));
    const size_t tiles = (n_blocks + BU_TILE - 1) / BU_TILE;
    for (int i = 0; i < launches; i++) {
        const size_t k = (size_t)i % n_buffers;
        const uint4* in = static_cast<const uint4*>(d_in[k]);
        switch (variant) {
        case 1: hipLaunchKernelGGL(bu_exp_kernel<1>, dim3((unsigned)tiles), dim3(BU_WG), 0, s, in, d_out[k], n_blocks, ctx->d_tables); break;
        case 2: hipLaunchKernelGGL(bu_exp_kernel<2>, dim3((unsigned)tiles), dim3(BU_WG), 0, s, in, d_out[k], n_blocks, ctx->d_tables); break;
        case 3: hipLaunchKernelGGL(bu_exp_kernel<3>, dim3((unsigned)tiles), dim3(BU_WG), 0, s, in, d_out[k], n_blocks, ctx->d_tables); break;
        case 100:  // plain one-lane-per-block kernel
            hipLaunchKernelGGL(bu_uastc_kernel<BU_TGT_BC7>, dim3(bu_grid_for(n_blocks, ctx->cu_count)), dim3(BU_WG), 0, s, in, d_out[k], n_blocks, 1u, 0ull,
                               (unsigned long long*)nullptr, ctx->d_tables);
            break;
#define SV(code, W, B, DIV, MINW, PF, DIR)                                                                                   \
    case code: {                                                                                                            \
        const size_t t_ = (n_blocks + (size_t)(W) * (B)-1) / ((size_t)(W) * (B));                                           \
        hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, W, B, MINW, PF, DIR>), dim3((unsigned)((t_ + (DIV)-1) / (DIV))), dim3(W), 0, s, in, d_out[k], \
                           (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);                   \
    } break;
            SV(0, 256, 4, 1, 1, true, false) SV(10, 256, 4, 1, 1, false, true) SV(11, 512, 4, 1, 1, false, false) SV(12, 512, 4, 1, 1, false, true)
            SV(13, 1024, 4, 1, 1, false, false) SV(14, 1024, 4, 1, 1, false, true) SV(15, 256, 8, 1, 1, false, true) SV(16, 512, 2, 1, 1, false, true)
        case 24: {
            const size_t t_ = (n_blocks + 1023) / 1024;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 512, 2, 1, false, false, 0>), dim3((unsigned)t_), dim3(512), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 25: {  // the shipped BC7 shape: 512 x 2, prefetch build, one tile per workgroup at 2^20 blocks
            const size_t t_ = (n_blocks + 1023) / 1024;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 512, 2, 1, true, false, 0>), dim3((unsigned)t_), dim3(512), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 30: {  // 1024 x 1, two persistent workgroups per CU, prefetch: two tiles per workgroup
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 1024, 1, 1, true, false, 0>), dim3((unsigned)(2 * ctx->cu_count)), dim3(1024), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 31: {  // 512 x 2, two persistent workgroups per CU, prefetch
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 512, 2, 1, true, false, 0>), dim3((unsigned)(2 * ctx->cu_count)), dim3(512), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 23: {
            const size_t t_ = (n_blocks + 2047) / 2048;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 1024, 2, 8, false, false, 0>), dim3((unsigned)t_), dim3(1024), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 22: {
            const size_t t_ = (n_blocks + 2047) / 2048;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 1024, 2, 1, false, false, 0>), dim3((unsigned)t_), dim3(1024), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 21: {
            const size_t t_ = (n_blocks + 2047) / 2048;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, 512, 4, 1, false, false, 40>), dim3((unsigned)t_), dim3(512), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
        case 40: {  // the shipped ETC1 shape
            const size_t t_ = (n_blocks + 4095) / 4096;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_ETC1, 1024, 4, 1, false, false, 0>), dim3((unsigned)t_), dim3(1024), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, 4096u, g_stamps);
        } break;
        case 41: {  // the shipped ETC2 shape
            const size_t t_ = (n_blocks + 2047) / 2048;
            hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_ETC2, 512, 4, 1, false, false, 0>), dim3((unsigned)t_), dim3(512), 0, s, in, d_out[k],
                               (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, (unsigned)0, g_stamps);
        } break;
#undef SV
        // round 4: rectangular 64-wide tiles (the headline's layout, bpr 1024), NT tiles per workgroup with every load up front
#define RV(code, W, B, MINW, PF, NT_, GRID)                                                                                                   \
    case code:                                                                                                                                \
        hipLaunchKernelGGL((bu_uastc_sorted_kernel<BU_TGT_BC7, W, B, MINW, PF, false, 0, BU_LAYOUT_RECT, NT_>), dim3(GRID), dim3(W), 0, s, in, d_out[k], \
                           (unsigned)n_blocks, 1024u, 0ull, (unsigned long long*)nullptr, ctx->d_tables, (unsigned)ctx->cu_count, 0x10000000u, g_stamps); \
        break;
            RV(60, 512, 2, 1, true, 1, 1024) RV(61, 1024, 1, 8, false, 2, 512) RV(62, 1024, 2, 8, true, 1, 512) RV(63, 1024, 4, 1, true, 1, 256)
            RV(64, 512, 4, 1, true, 1, 512) RV(65, 1024, 2, 1, false, 2, 256) RV(66, 512, 2, 1, false, 2, 512) RV(67, 512, 1, 1, false, 2, 1024)
            RV(68, 1024, 1, 1, false, 4, 256)
#undef RV
        default: return BU_ERR_ARGUMENT;
        }
    }
    BU_HIP(ctx, hipEventRecord(ctx->ev1, s));
    BU_HIP(ctx, hipEventSynchronize(ctx->ev1));
    BU_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return BU_OK;
}

// ---- ETC1S: codebooks staged in LDS against the shipped L2 gather (the north star's "LDS-staged codebook tables") ----------------
// One persistent 1024-thread workgroup per CU copies the endpoint codebook (4 B per entry) and the half of the selector codebook
// the target reads (4 B per entry: texel rows for RGBA32, ETC1 selector bytes for ETC1) into dynamic LDS, then walks the slice.
template <bool RGBA>
__global__ __launch_bounds__(1024) void bu_exp_etc1s_staged_kernel(const uint32_t* __restrict__ idx, unsigned nbx, size_t n_blocks,
                                                                   const uint32_t* __restrict__ endpoints, uint32_t n_ep,
                                                                   const uint2* __restrict__ selectors, uint32_t n_sel, uint8_t* __restrict__ out,
                                                                   const BuTablesAll* __restrict__ tables)
{
    extern __shared__ uint32_t lds[];
    uint32_t* s_ep = lds;
    uint32_t* s_sel = lds + n_ep;
    uint32_t* pal_lut = s_sel + n_sel;
    for (uint32_t i = threadIdx.x; i < n_ep; i += 1024) s_ep[i] = endpoints[i];
    for (uint32_t i = threadIdx.x; i < n_sel; i += 1024) s_sel[i] = RGBA ? selectors[i].x : selectors[i].y;
    if (RGBA && threadIdx.x < 256) pal_lut[threadIdx.x] = tables->t.etc1s_pal[threadIdx.x];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n_blocks; i += stride) {
        const uint32_t ix = __builtin_nontemporal_load(idx + i);
        const uint32_t e = ix & 0xFFFFu, sl = ix >> 16;
        if (e >= n_ep || sl >= n_sel) continue;
        const uint32_t ep = s_ep[e], sv = s_sel[sl];
        if constexpr (!RGBA) {
            const uint32_t inten = ep >> 24;
            bu_st_stream(reinterpret_cast<uint2*>(out) + i, make_uint2(((ep << 3) & 0x00F8F8F8u) | ((((inten << 5) | (inten << 2) | 3u) & 0xFFu) << 24), sv));
        } else {
            uint32_t px[16];
            bu_etc1s_block_rgba(pal_lut, ep, sv, false, 0u, 0u, px);
            const size_t by = i / nbx, bx = i - by * nbx;
            uint4* img = reinterpret_cast<uint4*>(out);
#pragma unroll
            for (int r = 0; r < 4; r++) bu_st_stream(img + (4 * by + r) * (size_t)nbx + bx, make_uint4(px[4 * r], px[4 * r + 1], px[4 * r + 2], px[4 * r + 3]));
        }
    }
}

// variant 0: the shipped kernels (bu_etc1s_*_device); 1: staged, one workgroup per CU; 2: staged, two workgroups per CU
extern "C" bu_status bu_exp_etc1s(bu_context* ctx, int variant, int rgba, const uint32_t* d_idx, unsigned nbx, size_t n_blocks, const uint32_t* d_ep,
                                  uint32_t n_ep, const void* d_sel, uint32_t n_sel, void* d_out, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (variant == 0) {
        if (rgba) return bu_etc1s_decode_rgba_device(ctx, d_idx, nullptr, nbx, n_blocks / nbx, d_ep, n_ep, d_sel, n_sel, d_out, nullptr, stream);
        return bu_etc1s_transcode_etc1_device(ctx, d_idx, n_blocks, d_ep, n_ep, d_sel, n_sel, d_out, nullptr, stream);
    }
    const size_t lds_bytes = ((size_t)n_ep + n_sel + 256) * 4;
    const unsigned grid = (unsigned)ctx->cu_count * (variant == 2 ? 2u : 1u);
    if (rgba)
        hipLaunchKernelGGL(bu_exp_etc1s_staged_kernel<true>, dim3(grid), dim3(1024), lds_bytes, s, d_idx, nbx, n_blocks, d_ep, n_ep,
                           static_cast<const uint2*>(d_sel), n_sel, static_cast<uint8_t*>(d_out), ctx->d_tables);
    else
        hipLaunchKernelGGL(bu_exp_etc1s_staged_kernel<false>, dim3(grid), dim3(1024), lds_bytes, s, d_idx, nbx, n_blocks, d_ep, n_ep,
                           static_cast<const uint2*>(d_sel), n_sel, static_cast<uint8_t*>(d_out), ctx->d_tables);
    BU_HIP(ctx, hipGetLastError());
    return BU_OK;
}
