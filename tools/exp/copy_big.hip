// EXPERIMENT: how fast can ANY traversal copy 512 MiB -> 512 MiB (BASELINE config 5's bytes) on this box, and does the transcoder's traversal -- a persistent grid walking
// 1024-block tiles, strips or 64 x 16-block rectangles at a 1024-block pitch, next tile's loads in flight -- cost bandwidth by itself?
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/exp/copy_big tools/exp/copy_big.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <cstdlib>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ inline uint4 ldnt(const uint4* p) { v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
__device__ inline void stnt(uint4* p, uint4 v) { v4u t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(p)); }
template <int EPT, bool NT>
__global__ void oneshot(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    uint4 v[EPT];
    const size_t base = (size_t)blockIdx.x * blockDim.x * EPT + threadIdx.x;
#pragma unroll
    for (int k = 0; k < EPT; k++) v[k] = NT ? ldnt(in + base + (size_t)k * blockDim.x) : in[base + (size_t)k * blockDim.x];
#pragma unroll
    for (int k = 0; k < EPT; k++) { if (NT) stnt(out + base + (size_t)k * blockDim.x, v[k]); else out[base + (size_t)k * blockDim.x] = v[k]; }
}
// persistent: gridDim.x workgroups of WGS threads walk 1024-element tiles t = blockIdx.x, + gridDim.x, ...; RECT: tile = 64 x 16 elements of a grid 1024 wide
template <int RW>
__device__ inline size_t tile_idx(unsigned t, unsigned l)
{
    if (RW >= 1024) return (size_t)t * 1024u + l;
    constexpr unsigned TPR = 1024u / RW, ROWS = 1024u / RW;  // tiles per row of the 1024-wide grid, rows per tile
    const unsigned ty = t / TPR, tx = t % TPR;
    return (size_t)(ROWS * ty + l / RW) * 1024u + RW * tx + (l % RW);
}
// XCD-aware renumbering: workgroup / walk index t (XCD = t % 8) -> tile whose ROW-in-a-group-of-eight is t % 8 and whose column (of 16) is (t / 8) % 16: an XCD then streams
// whole 256 KiB tile rows instead of two 1 KiB-wide columns of every row
__device__ inline unsigned xcd_rows(unsigned t) { return (t & ~0x7Fu) | ((t & 7u) << 4) | ((t >> 3) & 0xFu); }
template <int WGS, int RW, bool XR>
__global__ void oneshot_tile_x(const uint4* __restrict__ in, uint4* __restrict__ out)
{
    constexpr int BPT = 1024 / WGS;
    const unsigned t = XR ? xcd_rows(blockIdx.x) : blockIdx.x;
    uint4 v[BPT];
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(t, j * WGS + threadIdx.x));
#pragma unroll
    for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(t, j * WGS + threadIdx.x), v[j]);
}
template <int WGS, int RW, bool XR>
__global__ void persist_x(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned n_tiles)
{
    constexpr int BPT = 1024 / WGS;
    uint4 v[BPT], vn[BPT];
    unsigned w = blockIdx.x;
    if (w >= n_tiles) return;
    auto T = [&](unsigned x) { return XR ? xcd_rows(x) : x; };
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(T(w), j * WGS + threadIdx.x));
    for (; w < n_tiles; w += gridDim.x) {
        const unsigned nw = w + gridDim.x;
        if (nw < n_tiles) {
#pragma unroll
            for (int j = 0; j < BPT; j++) vn[j] = ldnt(in + tile_idx<RW>(T(nw), j * WGS + threadIdx.x));
        }
#pragma unroll
        for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(T(w), j * WGS + threadIdx.x), v[j]);
#pragma unroll
        for (int j = 0; j < BPT; j++) v[j] = vn[j];
    }
}
// one pass, one tile per workgroup, tiles RW wide
template <int WGS, int RW>
__global__ void oneshot_tile(const uint4* __restrict__ in, uint4* __restrict__ out)
{
    constexpr int BPT = 1024 / WGS;
    uint4 v[BPT];
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(blockIdx.x, j * WGS + threadIdx.x));
#pragma unroll
    for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(blockIdx.x, j * WGS + threadIdx.x), v[j]);
}
// persistent, tiles drawn off ONE atomic counter (reset by the host between launches)
template <int WGS, int RW>
__global__ void persist_ticket(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned n_tiles, unsigned* counter)
{
    constexpr int BPT = 1024 / WGS;
    __shared__ unsigned s_t;
    uint4 v[BPT], vn[BPT];
    if (threadIdx.x == 0) s_t = atomicAdd(counter, 1u);
    __syncthreads();
    unsigned t = s_t;
    if (t >= n_tiles) return;
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(t, j * WGS + threadIdx.x));
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_t = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned nt = s_t;
        if (nt < n_tiles) {
#pragma unroll
            for (int j = 0; j < BPT; j++) vn[j] = ldnt(in + tile_idx<RW>(nt, j * WGS + threadIdx.x));
        }
#pragma unroll
        for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(t, j * WGS + threadIdx.x), v[j]);
        if (nt >= n_tiles) break;
#pragma unroll
        for (int j = 0; j < BPT; j++) v[j] = vn[j];
        t = nt;
    }
}
// persistent fixed walk with a start-up delay per workgroup: SPREAD units of 64 x 16 clocks spread pseudo-randomly (or by residency generation) over the grid
template <int WGS, int RW, int MODE>
__global__ void persist_stagger(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned n_tiles, unsigned spread)
{
    constexpr int BPT = 1024 / WGS;
    unsigned d = MODE == 0 ? ((blockIdx.x * 2654435761u) >> 16) % (spread + 1u) : (blockIdx.x / 256u) * spread;
    for (; d; d--) __builtin_amdgcn_s_sleep(16);
    uint4 v[BPT], vn[BPT];
    unsigned t = blockIdx.x;
    if (t >= n_tiles) return;
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(t, j * WGS + threadIdx.x));
    for (; t < n_tiles; t += gridDim.x) {
        const unsigned nt = t + gridDim.x;
        if (nt < n_tiles) {
#pragma unroll
            for (int j = 0; j < BPT; j++) vn[j] = ldnt(in + tile_idx<RW>(nt, j * WGS + threadIdx.x));
        }
#pragma unroll
        for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(t, j * WGS + threadIdx.x), v[j]);
#pragma unroll
        for (int j = 0; j < BPT; j++) v[j] = vn[j];
    }
}
// persistent, tiles drawn off EIGHT counters 128 B apart (counter c serves tiles 8 d + c), the draw for the tile after the next issued one tile ahead (the library's scheme)
template <int WGS, int RW>
__global__ void persist_ticket8(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned n_tiles, unsigned* counters)
{
    constexpr int BPT = 1024 / WGS;
    __shared__ unsigned s_t[2];
    unsigned* const my = counters + (blockIdx.x & 7u) * 32u;
    const unsigned c = blockIdx.x & 7u;
    uint4 v[BPT], vn[BPT];
    unsigned t = blockIdx.x;  // first tile: the workgroup's own number (the counters start at gridDim.x / 8)
    unsigned draw = 0, par = 0;
    if (threadIdx.x == 0) draw = atomicAdd(my, 1u);
    if (t >= n_tiles) return;
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(t, j * WGS + threadIdx.x));
    for (;; par ^= 1u) {
        if (threadIdx.x == 0) s_t[par] = (gridDim.x / 8u + draw) * 8u + c;
        __syncthreads();
        const unsigned nt = s_t[par];
        if (threadIdx.x == 0 && nt < n_tiles) draw = atomicAdd(my, 1u);
        if (nt < n_tiles) {
#pragma unroll
            for (int j = 0; j < BPT; j++) vn[j] = ldnt(in + tile_idx<RW>(nt, j * WGS + threadIdx.x));
        }
#pragma unroll
        for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(t, j * WGS + threadIdx.x), v[j]);
        if (nt >= n_tiles) break;
#pragma unroll
        for (int j = 0; j < BPT; j++) v[j] = vn[j];
        t = nt;
    }
}
// semi-persistent: gridDim.x = n_tiles / K workgroups, each walks K tiles (prefetch) and exits; CONTIG: tiles K w .. K w + K - 1, else w, w + G, w + 2 G, ...
template <int WGS, int RW, bool CONTIG>
__global__ void semi(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned n_tiles, unsigned K)
{
    constexpr int BPT = 1024 / WGS;
    uint4 v[BPT], vn[BPT];
    const unsigned step = CONTIG ? 1u : gridDim.x;
    unsigned t = CONTIG ? blockIdx.x * K : blockIdx.x;
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = ldnt(in + tile_idx<RW>(t, j * WGS + threadIdx.x));
    for (unsigned k = 0; k < K; k++, t += step) {
        const unsigned nt = t + step;
        if (k + 1 < K) {
#pragma unroll
            for (int j = 0; j < BPT; j++) vn[j] = ldnt(in + tile_idx<RW>(nt, j * WGS + threadIdx.x));
        }
#pragma unroll
        for (int j = 0; j < BPT; j++) stnt(out + tile_idx<RW>(t, j * WGS + threadIdx.x), v[j]);
#pragma unroll
        for (int j = 0; j < BPT; j++) v[j] = vn[j];
    }
}
template <int WGS, bool RECT, bool PF, bool NT>
__global__ void persist(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned n_tiles)
{
    constexpr int BPT = 1024 / WGS;
    auto idx = [&](unsigned t, unsigned l) -> size_t {
        if (RECT) { const unsigned ty = t >> 4, tx = t & 15u; return (size_t)(16u * ty + l / 64u) * 1024u + 64u * tx + (l % 64u); }
        return (size_t)t * 1024u + l;
    };
    uint4 v[BPT], vn[BPT];
    unsigned t = blockIdx.x;
    if (t >= n_tiles) return;
#pragma unroll
    for (int j = 0; j < BPT; j++) v[j] = NT ? ldnt(in + idx(t, j * WGS + threadIdx.x)) : in[idx(t, j * WGS + threadIdx.x)];
    for (; t < n_tiles; t += gridDim.x) {
        const unsigned nt = t + gridDim.x;
        if (PF && nt < n_tiles) {
#pragma unroll
            for (int j = 0; j < BPT; j++) vn[j] = NT ? ldnt(in + idx(nt, j * WGS + threadIdx.x)) : in[idx(nt, j * WGS + threadIdx.x)];
        }
#pragma unroll
        for (int j = 0; j < BPT; j++) { if (NT) stnt(out + idx(t, j * WGS + threadIdx.x), v[j]); else out[idx(t, j * WGS + threadIdx.x)] = v[j]; }
        if (PF) {
#pragma unroll
            for (int j = 0; j < BPT; j++) v[j] = vn[j];
        } else if (nt < n_tiles) {
#pragma unroll
            for (int j = 0; j < BPT; j++) v[j] = NT ? ldnt(in + idx(nt, j * WGS + threadIdx.x)) : in[idx(nt, j * WGS + threadIdx.x)];
        }
    }
}
// inputs as the benches have them: incompressible pseudo-random bytes (COPY_BIG_FILL=const keeps the memset pattern: data-dependent power is part of the answer)
__global__ void fill_random(uint4* p, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; unsigned long long y = x * 0x94D049BB133111EBull; y ^= y >> 31;
        p[i] = make_uint4((unsigned)x, (unsigned)(x >> 32), (unsigned)y, (unsigned)(y >> 32));
    }
}
static std::vector<uint4*> g_in, g_out;
static size_t N;
static int g_reps = 8;
template <class F> void run(const char* name, F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t0 = std::chrono::steady_clock::now();
    int k = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.15) { launch(g_in[k % g_in.size()], g_out[k % g_in.size()]); k++; hipDeviceSynchronize(); }
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < g_reps; i++) launch(g_in[i % g_in.size()], g_out[i % g_in.size()]);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= g_reps; sum += ms; if (ms < best) best = ms;
    }
    printf("%-44s %8.2f us (avg %8.2f)  %6.3f TB/s  %.3f of 8 TB/s\n", name, best * 1e3, sum / 3 * 1e3, 32.0 * N / best / 1e9, 32.0 * N / best / 1e9 / 8);
    if (hipGetLastError() != hipSuccess) printf("  ERROR\n");
}
int main()
{
    N = (size_t)1 << (getenv("COPY_BIG_LG") ? atoi(getenv("COPY_BIG_LG")) : 25);  // uint4 elements = UASTC blocks
    const int pairs = getenv("COPY_BIG_PAIRS") ? atoi(getenv("COPY_BIG_PAIRS")) : 2;
    for (int i = 0; i < pairs; i++) { uint4 *a, *b; hipMalloc(&a, N * 16); hipMalloc(&b, N * 16); hipMemset(a, i + 1, N * 16);
        const char* f = getenv("COPY_BIG_FILL");
        if (!f || f[0] != 'c') hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, a, N, 77u + i);
        g_in.push_back(a); g_out.push_back(b); }
    hipDeviceSynchronize();
    printf("input: %s\n", (getenv("COPY_BIG_FILL") && getenv("COPY_BIG_FILL")[0] == 'c') ? "constant bytes" : "pseudo-random bytes");
    const unsigned tiles = (unsigned)(N / 1024);
    g_reps = (int)(((size_t)1 << 28) / N); if (g_reps < 8) g_reps = 8;
    run("oneshot 256 x 4 plain", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot<4, false>), dim3(N / 1024), dim3(256), 0, 0, a, b, N); });
    run("oneshot 256 x 4 nt", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot<4, true>), dim3(N / 1024), dim3(256), 0, 0, a, b, N); });
    run("oneshot 512 x 2 nt", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot<2, true>), dim3(N / 1024), dim3(512), 0, 0, a, b, N); });
    run("oneshot 1024 x 1 nt", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot<1, true>), dim3(N / 1024), dim3(1024), 0, 0, a, b, N); });
    run("oneshot 512 x 4 nt (the library's shape)", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot<4, true>), dim3(N / 2048), dim3(512), 0, 0, a, b, N); });
    run("oneshot 256 x 8 nt", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot<8, true>), dim3(N / 2048), dim3(256), 0, 0, a, b, N); });
    for (int per_cu : {2, 4, 6, 8}) {
        char nm[96];
        snprintf(nm, sizeof nm, "persist 512x2 strips pf nt, %d per CU", per_cu);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<512, false, true, true>), dim3(256 * per_cu), dim3(512), 0, 0, a, b, tiles); });
        snprintf(nm, sizeof nm, "persist 512x2 rect   pf nt, %d per CU", per_cu);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<512, true, true, true>), dim3(256 * per_cu), dim3(512), 0, 0, a, b, tiles); });
    }
    run("persist 512x2 rect nopf nt, 4 per CU", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<512, true, false, true>), dim3(1024), dim3(512), 0, 0, a, b, tiles); });
    run("persist 512x2 rect pf plain, 4 per CU", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<512, true, true, false>), dim3(1024), dim3(512), 0, 0, a, b, tiles); });
    run("persist 256x4 rect pf nt, 5 per CU", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<256, true, true, true>), dim3(1280), dim3(256), 0, 0, a, b, tiles); });
    run("persist 256x4 rect pf nt, 8 per CU", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<256, true, true, true>), dim3(2048), dim3(256), 0, 0, a, b, tiles); });
    run("oneshot tile 512x2, 64-wide rectangles", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile<512, 64>), dim3(tiles), dim3(512), 0, 0, a, b); });
    run("oneshot tile 512x2, 128-wide rectangles", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile<512, 128>), dim3(tiles), dim3(512), 0, 0, a, b); });
    run("oneshot tile 512x2, 256-wide rectangles", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile<512, 256>), dim3(tiles), dim3(512), 0, 0, a, b); });
    run("oneshot tile 512x2, strips", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile<512, 1024>), dim3(tiles), dim3(512), 0, 0, a, b); });
    run("oneshot tile 256x4, 64-wide rectangles", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile<256, 64>), dim3(tiles), dim3(256), 0, 0, a, b); });
    for (unsigned spread : {0u, 2u, 8u, 32u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "persist 512x2 64-wide 4/CU, random delay <= %u x 0.43 us", spread);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_stagger<512, 64, 0>), dim3(1024), dim3(512), 0, 0, a, b, tiles, spread); });
        snprintf(nm, sizeof nm, "persist 512x2 strips 4/CU, random delay <= %u x 0.43 us", spread);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_stagger<512, 1024, 0>), dim3(1024), dim3(512), 0, 0, a, b, tiles, spread); });
    }
    for (unsigned spread : {1u, 2u, 4u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "persist 512x2 64-wide 4/CU, generation delay %u x 0.43 us", spread);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_stagger<512, 64, 1>), dim3(1024), dim3(512), 0, 0, a, b, tiles, spread); });
    }
    for (unsigned K : {1u, 2u, 4u, 8u, 16u, 32u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "semi 512x2 64-wide, %u tiles per WG, strided", K);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((semi<512, 64, false>), dim3(tiles / K), dim3(512), 0, 0, a, b, tiles, K); });
        snprintf(nm, sizeof nm, "semi 512x2 64-wide, %u tiles per WG, contiguous", K);
        run(nm, [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((semi<512, 64, true>), dim3(tiles / K), dim3(512), 0, 0, a, b, tiles, K); });
    }
    run("oneshot tile 512x2 64-wide, as numbered", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile_x<512, 64, false>), dim3(tiles), dim3(512), 0, 0, a, b); });
    run("oneshot tile 512x2 64-wide, XCD = tile row", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((oneshot_tile_x<512, 64, true>), dim3(tiles), dim3(512), 0, 0, a, b); });
    run("persist 512x2 64-wide 4/CU, as numbered", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_x<512, 64, false>), dim3(1024), dim3(512), 0, 0, a, b, tiles); });
    run("persist 512x2 64-wide 4/CU, XCD = tile row", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_x<512, 64, true>), dim3(1024), dim3(512), 0, 0, a, b, tiles); });
    run("persist 512x2 strips 4/CU, as numbered", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_x<512, 1024, false>), dim3(1024), dim3(512), 0, 0, a, b, tiles); });
    run("persist 512x2 strips 4/CU, XCD = 8-tile group", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist_x<512, 1024, true>), dim3(1024), dim3(512), 0, 0, a, b, tiles); });
    unsigned* counters8; hipMalloc(&counters8, 8 * 128);
    auto tk8 = [&](auto kern, int wgs, int per_cu) { return [=](const uint4* a, uint4* b) { hipMemsetAsync(counters8, 0, 8 * 128, 0); hipLaunchKernelGGL(kern, dim3(256 * per_cu), dim3(wgs), 0, 0, a, b, tiles, counters8); }; };
    run("persist 8 tickets 512x2 64-wide, 4 per CU", tk8(persist_ticket8<512, 64>, 512, 4));
    run("persist 8 tickets 512x2 strips, 4 per CU", tk8(persist_ticket8<512, 1024>, 512, 4));
    run("persist 8 tickets 512x2 64-wide, 2 per CU", tk8(persist_ticket8<512, 64>, 512, 2));
    run("persist 8 tickets 256x4 64-wide, 5 per CU", tk8(persist_ticket8<256, 64>, 256, 5));
    run("persist 8 tickets 512x2 256-wide, 4 per CU", tk8(persist_ticket8<512, 256>, 512, 4));
    unsigned* counter; hipMalloc(&counter, 4);
    auto tk = [&](auto kern, int wgs, int per_cu) { return [=](const uint4* a, uint4* b) { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL(kern, dim3(256 * per_cu), dim3(wgs), 0, 0, a, b, tiles, counter); }; };
    run("persist ticket 512x2 64-wide, 4 per CU", tk(persist_ticket<512, 64>, 512, 4));
    run("persist ticket 512x2 strips, 4 per CU", tk(persist_ticket<512, 1024>, 512, 4));
    run("persist ticket 512x2 strips, 2 per CU", tk(persist_ticket<512, 1024>, 512, 2));
    run("persist ticket 256x4 64-wide, 5 per CU", tk(persist_ticket<256, 64>, 256, 5));
    run("persist ticket 256x4 strips, 5 per CU", tk(persist_ticket<256, 1024>, 256, 5));
    run("persist ticket 512x2 256-wide, 4 per CU", tk(persist_ticket<512, 256>, 512, 4));
    run("persist 1024x1 rect pf nt, 2 per CU", [&](const uint4* a, uint4* b) { hipLaunchKernelGGL((persist<1024, true, true, true>), dim3(512), dim3(1024), 0, 0, a, b, tiles); });
    return 0;
}
