// EXPERIMENT (round 4): how long does the dispatcher take to START the waves of a grid that fills every CU to 32 waves, and what
// does it depend on?  Each wave records s_memrealtime (100 MHz) on entry, optionally issues the transcoder's loads, then sleeps a
// few microseconds so that all workgroups of the launch are resident together.  Prints, per residency generation (blockIdx / 256),
// the median and maximum start time after the first wave of the launch.
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/ramp tools/exp/ramp.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int VREG>
__device__ __forceinline__ void touch()
{
    if constexpr (VREG == 32) asm volatile("v_mov_b32 v31, 0" ::: "v31");
    if constexpr (VREG == 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if constexpr (VREG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (VREG == 40) asm volatile("v_mov_b32 v39, 0" ::: "v39");
}

// 256 VOP3 instructions (8 bytes each) = 2 KiB of straight-line code
#define PAD16 "v_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\t" \
              "v_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\t" \
              "v_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\t" \
              "v_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\tv_add3_u32 %0, %0, %0, 1\n\t"
#define PAD256 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16 PAD16
template <int WGS, int VREG, int CODE_KIB = 0>
__global__ __launch_bounds__(WGS) void ramp(unsigned long long* t, const uint4* __restrict__ in, uint4* __restrict__ out, int loads, int sleep_loops,
                                            int lds_touch)
{
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    touch<VREG>();
    extern __shared__ uint4 lds[];
    const unsigned gw = blockIdx.x * (WGS / 64) + (threadIdx.x >> 6);
    uint4 v[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    const int flags = loads >> 4;
    loads &= 15;
    if (flags & 8) {
        if (blockIdx.x >= 768) __builtin_amdgcn_s_setprio(1);
        else if (blockIdx.x >= 512) __builtin_amdgcn_s_setprio(2);
        else if (blockIdx.x >= 256) __builtin_amdgcn_s_setprio(3);
    }
    uint4 tv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (flags & 2) {  // the transcoder's table staging: every workgroup reads the same 12 KiB
        const uint4* tab = in + ((size_t)48 << 16);
        tv[0] = tab[threadIdx.x];
        if (threadIdx.x < 256) tv[1] = tab[WGS + threadIdx.x];
    }
    for (int k = 0; k < loads; k++) v[k & 1] = in[(size_t)blockIdx.x * (WGS * 2) + k * WGS + threadIdx.x];
    unsigned long long t_issue;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_issue)::"memory");  // every load of this wave has been ISSUED
    if (flags & 2) {
        lds[threadIdx.x] = tv[0];
        if (threadIdx.x < 256) lds[WGS + threadIdx.x] = tv[1];
    }
    if (flags & 4) __syncthreads();
    if (flags & 16) {
        if (lds_touch) lds[threadIdx.x] = v[0];  // waits for the tile data
        if (flags & 4) __syncthreads();
    }
    if (lds_touch) lds[threadIdx.x] = v[0];
    if constexpr (CODE_KIB > 0) {
        unsigned x = threadIdx.x;
#pragma unroll
        for (int k = 0; k < CODE_KIB / 2; k++) asm volatile(PAD256 : "+v"(x));
        v[0].y += x;
    }
    for (int i = 0; i < sleep_loops; i++) __builtin_amdgcn_s_sleep(32);
    unsigned long long t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (loads) out[(size_t)blockIdx.x * (WGS * 2) + threadIdx.x] = make_uint4(v[0].x + v[1].x, v[0].y, v[0].z, v[0].w);
    if ((threadIdx.x & 63u) == 0) {
        t[2 * gw] = t0;
        t[2 * gw + 1] = t1;
        t[65536 + gw] = t_issue;
    }
}

template <int WGS, int VREG, int CODE_KIB = 0>
void run(const char* name, int grid, size_t lds_bytes, int loads, int sleep_loops, unsigned long long* d_t, uint4* in, uint4* out)
{
    const int waves = grid * (WGS / 64);
    std::vector<unsigned long long> h(65536 + (size_t)waves);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ramp<WGS, VREG, CODE_KIB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((ramp<WGS, VREG, CODE_KIB>), dim3(grid), dim3(WGS), lds_bytes, 0, d_t, in, out, loads, sleep_loops, lds_bytes >= (size_t)WGS * 16);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (int w = 0; w < waves; w++) first = std::min(first, h[2 * w]);
    printf("%-44s", name);
    const int wpg = 256 * (WGS / 64);  // waves per residency generation
    for (int g = 0; g * wpg < waves; g++) {
        std::vector<double> s;
        for (int w = g * wpg; w < std::min(waves, (g + 1) * wpg); w++) s.push_back((double)(h[2 * w] - first) / 100.0);
        std::sort(s.begin(), s.end());
        std::vector<double> q;
        for (int w = g * wpg; w < std::min(waves, (g + 1) * wpg); w++) q.push_back((double)(h[65536 + w] - first) / 100.0);
        std::sort(q.begin(), q.end());
        printf(" | gen %d start %.2f issued p50 %.2f p90 %.2f", g, s[s.size() / 2], q[q.size() / 2], q[q.size() * 9 / 10]);
    }
    double life = 0;
    for (int w = 0; w < waves; w++) life += (double)(h[2 * w + 1] - h[2 * w]) / 100.0;
    printf(" | mean life %.2f us\n", life / waves);
}

int main()
{
    unsigned long long* d_t;
    uint4 *in, *out;
    hipMalloc(&d_t, 8 * (65536 + 32768));
    hipMalloc(&in, (size_t)64 << 20);
    hipMalloc(&out, (size_t)64 << 20);
    hipMemset(in, 1, (size_t)64 << 20);
    const int S = 60;  // ~60 x 64 x 32 clocks... s_sleep 32 = 2048 clocks ~ 0.9 us each -> far too long; use few
    (void)S;
    // loads argument: low 4 bits = tile loads per lane; flags << 4: 2 = table staging (12 KiB common -> LDS), 4 = barrier,
    // 8 = s_setprio by generation, 16 = wait for the tile data (LDS store + barrier) before sleeping
    run<512, 56>("idle", 1024, 28672, 0, 6, d_t, in, out);
    run<512, 56>("tile loads", 1024, 28672, 2, 6, d_t, in, out);
    run<512, 56>("tile loads + wait", 1024, 28672, 2 | (16 << 4), 6, d_t, in, out);
    run<512, 56>("tables", 1024, 28672, 0 | (2 << 4), 6, d_t, in, out);
    run<512, 56>("tables + barrier", 1024, 28672, 0 | (6 << 4), 6, d_t, in, out);
    run<512, 56>("tables + tile loads", 1024, 28672, 2 | (2 << 4), 6, d_t, in, out);
    run<512, 56>("tables + tile loads + barrier", 1024, 28672, 2 | (6 << 4), 6, d_t, in, out);
    run<512, 56>("tables + tile loads + barrier + wait", 1024, 28672, 2 | (22 << 4), 6, d_t, in, out);
    run<512, 56>("tables + tile loads + barrier + wait + prio", 1024, 28672, 2 | (30 << 4), 6, d_t, in, out);
    run<512, 56>("tile loads + barrier + wait, short life", 1024, 28672, 2 | (20 << 4), 1, d_t, in, out);
    run<512, 56, 2>("2 KiB code", 1024, 28672, 0, 6, d_t, in, out);
    run<512, 56, 8>("8 KiB code", 1024, 28672, 0, 6, d_t, in, out);
    run<512, 56, 24>("24 KiB code", 1024, 28672, 0, 6, d_t, in, out);
    run<512, 56, 24>("24 KiB code + tables + tile loads + barrier", 1024, 28672, 2 | (6 << 4), 6, d_t, in, out);
    run<512, 56, 8>("8 KiB code behind tables + loads + barrier + wait", 1024, 28672, 2 | (22 << 4), 6, d_t, in, out);
    return 0;
}
