// EXPERIMENT: what is the fastest any kernel shape moves 16 MiB in + 16 MiB out (cold, 64 rotated buffer pairs)?
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/copy_shapes tools/exp/copy_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int EPT, bool NT>
__global__ void copyk(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    uint4 v[EPT];
    const size_t base = (size_t)blockIdx.x * blockDim.x * EPT + threadIdx.x;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const size_t i = base + (size_t)k * blockDim.x;
        if (i < n) {
            if (NT) { v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(in + i)); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
            else v[k] = in[i];
        }
    }
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const size_t i = base + (size_t)k * blockDim.x;
        if (i < n) {
            if (NT) { v4u t = {v[k].x, v[k].y, v[k].z, v[k].w}; __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(out + i)); }
            else out[i] = v[k];
        }
    }
}
template <int EPT, bool NT>
float timeit(int wg, const std::vector<uint4*>& in, const std::vector<uint4*>& out, size_t n, int launches)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n + (size_t)wg * EPT - 1) / ((size_t)wg * EPT));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < launches; i++) hipLaunchKernelGGL((copyk<EPT, NT>), dim3(grid), dim3(wg), 0, 0, in[i % in.size()], out[i % out.size()], n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms / launches * 1e3f < best ? ms / launches * 1e3f : best;
    }
    return best;
}
// grid-stride form: a fixed grid of `wgs_per_cu` workgroups per CU walks the array (what the transcoders' multi-tile path does)
template <int EPT>
__global__ void copy_stride(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    const size_t step = (size_t)gridDim.x * blockDim.x * EPT;
    for (size_t base = (size_t)blockIdx.x * blockDim.x * EPT + threadIdx.x; base < n; base += step) {
        uint4 v[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const size_t i = base + (size_t)k * blockDim.x;
            if (i < n) { v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(in + i)); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
        }
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const size_t i = base + (size_t)k * blockDim.x;
            if (i < n) { v4u t = {v[k].x, v[k].y, v[k].z, v[k].w}; __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(out + i)); }
        }
    }
}
template <int EPT>
float time_stride(int wg, int wgs_per_cu, const std::vector<uint4*>& in, const std::vector<uint4*>& out, size_t n, int launches)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < launches; i++) hipLaunchKernelGGL((copy_stride<EPT>), dim3(256 * wgs_per_cu), dim3(wg), 0, 0, in[i % in.size()], out[i % out.size()], n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms / launches * 1e3f < best ? ms / launches * 1e3f : best;
    }
    return best;
}
int main(int argc, char** argv)
{
    if (argc > 1) {  // large sizes: copy_shapes LOG2_BLOCKS
        const size_t n = (size_t)1 << atoi(argv[1]);
        const int nb = n >= ((size_t)1 << 24) ? 3 : 8;
        std::vector<uint4*> in(nb), out(nb);
        for (int k = 0; k < nb; k++) { hipMalloc(&in[k], n * 16); hipMalloc(&out[k], n * 16); hipMemset(in[k], k, n * 16); }
        hipDeviceSynchronize();
        const double gb = 32.0 * (double)n / 1e3;  // bytes per launch / 1e3 -> us * GB/s
        const int L = n >= ((size_t)1 << 24) ? 24 : 96;
        for (int rep = 0; rep < 2; rep++) {
        float a = timeit<1, true>(256, in, out, n, L), b = timeit<4, true>(512, in, out, n, L), c = timeit<8, true>(512, in, out, n, L);
        printf("2^%s blocks, one pass per thread: 256x1 %.1f us (%.0f GB/s)  512x4 %.1f us (%.0f GB/s)  512x8 %.1f us (%.0f GB/s)\n", argv[1], a, gb / a, b, gb / b, c, gb / c);
        for (int wpc : {2, 4, 8}) {
            float d = time_stride<1>(512, wpc, in, out, n, L), e = time_stride<2>(512, wpc, in, out, n, L), f = time_stride<4>(512, wpc, in, out, n, L), g = time_stride<4>(1024, wpc / 2 ? wpc / 2 : 1, in, out, n, L);
            printf("  grid-stride, %d x 512 threads per CU: ept1 %.1f us (%.0f GB/s)  ept2 %.1f (%.0f)  ept4 %.1f (%.0f) | %d x 1024 ept4 %.1f (%.0f)\n", wpc, d, gb / d, e, gb / e, f, gb / f, wpc / 2 ? wpc / 2 : 1, g, gb / g);
        }
        }
        return 0;
    }
    const size_t n = 1 << 20;
    std::vector<uint4*> in(64), out(64);
    for (int k = 0; k < 64; k++) { hipMalloc(&in[k], n * 16); hipMalloc(&out[k], n * 16); hipMemset(in[k], k, n * 16); }
    hipDeviceSynchronize();
    for (int wg : {256, 512, 1024}) {
        printf("wg %4d nt : ept1 %.2f  ept2 %.2f  ept4 %.2f  ept8 %.2f us\n", wg, timeit<1, true>(wg, in, out, n, 512), timeit<2, true>(wg, in, out, n, 512),
               timeit<4, true>(wg, in, out, n, 512), timeit<8, true>(wg, in, out, n, 512));
        printf("wg %4d pl : ept1 %.2f  ept2 %.2f  ept4 %.2f  ept8 %.2f us\n", wg, timeit<1, false>(wg, in, out, n, 512), timeit<2, false>(wg, in, out, n, 512),
               timeit<4, false>(wg, in, out, n, 512), timeit<8, false>(wg, in, out, n, 512));
    }
    return 0;
}
