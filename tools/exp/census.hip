// EXPERIMENT: where do the 1024 workgroups of a 512-thread launch land?  Prints, per workgroup, XCC_ID and the HW_ID fields.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void census(unsigned* out)
{
    __shared__ unsigned pad[7000];  // ~28 KiB like the BC7 kernel: four workgroups per CU
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t;
        out[blockIdx.x * 4 + 3] = pad[(threadIdx.x + 1) & 511];
    }
    // stay resident long enough that all four slots of a CU are occupied together
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(100);
}
int main()
{
    const int G = 1024;
    unsigned* d;
    hipMalloc(&d, G * 16);
    std::vector<unsigned> h(G * 4);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(census, dim3(G), dim3(512), 0, 0, d);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    printf("block: xcc se sh cu tg(19:16) wave simd  raw\n");
    std::map<unsigned, std::vector<int>> bycu;
    for (int b = 0; b < G; b++) {
        unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
        unsigned wave = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 15;
        if (b < 40 || (b % 97) == 0) printf("%4d: %u %u %u %2u %2u %2u %u  %08x  t=%u\n", b, xcc, se, sh, cu, tg, wave, simd, hw, h[b * 4 + 2]);
        bycu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("distinct (xcc,se,sh,cu): %zu\n", bycu.size());
    int shown = 0;
    for (auto& kv : bycu) {
        if (shown++ < 12) {
            printf("cu %06x:", kv.first);
            for (int b : kv.second) printf(" %d(tg%u)", b, (h[b * 4] >> 16) & 15);
            printf("\n");
        }
    }
    std::map<size_t, int> hist;
    for (auto& kv : bycu) hist[kv.second.size()]++;
    for (auto& kv : hist) printf("CUs holding %zu workgroups: %d\n", kv.first, kv.second);
    return 0;
}
