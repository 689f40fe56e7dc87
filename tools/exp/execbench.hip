// EXPERIMENT: does a wave64 VALU instruction with only the low 32 (or 16) lanes enabled issue faster on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* o, unsigned seed, int trips)
{
    unsigned a = threadIdx.x * seed, b = seed + 1, c = a ^ 0x55;
    unsigned long long mask = MODE == 0 ? ~0ull : MODE == 1 ? 0xFFFFFFFFull : MODE == 2 ? 0xFFFFull : MODE == 3 ? 0xFFFFFFFF00000000ull : 0x00000000FFFF0000ull;
    asm volatile("s_mov_b64 exec, %0" ::"s"(mask));
    for (int i = 0; i < trips; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 32; k2++) {
            asm volatile("v_and_or_b32 %0, %0, %3, %4\n\tv_and_or_b32 %1, %1, %3, %4\n\tv_and_or_b32 %2, %2, %3, %4\n\tv_perm_b32 %0, %0, %1, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(seed), "v"(i));
        }
    }
    asm volatile("s_mov_b64 exec, -1");
    if (seed == 0x1234567) o[threadIdx.x] = a + b + c;
}
int main()
{
    unsigned* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"all 64 lanes", "low 32 lanes", "low 16 lanes", "high 32 lanes", "lanes 16..31"};
    float ms[5];
#define RUN(M) { for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<M>, dim3(1024), dim3(512), 0, 0, d, 7u, 64); hipDeviceSynchronize(); hipEventRecord(e0); \
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<M>, dim3(1024), dim3(512), 0, 0, d, 7u, 64); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[M], e0, e1); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    for (int m = 0; m < 5; m++) printf("%-16s %8.1f us  %.2f clk/instr/SIMD\n", names[m], ms[m] * 200.0, ms[m] * 200.0 * 2400.0 / (8.0 * 64 * 32 * 4));
    return 0;
}
