// tools/exp/profiler_floor.hip: how fast can dispatches complete when they come from S streams -- plain, and under rocprofv3 --kernel-trace?
// A kernel that does nothing (one workgroup) and one that is busy for ~5 us on the whole chip, round-robin over S streams, N launches;
// prints the completion period.  Build: hipcc --offload-arch=gfx950 -O2 -o profiler_floor profiler_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_empty() {}
__global__ void k_busy(unsigned* p, int iters)
{
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) p[0] = x;
}
int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 2000, iters = argc > 2 ? atoi(argv[2]) : 180;
    unsigned* d;
    hipMalloc(&d, 4);
    for (int busy = 0; busy < 2; busy++)
        for (int S = 1; S <= 4; S *= 2) {
            hipStream_t st[4];
            for (int i = 0; i < S; i++) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
            for (int rep = 0; rep < 2; rep++) {
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; i++) {
                    if (busy) hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, st[i % S], d, iters);
                    else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[i % S]);
                }
                hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep) printf("%s kernel, %d stream(s): %.2f us per launch (%d launches, wall clock)\n", busy ? "busy " : "empty", S, us / N, N);
            }
            for (int i = 0; i < S; i++) hipStreamDestroy(st[i]);
        }
    return 0;
}
