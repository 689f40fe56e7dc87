// EXPERIMENT: does hipExtAnyOrderLaunch let consecutive kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Ten 100-us sleeping waves on one stream: 1 ms if they run one after another, ~0.1 ms if they overlap.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void sleeper(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 10; i++) {
                if (mode == 0) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, s, 10000ull);
                else hipExtLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 10000ull);
            }
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: 10 x 100 us sleepers on one stream took %.3f ms (%s)\n", mode ? "hipExtAnyOrderLaunch" : "plain launch", ms, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
