// EXPERIMENT: where a pageable host -> device -> host round trip of one 4096^2 atlas (16 MiB each way) spends its time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, reps, body)                                   \
    do {                                                       \
        body;                                                  \
        double best = 1e9;                                     \
        for (int r_ = 0; r_ < reps; r_++) {                    \
            double t0 = now();                                 \
            body;                                              \
            double t1 = now();                                 \
            if (t1 - t0 < best) best = t1 - t0;                \
        }                                                      \
        printf("%-64s %8.3f ms\n", label, best);               \
    } while (0)
int main()
{
    const size_t N = 16u << 20;
    void *d_in, *d_out;
    hipMalloc(&d_in, N);
    hipMalloc(&d_out, N);
    char* h_in = (char*)aligned_alloc(4096, N);
    char* h_out = (char*)aligned_alloc(4096, N);
    memset(h_in, 1, N);
    memset(h_out, 2, N);
    hipStream_t s0, s1;
    hipStreamCreate(&s0);
    hipStreamCreate(&s1);
    T("pageable hipMemcpy H2D 16 MiB", 8, hipMemcpy(d_in, h_in, N, hipMemcpyHostToDevice));
    T("pageable hipMemcpy D2H 16 MiB", 8, hipMemcpy(h_out, d_out, N, hipMemcpyDeviceToHost));
    T("pageable H2D then D2H (serial, one stream)", 8, { hipMemcpyAsync(d_in, h_in, N, hipMemcpyHostToDevice, s0); hipMemcpyAsync(h_out, d_out, N, hipMemcpyDeviceToHost, s0); hipStreamSynchronize(s0); });
    T("pageable H2D and D2H from two host threads", 8, {
        std::thread a([&] { hipMemcpy(d_in, h_in, N, hipMemcpyHostToDevice); });
        hipMemcpy(h_out, d_out, N, hipMemcpyDeviceToHost);
        a.join();
    });
    T("hipHostRegister 16 MiB + hipHostUnregister", 8, { hipHostRegister(h_in, N, hipHostRegisterDefault); hipHostUnregister(h_in); });
    T("hipHostRegister 16 MiB only (then unregister untimed)", 1, { hipHostRegister(h_in, N, hipHostRegisterDefault); });
    hipHostUnregister(h_in);
    {
        double t0 = now();
        hipHostRegister(h_in, N, hipHostRegisterDefault);
        double t1 = now();
        hipHostRegister(h_out, N, hipHostRegisterDefault);
        double t2 = now();
        printf("%-64s %8.3f ms, %8.3f ms\n", "register in, register out (single shots)", t1 - t0, t2 - t1);
    }
    T("registered H2D 16 MiB", 8, hipMemcpy(d_in, h_in, N, hipMemcpyHostToDevice));
    T("registered D2H 16 MiB", 8, hipMemcpy(h_out, d_out, N, hipMemcpyDeviceToHost));
    T("registered H2D and D2H on two streams", 8, { hipMemcpyAsync(d_in, h_in, N, hipMemcpyHostToDevice, s0); hipMemcpyAsync(h_out, d_out, N, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s0); hipStreamSynchronize(s1); });
    {
        double t0 = now();
        hipHostUnregister(h_in);
        double t1 = now();
        hipHostUnregister(h_out);
        double t2 = now();
        printf("%-64s %8.3f ms, %8.3f ms\n", "unregister in, out", t1 - t0, t2 - t1);
    }
    // staging by hand: CPU copy into a page-locked ring, 4 threads
    char *p_in, *p_out;
    hipHostMalloc((void**)&p_in, N, hipHostMallocDefault);
    hipHostMalloc((void**)&p_out, N, hipHostMallocDefault);
    T("memcpy 16 MiB pageable -> page-locked, 1 thread", 8, memcpy(p_in, h_in, N));
    for (int nt : {2, 4, 8}) {
        char label[96];
        snprintf(label, sizeof label, "memcpy 16 MiB pageable -> page-locked, %d threads", nt);
        T(label, 8, {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back([&, k] { memcpy(p_in + k * (N / nt), h_in + k * (N / nt), N / nt); });
            for (auto& x : th) x.join();
        });
    }
    T("page-locked H2D 16 MiB", 8, hipMemcpy(d_in, p_in, N, hipMemcpyHostToDevice));
    T("page-locked D2H 16 MiB", 8, hipMemcpy(p_out, d_out, N, hipMemcpyDeviceToHost));
    return 0;
}
