// EXPERIMENT (round 4): what does a 1024-block tile cost on its way THROUGH LDS, and does CDNA4's direct global -> LDS load
// (global_load_lds_dwordx4: no VGPRs, no ds_write) make it cheaper?  16 MiB in + 16 MiB out per launch, 512-thread workgroups on
// 1024-block tiles, four per CU (28 KiB of LDS each, like the BC7 kernel), cold rotation over 64 buffer pairs.
//   plain     : load -> store (no LDS)                                              = the copy in the transcoder's launch shape
//   regs_seq  : load -> ds_write_b128 -> barrier -> ds_read_b128 sequential -> store
//   regs_perm : the same, read back in a scattered order (what the sort's write-back does)
//   direct_*  : global_load_lds_dwordx4 -> s_waitcnt vmcnt(0) -> barrier -> ds_read_b128 -> store
//   twice_*   : two trips through LDS (write, barrier, read, write, barrier, read): the sorted kernel's traffic without its work
// hipcc --offload-arch=gfx950 -O3 -o tools/exp/lds_direct tools/exp/lds_direct.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldnt(const uint4* p) { v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void stnt(uint4* p, uint4 v) { v4u t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(p)); }
__device__ __forceinline__ unsigned perm(unsigned i) { return (i * 397u + 13u) & 1023u; }  // a bijection of 0..1023 that scatters neighbours

template <int MODE>
__global__ __launch_bounds__(512) void tilek(const uint4* __restrict__ in, uint4* __restrict__ out)
{
    __shared__ uint4 tile[1024 + 768];  // 28 KiB: four workgroups per CU, as the BC7 kernel
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * 1024;
    if constexpr (MODE == 8 || MODE == 9) {
        // half of the waves move four blocks per lane, the other half nothing (MODE 9: they still stage 8 KiB into LDS, like tables)
        if (wave >= 4) {
            if constexpr (MODE == 9) {
                tile[1024 + (tid - 256)] = in[tid & 255u];
                tile[1024 + 256 + (tid - 256)] = in[256 + (tid & 255u)];
                __syncthreads();
            }
            return;
        }
        uint4 q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = ldnt(in + base + k * 256 + tid);
        if constexpr (MODE == 9) __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) stnt(out + base + k * 256 + tid, q[k]);
        return;
    }
    if constexpr (MODE == 0) {
        const uint4 a = ldnt(in + base + tid), b = ldnt(in + base + 512 + tid);
        stnt(out + base + tid, a);
        stnt(out + base + 512 + tid, b);
        return;
    }
    if constexpr (MODE == 1 || MODE == 2 || MODE == 5 || MODE == 6) {
        const uint4 a = ldnt(in + base + tid), b = ldnt(in + base + 512 + tid);
        tile[tid] = a;
        tile[512 + tid] = b;
    } else {
        // LDS address = M0 (wave-uniform base) + lane * 16: a wave's 64 blocks land contiguously, in order
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        __builtin_amdgcn_global_load_lds((gptr_t)(in + base + tid), (lptr_t)(&tile[wave * 64]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(in + base + 512 + tid), (lptr_t)(&tile[512 + wave * 64]), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const bool scattered = MODE == 2 || MODE == 4 || MODE == 6 || MODE == 7;
    uint4 r0 = tile[scattered ? perm(tid) : tid], r1 = tile[scattered ? perm(512 + tid) : 512 + tid];
    if constexpr (MODE >= 5) {  // second trip: results back into LDS at the slot they were read from, then out in original order
        __syncthreads();
        r0.x ^= 1u;
        r1.x ^= 1u;
        tile[scattered ? perm(tid) : tid] = r0;
        tile[scattered ? perm(512 + tid) : 512 + tid] = r1;
        __syncthreads();
        r0 = tile[tid];
        r1 = tile[512 + tid];
    }
    stnt(out + base + tid, r0);
    stnt(out + base + 512 + tid, r1);
}
template <int MODE>
float timeit(const std::vector<uint4*>& in, const std::vector<uint4*>& out, size_t n, int launches)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < launches; i++) hipLaunchKernelGGL((tilek<MODE>), dim3((unsigned)(n / 1024)), dim3(512), 0, 0, in[i % in.size()], out[i % out.size()]);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms / launches * 1e3f < best ? ms / launches * 1e3f : best;
    }
    return best;
}
int main()
{
    const size_t n = 1 << 20; const int NB = 64;
    std::vector<uint4*> in(NB), out(NB);
    std::vector<uint4> h(n);
    for (size_t i = 0; i < n; i++) h[i] = make_uint4((unsigned)i, (unsigned)(i * 7), 3, 4);
    for (int k = 0; k < NB; k++) { hipMalloc(&in[k], n * 16); hipMalloc(&out[k], n * 16); hipMemcpy(in[k], h.data(), n * 16, hipMemcpyHostToDevice); }
    // correctness of the direct load: sequential read-back must reproduce the input
    hipLaunchKernelGGL((tilek<3>), dim3((unsigned)(n / 1024)), dim3(512), 0, 0, in[0], out[0]);
    std::vector<uint4> g(n); hipMemcpy(g.data(), out[0], n * 16, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < n; i++) bad += g[i].x != h[i].x || g[i].y != h[i].y;
    printf("direct global->LDS load, read back in order: %zu of %zu blocks differ\n", bad, n);
    for (int round = 0; round < 3; round++)
        printf("plain %.2f | one trip: regs_seq %.2f regs_perm %.2f direct_seq %.2f direct_perm %.2f | two trips: regs_seq %.2f regs_perm %.2f direct+regs_perm %.2f  (us per launch)\n",
               timeit<0>(in, out, n, 256), timeit<1>(in, out, n, 256), timeit<2>(in, out, n, 256), timeit<3>(in, out, n, 256), timeit<4>(in, out, n, 256),
               timeit<5>(in, out, n, 256), timeit<6>(in, out, n, 256), timeit<7>(in, out, n, 256));
    for (int round = 0; round < 3; round++)
        printf("512-thread workgroups on 1024-block tiles: every wave moves 2 blocks per lane %.2f | waves 0-3 move 4 per lane, waves 4-7 exit %.2f | ... waves 4-7 stage 8 KiB into LDS, one barrier %.2f\n",
               timeit<0>(in, out, n, 256), timeit<8>(in, out, n, 256), timeit<9>(in, out, n, 256));
    return 0;
}
