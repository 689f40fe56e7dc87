// EXPERIMENTS ONLY: one kernel per (target, mode) so that the static instruction count of each
// mode path can be read from the disassembly (tools/exp/mode_isa.py).
#include <hip/hip_runtime.h>
#include "../../basisu_rs_amd/csrc/bu_uastc_dispatch.hpp"

template <int TARGET, int M>
__global__ void mode_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, const BuTablesAll* __restrict__ tg)
{
    __shared__ BuTablesAll TA;
    for (unsigned i = threadIdx.x; i < sizeof(BuTablesAll) / 4; i += blockDim.x) ((uint32_t*)&TA)[i] = ((const uint32_t*)tg)[i];
    const BuTables& T = TA.t;
    __syncthreads();
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 v = in[i];
    BuBlk b{{v.x, v.y, v.z, v.w}};
    uint32_t o[16];
    const int st = bu_block_mode<TARGET, M>(T, b, o);
    constexpr int NW = TARGET == BU_TGT_RGBA ? 16 : (TARGET == BU_TGT_ETC1 ? 2 : 4);
    if (st == 0) {
        for (int k = 0; k < NW; k += (NW >= 4 ? 4 : 2)) {
            if (NW >= 4) out[i * (NW / 4) + k / 4] = make_uint4(o[k], o[k + 1], o[k + 2], o[k + 3]);
            else ((uint2*)out)[i] = make_uint2(o[0], o[1]);
        }
    }
}
#define INST(T, M) template __global__ void mode_kernel<T, M>(const uint4*, uint4*, const BuTablesAll*);
#define INST_T(T) INST(T,0) INST(T,1) INST(T,2) INST(T,3) INST(T,4) INST(T,5) INST(T,6) INST(T,7) INST(T,8) INST(T,9) \
    INST(T,10) INST(T,11) INST(T,12) INST(T,13) INST(T,14) INST(T,15) INST(T,16) INST(T,17) INST(T,18)
INST_T(0) INST_T(1) INST_T(2) INST_T(3) INST_T(4)
