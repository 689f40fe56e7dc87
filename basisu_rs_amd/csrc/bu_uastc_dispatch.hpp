// Per-block entry point of the UASTC kernels: 7-bit mode code -> mode-specialised transcoder.
// Targets mirror uastc::TargetTextureFormat (uastc.rs:41-47) plus the RGBA32 unpack (uastc.rs:89-110).
#pragma once
#include "bu_uastc_astc.hpp"
#include "bu_uastc_bc7.hpp"
#include "bu_uastc_etc.hpp"

enum { BU_TGT_ASTC = 0, BU_TGT_BC7 = 1, BU_TGT_ETC1 = 2, BU_TGT_ETC2 = 3, BU_TGT_RGBA = 4 };

template <int TARGET, int M>
BU_DEV int bu_block_mode(const BuTables& T, const BuBlk& b, uint32_t* out)
{
    if constexpr (TARGET == BU_TGT_ASTC) return bu_block_astc<M>(T, b, out);
    else if constexpr (TARGET == BU_TGT_BC7) return bu_block_bc7<M>(T, b, out);
    else if constexpr (TARGET == BU_TGT_ETC1) return bu_block_etc<M, false>(T, b, out);
    else if constexpr (TARGET == BU_TGT_ETC2) return bu_block_etc<M, true>(T, b, out);
    else return bu_block_rgba<M>(T, b, out);
}

// out: 4 words (ASTC/BC7/ETC2), 2 words (ETC1) or 16 words (RGBA, row-major texels of the block)
template <int TARGET>
BU_DEV int bu_block_any(const BuTables& T, uint32_t mode, const BuBlk& b, uint32_t* out)
{
    switch (mode) {
#define BU_CASE(m) \
    case m: return bu_block_mode<TARGET, m>(T, b, out);
        BU_CASE(0) BU_CASE(1) BU_CASE(2) BU_CASE(3) BU_CASE(4) BU_CASE(5) BU_CASE(6) BU_CASE(7) BU_CASE(8) BU_CASE(9)
        BU_CASE(10) BU_CASE(11) BU_CASE(12) BU_CASE(13) BU_CASE(14) BU_CASE(15) BU_CASE(16) BU_CASE(17) BU_CASE(18)
#undef BU_CASE
    default: return BU_ST_BAD_MODE;  // 7-bit code 69 -> LUT value 19 (uastc.rs:329-341)
    }
}
