// UASTC -> ETC1 and UASTC -> ETC2 RGBA (EAC alpha + ETC1 colour) for the gfx950 kernels.
// Replaces src/target_formats/etc.rs:11-341 of the reference:
//   :43-76    UASTC mode 8 uses its stored ETC1 flags directly
//   :78-111   full RGBA decode, optional transpose, per-half average
//   :113-158  bias (apply_etc1_bias :203-259), individual / differential base colours, header byte
//   :160-198  selectors by luma thresholds; Selector::set_selector bit planes (:363-393)
//   :261-341  EAC alpha block (solid / searched); the f32 centre is evaluated as exact integers
// The transpose is never materialised: the two ETC1 sub-blocks are sums of 2x2 quadrants, and the
// flip bit only chooses which quadrants pair up.
#pragma once
#include "bu_uastc_front.hpp"

BU_DEV int bu_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// etc.rs:261-275: {value, 0x1D, 0x92, 0x49, 0x24, 0x92, 0x49, 0x24}
BU_DEV void bu_eac_solid(uint32_t out[2], uint32_t value)
{
    out[0] = 0x49921D00u | value;
    out[1] = 0x24499224u;
}

// etc.rs:277-341
BU_DEV void bu_eac_block(const BuTables& T, uint32_t out[2], uint32_t etc2tm, const uint32_t px[16])
{
    uint32_t mn = 255, mx = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        const uint32_t a = px[i] >> 24;
        mn = a < mn ? a : mn;
        mx = a > mx ? a : mx;
    }
    if (etc2tm == 0) {
        bu_eac_solid(out, 255);
        return;
    }
    if (mn == mx) {
        bu_eac_solid(out, mn);
        return;
    }
    const uint32_t table = etc2tm & 15u;
    const int mult = (int)(etc2tm >> 4);
    const int mm = T.eac_mod_min[table], range = T.eac_range[table];
    // round(lerp(min, max, -mod_min/range)) == floor((2*(min*(range+mm) - max*mm) + range) / (2*range)):
    // every range is odd, so there are no .5 ties (tests/test_float_sites.py)
    const uint32_t num = (uint32_t)(2 * ((int)mn * (range + mm) - (int)mx * mm) + range);
    const int center = (int)((num * T.eac_magic[table]) >> 20);  // num <= 14791, 2*range <= 58: exact
    // nearest of the 8 table values, first minimum wins (min_by_key): key = 8*|value - a| + index via v_sad_u32, an
    // 8-way minimum as three v_min3_u32 + one v_min_u32.  The index is the low 3 bits of the winning key, and one
    // v_alignbit_b32 per texel funnels exactly those 3 bits into the top of an accumulator -- texels are visited in
    // descending column-major id, so id 0 ends up highest, as etc.rs:324-327 lays the 48-bit string out.
    uint32_t values8[8];
    BU_UNROLL
    for (int k = 0; k < 8; k++) values8[k] = 8u * (uint32_t)bu_clampi(center + T.etc2_amod[8 * table + k] * mult, 0, 255);
    uint32_t acc[2] = {0, 0};  // acc[0]: ids 0..7 in bits 8..31 (id 0 on top), acc[1]: ids 8..15
    BU_UNROLL
    for (int id = 15; id >= 0; id--) {
        const int i = (id % 4) * 4 + id / 4;  // column-major id -> row-major texel (etc.rs:324-327)
        const uint32_t a8 = (px[i] >> 21) & 0x7F8u;
        const uint32_t k0 = bu_sad<0>(values8[0], a8), k1 = bu_sad<1>(values8[1], a8), k2 = bu_sad<2>(values8[2], a8), k3 = bu_sad<3>(values8[3], a8);
        const uint32_t k4 = bu_sad<4>(values8[4], a8), k5 = bu_sad<5>(values8[5], a8), k6 = bu_sad<6>(values8[6], a8), k7 = bu_sad<7>(values8[7], a8);
        const uint32_t best = bu_umin(bu_umin3(k0, k1, k2), bu_umin3(k6, k7, bu_umin3(k3, k4, k5)));
        acc[id >> 3] = bu_alignbit(best, acc[id >> 3], 3);
    }
    // bytes 2..7 of the block are the 48-bit string big-endian: acc[0] bytes 3,2,1 then acc[1] bytes 3,2,1
    out[0] = bu_perm(acc[0], (uint32_t)center | (etc2tm << 8), 0x06070100u);
    out[1] = bu_perm(acc[1], acc[0], 0x05060701u);
}

// out: ETC1 -> out[0..1]; ETC2 -> out[0..1] alpha, out[2..3] colour (etc.rs:19-30)
template <int M, bool ETC2>
BU_DEV int bu_block_etc(const BuTables& T, const BuBlk& b, uint32_t out[4])
{
    uint32_t* col = ETC2 ? out + 2 : out;
    if constexpr (M == 8) {
        const uint32_t c = bu_bits(b, 5, 32);
        if constexpr (ETC2) bu_eac_solid(out, c >> 24);
        // uastc.rs:400-409: etc1d(1) etc1i(3) etc1s(2) etc1r(5) etc1g(5) etc1b(5) from bit 37
        const uint32_t d = bu_bits(b, 37, 1), in = bu_bits(b, 38, 3), s = bu_bits(b, 41, 2);
        const uint32_t r = bu_bits(b, 43, 5), g = bu_bits(b, 48, 5), bl = bu_bits(b, 53, 5);
        uint32_t b0, b1, b2;
        if (!d) {  // u8 arithmetic, release-build wrapping (etc.rs:54-56)
            b0 = ((r << 4) | r) & 0xFFu;
            b1 = ((g << 4) | g) & 0xFFu;
            b2 = ((bl << 4) | bl) & 0xFFu;
        } else {
            b0 = (r << 3) & 0xFFu;
            b1 = (g << 3) & 0xFFu;
            b2 = (bl << 3) & 0xFFu;
        }
        const uint32_t b3 = ((in << 5) | (in << 2) | (d << 1)) & 0xFFu;
        col[0] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        // selector id -> ETC1 code [3,2,0,1]; high plane in bytes 4-5, low plane in bytes 6-7 (etc.rs:68-73)
        const uint32_t code = (0x4Bu >> (2 * s)) & 3u;  // 0b01_00_10_11
        col[1] = ((code & 2u) ? 0x0000FFFFu : 0u) | ((code & 1u) ? 0xFFFF0000u : 0u);
        return BU_ST_OK;
    } else {
        using L = BuLayout<M>;
        uint32_t px[16];
        const int st = bu_block_rgba<M>(T, b, px);
        if (st) return st;
        const uint32_t f = bu_bits(b, L::pos_etc1f, 1), d = bu_bits(b, L::pos_etc1d, 1);
        const uint32_t i0 = bu_bits(b, L::pos_etc1i0, 3), i1 = bu_bits(b, L::pos_etc1i1, 3);
        if constexpr (ETC2) {
            if constexpr (L::has_alpha) bu_eac_block(T, out, bu_bits(b, L::pos_etc2tm, 8), px);
            else bu_eac_solid(out, 255);  // etc2tm = 0 for RGB modes (uastc.rs:430-434)
        }
        // quadrant sums: R and B in 16-bit lanes of one word, G separately
        uint32_t qrb[4] = {0, 0, 0, 0}, qg[4] = {0, 0, 0, 0};
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const int q = ((i >> 3) << 1) | ((i >> 1) & 1);  // row>=2, col>=2
            qrb[q] += px[i] & 0x00FF00FFu;
            qg[q] += (px[i] >> 8) & 0xFFu;
        }
        // flip (etc1f) : halves are rows 0-1 / 2-3; otherwise columns 0-1 / 2-3 (etc.rs:86-95)
        const uint32_t srb0 = qrb[0] + (f ? qrb[1] : qrb[2]), srb1 = qrb[3] + (f ? qrb[2] : qrb[1]);
        const uint32_t sg0 = qg[0] + (f ? qg[1] : qg[2]), sg1 = qg[3] + (f ? qg[2] : qg[1]);
        const int limit = d ? 31 : 15;
        int c[2][3];
        {
            const uint32_t sums[2][3] = {{srb0 & 0xFFFFu, sg0, srb0 >> 16}, {srb1 & 0xFFFFu, sg1, srb1 >> 16}};
            BU_UNROLL
            for (int sb = 0; sb < 2; sb++)
                BU_UNROLL
                for (int ch = 0; ch < 3; ch++) {
                    // (sum*limit + 1020) / 2040  (etc.rs:109) = floor(floor(x/8)/255)
                    const uint32_t y = (sums[sb][ch] * (uint32_t)limit + 1020u) >> 3;
                    c[sb][ch] = (int)((y + 1u + (y >> 8)) >> 8);
                }
        }
        if constexpr (!L::m1012) {
            // apply_etc1_bias (etc.rs:203-259): the six per-channel adjustments are one LUT read each
            const uint32_t p5 = (uint32_t)T.etc1_bias[bu_bits(b, L::pos_etc1bias, 5)] << 5, dsel = d << 7;
            BU_UNROLL
            for (int sb = 0; sb < 2; sb++)
                BU_UNROLL
                for (int ch = 0; ch < 3; ch++)
                    c[sb][ch] = (int)T.etc1_biasv[((p5 >> (2 * (sb * 3 + ch))) & 0x60u) | dsel | (uint32_t)c[sb][ch]];
        }
        // header bytes; cq[sb][ch] = the quantised base colour each half actually decodes with (etc.rs:113-158)
        uint32_t cq[2][3];
        uint32_t hdr = 0;
        BU_UNROLL
        for (int ch = 0; ch < 3; ch++) {
            // individual: 4+4 bits (etc.rs:122-129); differential: 5 bits + clamped 3-bit delta, and the second half decodes
            // from c0 + delta (etc.rs:130-149).  Branch-free: both forms are a few ALU ops, `d` only selects.
            const int dl = bu_clampi(c[1][ch] - c[0][ch], -4, 3);
            const uint32_t byte_i = (((uint32_t)c[0][ch] << 4) | (uint32_t)c[1][ch]) & 0xFFu;
            const uint32_t byte_d = (((uint32_t)c[0][ch] << 3) | ((uint32_t)dl & 7u)) & 0xFFu;
            hdr |= (d ? byte_d : byte_i) << (8 * ch);
            cq[0][ch] = (uint32_t)c[0][ch];
            cq[1][ch] = d ? (uint32_t)((c[0][ch] + dl) & 31) : (uint32_t)c[1][ch];
        }
        hdr |= (((i0 << 5) | (i1 << 2) | (d << 1) | f) & 0xFFu) << 24;  // etc.rs:151-158
        col[0] = hdr;

        // luma thresholds per half (etc.rs:165-177).  The factors (108, 366, 38) are all even, so every luma is even:
        // with lum = 2*L (L = 54r + 183g + 19b) the reference's test lum >= (lum_a + lum_b)/2 is exactly
        // L >= (L_a + L_b + 1) >> 1.  The four modified base colours of a half come from one LUT read per channel (byte k =
        // clamp(base + modifier k)); their lumas are three v_dot4_u32_u8 each with single-byte weight words.
        constexpr uint32_t LW = 54u | (183u << 8) | (19u << 16);
        uint32_t thr[2][3];
        BU_UNROLL
        for (int sb = 0; sb < 2; sb++) {
            const uint32_t sel = (d << 8) | ((sb ? i1 : i0) << 5);
            const uint32_t r4 = T.etc1_thrcol[sel | cq[sb][0]], g4 = T.etc1_thrcol[sel | cq[sb][1]], b4 = T.etc1_thrcol[sel | cq[sb][2]];
            uint32_t lum[4];
            BU_UNROLL
            for (int k = 0; k < 4; k++) lum[k] = bu_udot4(b4, 19u << (8 * k), bu_udot4(g4, 183u << (8 * k), bu_udot4(r4, 54u << (8 * k), 0u)));
            thr[sb][0] = (lum[0] + lum[1] + 1u) >> 1;
            thr[sb][1] = (lum[1] + lum[2] + 1u) >> 1;
            thr[sb][2] = (lum[2] + lum[3] + 1u) >> 1;
        }
        // the top-right and bottom-left 2x2 quadrants change half with the flip bit: pick their thresholds once.
        // Kept NEGATED: v_dot4_u32_u8 adds an accumulator for free, so dot4(texel, LW, -thr) = luma - thr in one instruction
        // and bit 31 of the result is the comparison (both operands < 2^31).
        uint32_t nthq[4][3];
        BU_UNROLL
        for (int k = 0; k < 3; k++) {
            const uint32_t n0 = 0u - thr[0][k], n1 = 0u - thr[1][k];
            nthq[0][k] = n0;
            nthq[1][k] = f ? n0 : n1;  // x >= 2, y < 2
            nthq[2][k] = f ? n1 : n0;  // x < 2, y >= 2
            nthq[3][k] = n1;
        }
        // sel = #thresholds <= luma; ETC1 code [3,2,0,1][sel]: high bit = sel < 2 = (luma < thr1), low bit = sel == 0 or
        // sel == 3 = NOT (luma < thr0 xor luma < thr2) (the thresholds are monotone).  Texels are visited in descending
        // pixel id (x*4 + y, etc.rs:376-392) and one v_alignbit_b32 per plane shifts the sign bit in from the right.
        uint32_t msbp = 0, lsbx = 0;
        BU_UNROLL
        for (int pid = 15; pid >= 0; pid--) {
            const int x = pid >> 2, y = pid & 3;
            const int q = ((y >> 1) << 1) | (x >> 1);
            const uint32_t t = px[y * 4 + x];
            const uint32_t d0 = bu_udot4(t, LW, nthq[q][0]), d1 = bu_udot4(t, LW, nthq[q][1]), d2 = bu_udot4(t, LW, nthq[q][2]);
            msbp = bu_alignbit(msbp, d1, 31);       // (msbp << 1) | (luma < thr1)
            lsbx = bu_alignbit(lsbx, d0 ^ d2, 31);  // (lsbx << 1) | (lt0 ^ lt2)
        }
        const uint32_t lsbp = ~lsbx & 0xFFFFu;
        msbp &= 0xFFFFu;
        col[1] = ((msbp >> 8) & 0xFFu) | ((msbp & 0xFFu) << 8) | (((lsbp >> 8) & 0xFFu) << 16) | ((lsbp & 0xFFu) << 24);
        return BU_ST_OK;
    }
}
