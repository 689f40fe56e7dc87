// UASTC -> ASTC 4x4 block repack for the gfx950 kernels.
// Replaces src/target_formats/astc.rs:8-181 of the reference:
//   :17-43    void-extent block for UASTC mode 8
//   :57-78    blue-contraction avoidance (swap quantised lo/hi pairs, complement the subset's weights)
//   :80-96    13-bit block mode, 10-bit partition seed, CEM
//   :98-141   endpoints re-packed from UASTC's "all digits, then all bits" BISE layout into ASTC's
//             interleaved layout (trit/quint encode LUTs astc.rs:208-264)
//   :143-178  weights written from bit 127 downward, each field bit-reversed (bitwriter.rs:56-116),
//             then the 2-bit component selector of dual-plane modes
// Weight trick: bit-reversing each w-bit field and stacking the fields downward from bit 127 is the
// same as bit-reversing the whole regular weight string once -- three v_bfrev_b32 per block.
#pragma once
#include "bu_uastc_front.hpp"

template <int M>
BU_DEV int bu_block_astc(const BuTables& T, const BuBlk& b, uint32_t out[4])
{
    if constexpr (M == 8) {
        const uint32_t c = bu_bits(b, 5, 32);
        out[0] = 0xFFFFFDFCu;  // 12-bit header 0xDFC + 20 one-bits (astc.rs:23-27)
        out[1] = 0xFFFFFFFFu;  // astc.rs:28
        out[2] = ((c & 0xFFu) * 257u) | ((((c >> 8) & 0xFFu) * 257u) << 16);
        out[3] = (((c >> 16) & 0xFFu) * 257u) | (((c >> 24) * 257u) << 16);
        return BU_ST_OK;
    } else if constexpr (M == 3) {
        // Mode 3: 3 subsets x RGB = 18 endpoint values in BISE range 7 (one trit + 2 bits each), 2-bit weights.  The generic
        // path below keeps 18 digits and 18 bit fields in 36 registers -- the register peak of the whole ASTC kernel.
        // Here both stay packed, value i at bits [2i, 2i+2) of a 36-bit string, and every step is SWAR or a LUT read:
        // lo/hi pairs are adjacent 2-bit fields, a subset is a 12-bit window, a BISE group is a 10-bit window.
        using L = BuLayout<3>;
        out[0] = out[1] = out[2] = out[3] = 0;
        const uint32_t pat = bu_bits(b, L::pos_pat, L::pat_bits);
        if (pat >= (uint32_t)L::pat_count) return BU_ST_BAD_PATTERN;
        const BuPart pr = T.part[L::part_base + pat];
        const uint32_t g0 = T.trit5[bu_bits(b, L::pos_ep, 8)], g1 = T.trit5[bu_bits(b, L::pos_ep + 8, 8)];
        const uint32_t g2 = T.trit5[bu_bits(b, L::pos_ep + 16, 8)], g3 = T.trit5[bu_bits(b, L::pos_ep + 24, 5)] & 0x3Fu;  // 3 digits
        uint32_t dlo = g0 | (g1 << 10) | (g2 << 20) | (g3 << 30), dhi = g3 >> 2;
        uint32_t elo = bu_bits(b, L::pos_epbits, 32), ehi = bu_bits(b, L::pos_epbits + 32, 4);
        // blue-contraction avoidance (astc.rs:57-78): per subset, sum(lo) > sum(hi) over R,G,B <=> sum(hi - lo) < 0;
        // one LUT read per lo/hi pair gives the difference of the two dequantised values
        int32_t ds[3] = {0, 0, 0};
        BU_UNROLL
        for (int p = 0; p < 9; p++) {
            const uint32_t d4 = p < 8 ? (dlo >> (4 * p)) & 15u : dhi, e4 = p < 8 ? (elo >> (4 * p)) & 15u : ehi;
            ds[p / 3] += (int32_t)T.pairdiff7[d4 | (e4 << 4)];
        }
        const bool i0 = ds[0] < 0, i1 = ds[1] < 0, i2 = ds[2] < 0;
        // swap the pairs of inverted subsets in both strings
        {
            const uint32_t mlo = (i0 ? 0x00000FFFu : 0u) | (i1 ? 0x00FFF000u : 0u) | (i2 ? 0xFF000000u : 0u), mhi = i2 ? 0xFu : 0u;
            const uint32_t k = 0x33333333u;
            dlo ^= (dlo ^ (((dlo & k) << 2) | ((dlo >> 2) & k))) & mlo;
            elo ^= (elo ^ (((elo & k) << 2) | ((elo >> 2) & k))) & mlo;
            dhi ^= (dhi ^ (((dhi & 3u) << 2) | (dhi >> 2))) & mhi;
            ehi ^= (ehi ^ (((ehi & 3u) << 2) | (ehi >> 2))) & mhi;
        }
        // header (astc.rs:80-96): block mode, partition seed + "one CEM for all", CEM 8 (RGB direct)
        bu_put(out, 0, 13, T.astc_mode13[3]);
        bu_put(out, 13, 10, pr.seed);
        bu_put(out, 25, 4, 8u);
        // endpoints in ASTC order (astc.rs:98-141): per group of five  m0 T01 m1 T23 m2 T4 m3 T56 m4 T7  = 18 bits; the last
        // group holds three values (its missing fields are zero)
        BU_UNROLL
        for (int g = 0; g < 4; g++) {
            const uint32_t dg = g < 3 ? (dlo >> (10 * g)) & 0x3FFu : (dlo >> 30) | (dhi << 2);
            const uint32_t e = g < 3 ? (elo >> (10 * g)) & 0x3FFu : (elo >> 30) | (ehi << 2);
            const uint32_t tsp = T.astc_trit_pk[dg];  // the trit byte, already spread (>> 2)
            const uint32_t v = (e & 3u) | ((e & 0xCu) << 2) | ((e & 0x30u) << 4) | ((e & 0xC0u) << 5) | ((e & 0x300u) << 7) | (tsp << 2);
            bu_put(out, 29 + 18 * g, 18, v);
        }
        // weights (astc.rs:143-178): complemented per inverted subset, then the whole string bit-reversed into the top
        uint32_t W[3];
        bu_decode_weights<3>(b, pr.uanch, W);
        const uint32_t m1 = (pr.upat & 0x55555555u) * 3u, m2 = ((pr.upat >> 1) & 0x55555555u) * 3u, m0 = ~(m1 | m2);
        W[0] ^= (i0 ? m0 : 0u) ^ (i1 ? m1 : 0u) ^ (i2 ? m2 : 0u);
        out[3] |= bu_brev(W[0]);
        return BU_ST_OK;
    } else if constexpr (M == 4 || M == 7) {
        // Modes 4 and 7: 2 subsets x RGB = 12 endpoint values in BISE range 12 (one quint + 3 bits each), 2-bit weights.
        // Same idea as mode 3, with 3-bit fields in 36-bit strings: lo/hi pairs are adjacent fields, a subset is an
        // 18-bit window, a BISE group a 9-bit window.  (With mode 3 these were the register peak of the ASTC kernel.)
        using L = BuLayout<M>;
        out[0] = out[1] = out[2] = out[3] = 0;
        const uint32_t pat = bu_bits(b, L::pos_pat, L::pat_bits);
        if (pat >= (uint32_t)L::pat_count) return BU_ST_BAD_PATTERN;
        const BuPart pr = T.part[L::part_base + pat];
        const uint32_t q0 = T.quint3[bu_bits(b, L::pos_ep, 7)], q1 = T.quint3[bu_bits(b, L::pos_ep + 7, 7)];
        const uint32_t q2 = T.quint3[bu_bits(b, L::pos_ep + 14, 7)], q3 = T.quint3[bu_bits(b, L::pos_ep + 21, 7)];
        uint64_t D = (uint64_t)(q0 | (q1 << 9) | (q2 << 18)) | ((uint64_t)q3 << 27);
        uint64_t E = (uint64_t)bu_bits(b, L::pos_epbits, 32) | ((uint64_t)bu_bits(b, L::pos_epbits + 32, 4) << 32);
        // blue-contraction avoidance (astc.rs:57-78): sum(lo) > sum(hi) over the subset's R,G,B <=> sum(hi - lo) < 0
        int32_t ds[2] = {0, 0};
        BU_UNROLL
        for (int i = 0; i < 12; i++) {
            const uint32_t tq = (uint32_t)(D >> (3 * i)) & 7u, eb = (uint32_t)(E >> (3 * i)) & 7u;
            const int32_t v = (int32_t)T.deq[bu_deq_ofs(12) + ((tq << 3) | eb)];
            ds[i / 6] += (i & 1) ? v : -v;
        }
        const bool i0 = ds[0] < 0, i1 = ds[1] < 0;
        {
            const uint64_t ev = 0x1C71C71C7ull;  // the even-numbered 3-bit fields
            const uint64_t mask = (i0 ? 0x3FFFFull : 0ull) | (i1 ? (0x3FFFFull << 18) : 0ull);
            D ^= (D ^ (((D & ev) << 3) | ((D >> 3) & ev))) & mask;
            E ^= (E ^ (((E & ev) << 3) | ((E >> 3) & ev))) & mask;
        }
        bu_put(out, 0, 13, T.astc_mode13[M]);
        bu_put(out, 13, 10, pr.seed);
        bu_put(out, 25, 4, 8u);
        // per group of three (astc.rs:98-141):  m0 Q[2:0] m1 Q[4:3] m2 Q[6:5]  = 16 bits
        BU_UNROLL
        for (int g = 0; g < 4; g++) {
            const uint32_t dg = (uint32_t)(D >> (9 * g)) & 0x1FFu, e = (uint32_t)(E >> (9 * g)) & 0x1FFu;
            const uint32_t v = (e & 7u) | ((e & 0x38u) << 3) | ((e & 0x1C0u) << 5) | (uint32_t)T.astc_quint_pk[dg];  // quint code already spread
            bu_put(out, 29 + 16 * g, 16, v);
        }
        uint32_t W[3];
        bu_decode_weights<M>(b, pr.uanch, W);
        const uint32_t m1 = (pr.upat & 0x55555555u) * 3u;
        W[0] ^= (i0 ? ~m1 : 0u) ^ (i1 ? m1 : 0u);
        out[3] |= bu_brev(W[0]);
        return BU_ST_OK;
    } else {
        using L = BuLayout<M>;
        constexpr int wb = L::d.wb, planes = L::d.planes, subsets = L::d.subsets, fmt = L::d.fmt;
        out[0] = out[1] = out[2] = out[3] = 0;
        uint32_t pat = 0;
        BuPart pr = {};
        if constexpr (L::pat_bits > 0) {
            pat = bu_bits(b, L::pos_pat, L::pat_bits);
            if (pat >= (uint32_t)L::pat_count) return BU_ST_BAD_PATTERN;
            pr = T.part[L::part_base + pat];
        }
        const uint32_t compsel = L::compsel_bits ? bu_bits(b, L::pos_compsel, 2) : 3u;

        uint32_t tq[18], eb[18];
        bu_decode_quant<M>(T, b, tq, eb);

        // ---- blue-contraction avoidance (astc.rs:57-78) -----------------------------------------
        bool inv[3] = {false, false, false};
        if constexpr (fmt != BU_FMT_LA) {
            constexpr int per = L::ep_count / subsets;
            BU_UNROLL
            for (int s = 0; s < subsets; s++) {
                uint32_t s0 = 0, s1 = 0;
                BU_UNROLL
                for (int c = 0; c < 3; c++) {
                    s0 += bu_deq<L::d.range>(T, tq[per * s + 2 * c], eb[per * s + 2 * c]);
                    s1 += bu_deq<L::d.range>(T, tq[per * s + 2 * c + 1], eb[per * s + 2 * c + 1]);
                }
                inv[s] = s0 > s1;
                BU_UNROLL
                for (int k = 0; k < per; k += 2) {
                    const uint32_t a = tq[per * s + k], c2 = eb[per * s + k];
                    tq[per * s + k] = inv[s] ? tq[per * s + k + 1] : a;
                    tq[per * s + k + 1] = inv[s] ? a : tq[per * s + k + 1];
                    eb[per * s + k] = inv[s] ? eb[per * s + k + 1] : c2;
                    eb[per * s + k + 1] = inv[s] ? c2 : eb[per * s + k + 1];
                }
            }
        }

        // ---- header (astc.rs:80-96) --------------------------------------------------------------
        int pos = 0;
        bu_put(out, 0, 13, T.astc_mode13[M]);
        pos = 13;
        if constexpr (L::pat_bits > 0) {
            bu_put(out, pos, 10, pr.seed);
            pos += 12;  // seed + 2 zero bits: all endpoints share one CEM
        }
        bu_put(out, pos, 4, fmt == BU_FMT_RGB ? 8u : (fmt == BU_FMT_RGBA ? 12u : 4u));
        pos += 4;

        // ---- endpoints in ASTC BISE order (astc.rs:98-141) ---------------------------------------
        constexpr int n = L::ep_count, bc = L::ebits;
        if constexpr (L::quints) {
            BU_UNROLL
            for (int g = 0; g * 3 < n; g++) {
                const uint32_t id = (tq[3 * g + 2] * 5u + tq[3 * g + 1]) * 5u + tq[3 * g];
                const uint32_t q = T.astc_quint[id];
                bu_put(out, pos, bc, eb[3 * g]);
                bu_put(out, pos + bc, 3, q & 7u);
                bu_put(out, pos + bc + 3, bc, eb[3 * g + 1]);
                bu_put(out, pos + 2 * bc + 3, 2, (q >> 3) & 3u);
                bu_put(out, pos + 2 * bc + 5, bc, eb[3 * g + 2]);
                bu_put(out, pos + 3 * bc + 5, 2, (q >> 5) & 3u);
                pos += 3 * bc + 7;
            }
        } else if constexpr (L::trits) {
            BU_UNROLL
            for (int g = 0; g * 5 < n; g++) {
                // entries past the endpoint count are zero (uastc.rs:623), so partial groups need no special case
                auto tv = [&](int i) -> uint32_t { return i < 18 ? tq[i < 18 ? i : 0] : 0u; };
                auto bv = [&](int i) -> uint32_t { return i < 18 ? eb[i < 18 ? i : 0] : 0u; };
                const uint32_t id = (((tv(5 * g + 4) * 3u + tv(5 * g + 3)) * 3u + tv(5 * g + 2)) * 3u + tv(5 * g + 1)) * 3u + tv(5 * g);
                const uint32_t t = T.astc_trit[id];
                bu_put(out, pos, bc, bv(5 * g));
                bu_put(out, pos + bc, 2, t & 3u);
                bu_put(out, pos + bc + 2, bc, bv(5 * g + 1));
                bu_put(out, pos + 2 * bc + 2, 2, (t >> 2) & 3u);
                bu_put(out, pos + 2 * bc + 4, bc, bv(5 * g + 2));
                bu_put(out, pos + 3 * bc + 4, 1, (t >> 4) & 1u);
                bu_put(out, pos + 3 * bc + 5, bc, bv(5 * g + 3));
                bu_put(out, pos + 4 * bc + 5, 2, (t >> 5) & 3u);
                bu_put(out, pos + 4 * bc + 7, bc, bv(5 * g + 4));
                bu_put(out, pos + 5 * bc + 7, 1, (t >> 7) & 1u);
                pos += 5 * bc + 8;
            }
        } else {
            BU_UNROLL
            for (int i = 0; i < n; i++) bu_put(out, pos + bc * i, bc, eb[i]);
        }

        // ---- weights (astc.rs:143-178) --------------------------------------------------------------
        uint32_t W[3];
        bu_decode_weights<M>(b, pr.uanch, W);
        if constexpr (subsets == 1) {
            // !weight on every field == complement of the whole string
            const uint32_t m = inv[0] ? 0xFFFFFFFFu : 0u;
            constexpr uint32_t last = (L::w_total & 31) ? ((1u << (L::w_total & 31)) - 1u) : 0xFFFFFFFFu;
            W[0] ^= L::w_words == 1 ? (m & last) : m;
            if constexpr (L::w_words > 1) W[1] ^= L::w_words == 2 ? (m & last) : m;
            if constexpr (L::w_words > 2) W[2] ^= m & last;
        } else if constexpr (wb == 3) {  // UASTC mode 2: 48-bit subset masks from the table
            const uint32_t m1lo = T.w3mask_u[pat][0], m1hi = T.w3mask_u[pat][1];
            W[0] ^= (inv[1] ? m1lo : 0u) ^ (inv[0] ? ~m1lo : 0u);
            W[1] ^= ((inv[1] ? m1hi : 0u) ^ (inv[0] ? ~m1hi : 0u)) & 0xFFFFu;
        } else {
            static_assert(subsets == 1 || wb == 3 || wb == 2, "multi-subset modes have 2- or 3-bit weights");
            const uint32_t m1 = (pr.upat & 0x55555555u) * 3u;
            const uint32_t m2 = ((pr.upat >> 1) & 0x55555555u) * 3u;
            const uint32_t m0 = ~(m1 | m2);
            W[0] ^= (inv[0] ? m0 : 0u) ^ (inv[1] ? m1 : 0u) ^ (inv[2] ? m2 : 0u);
        }
        out[3] |= bu_brev(W[0]);
        if constexpr (L::w_words > 1) out[2] |= bu_brev(W[1]);
        if constexpr (L::w_words > 2) out[1] |= bu_brev(W[2]);
        if constexpr (planes == 2) bu_put(out, 128 - L::w_total - 2, 2, compsel);  // CCS, not reversed (astc.rs:174-177)
        return BU_ST_OK;
    }
}
