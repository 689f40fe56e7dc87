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
