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
BU_DEV uint32_t bu_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// a * b + c with a, b < 2^24 (v_mad_u32_u24)
BU_DEV uint32_t bu_mad24(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(BU_GCN)
    return __umul24(a, b) + c;
#else
    return a * b + c;
#endif
}
// table reads by byte offset: the kernels keep indices pre-multiplied by the entry size (a shift that never has to be issued)
BU_DEV uint32_t bu_at_u8(const uint8_t* base, uint32_t off) { return base[off]; }
BU_DEV uint32_t bu_at_u16(const uint16_t* base, uint32_t off) { return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(base) + off); }
BU_DEV BuU2 bu_at_u2(const BuU2* base, uint32_t off) { return *reinterpret_cast<const BuU2*>(reinterpret_cast<const uint8_t*>(base) + off); }

// etc.rs:261-275: {value, 0x1D, 0x92, 0x49, 0x24, 0x92, 0x49, 0x24}
BU_DEV void bu_eac_solid(uint32_t out[2], uint32_t value)
{
    out[0] = 0x49921D00u | value;
    out[1] = 0x24499224u;
}

// etc.rs:277-341; px[i] = texel i with its alpha in byte 3.
// The reference takes, per texel, the first minimum of |value_k - a| over the table's eight values (min_by_key).  The values
// are monotone in a fixed order of k -- 3,2,1,0,4,5,6,7, every EAC table has four falling negative and four rising positive
// modifiers -- so the winner is a step function of a: rank = number of thresholds passed, threshold r = the point where
// value[rank r] starts to beat value[rank r-1].  A tie goes to the smaller k, which is the HIGHER rank among the first four and
// the LOWER rank from there on: thresholds 1..3 are ceil(mid), thresholds 4..7 floor(mid) + 1.  Clamped duplicates fall out
// right (equal values at 0 give threshold 0 = always passed, at 255 threshold 256 = never); the only other source of equal
// values is multiplier 0, where every texel takes k = 0.  k = 3 - t1 - t2 - t3 + 4 t4 + t5 + t6 + t7 is accumulated for two
// texels at a time in 16-bit lanes: (a | 0x100) - T has bit 8 set exactly when a >= T (T <= 256), three 2-clock instructions
// per threshold and texel pair instead of eight v_sad_u32 + four v_min3_u32 per texel.
// PAL = 0: every texel goes through the threshold count.  PAL = 4 / 8: the block's alphas come from a palette of that many
// byte entries (the 2-bit-weight modes, bu_block_unpack): the count runs over the palette, and a texel column picks its
// indices up with the same v_perm selector `asel[x]` that interpolated its alpha.
template <int PAL>
BU_DEV void bu_eac_block(const BuTables& T, uint32_t out[2], uint32_t etc2tm, const uint32_t px[16], const uint32_t apal[2], const uint32_t asel[4])
{
    uint32_t mn = px[0], mx = px[0];  // the alpha byte leads the comparison
    BU_UNROLL
    for (int i = 1; i < 16; i++) {
        mn = bu_umin(mn, px[i]);
        mx = bu_umax(mx, px[i]);
    }
    mn >>= 24;
    mx >>= 24;
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
    uint32_t val[8];
    BU_UNROLL
    for (int r = 0; r < 8; r++) val[r] = (uint32_t)bu_clampi(center + T.eac_mods[8 * table + r] * mult, 0, 255);
    uint32_t thr2[8];  // threshold in both 16-bit lanes
    BU_UNROLL
    for (int r = 1; r < 8; r++) thr2[r] = ((val[r - 1] + val[r] + (r < 4 ? 1u : 2u)) >> 1) * 0x10001u;
    const uint32_t keep = mult == 0 ? 0u : 0xFFFFFFFFu;
    // a2 = two alphas in 16-bit lanes -> index of the first << 8 | index of the second << 24
    auto count = [&](uint32_t a2) {
        const uint32_t b1 = a2 | 0x01000100u, b4 = a2 | 0x04000400u;
        uint32_t k = 0x03000300u;
        BU_UNROLL
        for (int r = 1; r < 4; r++) k -= (b1 - thr2[r]) & 0x01000100u;
        k += (b4 - thr2[4]) & 0x04000400u;
        BU_UNROLL
        for (int r = 5; r < 8; r++) k += (b1 - thr2[r]) & 0x01000100u;
        return k & keep;
    };
    uint32_t acc[2] = {0, 0};  // acc[0]: ids 0..7 in bits 8..31 (id 0 on top), acc[1]: ids 8..15
    if constexpr (PAL == 0) {
        BU_UNROLL
        for (int id = 14; id >= 0; id -= 2) {  // column-major ids id, id + 1 = texels (x, y), (x, y + 1)  (etc.rs:324-327)
            const int i = (id % 4) * 4 + id / 4;
            const uint32_t k = count(bu_perm(px[i + 4], px[i], 0x0C070C03u));  // alpha of id | alpha of id + 1 << 16
            // six bits into the top of the accumulator, id first
            acc[id >> 3] = bu_alignbit((k >> 5) | (k >> 24), acc[id >> 3], 6);
        }
    } else {
        uint32_t ipal[2] = {0, 0};  // the index of every palette entry, one per byte
        BU_UNROLL
        for (int h = 0; h < PAL / 4; h++) {
            const uint32_t k01 = count(bu_perm(0u, apal[h], 0x0C010C00u)), k23 = count(bu_perm(0u, apal[h], 0x0C030C02u));
            ipal[h] = bu_perm(k23, k01, 0x07050301u);
        }
        uint32_t c12[4];  // a column's four indices as 12 bits, row 0 on top (ids 4x .. 4x + 3)
        BU_UNROLL
        for (int x = 0; x < 4; x++) {
            const uint32_t ix = bu_perm(ipal[1], ipal[0], asel[x]);
            const uint32_t q = (ix << 3) | (ix >> 8);  // bits 0..5: row 0, row 1; bits 16..21: row 2, row 3
            c12[x] = ((q & 0x3Fu) << 6) | ((q >> 16) & 0x3Fu);
        }
        acc[0] = (c12[0] << 20) | (c12[1] << 8);
        acc[1] = (c12[2] << 20) | (c12[3] << 8);
    }
    // bytes 2..7 of the block are the 48-bit string big-endian: acc[0] bytes 3,2,1 then acc[1] bytes 3,2,1
    out[0] = bu_perm(acc[0], (uint32_t)center | (etc2tm << 8), 0x06070100u);
    out[1] = bu_perm(acc[1], acc[0], 0x05060701u);
}

// the ETC sink: texels as R | G << 8 | B << 16 (| A << 24 when the EAC half needs alpha), and the per-quadrant channel
// sums of etc.rs:97-111 on the way -- R and B in the 16-bit lanes of one word, G in another
template <bool ALPHA>
struct BuSinkEtc {
    uint32_t* px;
    uint32_t qrb[4], qg[4], qbb[4];  // qbb: B alone, only while the column form (cols) is summing
    uint32_t apal[2], asel[4];       // ALPHA: the alpha palette and the columns' selectors into it (bu_eac_block)
    BU_DEVM void alpha_palette(uint32_t lo, uint32_t hi)
    {
        apal[0] = lo;
        apal[1] = hi;
    }
    BU_DEVM void add(int i, uint32_t rb, uint32_t g)
    {
        const int q = ((i >> 3) << 1) | ((i >> 1) & 1);  // row >= 2, col >= 2
        if ((i & 5) == 0) {  // first texel of its quadrant
            qrb[q] = rb;
            qg[q] = g;
        } else {
            qrb[q] += rb;
            qg[q] += g;
        }
    }
    BU_DEVM void word(int, uint32_t) {}  // mode 8 never comes through here (bu_block_etc)
    template <int FMT>
    BU_DEVM void raw(int i, const uint32_t v[4])
    {
        const uint32_t r = v[0], g = FMT == BU_FMT_LA ? v[0] : v[1], bl = FMT == BU_FMT_LA ? v[0] : v[2];
        const uint32_t rb = bu_perm(bl, r, 0x0C060C02u), gg = g >> 16;
        uint32_t w = rb | (gg << 8);
        if constexpr (ALPHA) w = bu_perm(v[FMT == BU_FMT_LA ? 1 : 3], w, 0x06020100u);
        px[i] = w;
        add(i, rb, gg);
    }
    // a column of channel bytes: the sums of its upper and lower texel pair are one v_dot4 each, the texel words a byte transpose
    template <int FMT>
    BU_DEVM void cols(int x, const uint32_t ch[4], uint32_t alpha_sel)
    {
        if constexpr (ALPHA) asel[x] = alpha_sel;
        const int qt = x >> 1, qb = 2 + (x >> 1);  // quadrants of rows 0-1 and rows 2-3
        const bool first = (x & 1) == 0;
        const uint32_t r = ch[0], g = FMT == BU_FMT_LA ? ch[0] : ch[1], bl = FMT == BU_FMT_LA ? ch[0] : ch[2], a = ch[FMT == BU_FMT_LA ? 1 : 3];
        if constexpr (FMT == BU_FMT_LA) {
            qg[qt] = bu_udot4(g, 0x00000101u, first ? 0u : qg[qt]);
            qg[qb] = bu_udot4(g, 0x01010000u, first ? 0u : qg[qb]);
            if (!first) {
                qrb[qt] = qg[qt] * 0x10001u;
                qrb[qb] = qg[qb] * 0x10001u;
            }
        } else {
            // R in the low lane, B in the high lane: the weight word of B carries the 16-bit shift
            qrb[qt] = bu_udot4(r, 0x00000101u, first ? 0u : qrb[qt]);
            qrb[qb] = bu_udot4(r, 0x01010000u, first ? 0u : qrb[qb]);
            qbb[qt] = bu_udot4(bl, 0x00000101u, first ? 0u : qbb[qt]);
            qbb[qb] = bu_udot4(bl, 0x01010000u, first ? 0u : qbb[qb]);
            qg[qt] = bu_udot4(g, 0x00000101u, first ? 0u : qg[qt]);
            qg[qb] = bu_udot4(g, 0x01010000u, first ? 0u : qg[qb]);
            if (!first) {
                qrb[qt] |= qbb[qt] << 16;
                qrb[qb] |= qbb[qb] << 16;
            }
        }
        if constexpr (FMT == BU_FMT_LA) {
            px[x] = bu_perm(a, r, ALPHA ? 0x04000000u : 0x0C000000u);
            px[4 + x] = bu_perm(a, r, ALPHA ? 0x05010101u : 0x0C010101u);
            px[8 + x] = bu_perm(a, r, ALPHA ? 0x06020202u : 0x0C020202u);
            px[12 + x] = bu_perm(a, r, ALPHA ? 0x07030303u : 0x0C030303u);
        } else {
            const uint32_t t01 = bu_perm(g, r, 0x05010400u), t23 = bu_perm(g, r, 0x07030602u);  // R0 G0 R1 G1 / R2 G2 R3 G3
            if constexpr (ALPHA) {
                const uint32_t u01 = bu_perm(a, bl, 0x05010400u), u23 = bu_perm(a, bl, 0x07030602u);
                px[x] = bu_perm(u01, t01, 0x05040100u);
                px[4 + x] = bu_perm(u01, t01, 0x07060302u);
                px[8 + x] = bu_perm(u23, t23, 0x05040100u);
                px[12 + x] = bu_perm(u23, t23, 0x07060302u);
            } else {
                px[x] = bu_perm(bl, t01, 0x0C040100u);
                px[4 + x] = bu_perm(bl, t01, 0x0C050302u);
                px[8 + x] = bu_perm(bl, t23, 0x0C060100u);
                px[12 + x] = bu_perm(bl, t23, 0x0C070302u);
            }
        }
    }
};

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
        constexpr bool ALPHA = ETC2 && L::has_alpha;
        uint32_t px[16];
        BuSinkEtc<ALPHA> sink;
        sink.px = px;
        const int st = bu_block_unpack<M>(T, b, sink);
        if (st) return st;
        if constexpr (ETC2) {
            if constexpr (L::has_alpha) bu_eac_block<L::alpha_palette>(T, out, bu_bits(b, L::pos_etc2tm, 8), px, sink.apal, sink.asel);
            else bu_eac_solid(out, 255);  // etc2tm = 0 for RGB modes (uastc.rs:430-434)
        }
        // the eight flag bits flip, diff, inten0, inten1 are adjacent in every mode: one table read gives all their uses
        const uint32_t raw8 = bu_bits(b, L::pos_etc1f, 8);
        const BuU4 fl = T.etc1_flags[raw8];
        const bool f = (raw8 & 1u) != 0;
        // flip (etc1f) : halves are rows 0-1 / 2-3; otherwise columns 0-1 / 2-3 (etc.rs:86-95).  The transpose is the choice
        // of which off-diagonal quadrant joins which half.
        const uint32_t* qrb = sink.qrb;
        const uint32_t* qg = sink.qg;
        const uint32_t srb0 = qrb[0] + (f ? qrb[1] : qrb[2]), srb1 = qrb[3] + (f ? qrb[2] : qrb[1]);
        const uint32_t sg0 = qg[0] + (f ? qg[1] : qg[2]), sg1 = qg[3] + (f ? qg[2] : qg[1]);
        // (sum*limit + 1020) / 2040  (etc.rs:109) as one multiply-add and a shift (BU_Q_M); the value stays in bits 26..30
        const uint32_t sums[2][3] = {{srb0 & 0xFFFFu, sg0, srb0 >> 16}, {srb1 & 0xFFFFu, sg1, srb1 >> 16}};
        uint32_t x[2][3];
        BU_UNROLL
        for (int sb = 0; sb < 2; sb++)
            BU_UNROLL
            for (int ch = 0; ch < 3; ch++) x[sb][ch] = bu_mad24(sums[sb][ch], fl.w, 1020u * BU_Q_M);
        // e0 = 8 * (base colour of half 0), e1 = 2 * (half 1): the scalings the next two tables are indexed with
        uint32_t e0[3], e1[3];
        if constexpr (!L::m1012) {
            // apply_etc1_bias (etc.rs:203-259): the six per-channel adjustments are one LUT read each
            const BuU2 bias = T.etc1_bias2[bu_bits(b, L::pos_etc1bias, 5)];
            const uint32_t w0 = bias.x | fl.y, w1 = bias.y | fl.y;
            BU_UNROLL
            for (int ch = 0; ch < 3; ch++) {
                e0[ch] = bu_at_u8(T.etc1_biasv0, ((w0 >> (8 * ch)) & 0xE0u) | (x[0][ch] >> 26));
                e1[ch] = bu_at_u8(T.etc1_biasv1, ((w1 >> (8 * ch)) & 0xE0u) | (x[1][ch] >> 26));
            }
        } else {
            BU_UNROLL
            for (int ch = 0; ch < 3; ch++) {
                e0[ch] = (x[0][ch] >> 23) & 0xF8u;
                e1[ch] = (x[1][ch] >> 25) & 0x3Eu;
            }
        }
        // header bytes, and the base colour the second half actually decodes with (etc.rs:113-158): etc1_hdr
        uint32_t h[3];
        BU_UNROLL
        for (int ch = 0; ch < 3; ch++) h[ch] = bu_at_u16(T.etc1_hdr, (e0[ch] << 3) + (fl.z + e1[ch]));
        col[0] = bu_perm(h[1], h[0], 0x0C0C0400u) | bu_perm(fl.x, h[2], 0x07000C0Cu);

        // luma thresholds per half (etc.rs:165-177).  The factors (108, 366, 38) are all even, so every luma is even:
        // with lum = 2*L (L = 54r + 183g + 19b) the reference's test lum >= (lum_a + lum_b)/2 is exactly
        // L >= (L_a + L_b + 1) >> 1.  etc1_thr holds each channel's share of -(L0 + L1), L2 - L0 and L3 - L1 for the four
        // modified base colours of a half.  Kept NEGATED: v_dot4_u32_u8 adds an accumulator for free, so
        // dot4(texel, LW, -thr) = luma - thr in one instruction and bit 31 of the result is the comparison.
        uint32_t nthr[2][3];
        BU_UNROLL
        for (int sb = 0; sb < 2; sb++) {
            const uint32_t row = sb ? (fl.x >> 8) & 0xF00u : fl.x & 0xF00u;
            BuU2 t[3];
            BU_UNROLL
            for (int ch = 0; ch < 3; ch++) t[ch] = bu_at_u2(T.etc1_thr[ch], row | (sb ? h[ch] >> 8 : e0[ch]));
            const uint32_t n01 = t[0].x + t[1].x + t[2].x, dd = t[0].y + t[1].y + t[2].y;  // dd: two sums < 2^16, no carry between them
            const uint32_t n12 = n01 - (dd & 0xFFFFu), n23 = n12 - (dd >> 16);
            // -((S + 1) >> 1) = floor(-S / 2)
            nthr[sb][0] = (uint32_t)((int32_t)n01 >> 1);
            nthr[sb][1] = (uint32_t)((int32_t)n12 >> 1);
            nthr[sb][2] = (uint32_t)((int32_t)n23 >> 1);
        }
        // sel = #thresholds <= luma; ETC1 code [3,2,0,1][sel]: high bit = sel < 2 = (luma < thr1), low bit = sel == 0 or
        // sel == 3 = NOT (luma < thr0 xor luma < thr2) (the thresholds are monotone).  Two texels per instruction from the luma
        // on: rel = luma - thr1, saturated to i16 (v_cvt_pk_i16_i32), keeps its sign, and the outer thresholds are at most
        // (luma(+183) - luma(-47)) / 2 + 1 = 29 441 away from the middle one, so rel + (thr1 - thr0) and rel + (thr1 - thr2) with
        // signed saturation (v_pk_add_i16 clamp) have the signs of luma - thr0 and luma - thr2 whatever was cut off.
        // A word pairs pixel id k (x < 2) with k + 8 (x + 2, same row): the sign bytes of rel and of the xnor are the bits of
        // pixel k / k + 8 in the high / low plane, one v_perm lines the four up in the block's byte order (ids 8..15 of the high
        // plane first, etc.rs:376-392) and a shift + v_bfi per pair files them under the bits already there -- 70 instructions
        // where round 2 spent 96 (two dot4, add, xor, two v_alignbit per texel).
        // The top-right and bottom-left 2x2 quadrants change half with the flip bit: a row pair's two lanes take their
        // thresholds from half 0 | half 0 (flipped rows 0-1), half 1 | half 1 (flipped rows 2-3) or half 0 | half 1.
        constexpr uint32_t LW = 54u | (183u << 8) | (19u << 16);
        const uint32_t da0 = nthr[0][0] - nthr[0][1], da1 = nthr[1][0] - nthr[1][1];  // thr1 - thr0 >= 0
        const uint32_t db0 = nthr[0][2] - nthr[0][1], db1 = nthr[1][2] - nthr[1][1];  // thr1 - thr2 <= 0
        const uint32_t da01 = bu_perm(da1, da0, 0x05040100u), db01 = bu_perm(db1, db0, 0x05040100u);
        const uint32_t dpa[2] = {f ? bu_perm(da0, da0, 0x05040100u) : da01, f ? bu_perm(da1, da1, 0x05040100u) : da01};
        const uint32_t dpb[2] = {f ? bu_perm(db0, db0, 0x05040100u) : db01, f ? bu_perm(db1, db1, 0x05040100u) : db01};
        const uint32_t n1q[4] = {nthr[0][1], f ? nthr[0][1] : nthr[1][1], f ? nthr[1][1] : nthr[0][1], nthr[1][1]};  // by quadrant
        uint32_t planes = 0;
        BU_UNROLL
        for (int k = 0; k < 8; k++) {
            const int xx = k >> 2, y = k & 3, g = y >> 1;
            const uint32_t rel = bu_cvt_pk_i16((int32_t)bu_udot4(px[y * 4 + xx], LW, n1q[2 * g]), (int32_t)bu_udot4(px[y * 4 + xx + 2], LW, n1q[2 * g + 1]));
            const uint32_t mid = ~(bu_pk_add_i16_sat(rel, dpa[g]) ^ bu_pk_add_i16_sat(rel, dpb[g]));  // sign = NOT (lt0 xor lt2)
            const uint32_t four = bu_perm(mid, rel, 0x05070103u);  // bit 7 of: high plane id k + 8, id k, low plane id k + 8, id k
            planes = k == 0 ? four : bu_bfi(0x80808080u, four, planes >> 1);
        }
        col[1] = planes;  // bit j of byte 0 / 1 / 2 / 3 = high plane id 8 + j / id j / low plane id 8 + j / id j
        return BU_ST_OK;
    }
}
