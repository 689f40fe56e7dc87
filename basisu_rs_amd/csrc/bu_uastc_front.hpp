// UASTC block front-end for the gfx950 kernels: mode layout, field extraction, endpoint and weight
// decode.  Replaces (as register bit-field code, one lane per block) the reference's
//   src/uastc.rs:329-441   decode_mode / trans flags / compsel / pattern index
//   src/uastc.rs:585-695   decode_endpoints + unquant_endpoint
//   src/uastc.rs:721-740   decode_weights
//   src/uastc.rs:176-327   assemble_endpoint_pairs / astc_interpolate / decode_block_to_rgba
//   src/bitreader.rs:3-61  (all field positions are compile-time per mode, so there is no cursor)
//
// Everything is a template on the UASTC mode: a mode fixes every field position, so after inlining
// the code for one mode is straight-line bit-field extraction with immediate operands.  The same
// source compiles as plain C++ (BU_HOST_EMUL, tests only) so the exact kernel logic can be checked
// against the oracle on a machine without a GPU.
#pragma once
#include "bu_tables_dev.hpp"

// Under hipcc the per-block code is host + device: the device pass (BU_GCN) maps the helpers below to gfx950 instructions, the host
// pass compiles their portable forms -- that is the code the per-block API of the C ABI runs on the CPU (bu_capi_slice.hpp:
// one 16-byte block is not worth a kernel launch, lib.rs:29-53).
#if defined(__HIPCC__)
#define BU_DEV __host__ __device__ __forceinline__
#define BU_DEVM __host__ __device__ __forceinline__  // member functions
#define BU_UNROLL _Pragma("unroll")
#if defined(__HIP_DEVICE_COMPILE__)
#define BU_GCN 1
#endif
#else
#define BU_DEV static inline __attribute__((always_inline))
#define BU_DEVM inline __attribute__((always_inline))
#define BU_UNROLL _Pragma("GCC unroll 32")
#endif

enum { BU_ST_OK = 0, BU_ST_BAD_MODE = 1, BU_ST_BAD_PATTERN = 2 };
enum { BU_FMT_RGB = 0, BU_FMT_RGBA = 1, BU_FMT_LA = 2 };

struct BuBlk {
    uint32_t w[4];
};

BU_DEV uint32_t bu_popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
BU_DEV uint32_t bu_brev(uint32_t v)
{
#if defined(__HIPCC__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

// n bits at bit position pos of the 128-bit block; bits past the end read as 0 (bitreader.rs:45,55).
// pos and n are compile-time constants at every call site after unrolling, so this folds to one
// shift/alignbit + and.
BU_DEV uint32_t bu_bits(const BuBlk& b, int pos, int n)
{
    if (n <= 0 || pos >= 128) return 0;
    const int wi = pos >> 5, sh = pos & 31;
    const uint32_t lo = b.w[wi];
    const uint32_t hi = (wi < 3) ? b.w[wi < 3 ? wi + 1 : 3] : 0u;
    uint32_t v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
    return n < 32 ? (v & ((1u << n) - 1u)) : v;
}

// OR an n-bit value into a zeroed 128-bit output at a compile-time position (bitwriter.rs:23-51).
// v must already be < 2^n.
BU_DEV void bu_put(uint32_t out[4], int pos, int n, uint32_t v)
{
    if (n <= 0 || pos >= 128) return;
    const int wi = pos >> 5, sh = pos & 31;
    out[wi] |= v << sh;
    if (sh + n > 32 && wi < 3) out[wi < 3 ? wi + 1 : 3] |= v >> (32 - sh);
}

// ------------------------------------------------------------------------------------------------
// Mode facts (uastc.rs:528-557) and the field layout they imply.
struct BuModeDesc {
    int code_size, range, fmt, wb, planes, subsets, tf_bits;
};
constexpr BuModeDesc BU_MODE_DESC[19] = {
    {4, 19, BU_FMT_RGB, 4, 1, 1, 15},  {6, 20, BU_FMT_RGB, 2, 1, 1, 15},  {5, 8, BU_FMT_RGB, 3, 1, 2, 15},
    {5, 7, BU_FMT_RGB, 2, 1, 3, 15},   {5, 12, BU_FMT_RGB, 2, 1, 2, 15},  {5, 20, BU_FMT_RGB, 3, 1, 1, 15},
    {5, 18, BU_FMT_RGB, 2, 2, 1, 15},  {5, 12, BU_FMT_RGB, 2, 1, 2, 15},  {5, 0, BU_FMT_RGBA, 0, 1, 1, 0},
    {5, 8, BU_FMT_RGBA, 2, 1, 2, 23},  {3, 13, BU_FMT_RGBA, 4, 1, 1, 17}, {2, 13, BU_FMT_RGBA, 2, 2, 1, 17},
    {3, 19, BU_FMT_RGBA, 3, 1, 1, 17}, {5, 20, BU_FMT_RGBA, 1, 2, 1, 23}, {5, 20, BU_FMT_RGBA, 2, 1, 1, 23},
    {7, 20, BU_FMT_LA, 4, 1, 1, 23},   {6, 20, BU_FMT_LA, 2, 1, 2, 23},   {6, 20, BU_FMT_LA, 2, 2, 1, 23},
    {4, 11, BU_FMT_RGB, 5, 1, 1, 15},
};

// BISE range facts for the ranges UASTC uses (astc.rs:309-331)
constexpr int bu_bise_bits(int r) { return r == 7 ? 2 : r == 8 ? 4 : r == 11 ? 5 : r == 12 ? 3 : r == 13 ? 4 : r == 18 ? 5 : r == 19 ? 6 : r == 20 ? 8 : 0; }
constexpr bool bu_bise_trits(int r) { return r == 7 || r == 13 || r == 19; }
constexpr bool bu_bise_quints(int r) { return r == 12 || r == 18; }

template <int M>
struct BuLayout {
    static constexpr BuModeDesc d = BU_MODE_DESC[M];
    static constexpr int channels = d.fmt == BU_FMT_RGB ? 3 : d.fmt == BU_FMT_RGBA ? 4 : 2;
    static constexpr int ep_count = channels * d.subsets * 2;
    static constexpr int ebits = bu_bise_bits(d.range);
    static constexpr bool trits = bu_bise_trits(d.range);
    static constexpr bool quints = bu_bise_quints(d.range);
    static constexpr bool m1012 = (M >= 10 && M <= 12);
    static constexpr bool has_alpha = d.fmt != BU_FMT_RGB;
    // transcoding flags (uastc.rs:411-436), positions relative to the block
    static constexpr int pos_tf = d.code_size;
    static constexpr int pos_etc1f = pos_tf + (m1012 ? 1 : 2);
    static constexpr int pos_etc1d = pos_etc1f + 1;
    static constexpr int pos_etc1i0 = pos_etc1d + 1;
    static constexpr int pos_etc1i1 = pos_etc1i0 + 3;
    static constexpr int pos_etc1bias = pos_etc1i1 + 3;            // 5 bits, absent for modes 10-12
    static constexpr int pos_etc2tm = pos_etc1bias + (m1012 ? 0 : 5);  // 8 bits if has_alpha
    static constexpr int pos_compsel = pos_tf + d.tf_bits;
    static constexpr int compsel_bits = (d.planes == 2 && d.fmt != BU_FMT_LA) ? 2 : 0;
    static constexpr int pos_pat = pos_compsel + compsel_bits;
    static constexpr int pat_bits = (M == 7 || d.subsets == 2) ? 5 : (d.subsets == 3 ? 4 : 0);
    static constexpr int pat_count = M == 7 ? 19 : (d.subsets == 2 ? 30 : (d.subsets == 3 ? 11 : 1));
    static constexpr int part_base = M == 7 ? BU_PART_BASE23 : (d.subsets == 2 ? BU_PART_BASE2 : (d.subsets == 3 ? BU_PART_BASE3 : BU_PART_MODE1));
    static constexpr int pos_ep = pos_pat + pat_bits;
    static constexpr int tq_rem = quints ? ep_count % 3 : (trits ? ep_count % 5 : 0);
    static constexpr int tq_full = quints ? ep_count / 3 : (trits ? ep_count / 5 : 0);
    static constexpr int tq_rem_bits = quints ? (tq_rem == 1 ? 3 : tq_rem == 2 ? 5 : 0)
                                              : (trits ? (tq_rem == 1 ? 2 : tq_rem == 2 ? 4 : tq_rem == 3 ? 5 : tq_rem == 4 ? 7 : 0) : 0);
    static constexpr int tq_bits = tq_full * (quints ? 7 : 8) + tq_rem_bits;
    static constexpr int pos_epbits = pos_ep + tq_bits;
    static constexpr int pos_w = pos_epbits + ep_count * ebits;
    static constexpr int n_anch = d.subsets == 1 ? d.planes : d.subsets;  // weight MSBs that are not stored
    static constexpr int w_total = 16 * d.planes * d.wb;                   // bits after regularisation
    static constexpr int w_raw = w_total - n_anch;
    static constexpr int w_words = (w_total + 31) / 32;
    static_assert(M == 8 || pos_w + w_raw <= 128, "mode does not fit in 128 bits");
    // texel unpack through byte palettes (bu_block_unpack): entries of the alpha palette, 0 = the mode takes the generic path
    static constexpr int alpha_palette = (M != 8 && has_alpha && d.wb <= 2 && (d.planes == 2 || d.wb == 2)) ? (d.subsets == 2 ? 8 : 4)
                                         : (M != 8 && has_alpha && d.wb == 3 && d.planes == 1 && d.subsets == 1) ? 8 : 0;  // (mode 12)
};

// ------------------------------------------------------------------------------------------------
// Quantised endpoints (uastc.rs:616-695): tq[i] = trit/quint digit, eb[i] = plain bits.
template <int M>
BU_DEV void bu_decode_quant(const BuTables& T, const BuBlk& b, uint32_t tq[18], uint32_t eb[18])
{
    using L = BuLayout<M>;
    BU_UNROLL
    for (int i = 0; i < 18; i++) {
        tq[i] = 0;
        eb[i] = 0;
    }
    if constexpr (L::quints) {
        BU_UNROLL
        for (int g = 0; g < L::tq_full; g++) {
            const uint32_t dg = T.quint3[bu_bits(b, L::pos_ep + 7 * g, 7)];
            BU_UNROLL
            for (int k = 0; k < 3; k++) tq[3 * g + k] = (dg >> (3 * k)) & 7u;
        }
        if constexpr (L::tq_rem > 0) {
            const uint32_t dg = T.quint3[bu_bits(b, L::pos_ep + 7 * L::tq_full, L::tq_rem_bits)];
            BU_UNROLL
            for (int k = 0; k < L::tq_rem; k++) tq[3 * L::tq_full + k] = (dg >> (3 * k)) & 7u;
        }
    }
    if constexpr (L::trits) {
        BU_UNROLL
        for (int g = 0; g < L::tq_full; g++) {
            const uint32_t dg = T.trit5[bu_bits(b, L::pos_ep + 8 * g, 8)];
            BU_UNROLL
            for (int k = 0; k < 5; k++) tq[5 * g + k] = (dg >> (2 * k)) & 3u;
        }
        if constexpr (L::tq_rem > 0) {
            const uint32_t dg = T.trit5[bu_bits(b, L::pos_ep + 8 * L::tq_full, L::tq_rem_bits)];
            BU_UNROLL
            for (int k = 0; k < L::tq_rem; k++) tq[5 * L::tq_full + k] = (dg >> (2 * k)) & 3u;
        }
    }
    BU_UNROLL
    for (int i = 0; i < L::ep_count; i++) eb[i] = bu_bits(b, L::pos_epbits + L::ebits * i, L::ebits);
}

// Dequantised endpoint (uastc.rs:585-614): bit replication for the plain-bit ranges, LUT otherwise.
template <int RANGE>
BU_DEV uint32_t bu_deq(const BuTables& T, uint32_t tq, uint32_t eb)
{
    if constexpr (RANGE == 20) return eb;
    else if constexpr (RANGE == 8) return eb * 17u;
    else if constexpr (RANGE == 11) return (eb << 3) | (eb >> 2);
    else return T.deq[bu_deq_ofs(RANGE) + ((tq << bu_bise_bits(RANGE)) | eb)];
}

template <int M>
BU_DEV void bu_decode_endpoints(const BuTables& T, const BuBlk& b, uint32_t e[18])
{
    using L = BuLayout<M>;
    uint32_t tq[18], eb[18];
    bu_decode_quant<M>(T, b, tq, eb);
    BU_UNROLL
    for (int i = 0; i < 18; i++) e[i] = i < L::ep_count ? bu_deq<L::d.range>(T, tq[i], eb[i]) : 0u;
}

// ------------------------------------------------------------------------------------------------
// Weights (uastc.rs:721-740).  The stored stream drops the MSB of one anchor weight per subset (per
// plane).  We re-insert those zero bits so that texel i / plane p sits at bit (i*planes+p)*wb of a
// regular bit string W (up to 80 bits in 3 words): every later step is then plain SWAR.
BU_DEV void bu_ins0_static(uint32_t W[3], int nw, int q)  // nw, q compile-time
{
    const int k = q >> 5, s = q & 31;
    for (int j = nw - 1; j > k; j--) W[j] = (W[j] << 1) | (W[j - 1] >> 31);
    const uint32_t low = (1u << s) - 1u;
    W[k] = (W[k] & low) | ((W[k] & ~low) << 1);
}
BU_DEV uint32_t bu_ins0_rt32(uint32_t x, uint32_t q) { return x + (x & (0xFFFFFFFFu << q)); }
BU_DEV void bu_ins0_rt64(uint32_t W[3], uint32_t q)
{
    uint64_t x = (uint64_t)W[0] | ((uint64_t)W[1] << 32);
    x = x + (x & (~0ull << q));
    W[0] = (uint32_t)x;
    W[1] = (uint32_t)(x >> 32);
}

// uanch: UASTC anchor texels, nibble s = anchor of subset s (multi-subset modes only)
template <int M>
BU_DEV void bu_decode_weights(const BuBlk& b, uint32_t uanch, uint32_t W[3])
{
    using L = BuLayout<M>;
    constexpr int wb = L::d.wb;
    W[0] = bu_bits(b, L::pos_w, L::w_raw < 32 ? L::w_raw : 32);
    W[1] = L::w_raw > 32 ? bu_bits(b, L::pos_w + 32, L::w_raw - 32 < 32 ? L::w_raw - 32 : 32) : 0u;
    W[2] = L::w_raw > 64 ? bu_bits(b, L::pos_w + 64, L::w_raw - 64) : 0u;
    // texel 0 is an anchor in every pattern (uastc.rs:792-811 always list a 0)
    bu_ins0_static(W, L::w_words, wb - 1);
    if constexpr (L::d.planes == 2) bu_ins0_static(W, L::w_words, 2 * wb - 1);
    if constexpr (L::d.subsets == 2) {
        const uint32_t a = (uanch | (uanch >> 4)) & 15u;  // the non-zero one of the two
        if constexpr (L::w_words == 1) W[0] = bu_ins0_rt32(W[0], a * wb + (wb - 1));
        else bu_ins0_rt64(W, a * wb + (wb - 1));
    }
    if constexpr (L::d.subsets == 3) {
        const uint32_t x = uanch & 15u, y = (uanch >> 4) & 15u, z = (uanch >> 8) & 15u;
        uint32_t hi = x > y ? x : y;
        hi = hi > z ? hi : z;
        const uint32_t lo = x + y + z - hi;
        static_assert(L::d.subsets != 3 || L::w_words == 1, "3-subset modes have 2-bit weights");
        W[0] = bu_ins0_rt32(W[0], lo * wb + (wb - 1));
        W[0] = bu_ins0_rt32(W[0], hi * wb + (wb - 1));
    }
}

// raw weight k (k = texel*planes + plane) out of the regular string; k compile-time
template <int WB>
BU_DEV uint32_t bu_wfield(const uint32_t W[3], int k)
{
    const int pos = k * WB, wi = pos >> 5, sh = pos & 31;
    uint32_t v = W[wi] >> sh;
    if (sh + WB > 32) v |= W[wi + 1] << (32 - sh);
    return v & ((1u << WB) - 1u);
}

// uastc.rs:218-235 as one dot product.  With lo16 = l*257, hi16 = h*257 packed in a word and the
// weights scaled by 4, ((lo16*(64-w) + hi16*w + 32) >> 6) >> 8 is byte 2 of
//   lo16*(256-4w) + hi16*4w + 128      (v_dot2_u32_u16, max 65535*256+128 < 2^32)
BU_DEV uint32_t bu_udot2(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(BU_GCN)
    typedef unsigned short bu_us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(bu_us2, a), __builtin_bit_cast(bu_us2, b), c, false);
#else
    return (a & 0xFFFFu) * (b & 0xFFFFu) + (a >> 16) * (b >> 16) + c;
#endif
}
// 4 x u8 dot product + accumulator (v_dot4_u32_u8)
BU_DEV uint32_t bu_udot4(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(BU_GCN)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    uint32_t r = c;
    for (int k = 0; k < 4; k++) r += ((a >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
    return r;
#endif
}
BU_DEV uint32_t bu_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// ({hi, lo} >> sh) & 0xFFFFFFFF for 0 < sh < 32 (v_alignbit_b32)
BU_DEV uint32_t bu_alignbit(uint32_t hi, uint32_t lo, int sh)
{
#if defined(BU_GCN)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
// bytes of a:b selected into one word (v_perm_b32): sel byte k picks byte (sel>>8k)&7 of {b (0-3), a (4-7)}, 0x0C = zero, 0x0D.. = 0xFF
BU_DEV uint32_t bu_perm(uint32_t a, uint32_t b, uint32_t sel)
{
#if defined(BU_GCN)
    return __builtin_amdgcn_perm(a, b, sel);
#else
    const uint64_t v = ((uint64_t)a << 32) | b;
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t sk = (sel >> (8 * k)) & 0xFFu;
        const uint32_t byte = sk < 8 ? (uint32_t)((v >> (8 * sk)) & 0xFFu) : (sk >= 0x0Du ? 0xFFu : 0u);  // 8..11 (sign fills) are not used
        r |= byte << (8 * k);
    }
    return r;
#endif
}

// two i32 -> two i16 with signed saturation, the first in the low half (v_cvt_pk_i16_i32)
BU_DEV uint32_t bu_cvt_pk_i16(int32_t lo, int32_t hi)
{
#if defined(BU_GCN)
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(lo, hi));
#else
    const int32_t a = lo < -32768 ? -32768 : (lo > 32767 ? 32767 : lo), b = hi < -32768 ? -32768 : (hi > 32767 ? 32767 : hi);
    return ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
#endif
}
// i16 lanes, a + b with signed saturation (v_pk_add_i16 ... clamp)
BU_DEV uint32_t bu_pk_add_i16_sat(uint32_t a, uint32_t b)
{
#if defined(BU_GCN)
    typedef short bu_s2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(bu_s2, a), __builtin_bit_cast(bu_s2, b)));
#else
    uint32_t r = 0;
    for (int k = 0; k < 2; k++) {
        int32_t v = (int32_t)(int16_t)(a >> (16 * k)) + (int32_t)(int16_t)(b >> (16 * k));
        v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
        r |= ((uint32_t)v & 0xFFFFu) << (16 * k);
    }
    return r;
#endif
}

// (a & m) | (b & ~m)  (v_bfi_b32)
BU_DEV uint32_t bu_bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

// ------------------------------------------------------------------------------------------------
// Texel unpack (uastc.rs:237-327; color.rs:22-24).  The interpolation leaves every channel in byte 2 of its own word; what
// happens to those words is the caller's business (a "sink"): RGBA32 gathers them into R,G,B,A bytes, the ETC path wants
// R and B in 16-bit lanes for its sums.  A sink provides
//   raw<FMT>(i, v)   v[0..2] = R,G,B (FMT RGB, alpha is 255), v[0..3] = R,G,B,A (RGBA), v[0..1] = L,A (LA), value = byte 2, byte 3 = 0
//   word(i, px)      an assembled texel R | G << 8 | B << 16 | A << 24 (mode 8)
//   cols<FMT>(x, ch, asel)  one block COLUMN: ch[c] = channel c (as in raw) of texels (x, 0..3) in bytes 0..3; asel = the v_perm
//                    selector that read the alpha channel out of alpha_palette(lo, hi)  (2-bit weights, below)
// with i = row-major texel index, compile-time after unrolling.
//
// Two-bit weights (modes 1, 3, 4, 7, 9, 14, 16; with two planes 6, 11, 17, and 13 with one-bit weights; three-bit weights: further down): a channel takes one of four values per subset, and the weights
// of a column are already byte-aligned -- texel (x, y) sits at bits 8y + 2x of the weight word, so (W >> 2x) & 0x03030303 is
// the column's four weights, one per byte.  That is exactly a v_perm_b32 selector: with the four values of a channel in the
// bytes of one register (two subsets: eight values in a register pair, selector |= subset << 2) ONE instruction
// interpolates the channel for four texels.  The palette costs two v_dot2 per channel and subset (the outer two entries are
// the endpoints themselves); per texel this path issues about four instructions where the generic one issues nine to twenty.
template <int M, class SINK>
BU_DEV int bu_block_unpack(const BuTables& T, const BuBlk& b, SINK& sink)
{
    using L = BuLayout<M>;
    if constexpr (M == 8) {
        const uint32_t c = bu_bits(b, 5, 32);  // R,G,B,A bytes right after the 5-bit code (uastc.rs:387-394)
        BU_UNROLL
        for (int i = 0; i < 16; i++) sink.word(i, c);
        return BU_ST_OK;
    } else {
        constexpr int wb = L::d.wb, planes = L::d.planes, subsets = L::d.subsets, fmt = L::d.fmt;
        uint32_t pat = 0, upat = 0, uanch = 0;
        if constexpr (L::pat_bits > 0) {
            pat = bu_bits(b, L::pos_pat, L::pat_bits);
            if (pat >= (uint32_t)L::pat_count) return BU_ST_BAD_PATTERN;  // uastc.rs:360-365
            upat = T.part[L::part_base + pat].upat;
            uanch = T.part[L::part_base + pat].uanch;
        }
        const uint32_t compsel = L::compsel_bits ? bu_bits(b, L::pos_compsel, 2) : 3u;  // uastc.rs:343-350

        uint32_t e[18];
        bu_decode_endpoints<M>(T, b, e);
        uint32_t W[3];
        bu_decode_weights<M>(b, uanch, W);

        // per subset, per interpolated channel: lo*257 | (hi*257) << 16 (uastc.rs:176-216 gives the channel order)
        constexpr int NC = fmt == BU_FMT_RGB ? 3 : (fmt == BU_FMT_RGBA ? 4 : 2);  // distinct interpolations
        uint32_t A[3][4];
        BU_UNROLL
        for (int s = 0; s < subsets; s++) {
            BU_UNROLL
            for (int c = 0; c < NC; c++) {
                const uint32_t lo = e[(2 * L::channels) * s + 2 * c], hi = e[(2 * L::channels) * s + 2 * c + 1];
                A[s][c] = (lo * 257u) | ((hi * 257u) << 16);
            }
        }
        uint32_t X1[4], X2[4];  // xor deltas for the branch-free subset blend
        if constexpr (subsets == 3) {
            BU_UNROLL
            for (int c = 0; c < NC; c++) {
                X1[c] = A[0][c] ^ A[1][c];
                X2[c] = A[0][c] ^ A[2][c];
            }
        }
        if constexpr (planes == 2 && wb <= 2) {
            // Dual plane with 1- or 2-bit weights (modes 6, 11, 13, 17): the palette form again.  The two planes' weights
            // interleave (texel i, plane p = field 2i + p), so a column's selector is two masked shifts and -- for 2-bit
            // weights, where a block row is 16 bits -- a v_perm that brings rows 0..3 back into byte order.  The channel
            // named by compsel reads the plane-1 selector (uastc.rs:293-296); LA always interpolates alpha on plane 1.
            constexpr uint32_t BW1 = 21u * 0x3FFFCu + 256u, BW2 = 43u * 0x3FFFCu + 256u;
            uint32_t pal[4];
            BU_UNROLL
            for (int c = 0; c < NC; c++) {
                const uint32_t lo = e[2 * c], hi = e[2 * c + 1];
                if constexpr (wb == 1) {
                    pal[c] = lo | (hi << 8);
                } else {
                    const uint32_t v1 = bu_udot2(A[0][c], BW1, 128u), v2 = bu_udot2(A[0][c], BW2, 128u);
                    pal[c] = bu_perm(v2, v1, 0x0C06020Cu) | lo | (hi << 24);
                }
            }
            if constexpr (L::has_alpha) sink.alpha_palette(pal[NC - 1], 0u);
            BU_UNROLL
            for (int x = 0; x < 4; x++) {
                uint32_t sel[2];
                BU_UNROLL
                for (int p2 = 0; p2 < 2; p2++) {
                    if constexpr (wb == 1) {
                        sel[p2] = (W[0] >> (2 * x + p2)) & 0x01010101u;
                    } else {
                        const uint32_t t0 = (W[0] >> (4 * x + 2 * p2)) & 0x00030003u, t1 = (W[1] >> (4 * x + 2 * p2)) & 0x00030003u;  // rows 0,1 / 2,3 in bytes 0 and 2
                        sel[p2] = bu_perm(t1, t0, 0x06040200u);
                    }
                }
                uint32_t ch[4] = {0, 0, 0, 0}, sc[4] = {0, 0, 0, 0};
                BU_UNROLL
                for (int c = 0; c < NC; c++) {
                    sc[c] = fmt == BU_FMT_LA ? sel[c] : (compsel == (uint32_t)c ? sel[1] : sel[0]);
                    ch[c] = bu_perm(0u, pal[c], sc[c]);
                }
                sink.template cols<fmt>(x, ch, sc[NC - 1]);
            }
            return BU_ST_OK;
        }
        // (every dual-plane mode has 1- or 2-bit weights: from here on there is one plane)
        if constexpr (wb == 2 && planes == 1) {
            // weights 1 and 2 of LUT2 (21, 43) in the operand form of the interpolation; weights 0 and 3 return the endpoints
            constexpr uint32_t BW1 = 21u * 0x3FFFCu + 256u, BW2 = 43u * 0x3FFFCu + 256u;
            uint32_t pal[3][4];
            BU_UNROLL
            for (int s = 0; s < subsets; s++) {
                BU_UNROLL
                for (int c = 0; c < NC; c++) {
                    const uint32_t lo = e[(2 * L::channels) * s + 2 * c], hi = e[(2 * L::channels) * s + 2 * c + 1];
                    const uint32_t v1 = bu_udot2(A[s][c], BW1, 128u), v2 = bu_udot2(A[s][c], BW2, 128u);
                    pal[s][c] = bu_perm(v2, v1, 0x0C06020Cu) | lo | (hi << 24);
                }
            }
            if constexpr (L::has_alpha) sink.alpha_palette(pal[0][NC - 1], subsets == 2 ? pal[1][NC - 1] : 0u);
            BU_UNROLL
            for (int x = 0; x < 4; x++) {
                const uint32_t wsel = (W[0] >> (2 * x)) & 0x03030303u;
                uint32_t ch[4] = {0, 0, 0, 0}, asel = wsel;
                if constexpr (subsets == 1) {
                    BU_UNROLL
                    for (int c = 0; c < NC; c++) ch[c] = bu_perm(0u, pal[0][c], wsel);
                } else if constexpr (subsets == 2) {
                    const uint32_t sel = wsel | (((upat >> (2 * x)) & 0x01010101u) << 2);
                    asel = sel;
                    BU_UNROLL
                    for (int c = 0; c < NC; c++) ch[c] = bu_perm(pal[1][c], pal[0][c], sel);
                } else {
                    const uint32_t sid = (upat >> (2 * x)) & 0x03030303u;
                    const uint32_t sel = wsel | ((sid & 0x01010101u) << 2), is2 = ((sid >> 1) & 0x01010101u) * 255u;
                    BU_UNROLL
                    for (int c = 0; c < NC; c++) ch[c] = bu_bfi(is2, bu_perm(0u, pal[2][c], wsel), bu_perm(pal[1][c], pal[0][c], sel));
                }
                sink.template cols<fmt>(x, ch, asel);
            }
            return BU_ST_OK;
        }
        if constexpr (wb == 3 && planes == 1) {
            // Three-bit weights (modes 2, 5, 12): eight values per channel and subset -- exactly the eight source bytes of ONE
            // v_perm_b32.  Entries 0 and 7 are the endpoints, entries 1..6 a v_dot2 each (LUT3 = 0, 9, 18, 27, 37, 46, 55, 64).
            // A block row is 12 bits of the weight string: rows 0 | 1 and 2 | 3 go into the 16-bit lanes of two words once, and a
            // column's selector is a shift + mask of each and a v_perm that lines the four rows up.  Per block 36 (RGB) / 48
            // (RGBA) instructions of palette + 8 / 9 per column where the generic path below issues 9 to 12 per TEXEL.
            constexpr uint32_t LUT3[6] = {9, 18, 27, 37, 46, 55};
            uint32_t pal[2][4][2];
            BU_UNROLL
            for (int s = 0; s < subsets; s++) {
                BU_UNROLL
                for (int c = 0; c < NC; c++) {
                    const uint32_t lo = e[(2 * L::channels) * s + 2 * c], hi = e[(2 * L::channels) * s + 2 * c + 1];
                    uint32_t v[6];
                    BU_UNROLL
                    for (int k = 0; k < 6; k++) v[k] = bu_udot2(A[s][c], LUT3[k] * 0x3FFFCu + 256u, 128u);
                    pal[s][c][0] = bu_perm(v[0], lo, 0x0C0C0600u) | bu_perm(v[2], v[1], 0x06020C0Cu);
                    pal[s][c][1] = bu_perm(v[4], v[3], 0x0C0C0602u) | bu_perm(hi, v[5], 0x04020C0Cu);
                }
            }
            if constexpr (L::has_alpha) sink.alpha_palette(pal[0][NC - 1][0], pal[0][NC - 1][1]);
            const uint32_t w23 = bu_alignbit(W[1], W[0], 24);  // rows 2 and 3
            const uint32_t q01 = (W[0] & 0xFFFu) | ((W[0] << 4) & 0x0FFF0000u), q23 = (w23 & 0xFFFu) | ((w23 << 4) & 0x0FFF0000u);
            BU_UNROLL
            for (int x = 0; x < 4; x++) {
                const uint32_t sel = bu_perm((q23 >> (3 * x)) & 0x00070007u, (q01 >> (3 * x)) & 0x00070007u, 0x06040200u);
                uint32_t ch[4] = {0, 0, 0, 0};
                if constexpr (subsets == 1) {
                    BU_UNROLL
                    for (int c = 0; c < NC; c++) ch[c] = bu_perm(pal[0][c][1], pal[0][c][0], sel);
                } else {
                    static_assert(subsets <= 2, "three-subset modes have 2-bit weights");
                    const uint32_t is1 = ((upat >> (2 * x)) & 0x01010101u) * 255u;
                    BU_UNROLL
                    for (int c = 0; c < NC; c++) ch[c] = bu_bfi(is1, bu_perm(pal[1][c][1], pal[1][c][0], sel), bu_perm(pal[0][c][1], pal[0][c][0], sel));
                }
                sink.template cols<fmt>(x, ch, sel);
            }
            return BU_ST_OK;
        }
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            uint32_t sid = 0;
            if constexpr (subsets > 1) sid = (upat >> (2 * i)) & 3u;
            // weights x4, packed (256-4w) | 4w << 16: one LUT read per weight (dequantisation and packing folded in)
            const uint32_t b0 = T.wpack[(1 << wb) - 2 + bu_wfield<wb>(W, i)];
            uint32_t v[4] = {0, 0, 0, 0};
            BU_UNROLL
            for (int c = 0; c < NC; c++) {
                uint32_t a = A[0][c];
                if constexpr (subsets == 2) a = sid == 1 ? A[1][c] : a;
                if constexpr (subsets == 3) {
                    // mask blend, not a select chain: hipcc turns `sid==2 ? a[2] : sid==1 ? a[1] : a[0]` back
                    // into a dynamically indexed array and parks it in LDS (48 KiB per workgroup)
                    const uint32_t k1 = 0u - (sid & 1u), k2 = 0u - (sid >> 1);
                    a ^= (X1[c] & k1) ^ (X2[c] & k2);
                }
                v[c] = bu_udot2(a, b0, 128u);  // result = byte 2
            }
            sink.template raw<fmt>(i, v);
        }
        return BU_ST_OK;
    }
}

// the RGBA32 sink: 16 texels, row-major inside the block, little-endian R,G,B,A
struct BuSinkRgba {
    uint32_t* px;
    BU_DEVM void word(int i, uint32_t w) { px[i] = w; }
    template <int FMT>
    BU_DEVM void raw(int i, const uint32_t v[4])
    {
        // gather byte 2 of each channel word
        if constexpr (FMT == BU_FMT_RGB) px[i] = bu_perm(v[1], v[0], 0x0C0C0602u) | ((v[2] & 0x00FF0000u) | 0xFF000000u);
        else if constexpr (FMT == BU_FMT_RGBA) px[i] = bu_perm(v[1], v[0], 0x0C0C0602u) | (bu_perm(v[3], v[2], 0x06020C0Cu));
        else px[i] = bu_perm(v[1], v[0], 0x06020202u);
    }
    // a column of channel bytes -> four texel words (a 4x4 byte transpose, two v_perm levels)
    BU_DEVM void alpha_palette(uint32_t, uint32_t) {}
    template <int FMT>
    BU_DEVM void cols(int x, const uint32_t ch[4], uint32_t)
    {
        if constexpr (FMT == BU_FMT_LA) {
            px[x] = bu_perm(ch[1], ch[0], 0x04000000u);
            px[4 + x] = bu_perm(ch[1], ch[0], 0x05010101u);
            px[8 + x] = bu_perm(ch[1], ch[0], 0x06020202u);
            px[12 + x] = bu_perm(ch[1], ch[0], 0x07030303u);
        } else {
            const uint32_t t01 = bu_perm(ch[1], ch[0], 0x05010400u), t23 = bu_perm(ch[1], ch[0], 0x07030602u);  // R0 G0 R1 G1 / R2 G2 R3 G3
            if constexpr (FMT == BU_FMT_RGB) {
                px[x] = bu_perm(ch[2], t01, 0x0D040100u);  // R G B 255
                px[4 + x] = bu_perm(ch[2], t01, 0x0D050302u);
                px[8 + x] = bu_perm(ch[2], t23, 0x0D060100u);
                px[12 + x] = bu_perm(ch[2], t23, 0x0D070302u);
            } else {
                const uint32_t u01 = bu_perm(ch[3], ch[2], 0x05010400u), u23 = bu_perm(ch[3], ch[2], 0x07030602u);  // B0 A0 B1 A1 / ...
                px[x] = bu_perm(u01, t01, 0x05040100u);
                px[4 + x] = bu_perm(u01, t01, 0x07060302u);
                px[8 + x] = bu_perm(u23, t23, 0x05040100u);
                px[12 + x] = bu_perm(u23, t23, 0x07060302u);
            }
        }
    }
};
template <int M>
BU_DEV int bu_block_rgba(const BuTables& T, const BuBlk& b, uint32_t px[16])
{
    BuSinkRgba sink{px};
    return bu_block_unpack<M>(T, b, sink);
}
