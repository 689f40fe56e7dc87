/*
 * bu_oracle.c -- CPU ORACLE for the UASTC / ETC1S block-transcode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is a plain-C restatement of the reference's
 * (JakubValtar/basisu_rs) per-block algorithms.  It exists so that the HIP product path can be
 * checked bit-for-bit, and so that bench.py can time "the reference CPU path" on the GPU box's
 * host cores (cpu_baseline.kind = "port").  Nothing under basisu_rs_amd/ links, loads or calls
 * it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinning: the reference is Rust and no Rust toolchain exists in the build image, so the
 * reference itself cannot be run.  The oracle is pinned against the reference's own 3 040
 * known-answer vectors (tests/block_test_cases/uastc_{astc,bc7,etc1,etc2,rgba}.rs, asserted by
 * tests/transcode_uastc_block.rs:35-78), committed as tests/golden/uastc_kat.bin.  The ETC1S
 * back-end has no vectors in the reference ("parity by reading" -- see DESIGN.md).
 *
 * Every function cites the reference file:line it follows.  The three f32 sites of the reference
 * (bc7.rs:408-553, etc.rs:297-307) are restated with IEEE binary32 operations in the reference's
 * order (build with -ffp-contract=off, no -ffast-math); integer forms used by the GPU are
 * exported next to them so tests can prove the equivalence exhaustively.
 * Release-build (wrapping) integer semantics are pinned (hard part 4).
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bu_oracle_tables.h"

#define ORC_OK 0
#define ORC_ERR_MODE 1    /* "invalid mode index"          uastc.rs:336 */
#define ORC_ERR_PATTERN 2 /* "block pattern is not valid"  uastc.rs:364 */
#define ORC_ERR_LENGTH 3  /* "data length is not divisible by UASTC block size (16)" uastc.rs:56 */

/* lib.rs:57-61  mask!(n) */
static uint32_t mask32(unsigned n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

/* ------------------------------------------------------------------ bitreader.rs:3-61 */
typedef struct {
    const uint8_t *bytes;
    size_t len;
    size_t bit_pos;
} reader_t;

static uint8_t rd_byte(const reader_t *r, size_t i) { return i < r->len ? r->bytes[i] : 0; } /* :45,55 */

static uint32_t rd_peek(const reader_t *r, unsigned count) /* bitreader.rs:37-60 */
{
    size_t byte = r->bit_pos / 8;
    unsigned bit = (unsigned)(r->bit_pos % 8);
    uint32_t result = (uint32_t)(rd_byte(r, byte) >> bit);
    unsigned read = 8 - bit;
    byte++;
    while (read < count) {
        /* the reference shifts a u32 by `read` (< 32 because count <= 32) */
        result |= (uint32_t)rd_byte(r, byte) << read;
        read += 8;
        byte++;
    }
    return result & mask32(count);
}
static void rd_remove(reader_t *r, unsigned count) { r->bit_pos += count; } /* :33-35 */
static uint32_t rd_read(reader_t *r, unsigned count)                        /* :27-31 */
{
    uint32_t v = rd_peek(r, count);
    rd_remove(r, count);
    return v;
}

/* ------------------------------------------------------------------ bitwriter.rs:3-54 */
typedef struct {
    uint8_t *bytes;
    size_t len;
    size_t bit_pos;
} writer_t;

static void wr_or(uint8_t *bytes, size_t len, size_t i, uint8_t v)
{
    if (i < len) bytes[i] |= v; /* out-of-range writes go to `trash` (bitwriter.rs:34,46) */
}

static void wr_at(uint8_t *bytes, size_t len, size_t bit_pos, unsigned count, uint32_t v)
{
    v &= mask32(count);
    size_t byte = bit_pos / 8;
    unsigned bit = (unsigned)(bit_pos % 8);
    wr_or(bytes, len, byte, (uint8_t)(v << bit));
    unsigned written = 8 - bit;
    byte++;
    while (written < count) {
        wr_or(bytes, len, byte, (uint8_t)(v >> written));
        written += 8;
        byte++;
    }
}

static void wr_write(writer_t *w, unsigned count, uint32_t v) /* bitwriter.rs:23-51 */
{
    wr_at(w->bytes, w->len, w->bit_pos, count, v);
    w->bit_pos += count;
}

/* ------------------------------------------------------------------ bitwriter.rs:56-116 */
typedef struct {
    uint8_t *bytes;
    size_t len;
    size_t bit_pos;
} writer_rev_t;

static void wrr_write(writer_rev_t *w, unsigned count, uint32_t v) /* :87-113 */
{
    w->bit_pos -= count; /* wrapping_sub; an underflow lands out of range and is dropped */
    wr_at(w->bytes, w->len, w->bit_pos, count, v);
}
static uint32_t rev_bits32(uint32_t v)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++)
        if (v & (1u << i)) r |= 1u << (31 - i);
    return r;
}
static void wrr_write_rev(writer_rev_t *w, unsigned count, uint32_t v) /* :72-75 */
{
    /* v.reverse_bits().wrapping_shr(32 - count): wrapping_shr masks the amount to 5 bits */
    uint32_t r = rev_bits32(v) >> ((32 - count) & 31);
    wrr_write(w, count, r);
}

/* ------------------------------------------------------------------ uastc.rs:443-557 */
enum { FMT_RGB = 0, FMT_RGBA = 1, FMT_LA = 2 };
typedef struct {
    uint8_t id, code_size, range, format, weight_bits, planes, subsets, trans_flags_bits;
} mode_t_;

/* uastc.rs:528-557 (format facts of the 19 UASTC modes) */
static const mode_t_ MODES[19] = {
    {0, 4, 19, FMT_RGB, 4, 1, 1, 15},  {1, 6, 20, FMT_RGB, 2, 1, 1, 15},  {2, 5, 8, FMT_RGB, 3, 1, 2, 15},
    {3, 5, 7, FMT_RGB, 2, 1, 3, 15},   {4, 5, 12, FMT_RGB, 2, 1, 2, 15},  {5, 5, 20, FMT_RGB, 3, 1, 1, 15},
    {6, 5, 18, FMT_RGB, 2, 2, 1, 15},  {7, 5, 12, FMT_RGB, 2, 1, 2, 15},  {8, 5, 0, FMT_RGBA, 0, 1, 1, 0},
    {9, 5, 8, FMT_RGBA, 2, 1, 2, 23},  {10, 3, 13, FMT_RGBA, 4, 1, 1, 17}, {11, 2, 13, FMT_RGBA, 2, 2, 1, 17},
    {12, 3, 19, FMT_RGBA, 3, 1, 1, 17}, {13, 5, 20, FMT_RGBA, 1, 2, 1, 23}, {14, 5, 20, FMT_RGBA, 2, 1, 1, 23},
    {15, 7, 20, FMT_LA, 4, 1, 1, 23},  {16, 6, 20, FMT_LA, 2, 1, 2, 23},  {17, 6, 20, FMT_LA, 2, 2, 1, 23},
    {18, 4, 11, FMT_RGB, 5, 1, 1, 15},
};

static int mode_has_alpha(const mode_t_ *m) { return m->format != FMT_RGB; }   /* :456-461 */
static int mode_has_blue(const mode_t_ *m) { return m->format != FMT_LA; }     /* :463-468 */
static unsigned mode_channels(const mode_t_ *m)                                 /* :470-476 */
{
    return m->format == FMT_RGB ? 3 : (m->format == FMT_RGBA ? 4 : 2);
}
static unsigned mode_endpoint_count(const mode_t_ *m) { return mode_channels(m) * m->subsets * 2; } /* :478-480 */
static unsigned mode_weight_count(const mode_t_ *m) { return m->planes * 16u; }                     /* :482-484 */

/* uastc.rs:329-341 */
static int decode_mode(reader_t *r, const mode_t_ **out)
{
    uint32_t code = rd_peek(r, 7);
    unsigned idx = ORC_MODE_LUT[code];
    if (idx >= 19) return ORC_ERR_MODE;
    *out = &MODES[idx];
    rd_remove(r, (*out)->code_size);
    return ORC_OK;
}

/* uastc.rs:343-350 */
static unsigned decode_compsel(reader_t *r, const mode_t_ *m)
{
    if (m->planes == 2 && m->format == FMT_LA) return 3;
    if (m->planes == 2) return rd_read(r, 2);
    return 0;
}

/* uastc.rs:352-366 */
static int decode_pattern_index(reader_t *r, const mode_t_ *m, unsigned *pat)
{
    unsigned idx, count;
    if (m->id == 7) {
        idx = rd_read(r, 5);
        count = 19;
    } else if (m->subsets == 1) {
        *pat = 0;
        return ORC_OK;
    } else if (m->subsets == 2) {
        idx = rd_read(r, 5);
        count = 30;
    } else {
        idx = rd_read(r, 4);
        count = 11;
    }
    if (idx < count) {
        *pat = idx;
        return ORC_OK;
    }
    return ORC_ERR_PATTERN;
}

static const uint8_t ZERO16[16] = {0};
static const uint8_t ZERO1[1] = {0};

/* uastc.rs:368-376 */
static const uint8_t *get_pattern(const mode_t_ *m, unsigned pat)
{
    if (m->id == 7) return ORC_PAT23[pat];
    if (m->subsets == 1) return ZERO16;
    if (m->subsets == 2) return ORC_PAT2[pat];
    return ORC_PAT3[pat];
}

/* uastc.rs:378-385 */
static const uint8_t *get_anchors(const mode_t_ *m, unsigned pat, unsigned *n)
{
    if (m->id == 7) {
        *n = 2;
        return ORC_PAT23_ANCH[pat];
    }
    if (m->subsets == 1) {
        *n = 1;
        return ZERO1;
    }
    if (m->subsets == 2) {
        *n = 2;
        return ORC_PAT2_ANCH[pat];
    }
    *n = 3;
    return ORC_PAT3_ANCH[pat];
}

typedef struct {
    uint8_t c[4];
} color_t; /* color.rs:5-10, RGBA order */

/* uastc.rs:387-394 */
static color_t decode_mode8_rgba(reader_t *r)
{
    color_t c;
    c.c[0] = (uint8_t)rd_read(r, 8);
    c.c[1] = (uint8_t)rd_read(r, 8);
    c.c[2] = (uint8_t)rd_read(r, 8);
    c.c[3] = (uint8_t)rd_read(r, 8);
    return c;
}

typedef struct {
    uint8_t etc1d, etc1i, etc1s, etc1r, etc1g, etc1b;
} mode8_flags_t; /* uastc.rs:19-27 */

/* uastc.rs:400-409 */
static mode8_flags_t decode_mode8_etc1_flags(reader_t *r)
{
    mode8_flags_t f;
    f.etc1d = (uint8_t)rd_read(r, 1);
    f.etc1i = (uint8_t)rd_read(r, 3);
    f.etc1s = (uint8_t)rd_read(r, 2);
    f.etc1r = (uint8_t)rd_read(r, 5);
    f.etc1g = (uint8_t)rd_read(r, 5);
    f.etc1b = (uint8_t)rd_read(r, 5);
    return f;
}

typedef struct {
    uint8_t bc1h0, bc1h1, etc1f, etc1d, etc1i0, etc1i1;
    int has_bias;
    uint8_t etc1bias, etc2tm;
} trans_flags_t; /* uastc.rs:29-39 */

/* uastc.rs:411-436 */
static trans_flags_t decode_trans_flags(reader_t *r, const mode_t_ *m)
{
    trans_flags_t f;
    int m1012 = m->id >= 10 && m->id <= 12;
    f.bc1h0 = (uint8_t)rd_read(r, 1);
    f.bc1h1 = m1012 ? 0 : (uint8_t)rd_read(r, 1);
    f.etc1f = (uint8_t)rd_read(r, 1);
    f.etc1d = (uint8_t)rd_read(r, 1);
    f.etc1i0 = (uint8_t)rd_read(r, 3);
    f.etc1i1 = (uint8_t)rd_read(r, 3);
    f.has_bias = !m1012;
    f.etc1bias = m1012 ? 0 : (uint8_t)rd_read(r, 5);
    f.etc2tm = mode_has_alpha(m) ? (uint8_t)rd_read(r, 8) : 0;
    return f;
}

/* uastc.rs:438-441 */
static void skip_trans_flags(reader_t *r, const mode_t_ *m) { rd_remove(r, m->trans_flags_bits); }

typedef struct {
    uint8_t trit_quint, bits;
} quant_ep_t; /* uastc.rs:579-583 */

/* uastc.rs:585-614 */
static uint8_t unquant_endpoint(quant_ep_t q, unsigned range_index)
{
    const orc_bise *range = &ORC_BISE[range_index];
    uint16_t quant_bits = q.bits;
    if (range->trits == 0 && range->quints == 0 && range->bits > 0) {
        uint16_t bits_la = (uint16_t)(quant_bits << (8 - range->bits));
        uint16_t val = 0;
        while (bits_la > 0) {
            val |= bits_la;
            bits_la >>= range->bits;
        }
        return (uint8_t)val;
    } else {
        uint16_t a = (quant_bits & 1) ? 511 : 0;
        uint16_t b = 0;
        for (int j = 0; j < 9; j++) {
            b <<= 1;
            char shift = range->deq_b[j];
            if (shift != '0') b |= (quant_bits >> (shift - 'a')) & 1;
        }
        uint16_t c = range->deq_c;
        uint16_t d = q.trit_quint;
        uint16_t val = (uint16_t)(d * c + b);
        val ^= a;
        return (uint8_t)((a & 0x80) | (val >> 2));
    }
}

/* uastc.rs:616-695 */
static void decode_endpoints(reader_t *r, unsigned range_index, unsigned value_count, quant_ep_t out[18])
{
    memset(out, 0, 18 * sizeof(quant_ep_t));
    const orc_bise *range = &ORC_BISE[range_index];
    unsigned bit_count = range->bits;

    if (range->quints > 0) {
        unsigned out_pos = 0;
        for (unsigned g = 0; g < value_count / 3; g++) {
            uint8_t quints = (uint8_t)rd_read(r, 7);
            for (int k = 0; k < 3; k++) {
                out[out_pos].trit_quint = quints % 5;
                quints /= 5;
                out_pos++;
            }
        }
        unsigned remaining = value_count - out_pos;
        if (remaining > 0) {
            unsigned bits_used = remaining == 1 ? 3 : 5;
            uint8_t quints = (uint8_t)rd_read(r, bits_used);
            for (unsigned k = 0; k < remaining; k++) {
                out[out_pos].trit_quint = quints % 5;
                quints /= 5;
                out_pos++;
            }
        }
    }
    if (range->trits > 0) {
        unsigned out_pos = 0;
        for (unsigned g = 0; g < value_count / 5; g++) {
            uint8_t trits = (uint8_t)rd_read(r, 8);
            for (int k = 0; k < 5; k++) {
                out[out_pos].trit_quint = trits % 3;
                trits /= 3;
                out_pos++;
            }
        }
        unsigned remaining = value_count - out_pos;
        if (remaining > 0) {
            static const unsigned used[5] = {0, 2, 4, 5, 7};
            uint8_t trits = (uint8_t)rd_read(r, used[remaining]);
            for (unsigned k = 0; k < remaining; k++) {
                out[out_pos].trit_quint = trits % 3;
                trits /= 3;
                out_pos++;
            }
        }
    }
    if (bit_count > 0)
        for (unsigned i = 0; i < value_count; i++) out[i].bits = (uint8_t)rd_read(r, bit_count);
}

/* uastc.rs:697-719 */
static void unquant_weights(uint8_t *w, unsigned n, unsigned weight_bits)
{
    const uint8_t *lut = weight_bits == 1   ? ORC_WLUT1
                         : weight_bits == 2 ? ORC_WLUT2
                         : weight_bits == 3 ? ORC_WLUT3
                         : weight_bits == 4 ? ORC_WLUT4
                                            : ORC_WLUT5;
    for (unsigned i = 0; i < n; i++) w[i] = lut[w[i]];
}

/* uastc.rs:721-740; out[planes*i + plane] = raw weight */
static void decode_weights(reader_t *r, const mode_t_ *m, unsigned pat, uint8_t out[32])
{
    unsigned n_anch;
    const uint8_t *anchors = get_anchors(m, pat, &n_anch);
    uint8_t bits[16];
    for (int i = 0; i < 16; i++) bits[i] = m->weight_bits;
    for (unsigned a = 0; a < n_anch; a++) bits[anchors[a]] = (uint8_t)(m->weight_bits - 1);
    for (unsigned i = 0; i < 16; i++)
        for (unsigned plane = 0; plane < m->planes; plane++)
            out[m->planes * i + plane] = (uint8_t)rd_read(r, bits[i]);
}

/* uastc.rs:176-216 */
static void assemble_endpoint_pairs(const mode_t_ *m, const uint8_t e[18], color_t pairs[3][2])
{
    memset(pairs, 0, 6 * sizeof(color_t));
    if (m->format == FMT_RGB) {
        for (int s = 0; s < 3; s++) {
            const uint8_t *b = e + 6 * s;
            color_t lo = {{b[0], b[2], b[4], 0xFF}}, hi = {{b[1], b[3], b[5], 0xFF}};
            pairs[s][0] = lo;
            pairs[s][1] = hi;
        }
    } else if (m->format == FMT_RGBA) {
        for (int s = 0; s < 2; s++) { /* 18 bytes hold two whole chunks of 8 */
            const uint8_t *b = e + 8 * s;
            color_t lo = {{b[0], b[2], b[4], b[6]}}, hi = {{b[1], b[3], b[5], b[7]}};
            pairs[s][0] = lo;
            pairs[s][1] = hi;
        }
    } else {
        for (int s = 0; s < 3; s++) { /* chunks_exact(4) over 18 bytes: 4 chunks, zip stops at 3 */
            const uint8_t *b = e + 4 * s;
            color_t lo = {{b[0], b[0], b[0], b[2]}}, hi = {{b[1], b[1], b[1], b[3]}};
            pairs[s][0] = lo;
            pairs[s][1] = hi;
        }
    }
}

/* uastc.rs:218-235 (srgb is always false, :271) */
static uint8_t astc_interpolate(uint8_t l8, uint8_t h8, uint8_t w8)
{
    uint32_t l = l8, h = h8, w = w8;
    l = (l << 8) | l;
    h = (h << 8) | h;
    uint32_t k = (l * (64 - w) + h * w + 32) >> 6;
    return (uint8_t)(k >> 8);
}

/* uastc.rs:237-327 */
static int decode_block_to_rgba(const uint8_t bytes[16], color_t out[16])
{
    reader_t r = {bytes, 16, 0};
    const mode_t_ *m;
    int st = decode_mode(&r, &m);
    if (st) return st;
    if (m->id == 8) {
        color_t c = decode_mode8_rgba(&r);
        for (int i = 0; i < 16; i++) out[i] = c;
        return ORC_OK;
    }
    skip_trans_flags(&r, m);
    unsigned compsel = decode_compsel(&r, m);
    unsigned pat;
    st = decode_pattern_index(&r, m, &pat);
    if (st) return st;

    unsigned endpoint_count = mode_endpoint_count(m);
    unsigned weight_count = mode_weight_count(m);
    uint8_t endpoints[18] = {0};
    uint8_t weights[32] = {0};
    quant_ep_t q[18];
    decode_endpoints(&r, m->range, endpoint_count, q);
    for (unsigned i = 0; i < endpoint_count; i++) endpoints[i] = unquant_endpoint(q[i], m->range);
    decode_weights(&r, m, pat, weights);
    unquant_weights(weights, weight_count, m->weight_bits);

    color_t e[3][2];
    assemble_endpoint_pairs(m, endpoints, e);
    if (m->subsets == 1) {
        const color_t e0 = e[0][0], e1 = e[0][1];
        if (m->planes == 1) {
            for (int i = 0; i < 16; i++)
                for (int c = 0; c < 4; c++) out[i].c[c] = astc_interpolate(e0.c[c], e1.c[c], weights[i]);
        } else {
            for (int i = 0; i < 16; i++) {
                const uint8_t *ws = weights + 2 * i;
                for (unsigned c = 0; c < 4; c++) {
                    uint8_t w = (compsel == c) ? ws[1] : ws[0];
                    out[i].c[c] = astc_interpolate(e0.c[c], e1.c[c], w);
                }
            }
        }
    } else {
        const uint8_t *pattern = get_pattern(m, pat);
        for (int i = 0; i < 16; i++) {
            const color_t e0 = e[pattern[i]][0], e1 = e[pattern[i]][1];
            for (int c = 0; c < 4; c++) out[i].c[c] = astc_interpolate(e0.c[c], e1.c[c], weights[i]);
        }
    }
    return ORC_OK;
}

/* ================================================================== ASTC (astc.rs:8-181) */
static int convert_astc(const uint8_t bytes[16], uint8_t output[16])
{
    reader_t r = {bytes, 16, 0};
    const mode_t_ *m;
    int st = decode_mode(&r, &m);
    if (st) return st;
    memset(output, 0, 16);
    writer_t w = {output, 16, 0};

    if (m->id == 8) { /* :17-43 void extent */
        color_t rgba = decode_mode8_rgba(&r);
        wr_write(&w, 12, 0xDFC);
        wr_write(&w, 20, 0x000FFFFF);
        wr_write(&w, 32, 0xFFFFFFFF);
        for (int c = 0; c < 4; c++) {
            uint16_t v = rgba.c[c];
            wr_write(&w, 16, (uint16_t)(v << 8 | v));
        }
        return ORC_OK;
    }
    skip_trans_flags(&r, m);
    unsigned compsel = decode_compsel(&r, m);
    unsigned pat;
    st = decode_pattern_index(&r, m, &pat);
    if (st) return st;

    unsigned endpoint_count = mode_endpoint_count(m);
    quant_ep_t q[18];
    decode_endpoints(&r, m->range, endpoint_count, q);

    int invert[3] = {0, 0, 0};
    if (mode_has_blue(m)) { /* :57-78 blue-contraction avoidance */
        unsigned per_subset = endpoint_count / m->subsets;
        for (unsigned s = 0; s < m->subsets; s++) {
            quant_ep_t *qs = q + s * per_subset;
            uint8_t e[6] = {0};
            for (unsigned i = 0; i < 6 && i < per_subset; i++) e[i] = unquant_endpoint(qs[i], m->range);
            uint32_t s0 = (uint32_t)e[0] + e[2] + e[4];
            uint32_t s1 = (uint32_t)e[1] + e[3] + e[5];
            if (s0 > s1) {
                invert[s] = 1;
                for (unsigned i = 0; i + 1 < per_subset; i += 2) {
                    quant_ep_t t = qs[i];
                    qs[i] = qs[i + 1];
                    qs[i + 1] = t;
                }
            }
        }
    }

    /* :80-96 block mode and config */
    wr_write(&w, 13, ORC_ASTC_BLOCK_MODE13[m->id]);
    if (m->id == 7) {
        wr_write(&w, 10, ORC_PAT23_ASTC[pat]);
        wr_write(&w, 2, 0);
    } else if (m->subsets == 2) {
        wr_write(&w, 10, ORC_PAT2_ASTC[pat]);
        wr_write(&w, 2, 0);
    } else if (m->subsets == 3) {
        wr_write(&w, 10, ORC_PAT3_ASTC[pat]);
        wr_write(&w, 2, 0);
    }
    wr_write(&w, 4, m->format == FMT_RGB ? 8 : (m->format == FMT_RGBA ? 12 : 4));

    { /* :98-141 endpoints; the reference walks all 18 array entries */
        const orc_bise *range = &ORC_BISE[m->range];
        unsigned bc = range->bits;
        if (range->quints > 0) {
            for (unsigned base = 0; base < 18; base += 3) {
                unsigned n = 18 - base < 3 ? 18 - base : 3;
                uint8_t id = 0;
                for (int k = (int)n - 1; k >= 0; k--) id = (uint8_t)(id * 5 + q[base + k].trit_quint);
                uint8_t qv = ORC_ASTC_QUINT_ENC[id];
                wr_write(&w, bc, n > 0 ? q[base + 0].bits : 0);
                wr_write(&w, 3, qv);
                wr_write(&w, bc, n > 1 ? q[base + 1].bits : 0);
                wr_write(&w, 2, (uint8_t)(qv >> 3));
                wr_write(&w, bc, n > 2 ? q[base + 2].bits : 0);
                wr_write(&w, 2, (uint8_t)(qv >> 5));
            }
        } else if (range->trits > 0) {
            for (unsigned base = 0; base < 18; base += 5) {
                unsigned n = 18 - base < 5 ? 18 - base : 5;
                uint8_t id = 0;
                for (int k = (int)n - 1; k >= 0; k--) id = (uint8_t)(id * 3 + q[base + k].trit_quint);
                uint8_t t = ORC_ASTC_TRIT_ENC[id];
                wr_write(&w, bc, n > 0 ? q[base + 0].bits : 0);
                wr_write(&w, 2, t);
                wr_write(&w, bc, n > 1 ? q[base + 1].bits : 0);
                wr_write(&w, 2, (uint8_t)(t >> 2));
                wr_write(&w, bc, n > 2 ? q[base + 2].bits : 0);
                wr_write(&w, 1, (uint8_t)(t >> 4));
                wr_write(&w, bc, n > 3 ? q[base + 3].bits : 0);
                wr_write(&w, 2, (uint8_t)(t >> 5));
                wr_write(&w, bc, n > 4 ? q[base + 4].bits : 0);
                wr_write(&w, 1, (uint8_t)(t >> 7));
            }
        } else {
            for (unsigned i = 0; i < 18; i++) wr_write(&w, bc, q[i].bits);
        }
    }

    { /* :143-178 weights and CCS, filled from the end */
        writer_rev_t wr = {output, 16, 128};
        uint8_t weights[32];
        decode_weights(&r, m, pat, weights);
        const uint8_t *pattern = get_pattern(m, pat);
        unsigned n = mode_weight_count(m);
        for (unsigned i = 0; i < n; i++) {
            unsigned texel = i / m->planes;
            unsigned subset = m->subsets == 1 ? 0 : pattern[texel];
            uint8_t wv = weights[i];
            if (invert[subset]) wv = (uint8_t)~wv;
            wrr_write_rev(&wr, m->weight_bits, wv);
        }
        if (m->planes != 1) wrr_write(&wr, 2, compsel);
    }
    return ORC_OK;
}

/* ================================================================== BC7 (bc7.rs) */
typedef struct {
    uint8_t id, pat_bits, endpoint_count, color_bits, alpha_bits, weight_bits, planes, subsets, p_bits, sp_bits;
} bc7_mode_t;

/* bc7.rs:570-579 (BC7 format facts) */
static const bc7_mode_t BC7_MODES[8] = {
    {0, 4, 18, 4, 0, 3, 1, 3, 1, 0}, {1, 6, 12, 6, 0, 3, 1, 2, 0, 1}, {2, 6, 18, 5, 0, 2, 1, 3, 0, 0},
    {3, 6, 12, 7, 0, 2, 1, 2, 1, 0}, {4, 0, 8, 5, 6, 2, 2, 1, 0, 0},  {5, 0, 8, 7, 8, 2, 2, 1, 0, 0},
    {6, 0, 8, 7, 7, 4, 1, 1, 1, 0},  {7, 6, 16, 5, 5, 2, 1, 2, 1, 0},
};

/* bc7.rs:1126-1136 */
static const uint8_t *mode6_opt(uint8_t c, int p) { return ORC_BC7_M6_OPT[(unsigned)c + (p ? 0 : 1)]; }
static uint32_t mode6_opt_err(uint8_t c, int p) { return ((c == 0 && p) || (c == 255 && !p)) ? 1 : 0; }

/* bc7.rs:312-375 */
static void convert_mode8_bc7(color_t solid, unsigned *mode, color_t endpoint[2], uint8_t p_bits[2], uint8_t weights[2])
{
    uint32_t best_err0 = 0, best_err1 = 0;
    for (int c = 0; c < 4; c++) {
        best_err0 += mode6_opt_err(solid.c[c], 0);
        best_err1 += mode6_opt_err(solid.c[c], 1);
    }
    memset(endpoint, 0, 2 * sizeof(color_t));
    p_bits[0] = p_bits[1] = 0;
    weights[0] = weights[1] = 0;
    if (best_err0 > 0 && best_err1 > 0) {
        *mode = 5;
        for (int c = 0; c < 3; c++) {
            endpoint[0].c[c] = ORC_BC7_M5_OPT[solid.c[c]][0];
            endpoint[1].c[c] = ORC_BC7_M5_OPT[solid.c[c]][1];
        }
        endpoint[0].c[3] = solid.c[3];
        endpoint[1].c[3] = solid.c[3];
        weights[0] = 1; /* BC7ENC_MODE_5_OPTIMAL_INDEX */
        weights[1] = 0;
    } else {
        *mode = 6;
        int best_p = best_err1 < best_err0;
        for (int c = 0; c < 4; c++) {
            endpoint[0].c[c] = mode6_opt(solid.c[c], best_p)[0];
            endpoint[1].c[c] = mode6_opt(solid.c[c], best_p)[1];
        }
        p_bits[0] = p_bits[1] = (uint8_t)best_p;
        weights[0] = weights[1] = 5; /* BC7ENC_MODE_6_OPTIMAL_INDEX */
    }
}

/* bc7.rs:377-398 */
static void convert_weights_to_bc7(uint8_t w[16], unsigned ub, unsigned bb)
{
    const uint8_t *lut;
    if (ub == 1 && bb == 2) lut = ORC_W1_BC7_2;
    else if (ub == 2 && bb == 4) lut = ORC_W2_BC7_4;
    else if (ub == 3 && bb == 4) lut = ORC_W3_BC7_4;
    else if (ub == 5 && bb == 4) lut = ORC_W5_BC7_4;
    else return; /* a == b */
    for (int i = 0; i < 16; i++) w[i] = lut[w[i]];
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* bc7.rs:408-475, f32-faithful.  Returns the shared p-bit; rewrites the endpoint pair. */
int bu_oracle_shared_pbits_f32(unsigned total_comps, unsigned comp_bits, uint8_t xl_col[4], uint8_t xh_col[4])
{
    unsigned total_bits = comp_bits + 1;
    int iscalep = (1 << total_bits) - 1;
    float scalep = (float)iscalep;
    float xl[4], xh[4];
    for (int c = 0; c < 4; c++) {
        xl[c] = (float)xl_col[c] / 255.0f;
        xh[c] = (float)xh_col[c] / 255.0f;
    }
    memset(xl_col, 0, 4);
    memset(xh_col, 0, 4);
    float best_err = 1e+9f;
    int s_bit = 0;
    for (int p = 0; p < 2; p++) {
        uint8_t x_min[4], x_max[4];
        for (int c = 0; c < 4; c++) {
            x_min[c] = (uint8_t)clampi((int)((xl[c] * scalep - (float)p) / 2.0f + 0.5f) * 2 + p, p, iscalep - 1 + p);
            x_max[c] = (uint8_t)clampi((int)((xh[c] * scalep - (float)p) / 2.0f + 0.5f) * 2 + p, p, iscalep - 1 + p);
        }
        uint8_t scaled_low[4], scaled_high[4];
        for (int i = 0; i < 4; i++) {
            scaled_low[i] = (uint8_t)(x_min[i] << (8 - total_bits));
            scaled_low[i] |= (uint8_t)(scaled_low[i] >> total_bits);
            scaled_high[i] = (uint8_t)(x_max[i] << (8 - total_bits));
            scaled_high[i] |= (uint8_t)(scaled_high[i] >> total_bits);
        }
        float err = 0.0f;
        for (unsigned i = 0; i < total_comps; i++) {
            float dl = (float)scaled_low[i] / 255.0f - xl[i];
            float dh = (float)scaled_high[i] / 255.0f - xh[i];
            /* err += a.powi(2) + b.powi(2): the sum of squares is formed first, then added */
            float t = dl * dl + dh * dh;
            err = err + t;
        }
        if (err < best_err) {
            best_err = err;
            s_bit = p;
            for (int j = 0; j < 4; j++) {
                xl_col[j] = x_min[j] >> 1;
                xh_col[j] = x_max[j] >> 1;
            }
        }
    }
    return s_bit;
}

/* Integer form of the same decision, used by the GPU path: exact rational arithmetic.
 * quantiser: q_p(x) = clamp(2*floor((x*S/255 - p)/2 + 1/2) + p, p, S-1+p), S = 2^tb - 1
 *   floor((x*S - 255p)/510 + 1/2) = floor((x*S - 255p + 255)/510)   (numerator >= 0)
 * error: sum over channels of (scaled - x)^2, p = 1 wins only if strictly smaller. */
static int quant_p_int(int x, int S, int p)
{
    int num = x * S - 255 * p + 255;
    int q = (num / 510) * 2 + p; /* num >= 0 so C truncation == floor */
    return clampi(q, p, S - 1 + p);
}
int bu_oracle_shared_pbits_int(unsigned total_comps, unsigned comp_bits, uint8_t xl_col[4], uint8_t xh_col[4])
{
    unsigned tb = comp_bits + 1;
    int S = (1 << tb) - 1;
    int best = 0;
    long best_err = -1;
    uint8_t ol[4] = {0}, oh[4] = {0};
    for (int p = 0; p < 2; p++) {
        int qmin[4], qmax[4];
        long err = 0;
        for (int c = 0; c < 4; c++) {
            qmin[c] = quant_p_int(xl_col[c], S, p);
            qmax[c] = quant_p_int(xh_col[c], S, p);
        }
        for (unsigned c = 0; c < total_comps; c++) {
            int sl = (uint8_t)(qmin[c] << (8 - tb));
            sl |= sl >> tb;
            int sh = (uint8_t)(qmax[c] << (8 - tb));
            sh |= sh >> tb;
            err += (long)(sl - xl_col[c]) * (sl - xl_col[c]) + (long)(sh - xh_col[c]) * (sh - xh_col[c]);
        }
        if (best_err < 0 || err < best_err) {
            best_err = err;
            best = p;
            for (int c = 0; c < 4; c++) {
                ol[c] = (uint8_t)(qmin[c] >> 1);
                oh[c] = (uint8_t)(qmax[c] >> 1);
            }
        }
    }
    memcpy(xl_col, ol, 4);
    memcpy(xh_col, oh, 4);
    return best;
}

/* bc7.rs:478-553, f32-faithful.  p_out[0] for the low endpoint, p_out[1] for the high one. */
void bu_oracle_unique_pbits_f32(unsigned total_comps, unsigned comp_bits, uint8_t xl_col[4], uint8_t xh_col[4], uint8_t p_out[2])
{
    unsigned total_bits = comp_bits + 1;
    int iscalep = (1 << total_bits) - 1;
    float scalep = (float)iscalep;
    float xl[4], xh[4];
    for (int c = 0; c < 4; c++) {
        xl[c] = (float)xl_col[c] / 255.0f;
        xh[c] = (float)xh_col[c] / 255.0f;
    }
    memset(xl_col, 0, 4);
    memset(xh_col, 0, 4);
    float best_err0 = 1e+9f, best_err1 = 1e+9f;
    p_out[0] = p_out[1] = 0;
    for (int p = 0; p < 2; p++) {
        uint8_t x_min[4], x_max[4];
        for (int c = 0; c < 4; c++) {
            x_min[c] = (uint8_t)clampi((int)((xl[c] * scalep - (float)p) / 2.0f + 0.5f) * 2 + p, p, iscalep - 1 + p);
            x_max[c] = (uint8_t)clampi((int)((xh[c] * scalep - (float)p) / 2.0f + 0.5f) * 2 + p, p, iscalep - 1 + p);
        }
        uint8_t scaled_low[4], scaled_high[4];
        for (int i = 0; i < 4; i++) {
            /* u8 wrapping_shr masks the amount to 3 bits: total_bits == 8 shifts by 0 */
            scaled_low[i] = (uint8_t)(x_min[i] << (8 - total_bits));
            scaled_low[i] |= (uint8_t)(scaled_low[i] >> (total_bits & 7));
            scaled_high[i] = (uint8_t)(x_max[i] << (8 - total_bits));
            scaled_high[i] |= (uint8_t)(scaled_high[i] >> (total_bits & 7));
        }
        float err0 = 0.0f, err1 = 0.0f;
        for (unsigned i = 0; i < total_comps; i++) {
            float d0 = (float)scaled_low[i] - xl[i] * 255.0f;
            float d1 = (float)scaled_high[i] - xh[i] * 255.0f;
            err0 = err0 + d0 * d0;
            err1 = err1 + d1 * d1;
        }
        if (err0 < best_err0) {
            best_err0 = err0;
            p_out[0] = (uint8_t)p;
            for (int j = 0; j < 4; j++) xl_col[j] = x_min[j] >> 1;
        }
        if (err1 < best_err1) {
            best_err1 = err1;
            p_out[1] = (uint8_t)p;
            for (int j = 0; j < 4; j++) xh_col[j] = x_max[j] >> 1;
        }
    }
}

/* Integer form of determine_unique_pbits for ONE endpoint (both endpoints are independent). */
int bu_oracle_unique_pbit_int(unsigned total_comps, unsigned comp_bits, uint8_t x_col[4])
{
    unsigned tb = comp_bits + 1;
    int S = (1 << tb) - 1;
    int best = 0;
    long best_err = -1;
    uint8_t o[4] = {0};
    for (int p = 0; p < 2; p++) {
        int q[4];
        long err = 0;
        for (int c = 0; c < 4; c++) q[c] = quant_p_int(x_col[c], S, p);
        for (unsigned c = 0; c < total_comps; c++) {
            int s = (uint8_t)(q[c] << (8 - tb));
            s |= s >> (tb & 7);
            err += (long)(s - x_col[c]) * (s - x_col[c]);
        }
        if (best_err < 0 || err < best_err) {
            best_err = err;
            best = p;
            for (int c = 0; c < 4; c++) o[c] = (uint8_t)(q[c] >> 1);
        }
    }
    memcpy(x_col, o, 4);
    return best;
}

/* bc7.rs:9-310 */
static int convert_bc7(const uint8_t bytes[16], uint8_t output[16])
{
    reader_t r = {bytes, 16, 0};
    const mode_t_ *m;
    int st = decode_mode(&r, &m);
    if (st) return st;
    memset(output, 0, 16);
    writer_t w = {output, 16, 0};
    enum { ALPHA = 3 };

    if (m->id == 8) { /* :18-59 */
        color_t rgba = decode_mode8_rgba(&r);
        unsigned bmode;
        color_t endpoint[2];
        uint8_t p_bits[2], wts[2];
        convert_mode8_bc7(rgba, &bmode, endpoint, p_bits, wts);
        const bc7_mode_t *bm = &BC7_MODES[bmode];
        wr_write(&w, bmode + 1, 1u << bmode);
        if (bmode == 5) wr_write(&w, 2, 0);
        for (int ch = 0; ch < 4; ch++) {
            unsigned bc = ch != ALPHA ? bm->color_bits : bm->alpha_bits;
            wr_write(&w, bc, endpoint[0].c[ch]);
            wr_write(&w, bc, endpoint[1].c[ch]);
        }
        if (bmode == 6) wr_write(&w, 2, (uint8_t)((p_bits[1] << 1) | p_bits[0]));
        for (unsigned pl = 0; pl < bm->planes; pl++) {
            wr_write(&w, bm->weight_bits - 1u, wts[pl]);
            for (int k = 0; k < 15; k++) wr_write(&w, bm->weight_bits, wts[pl]);
        }
        return ORC_OK;
    }

    unsigned bc7_mode_index = ORC_UASTC_TO_BC7_MODE[m->id];
    const bc7_mode_t *bm = &BC7_MODES[bc7_mode_index];
    skip_trans_flags(&r, m);
    unsigned compsel = decode_compsel(&r, m);
    unsigned uastc_pat;
    st = decode_pattern_index(&r, m, &uastc_pat);
    if (st) return st;

    unsigned per_channel = 2u * bm->subsets;
    unsigned bc7_channels = bm->endpoint_count / per_channel;

    color_t endpoints[3][2];
    {
        unsigned endpoint_count = mode_endpoint_count(m);
        quant_ep_t q[18];
        decode_endpoints(&r, m->range, endpoint_count, q);
        uint8_t un[18] = {0};
        for (unsigned i = 0; i < endpoint_count; i++) un[i] = unquant_endpoint(q[i], m->range);
        assemble_endpoint_pairs(m, un, endpoints);
    }

    uint8_t weights[2][16];
    memset(weights, 0, sizeof weights);
    {
        uint8_t raw[32];
        decode_weights(&r, m, uastc_pat, raw);
        if (m->planes == 1) {
            for (int i = 0; i < 16; i++) weights[0][i] = raw[i];
            convert_weights_to_bc7(weights[0], m->weight_bits, bm->weight_bits);
        } else {
            for (int i = 0; i < 32; i++) weights[i & 1][i >> 1] = raw[i];
            convert_weights_to_bc7(weights[0], m->weight_bits, bm->weight_bits);
            convert_weights_to_bc7(weights[1], m->weight_bits, bm->weight_bits);
        }
    }

    unsigned n_sub = bm->subsets;   /* endpoints[0..n_sub] */
    unsigned n_planes = bm->planes; /* weights[0..n_planes] */

    wr_write(&w, bc7_mode_index + 1, 1u << bc7_mode_index); /* :109 */

    static const uint8_t ANCH0[1] = {0};
    const uint8_t *bc7_anchors = ANCH0;
    unsigned n_bc7_anchors = 1;

    if (bm->subsets != 1) { /* :116-195 */
        unsigned bc7_pat;
        const uint8_t *pattern, *anchors, *perm;
        unsigned n_anch, n_perm;
        static const uint8_t P00[2] = {0, 0}, P01[2] = {0, 1}, P10[2] = {1, 0};
        if (m->id == 1) {
            bc7_pat = ORC_PAT2_BC7_INDEX_INV[0][0];
            pattern = ORC_PAT2_BC7[uastc_pat];
            anchors = ORC_BC7_ANCH2[bc7_pat];
            n_anch = 2;
            perm = P00;
            n_perm = 2;
        } else if (m->id == 7) {
            bc7_pat = ORC_PAT23_BC7_INDEX_PERM[uastc_pat][0];
            perm = ORC_PAT23_BC7_PERMS[ORC_PAT23_BC7_INDEX_PERM[uastc_pat][1]];
            n_perm = 3;
            pattern = ORC_PAT23_BC7[uastc_pat];
            anchors = ORC_BC7_ANCH3[bc7_pat];
            n_anch = 3;
        } else if (m->subsets == 2) {
            bc7_pat = ORC_PAT2_BC7_INDEX_INV[uastc_pat][0];
            pattern = ORC_PAT2_BC7[uastc_pat];
            anchors = ORC_BC7_ANCH2[bc7_pat];
            n_anch = 2;
            perm = ORC_PAT2_BC7_INDEX_INV[uastc_pat][1] ? P10 : P01;
            n_perm = 2;
        } else {
            bc7_pat = ORC_PAT3_BC7_INDEX_PERM[uastc_pat][0];
            perm = ORC_PAT3_BC7_PERMS[ORC_PAT3_BC7_INDEX_PERM[uastc_pat][1]];
            n_perm = 3;
            pattern = ORC_PAT3_BC7[uastc_pat];
            anchors = ORC_BC7_ANCH3[bc7_pat];
            n_anch = 3;
        }
        bc7_anchors = anchors;
        n_bc7_anchors = n_anch;
        wr_write(&w, bm->pat_bits, bc7_pat);

        { /* :163-169 permute: dst[X] = src[src_for_dst[X]], src = endpoints[0..n_sub] */
            color_t permuted[3][2];
            memset(permuted, 0, sizeof permuted);
            for (unsigned x = 0; x < n_perm && x < 3; x++) {
                permuted[x][0] = endpoints[perm[x]][0];
                permuted[x][1] = endpoints[perm[x]][1];
            }
            for (unsigned s = 0; s < n_sub; s++) {
                endpoints[s][0] = permuted[s][0];
                endpoints[s][1] = permuted[s][1];
            }
        }
        { /* :171-195 */
            uint8_t weight_mask = (uint8_t)mask32(bm->weight_bits);
            uint8_t msb = (uint8_t)(1u << (bm->weight_bits - 1));
            int inv[3] = {0, 0, 0};
            for (unsigned a = 0; a < n_anch && a < 3; a++) inv[a] = (weights[0][anchors[a]] & msb) != 0;
            for (unsigned s = 0; s < n_sub; s++)
                if (inv[s]) {
                    color_t t = endpoints[s][0];
                    endpoints[s][0] = endpoints[s][1];
                    endpoints[s][1] = t;
                }
            for (int i = 0; i < 16; i++)
                if (inv[pattern[i]]) weights[0][i] = (uint8_t)(~weights[0][i] & weight_mask);
        }
    } else { /* :196-247 */
        uint8_t weight_mask = (uint8_t)mask32(bm->weight_bits);
        uint8_t msb = (uint8_t)(1u << (bm->weight_bits - 1));
        if (m->planes == 1) {
            if (weights[0][0] & msb) {
                color_t t = endpoints[0][0];
                endpoints[0][0] = endpoints[0][1];
                endpoints[0][1] = t;
                for (int i = 0; i < 16; i++) weights[0][i] = (uint8_t)(~weights[0][i] & weight_mask);
            }
        } else {
            int inv0 = (weights[0][0] & msb) != 0, inv1 = (weights[1][0] & msb) != 0;
            for (int e = 0; e < 2; e++) { /* channel rotation :218-219 */
                uint8_t t = endpoints[0][e].c[compsel];
                endpoints[0][e].c[compsel] = endpoints[0][e].c[ALPHA];
                endpoints[0][e].c[ALPHA] = t;
            }
            if (inv0) {
                color_t t = endpoints[0][0];
                endpoints[0][0] = endpoints[0][1];
                endpoints[0][1] = t;
            }
            if (inv0 != inv1) {
                uint8_t t = endpoints[0][0].c[ALPHA];
                endpoints[0][0].c[ALPHA] = endpoints[0][1].c[ALPHA];
                endpoints[0][1].c[ALPHA] = t;
            }
            if (inv0)
                for (int i = 0; i < 16; i++) weights[0][i] = (uint8_t)(~weights[0][i] & weight_mask);
            if (inv1 && n_planes > 1)
                for (int i = 0; i < 16; i++) weights[1][i] = (uint8_t)(~weights[1][i] & weight_mask);
            wr_write(&w, 2, (compsel + 1) & 3);
            if (bm->id == 4) wr_write(&w, 1, 0);
        }
    }

    unsigned color_bits = bm->color_bits, alpha_bits = bm->alpha_bits;
    uint8_t p_bits[3][2];
    memset(p_bits, 0, sizeof p_bits);
    if (bm->p_bits != 0) { /* :253-256 */
        for (unsigned s = 0; s < n_sub; s++)
            bu_oracle_unique_pbits_f32(bc7_channels, bm->color_bits, endpoints[s][0].c, endpoints[s][1].c, p_bits[s]);
    } else if (bm->sp_bits != 0) { /* :257-260 */
        for (unsigned s = 0; s < n_sub; s++) {
            int sb = bu_oracle_shared_pbits_f32(bc7_channels, bm->color_bits, endpoints[s][0].c, endpoints[s][1].c);
            p_bits[s][0] = p_bits[s][1] = (uint8_t)sb;
        }
    } else { /* :261-273 */
        for (unsigned s = 0; s < n_sub; s++)
            for (int e = 0; e < 2; e++) {
                for (int ch = 0; ch < 3; ch++)
                    endpoints[s][e].c[ch] = (uint8_t)(((uint32_t)endpoints[s][e].c[ch] * mask32(color_bits) + 127) / 255);
                endpoints[s][e].c[ALPHA] = (uint8_t)(((uint32_t)endpoints[s][e].c[ALPHA] * mask32(alpha_bits) + 127) / 255);
            }
    }

    for (unsigned ch = 0; ch < bc7_channels; ch++) { /* :276-286 */
        unsigned bc = ch != ALPHA ? color_bits : alpha_bits;
        for (unsigned s = 0; s < n_sub; s++) {
            wr_write(&w, bc, endpoints[s][0].c[ch]);
            wr_write(&w, bc, endpoints[s][1].c[ch]);
        }
    }
    if (bm->p_bits != 0) { /* :288-294 */
        for (unsigned s = 0; s < n_sub; s++) wr_write(&w, 2, (uint8_t)((p_bits[s][1] << 1) | p_bits[s][0]));
    } else if (bm->sp_bits != 0) {
        wr_write(&w, 2, (uint8_t)((p_bits[1][0] << 1) | p_bits[0][0]));
    }
    { /* :296-307 */
        uint8_t bit_counts[16];
        for (int i = 0; i < 16; i++) bit_counts[i] = bm->weight_bits;
        for (unsigned a = 0; a < n_bc7_anchors; a++) bit_counts[bc7_anchors[a]] -= 1;
        for (unsigned pl = 0; pl < n_planes; pl++)
            for (int i = 0; i < 16; i++) wr_write(&w, bit_counts[i], weights[pl][i]);
    }
    return ORC_OK;
}

/* ================================================================== ETC (etc.rs) */
/* etc.rs:343-394 */
typedef struct {
    uint8_t selectors[4];
    uint8_t etc1_bytes[4];
} selector_t;

static unsigned sel_get(const selector_t *s, unsigned x, unsigned y) { return (s->selectors[y] >> (2 * x)) & 3; } /* :354-361 */

static void sel_set(selector_t *s, unsigned x, unsigned y, uint8_t val) /* :363-393 */
{
    unsigned shift = 2 * x;
    s->selectors[y] &= (uint8_t)~(3u << shift);
    s->selectors[y] |= (uint8_t)(val << shift);
    uint8_t mod_id = ORC_SEL_TO_ETC1[val];
    unsigned pixel_id = x * 4 + y;
    unsigned ms_byte = 1 - (pixel_id / 8);
    unsigned ls_byte = ms_byte + 2;
    unsigned bit = pixel_id % 8;
    s->etc1_bytes[ls_byte] &= (uint8_t)~(1u << bit);
    s->etc1_bytes[ls_byte] |= (uint8_t)((mod_id % 2) << bit);
    s->etc1_bytes[ms_byte] &= (uint8_t)~(1u << bit);
    s->etc1_bytes[ms_byte] |= (uint8_t)((mod_id / 2) << bit);
}

/* etc.rs:396-418 */
static color_t color_5_to_8(color_t c5)
{
    color_t o;
    for (int i = 0; i < 3; i++) o.c[i] = (uint8_t)((c5.c[i] << 3) | (c5.c[i] >> 2));
    o.c[3] = 255;
    return o;
}
static color_t color_4_to_8(color_t c4)
{
    color_t o;
    for (int i = 0; i < 3; i++) o.c[i] = (uint8_t)((c4.c[i] << 4) | c4.c[i]);
    o.c[3] = 255;
    return o;
}
/* etc.rs:420-431 */
static void apply_mod_to_base_color(color_t base, unsigned inten, color_t out[4])
{
    for (int k = 0; k < 4; k++) {
        int16_t md = ORC_ETC1_MOD[inten][k];
        for (int i = 0; i < 3; i++) out[k].c[i] = (uint8_t)clampi((int16_t)base.c[i] + md, 0, 255);
        out[k].c[3] = 255;
    }
}

/* etc.rs:203-259 */
static color_t apply_etc1_bias(color_t block_color, unsigned bias, uint32_t limit, unsigned subblock)
{
    static const uint8_t S_DIVS[3] = {1, 3, 9};
    for (unsigned c = 0; c < 3; c++) {
        int delta;
        switch (bias) {
        case 2: delta = subblock == 1 ? 0 : (c == 0 ? -1 : 0); break;
        case 5: delta = subblock == 1 ? 0 : (c == 1 ? -1 : 0); break;
        case 6: delta = subblock == 1 ? 0 : (c == 2 ? -1 : 0); break;
        case 7: delta = subblock == 1 ? 0 : (c == 0 ? 1 : 0); break;
        case 11: delta = subblock == 1 ? 0 : (c == 1 ? 1 : 0); break;
        case 15: delta = subblock == 1 ? 0 : (c == 2 ? 1 : 0); break;
        case 18: delta = subblock == 1 ? (c == 0 ? -1 : 0) : 0; break;
        case 19: delta = subblock == 1 ? (c == 1 ? -1 : 0) : 0; break;
        case 20: delta = subblock == 1 ? (c == 2 ? -1 : 0) : 0; break;
        case 21: delta = subblock == 1 ? (c == 0 ? 1 : 0) : 0; break;
        case 24: delta = subblock == 1 ? (c == 1 ? 1 : 0) : 0; break;
        case 8: delta = subblock == 1 ? (c == 2 ? 1 : 0) : 0; break;
        case 10: delta = -2; break;
        case 27: delta = subblock == 1 ? 0 : -1; break;
        case 28: delta = subblock == 1 ? -1 : 1; break;
        case 29: delta = subblock == 1 ? 1 : 0; break;
        case 30: delta = subblock == 1 ? -1 : 0; break;
        case 31: delta = subblock == 1 ? 0 : 1; break;
        default: delta = (int)((bias / S_DIVS[c]) % 3) - 1; break;
        }
        int v = block_color.c[c];
        if (v == 0) {
            if (delta == -2) v += 3;
            else v += delta + 1;
        } else if (v == (int)limit) {
            v += delta - 1;
        } else {
            v += delta;
            if (v < 0 || v > (int)limit) v = (v - delta) - delta;
        }
        block_color.c[c] = (uint8_t)v;
    }
    return block_color;
}

/* etc.rs:261-275 */
static void write_solid_etc2_alpha_block(uint8_t out[8], uint8_t value)
{
    static const uint8_t tail[7] = {(1 << 4) | 13, 0x92, 0x49, 0x24, 0x92, 0x49, 0x24};
    out[0] = value;
    memcpy(out + 1, tail, 7);
}

/* etc.rs:297-307: centre of the EAC modifier table, f32-faithful */
int bu_oracle_eac_center_f32(int min_alpha, int max_alpha, unsigned table_index)
{
    int mod_min = ORC_ETC2_ALPHA_MOD[table_index][3];
    int mod_max = ORC_ETC2_ALPHA_MOD[table_index][7];
    int range = mod_max - mod_min;
    float amt = -((float)mod_min) / (float)range;
    float a = (float)min_alpha, b = (float)max_alpha;
    float v = a * (1.0f - amt) + b * amt;
    return (int)roundf(v); /* f32::round = half away from zero */
}
/* integer form used by the GPU path */
int bu_oracle_eac_center_int(int min_alpha, int max_alpha, unsigned table_index)
{
    int mod_min = ORC_ETC2_ALPHA_MOD[table_index][3];
    int mod_max = ORC_ETC2_ALPHA_MOD[table_index][7];
    int range = mod_max - mod_min;
    return (2 * (min_alpha * (range + mod_min) - max_alpha * mod_min) + range) / (2 * range);
}

/* etc.rs:277-341 */
static void write_etc2_alpha_block(uint8_t out[8], uint8_t etc2tm, const color_t rgba[16])
{
    if (etc2tm == 0) {
        write_solid_etc2_alpha_block(out, 255);
        return;
    }
    uint8_t min_alpha = 255, max_alpha = 0;
    for (int i = 0; i < 16; i++) {
        if (rgba[i].c[3] < min_alpha) min_alpha = rgba[i].c[3];
        if (rgba[i].c[3] > max_alpha) max_alpha = rgba[i].c[3];
    }
    if (min_alpha == max_alpha) {
        write_solid_etc2_alpha_block(out, min_alpha);
        return;
    }
    unsigned table_index = etc2tm & 15;
    int multiplier = etc2tm >> 4;
    const int8_t *mod_table = ORC_ETC2_ALPHA_MOD[table_index];
    int center = bu_oracle_eac_center_f32(min_alpha, max_alpha, table_index);
    uint8_t values[8];
    for (int k = 0; k < 8; k++) values[k] = (uint8_t)clampi(center + mod_table[k] * multiplier, 0, 255);
    uint64_t selectors = 0;
    for (int i = 0; i < 16; i++) {
        int a = rgba[i].c[3];
        int best = 0, best_d = 1 << 30;
        for (int k = 0; k < 8; k++) { /* min_by_key: first minimum wins */
            int d = abs((int)values[k] - a);
            if (d < best_d) {
                best_d = d;
                best = k;
            }
        }
        unsigned x = (unsigned)i / 4, y = (unsigned)i % 4;
        unsigned id = y * 4 + x;
        selectors |= (uint64_t)best << (45 - id * 3);
    }
    out[0] = (uint8_t)center;
    out[1] = etc2tm;
    for (int k = 2; k < 8; k++) out[k] = (uint8_t)(selectors >> (8 * (7 - k))); /* to_be_bytes()[2..8] */
}

/* etc.rs:32-201; output = 8-byte ETC1 colour block, alpha = optional 8-byte EAC block */
static int convert_etc(const uint8_t bytes[16], uint8_t output[8], uint8_t *alpha)
{
    reader_t r = {bytes, 16, 0};
    const mode_t_ *m;
    int st = decode_mode(&r, &m);
    if (st) return st;
    memset(output, 0, 8);
    writer_t w = {output, 8, 0};

    if (m->id == 8) { /* :43-76 */
        if (alpha) {
            color_t rgba = decode_mode8_rgba(&r);
            write_solid_etc2_alpha_block(alpha, rgba.c[3]);
        } else {
            rd_remove(&r, 32);
        }
        mode8_flags_t f = decode_mode8_etc1_flags(&r);
        if (!f.etc1d) { /* u8 arithmetic, release wrapping */
            wr_write(&w, 8, (uint8_t)((uint8_t)(f.etc1r << 4) | f.etc1r));
            wr_write(&w, 8, (uint8_t)((uint8_t)(f.etc1g << 4) | f.etc1g));
            wr_write(&w, 8, (uint8_t)((uint8_t)(f.etc1b << 4) | f.etc1b));
        } else {
            wr_write(&w, 8, (uint8_t)(f.etc1r << 3));
            wr_write(&w, 8, (uint8_t)(f.etc1g << 3));
            wr_write(&w, 8, (uint8_t)(f.etc1b << 3));
        }
        wr_write(&w, 8, (uint8_t)((uint8_t)(f.etc1i << 5) | (uint8_t)(f.etc1i << 2) | (uint8_t)(f.etc1d << 1)));
        static const uint8_t SEL[4] = {3, 2, 0, 1};
        uint8_t selector = SEL[f.etc1s];
        uint16_t s_lo = selector & 1, s_hi = selector >> 1;
        wr_write(&w, 16, (uint16_t)(0 - s_hi));
        wr_write(&w, 16, (uint16_t)(0 - s_lo));
        return ORC_OK;
    }

    trans_flags_t tf = decode_trans_flags(&r, m);
    color_t rgba[16];
    st = decode_block_to_rgba(bytes, rgba);
    if (st) return st;
    if (alpha) write_etc2_alpha_block(alpha, tf.etc2tm, rgba);

    if (!tf.etc1f) { /* :86-95 transpose */
        for (int y = 0; y < 3; y++)
            for (int x = y + 1; x < 4; x++) {
                color_t t = rgba[y * 4 + x];
                rgba[y * 4 + x] = rgba[x * 4 + y];
                rgba[x * 4 + y] = t;
            }
    }
    unsigned color_bits = !tf.etc1d ? 4 : 5;
    uint32_t limit = mask32(color_bits);

    color_t avg[2];
    memset(avg, 0, sizeof avg);
    for (int sb = 0; sb < 2; sb++) { /* :100-111 */
        uint16_t sum[4] = {0, 0, 0, 0};
        for (int i = 0; i < 8; i++)
            for (int c = 0; c < 4; c++) sum[c] = (uint16_t)(sum[c] + rgba[sb * 8 + i].c[c]);
        for (int c = 0; c < 3; c++) avg[sb].c[c] = (uint8_t)(((uint32_t)sum[c] * limit + 1020) / (8 * 255));
    }
    color_t c0, c1;
    if (tf.has_bias) {
        c0 = apply_etc1_bias(avg[0], tf.etc1bias, limit, 0);
        c1 = apply_etc1_bias(avg[1], tf.etc1bias, limit, 1);
    } else {
        c0 = avg[0];
        c1 = avg[1];
    }
    color_t block_colors[2][4];
    if (!tf.etc1d) { /* :122-129 */
        for (int c = 0; c < 3; c++) wr_write(&w, 8, (uint8_t)((uint8_t)(c0.c[c] << 4) | c1.c[c]));
        apply_mod_to_base_color(color_4_to_8(c0), tf.etc1i0, block_colors[0]);
        apply_mod_to_base_color(color_4_to_8(c1), tf.etc1i1, block_colors[1]);
    } else { /* :130-149 */
        int16_t d[3];
        for (int c = 0; c < 3; c++) d[c] = (int16_t)clampi((int16_t)c1.c[c] - (int16_t)c0.c[c], -4, 3);
        for (int c = 0; c < 3; c++) wr_write(&w, 8, (uint8_t)((uint8_t)(c0.c[c] << 3) | (uint8_t)(d[c] & 7)));
        color_t c1d;
        for (int c = 0; c < 3; c++) c1d.c[c] = (uint8_t)((int16_t)c0.c[c] + d[c]);
        c1d.c[3] = 255;
        apply_mod_to_base_color(color_5_to_8(c0), tf.etc1i0, block_colors[0]);
        apply_mod_to_base_color(color_5_to_8(c1d), tf.etc1i1, block_colors[1]);
    }
    wr_write(&w, 8, (uint8_t)((uint8_t)(tf.etc1i0 << 5) | (uint8_t)(tf.etc1i1 << 2) | (uint8_t)(tf.etc1d << 1) | tf.etc1f)); /* :151-158 */

    selector_t sel;
    memset(&sel, 0, sizeof sel);
    static const int LUM[3] = {108, 366, 38};
    for (unsigned sb = 0; sb < 2; sb++) { /* :162-196 */
        int lums[4];
        for (int k = 0; k < 4; k++) {
            lums[k] = 0;
            for (int c = 0; c < 3; c++) lums[k] += block_colors[sb][k].c[c] * LUM[c];
        }
        int l01 = (lums[0] + lums[1]) / 2, l12 = (lums[1] + lums[2]) / 2, l23 = (lums[2] + lums[3]) / 2;
        for (unsigned i = 0; i < 8; i++) {
            const color_t *c = &rgba[sb * 8 + i];
            int lum = c->c[0] * LUM[0] + c->c[1] * LUM[1] + c->c[2] * LUM[2];
            uint8_t s = (uint8_t)((lum >= l01) + (lum >= l12) + (lum >= l23));
            unsigned x = i & 3, y = 2 * sb + (i >> 2);
            if (tf.etc1f) sel_set(&sel, x, y, s);
            else sel_set(&sel, y, x, s);
        }
    }
    uint32_t sb32 = (uint32_t)sel.etc1_bytes[0] | (uint32_t)sel.etc1_bytes[1] << 8 | (uint32_t)sel.etc1_bytes[2] << 16 |
                    (uint32_t)sel.etc1_bytes[3] << 24;
    wr_write(&w, 32, sb32);
    return ORC_OK;
}

/* ================================================================== exported per-block API (lib.rs:29-53) */
enum { BU_T_ASTC = 0, BU_T_BC7 = 1, BU_T_ETC1 = 2, BU_T_ETC2 = 3, BU_T_RGBA = 4 };

int bu_oracle_block_to_rgba(const uint8_t in[16], uint8_t out[64])
{
    color_t px[16];
    int st = decode_block_to_rgba(in, px);
    if (st) return st;
    for (int i = 0; i < 16; i++) memcpy(out + 4 * i, px[i].c, 4);
    return ORC_OK;
}
int bu_oracle_block_to_astc(const uint8_t in[16], uint8_t out[16]) { return convert_astc(in, out); }
int bu_oracle_block_to_bc7(const uint8_t in[16], uint8_t out[16]) { return convert_bc7(in, out); }
int bu_oracle_block_to_etc1(const uint8_t in[16], uint8_t out[8]) { return convert_etc(in, out, NULL); } /* etc.rs:11-17 */
int bu_oracle_block_to_etc2(const uint8_t in[16], uint8_t out[16])                                       /* etc.rs:19-30 */
{
    memset(out, 0, 16);
    return convert_etc(in, out + 8, out);
}

static size_t out_block_size(int target) { return target == BU_T_ETC1 ? 8 : (target == BU_T_RGBA ? 64 : 16); }

static int block_any(int target, const uint8_t *in, uint8_t *out)
{
    switch (target) {
    case BU_T_ASTC: return convert_astc(in, out);
    case BU_T_BC7: return convert_bc7(in, out);
    case BU_T_ETC1: return convert_etc(in, out, NULL);
    case BU_T_ETC2: return bu_oracle_block_to_etc2(in, out);
    default: return bu_oracle_block_to_rgba(in, out);
    }
}

/* uastc.rs:112-165 Decoder::transcode / transcode_into: first Err aborts the slice.
 * Returns the status; *first_bad = index of the failing block. */
int bu_oracle_transcode(int target, const uint8_t *in, size_t in_bytes, uint8_t *out, size_t *first_bad)
{
    if (in_bytes % 16) return ORC_ERR_LENGTH; /* uastc.rs:54-59 */
    size_t n = in_bytes / 16, obs = out_block_size(target);
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp[64];
        int st = block_any(target, in + 16 * i, tmp);
        if (st) {
            if (first_bad) *first_bad = i;
            return st;
        }
        memcpy(out + obs * i, tmp, obs);
    }
    return ORC_OK;
}

/* uastc.rs:89-110 Decoder::decode_to_rgba: row-major image, pitch 4*blocks_per_row pixels */
int bu_oracle_decode_to_rgba(const uint8_t *in, size_t in_bytes, size_t blocks_per_row, uint8_t *out, size_t *first_bad)
{
    if (in_bytes % 16) return ORC_ERR_LENGTH;
    size_t n = in_bytes / 16;
    size_t stride = 4 * blocks_per_row;
    for (size_t i = 0; i < n; i++) {
        color_t px[16];
        int st = decode_block_to_rgba(in + 16 * i, px);
        if (st) {
            if (first_bad) *first_bad = i;
            return st;
        }
        size_t bx = i % blocks_per_row, by = i / blocks_per_row;
        for (size_t y = 0; y < 4; y++) {
            size_t start = (4 * by + y) * stride + 4 * bx;
            memcpy(out + 4 * start, px[4 * y].c, 16);
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ threaded slice driver (CPU baseline timing) */
typedef struct {
    int target;
    const uint8_t *in;
    uint8_t *out;
    size_t begin, end;
    int status;
} job_t;

static void *job_main(void *p)
{
    job_t *j = (job_t *)p;
    size_t obs = out_block_size(j->target);
    j->status = 0;
    for (size_t i = j->begin; i < j->end; i++) {
        uint8_t tmp[64];
        int st = block_any(j->target, j->in + 16 * i, tmp);
        if (st) {
            j->status = st;
            break;
        }
        memcpy(j->out + obs * i, tmp, obs);
    }
    return NULL;
}

/* Same work as bu_oracle_transcode for block-linear targets, blocks split into `threads`
 * equal contiguous ranges.  RGBA here is block-linear (64 B per block). */
int bu_oracle_transcode_mt(int target, const uint8_t *in, size_t in_bytes, uint8_t *out, int threads)
{
    if (in_bytes % 16) return ORC_ERR_LENGTH;
    size_t n = in_bytes / 16;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t].target = target;
        jobs[t].in = in;
        jobs[t].out = out;
        jobs[t].begin = n * (size_t)t / (size_t)threads;
        jobs[t].end = n * (size_t)(t + 1) / (size_t)threads;
        jobs[t].status = 0;
        if (t > 0) pthread_create(&th[t], NULL, job_main, &jobs[t]);
    }
    job_main(&jobs[0]);
    int st = jobs[0].status;
    for (int t = 1; t < threads; t++) {
        pthread_join(th[t], NULL);
        if (!st) st = jobs[t].status;
    }
    return st;
}

/* ================================================================== ETC1S block back-end (basis_lz/mod.rs) */
/* Codebook entry layouts of this oracle:
 *   endpoint: 4 bytes {r5, g5, b5, inten}            (basis_lz/mod.rs:518-522)
 *   selector: 8 bytes {rows[4], etc1_bytes[4]}       (etc.rs:343-350)
 * idx: per block {endpoint_index u16, selector_index u16}, raster order (mod.rs:43-48). */

/* etc.rs:363-393 applied to 4 raw selector rows (mod.rs:527-583 builds codebook entries this way) */
void bu_oracle_selector_from_rows(const uint8_t rows[4], uint8_t out8[8])
{
    selector_t s;
    memset(&s, 0, sizeof s);
    for (unsigned y = 0; y < 4; y++)
        for (unsigned x = 0; x < 4; x++) sel_set(&s, x, y, (rows[y] >> (x * 2)) & 3);
    memcpy(out8, s.selectors, 4);
    memcpy(out8 + 4, s.etc1_bytes, 4);
}

/* basis_lz/mod.rs:153-186 (closure block_to_etc1 :163-181) */
void bu_oracle_etc1s_to_etc1(const uint16_t *idx, size_t n_blocks, const uint8_t *endpoints, const uint8_t *selectors, uint8_t *out)
{
    for (size_t i = 0; i < n_blocks; i++) {
        const uint8_t *ep = endpoints + 4 * (size_t)idx[2 * i];
        const uint8_t *sl = selectors + 8 * (size_t)idx[2 * i + 1];
        uint8_t *b = out + 8 * i;
        b[0] = (uint8_t)(ep[0] << 3);
        b[1] = (uint8_t)(ep[1] << 3);
        b[2] = (uint8_t)(ep[2] << 3);
        b[3] = (uint8_t)((uint8_t)(ep[3] << 5) | (uint8_t)(ep[3] << 2) | 3);
        memcpy(b + 4, sl + 4, 4);
    }
}

/* basis_lz/mod.rs:97-151 (closure block_to_rgba :122-146); alpha_idx may be NULL */
void bu_oracle_etc1s_to_rgba(const uint16_t *idx, const uint16_t *alpha_idx, size_t nbx, size_t nby, const uint8_t *endpoints,
                             const uint8_t *selectors, uint8_t *out)
{
    size_t stride = nbx * 4;
    for (int pass = 0; pass < 2; pass++) {
        const uint16_t *ix = pass == 0 ? idx : alpha_idx;
        if (!ix) break;
        for (size_t by = 0; by < nby; by++)
            for (size_t bx = 0; bx < nbx; bx++) {
                size_t i = by * nbx + bx;
                const uint8_t *ep = endpoints + 4 * (size_t)ix[2 * i];
                const uint8_t *sl = selectors + 8 * (size_t)ix[2 * i + 1];
                color_t c5 = {{ep[0], ep[1], ep[2], 0}};
                color_t colors[4];
                apply_mod_to_base_color(color_5_to_8(c5), ep[3], colors);
                selector_t s;
                memcpy(s.selectors, sl, 4);
                memcpy(s.etc1_bytes, sl + 4, 4);
                for (unsigned y = 0; y < 4; y++)
                    for (unsigned x = 0; x < 4; x++) {
                        unsigned k = sel_get(&s, x, y);
                        size_t gid = (bx * 4 + x) + (by * 4 + y) * stride;
                        if (pass == 0) memcpy(out + 4 * gid, colors[k].c, 4);
                        else out[4 * gid + 3] = colors[k].c[1];
                    }
            }
    }
}

/* Independent ETC1 decoder for the self-consistency check of SURVEY.md 8c: decode an ETC1 block
 * whose diff bit is set (differential mode, no overflow checks needed for delta 0) or clear. */
void bu_oracle_decode_etc1_block(const uint8_t b[8], uint8_t out[64])
{
    int diff = (b[3] >> 1) & 1, flip = b[3] & 1;
    color_t base[2];
    if (diff) {
        color_t c0 = {{(uint8_t)(b[0] >> 3), (uint8_t)(b[1] >> 3), (uint8_t)(b[2] >> 3), 0}};
        color_t c1;
        for (int c = 0; c < 3; c++) {
            int d = b[c] & 7;
            if (d >= 4) d -= 8;
            c1.c[c] = (uint8_t)(c0.c[c] + d);
        }
        c1.c[3] = 0;
        base[0] = color_5_to_8(c0);
        base[1] = color_5_to_8(c1);
    } else {
        color_t c0 = {{(uint8_t)(b[0] >> 4), (uint8_t)(b[1] >> 4), (uint8_t)(b[2] >> 4), 0}};
        color_t c1 = {{(uint8_t)(b[0] & 15), (uint8_t)(b[1] & 15), (uint8_t)(b[2] & 15), 0}};
        base[0] = color_4_to_8(c0);
        base[1] = color_4_to_8(c1);
    }
    unsigned t0 = (b[3] >> 5) & 7, t1 = (b[3] >> 2) & 7;
    color_t cols[2][4];
    apply_mod_to_base_color(base[0], t0, cols[0]);
    apply_mod_to_base_color(base[1], t1, cols[1]);
    static const uint8_t ETC1_TO_SEL[4] = {2, 3, 1, 0}; /* inverse of etc.rs:433 */
    for (unsigned x = 0; x < 4; x++)
        for (unsigned y = 0; y < 4; y++) {
            unsigned pix = x * 4 + y;
            unsigned msb = (b[4 + 1 - pix / 8] >> (pix % 8)) & 1; /* etc.rs:379-392 */
            unsigned lsb = (b[4 + 3 - pix / 8] >> (pix % 8)) & 1;
            unsigned k = ETC1_TO_SEL[msb * 2 + lsb];
            unsigned sb = flip ? (y >= 2) : (x >= 2);
            memcpy(out + 4 * (y * 4 + x), cols[sb][k].c, 4);
        }
}

/* ================================================================== test helpers */
/* block-linear batch that does not stop at errors: st[i] = status of block i (fuzz parity) */
void bu_oracle_batch(int target, const uint8_t *in, size_t n_blocks, uint8_t *out, uint8_t *st)
{
    size_t obs = out_block_size(target);
    for (size_t i = 0; i < n_blocks; i++) {
        uint8_t tmp[64];
        memset(tmp, 0, sizeof tmp);
        st[i] = (uint8_t)block_any(target, in + 16 * i, tmp);
        memcpy(out + obs * i, tmp, obs);
    }
}

/* Exhaustive proof helper: determine_shared_pbits, f32 form vs integer form,
 * over every RGB endpoint pair whose channels are multiples of 17 (the only inputs UASTC mode 2 can
 * produce).  Returns the number of (decision or endpoint) mismatches; *ties = exact-tie count. */
uint64_t bu_oracle_prove_shared_pbits(uint64_t *ties)
{
    uint64_t bad = 0, nt = 0;
    for (unsigned v = 0; v < (1u << 24); v++) {
        uint8_t lo[4], hi[4], lo2[4], hi2[4];
        for (int c = 0; c < 3; c++) {
            lo[c] = lo2[c] = (uint8_t)(17 * ((v >> (4 * c)) & 15));
            hi[c] = hi2[c] = (uint8_t)(17 * ((v >> (12 + 4 * c)) & 15));
        }
        lo[3] = lo2[3] = hi[3] = hi2[3] = 255;
        int p1 = bu_oracle_shared_pbits_f32(3, 6, lo, hi);
        int p2 = bu_oracle_shared_pbits_int(3, 6, lo2, hi2);
        if (p1 != p2 || memcmp(lo, lo2, 3) || memcmp(hi, hi2, 3)) bad++;
        (void)nt;
    }
    if (ties) *ties = nt;
    return bad;
}

/* determine_unique_pbits f32 vs integer over random endpoints (decision is per endpoint) */
uint64_t bu_oracle_prove_unique_pbits(unsigned total_comps, unsigned comp_bits, uint64_t n, uint64_t seed)
{
    uint64_t bad = 0, s = seed * 6364136223846793005ull + 1442695040888963407ull;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t lo[4], hi[4], lo2[4], hi2[4], p[2];
        for (int c = 0; c < 4; c++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            lo[c] = lo2[c] = (uint8_t)(s >> 33);
            hi[c] = hi2[c] = (uint8_t)(s >> 41);
        }
        bu_oracle_unique_pbits_f32(total_comps, comp_bits, lo, hi, p);
        int q0 = bu_oracle_unique_pbit_int(total_comps, comp_bits, lo2);
        int q1 = bu_oracle_unique_pbit_int(total_comps, comp_bits, hi2);
        if (q0 != p[0] || q1 != p[1] || memcmp(lo, lo2, 4) || memcmp(hi, hi2, 4)) bad++;
    }
    return bad;
}

/* EAC centre, f32 vs integer, all 16 tables x all 0 <= min < max <= 255 */
uint64_t bu_oracle_prove_eac_center(void)
{
    uint64_t bad = 0;
    for (unsigned t = 0; t < 16; t++)
        for (int mn = 0; mn < 256; mn++)
            for (int mx = mn + 1; mx < 256; mx++)
                if (bu_oracle_eac_center_f32(mn, mx, t) != bu_oracle_eac_center_int(mn, mx, t)) bad++;
    return bad;
}

/* ------------------------------------------------------------------ self-test of the bit readers / writers above
 * The reference's own unit tests for its bit I/O (bitreader.rs:63-100 test_bitreader_lsb; bitwriter.rs:118-225
 * test_bitwriter_lsb, _msb_rev_bytes_rev_bits, _msb_rev_bytes) run the 16 patterns "64 alternating bits with the four 16-bit
 * segments selectively inverted" through every (offset, length) pair below 32.  The same procedure on the restatements: the
 * known answers are arithmetic on the pattern, so this pins rd_* / wr_* / wrr_* to the reference's test suite without a corpus.
 * Returns the number of mismatches (0 = all 4 x 16 x 32 x 32 cases agree). */
static uint64_t selftest_pattern(uint64_t i)
{
    const uint64_t pattern = 0x5555555555555555ull, segment = 0xFFFFull;
    const uint64_t x = (segment * ((i >> 3) & 1)) << 48 | (segment * ((i >> 2) & 1)) << 32 | (segment * ((i >> 1) & 1)) << 16 | (segment * (i & 1));
    return pattern ^ x;
}
static uint64_t mask64(unsigned n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
static uint64_t le64(const uint8_t b[8])
{
    uint64_t v = 0;
    for (int k = 7; k >= 0; k--) v = (v << 8) | b[k];
    return v;
}
static uint64_t rev64(uint64_t v)
{
    uint64_t r = 0;
    for (int k = 0; k < 64; k++)
        if (v & (1ull << k)) r |= 1ull << (63 - k);
    return r;
}
uint64_t bu_oracle_selftest_bitio(void)
{
    uint64_t bad = 0;
    for (uint64_t i = 0; i < 16; i++) {
        const uint64_t data = selftest_pattern(i);
        uint8_t src[8];
        for (int k = 0; k < 8; k++) src[k] = (uint8_t)(data >> (8 * k));
        for (unsigned len = 0; len < 32; len++)
            for (unsigned offset = 0; offset < 32; offset++) {
                /* bitreader.rs:63-100 */
                reader_t r = {src, 8, 0};
                if (rd_read(&r, offset) != (uint32_t)(data & mask64(offset))) bad++;
                if (rd_read(&r, len) != (uint32_t)((data >> offset) & mask64(len))) bad++;
                /* bitwriter.rs:132-158 (LSB first) */
                uint8_t out[8] = {0};
                writer_t w = {out, 8, 0};
                wr_write(&w, offset, (uint32_t)data);
                wr_write(&w, len, (uint32_t)(data >> offset));
                if (le64(out) != (data & mask64(offset + len))) bad++;
                /* bitwriter.rs:160-190 (MSB first from the end of the buffer, values bit-reversed) */
                uint8_t out2[8] = {0};
                writer_rev_t wr = {out2, 8, 64};
                wrr_write_rev(&wr, offset, (uint32_t)data);
                wrr_write_rev(&wr, len, (uint32_t)(data >> offset));
                if (rev64(le64(out2)) != (data & mask64(offset + len))) bad++;
                /* bitwriter.rs:192-224 (MSB first, values as they are; wrapping_shr masks the shift to 6 bits) */
                uint8_t out3[8] = {0};
                writer_rev_t w3 = {out3, 8, 64};
                wrr_write(&w3, offset, (uint32_t)(data >> ((64 - offset) & 63)));
                wrr_write(&w3, len, (uint32_t)(data >> ((64 - offset - len) & 63)));
                if (le64(out3) != (data & ~mask64(64 - offset - len))) bad++;
            }
    }
    return bad;
}
