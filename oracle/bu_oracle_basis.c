/*
 * bu_oracle_basis.c -- CPU ORACLE, container + BasisLZ half (TEST INFRASTRUCTURE ONLY, see bu_oracle.c).
 *
 * Plain-C restatement of the reference's host side:
 *   src/basis.rs            header (77 B), slice descs (23 B), crc16, the six read_to_* drivers
 *   src/bytereader.rs       little-endian field reads
 *   src/basis_lz/huffman.rs Huffman table records and canonical decoding tables
 *   src/basis_lz/mod.rs     codebook decode, tables section, decode_blocks (the serial symbol loop)
 * Each function cites the lines it follows.  Pinning: the header field layout is pinned by the
 * reference's own unit test (basis.rs:578-620, bytes 0..76 -> fields; tests/test_container.py); crc16 is
 * CRC-16/GENIBUS as the reference's comment says (basis.rs:422) and is pinned by that algorithm's
 * published check value 0xD64E for "123456789".  The BasisLZ path has no runnable vectors in the
 * reference (its corpus tests are #[ignore]d, corpus absent): parity unpinned, restated by reading.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- from bu_oracle.c (same shared object) ---- */
int bu_oracle_transcode(int target, const uint8_t *in, size_t in_bytes, uint8_t *out, size_t *first_bad);
int bu_oracle_decode_to_rgba(const uint8_t *in, size_t in_bytes, size_t blocks_per_row, uint8_t *out, size_t *first_bad);
void bu_oracle_selector_from_rows(const uint8_t rows[4], uint8_t out8[8]);
void bu_oracle_etc1s_to_etc1(const uint16_t *idx, size_t n_blocks, const uint8_t *endpoints, const uint8_t *selectors, uint8_t *out);
void bu_oracle_etc1s_to_rgba(const uint16_t *idx, const uint16_t *alpha_idx, size_t nbx, size_t nby, const uint8_t *endpoints,
                             const uint8_t *selectors, uint8_t *out);

/* statuses of this half (0 = ok); 1..3 are the block statuses of bu_oracle.c */
enum {
    OB_OK = 0,
    OB_ERR_SIG = 9,             /* "Sig mismatch, not a Basis Universal file"  basis.rs:309 */
    OB_ERR_HEADER_TRUNCATED = 10, /* "Expected at least 77 byte header, ..."   basis.rs:313 */
    OB_ERR_HEADER_SIZE = 11,    /* "File specified unexpected header size..." basis.rs:323 */
    OB_ERR_HEADER_CRC = 12,     /* "Header CRC16 failed"                      basis.rs:332 */
    OB_ERR_DATA_CRC = 13,       /* "Data CRC16 failed"                        basis.rs:12 */
    OB_ERR_TEX_FORMAT = 14,     /* "Unknown texture format"                   basis.rs:404 */
    OB_ERR_SLICE_DESC = 15,     /* "Expected 23 byte slice desc at pos ..."   basis.rs:350 */
    OB_ERR_ALPHA_SLICES = 16,   /* odd slice count / missing alpha flag / size mismatch  basis.rs:19,29,34 */
    OB_ERR_UNSUPPORTED = 17,    /* unimplemented!()                           basis.rs:88,141,171,200,229,258 */
    OB_ERR_BASISLZ = 18,        /* any Err of basis_lz (huffman.rs, mod.rs:531-537) */
    OB_ERR_BOUNDS = 19          /* slice/section outside the file or an assert! of decode_blocks: the reference panics */
};

/* ------------------------------------------------------------------ bytereader.rs */
static uint32_t rd_le(const uint8_t *p, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

/* basis.rs:364-372 */
uint16_t bu_oracle_crc16(const uint8_t *r, size_t len, uint16_t crc)
{
    crc = (uint16_t)~crc;
    for (size_t i = 0; i < len; i++) {
        uint16_t q = (uint16_t)((uint16_t)r[i] ^ (crc >> 8));
        uint16_t k = (uint16_t)((q >> 4) ^ q);
        crc = (uint16_t)((((crc << 8) ^ k) ^ (k << 5)) ^ (k << 12));
    }
    return (uint16_t)~crc;
}

/* basis.rs:417-454: 26 fields, all widened to u32 in declaration order */
typedef struct {
    uint32_t f[26];
} ob_header;
enum {
    H_SIG, H_VER, H_HEADER_SIZE, H_HEADER_CRC16, H_DATA_SIZE, H_DATA_CRC16, H_TOTAL_SLICES, H_TOTAL_IMAGES, H_TEX_FORMAT, H_FLAGS,
    H_TEX_TYPE, H_US_PER_FRAME, H_RESERVED, H_USERDATA0, H_USERDATA1, H_TOTAL_ENDPOINTS, H_ENDPOINT_OFS, H_ENDPOINT_SIZE,
    H_TOTAL_SELECTORS, H_SELECTOR_OFS, H_SELECTOR_SIZE, H_TABLES_OFS, H_TABLES_SIZE, H_SLICE_DESC_OFS, H_EXT_OFS, H_EXT_SIZE
};
/* byte width of each field in file order (basis.rs:475-516) */
static const uint8_t H_WIDTH[26] = {2, 2, 2, 2, 4, 2, 3, 3, 1, 2, 1, 3, 4, 4, 4, 2, 4, 3, 2, 4, 3, 4, 4, 4, 4, 4};

/* basis.rs:475-516 Header::from_file_bytes (needs >= 77 bytes) */
void bu_oracle_header_from_bytes(const uint8_t *buf, uint32_t out26[26])
{
    size_t pos = 0;
    for (int i = 0; i < 26; i++) {
        out26[i] = rd_le(buf + pos, H_WIDTH[i]);
        pos += H_WIDTH[i];
    }
}

/* basis.rs:307-336 */
int bu_oracle_read_header(const uint8_t *bytes, size_t len, uint32_t out26[26])
{
    if (len < 2 || rd_le(bytes, 2) != 0x4273) return OB_ERR_SIG; /* LE::read_u16 would panic below 2 bytes */
    if (len < 77) return OB_ERR_HEADER_TRUNCATED;
    bu_oracle_header_from_bytes(bytes, out26);
    if (out26[H_HEADER_SIZE] != 77) return OB_ERR_HEADER_SIZE;
    if (bu_oracle_crc16(bytes + 8, 77 - 8, 0) != out26[H_HEADER_CRC16]) return OB_ERR_HEADER_CRC;
    return OB_OK;
}

/* basis.rs:519-572: image_index u24, level u8, flags u8, orig_w u16, orig_h u16, nbx u16, nby u16, file_ofs u32, file_size u32, crc16 u16 */
typedef struct {
    uint32_t image_index, level_index, flags, orig_width, orig_height, num_blocks_x, num_blocks_y, file_ofs, file_size, crc16;
} ob_slice;

/* basis.rs:343-362 */
int bu_oracle_read_slice_descs(const uint8_t *bytes, size_t len, const uint32_t hdr[26], ob_slice *out, size_t max)
{
    size_t start = hdr[H_SLICE_DESC_OFS], count = hdr[H_TOTAL_SLICES];
    for (size_t i = 0; i < count; i++) {
        size_t s = start + i * 23;
        if (s > len) return OB_ERR_BOUNDS; /* &bytes[slice_start..] panics */
        if (len - s < 23) return OB_ERR_SLICE_DESC;
        if (i >= max) return OB_ERR_BOUNDS;
        const uint8_t *p = bytes + s;
        ob_slice d = {rd_le(p, 3),      p[3],           p[4],           rd_le(p + 5, 2),  rd_le(p + 7, 2),
                      rd_le(p + 9, 2),  rd_le(p + 11, 2), rd_le(p + 13, 4), rd_le(p + 17, 4), rd_le(p + 21, 2)};
        out[i] = d;
    }
    return OB_OK;
}

/* ------------------------------------------------------------------ bitreader.rs (arbitrary length) */
typedef struct {
    const uint8_t *bytes;
    size_t len, bit_pos;
} breader;
static uint32_t br_peek(const breader *r, unsigned count)
{
    size_t byte = r->bit_pos / 8;
    unsigned bit = (unsigned)(r->bit_pos % 8);
    uint32_t result = (byte < r->len ? r->bytes[byte] : 0) >> bit;
    unsigned read = 8 - bit;
    byte++;
    while (read < count) {
        result |= (uint32_t)(byte < r->len ? r->bytes[byte] : 0) << read;
        read += 8;
        byte++;
    }
    return count >= 32 ? result : (result & ((1u << count) - 1u));
}
static uint32_t br_read(breader *r, unsigned count)
{
    uint32_t v = br_peek(r, count);
    r->bit_pos += count;
    return v;
}

/* ------------------------------------------------------------------ huffman.rs */
typedef struct {
    uint16_t symbol;
    uint8_t code_size;
} hentry;
typedef struct {
    hentry *lookup;
    unsigned max_code_size;
} htable;

static uint32_t rev32(uint32_t v)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++)
        if (v & (1u << i)) r |= 1u << (31 - i);
    return r;
}

/* huffman.rs:133-184 */
static int htable_from_sizes(const uint8_t *code_sizes, size_t n, htable *t)
{
    uint32_t syms_using[17] = {0};
    unsigned max_code_size = 0;
    for (size_t i = 0; i < n; i++) {
        if (code_sizes[i] > 16) return OB_ERR_BOUNDS; /* index out of bounds panic */
        syms_using[code_sizes[i]]++;
        if (code_sizes[i] > max_code_size) max_code_size = code_sizes[i];
    }
    uint32_t total = 0, next_code[17] = {0};
    syms_using[0] = 0;
    for (int bits = 1; bits < 17; bits++) {
        total = (total + syms_using[bits - 1]) << 1;
        next_code[bits] = total;
    }
    t->lookup = (hentry *)calloc((size_t)1 << max_code_size, sizeof(hentry));
    t->max_code_size = max_code_size;
    for (size_t sym = 0; sym < n; sym++) {
        unsigned size = code_sizes[sym];
        if (size != 0) {
            uint16_t code = (uint16_t)(rev32(next_code[size]) >> (32 - size));
            uint16_t variant_count = (uint16_t)(1u << (max_code_size - size));
            for (uint16_t fill = 0; fill < variant_count; fill++) {
                size_t id = (uint16_t)((uint16_t)(fill << (size & 15)) | code); /* u16 wrapping_shl */
                if (id >= ((size_t)1 << max_code_size)) { /* lookup[id] panics */
                    free(t->lookup);
                    t->lookup = NULL;
                    return OB_ERR_BOUNDS;
                }
                t->lookup[id].symbol = (uint16_t)sym;
                t->lookup[id].code_size = (uint8_t)size;
            }
            next_code[size]++;
        }
    }
    for (int i = 0; i < 17; i++)
        if (next_code[i] > 65536u) {
            free(t->lookup);
            t->lookup = NULL;
            return OB_ERR_BASISLZ; /* "Code lengths are invalid, codes don't fit into 16 bits" */
        }
    return OB_OK;
}
static void htable_free(htable *t)
{
    free(t->lookup);
    t->lookup = NULL;
}
/* huffman.rs:186-198 */
static int htable_decode(const htable *t, breader *r, uint16_t *sym)
{
    uint32_t bits = br_peek(r, t->max_code_size);
    hentry e = t->lookup[bits];
    if (e.code_size > 0) {
        r->bit_pos += e.code_size;
        *sym = e.symbol;
        return OB_OK;
    }
    return OB_ERR_BASISLZ; /* "No matching code found in the decoding table" */
}

/* huffman.rs:43-118 */
static int read_huffman_table(breader *r, htable *out)
{
    size_t total_used_syms = br_read(r, 14);
    htable cl;
    {
        size_t num_cl = br_read(r, 5);
        static const uint8_t indices[21] = {17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16};
        uint8_t sizes[21] = {0};
        if (num_cl > 21) return OB_ERR_BOUNDS; /* indices[i] panics */
        for (size_t i = 0; i < num_cl; i++) sizes[indices[i]] = (uint8_t)br_read(r, 3);
        int st = htable_from_sizes(sizes, 21, &cl);
        if (st) return st;
    }
    uint8_t *sym_sizes = (uint8_t *)malloc(total_used_syms + 160);
    size_t n = 0;
    int st = OB_OK;
    while (n < total_used_syms) {
        uint16_t s;
        st = htable_decode(&cl, r, &s);
        if (st) break;
        if (s <= 16) {
            sym_sizes[n++] = (uint8_t)s;
        } else if (s == 17 || s == 18) {
            size_t count = s == 17 ? 3 + br_read(r, 3) : 11 + br_read(r, 7);
            for (size_t k = 0; k < count; k++) sym_sizes[n++] = 0;
        } else {
            if (n == 0 || sym_sizes[n - 1] == 0) { /* huffman.rs:82-107 */
                st = OB_ERR_BASISLZ;
                break;
            }
            uint8_t prev = sym_sizes[n - 1];
            size_t count = s == 19 ? 3 + br_read(r, 2) : 7 + br_read(r, 7);
            for (size_t k = 0; k < count; k++) sym_sizes[n++] = prev;
        }
    }
    htable_free(&cl);
    if (!st) st = htable_from_sizes(sym_sizes, n, out); /* the Vec may have grown past total_used_syms */
    free(sym_sizes);
    return st;
}

/* ------------------------------------------------------------------ basis_lz/mod.rs */
typedef struct {
    htable endpoint_pred, delta_endpoint, selector, history_rle;
    uint32_t history_size;
    int is_video;
    size_t n_endpoints, n_selectors;
    uint8_t *endpoints; /* 4 B each: r5,g5,b5,inten  (mod.rs:518-522) */
    uint8_t *selectors; /* 8 B each: rows[4], etc1_bytes[4] (etc.rs:343-350) */
} lz_decoder;

static void lz_free(lz_decoder *d)
{
    htable_free(&d->endpoint_pred);
    htable_free(&d->delta_endpoint);
    htable_free(&d->selector);
    htable_free(&d->history_rle);
    free(d->endpoints);
    free(d->selectors);
    memset(d, 0, sizeof *d);
}

/* mod.rs:461-516 */
static int lz_decode_endpoints(size_t num, const uint8_t *bytes, size_t len, uint8_t *out)
{
    breader r = {bytes, len, 0};
    htable m0 = {0}, m1 = {0}, m2 = {0}, mi = {0};
    int st = read_huffman_table(&r, &m0);
    if (!st) st = read_huffman_table(&r, &m1);
    if (!st) st = read_huffman_table(&r, &m2);
    if (!st) st = read_huffman_table(&r, &mi);
    if (!st) {
        int grayscale = (int)br_read(&r, 1);
        uint8_t prev_color[3] = {16, 16, 16};
        uint32_t prev_inten = 0;
        for (size_t i = 0; i < num && !st; i++) {
            uint16_t s;
            st = htable_decode(&mi, &r, &s);
            if (st) break;
            uint8_t *e = out + 4 * i;
            e[3] = (uint8_t)((s + prev_inten) & 7);
            prev_inten = e[3];
            int channels = grayscale ? 1 : 3;
            for (int c = 0; c < channels; c++) {
                const htable *m = prev_color[c] <= 9 ? &m0 : (prev_color[c] <= 21 ? &m1 : &m2);
                st = htable_decode(m, &r, &s);
                if (st) break;
                uint8_t v = (uint8_t)((uint8_t)(prev_color[c] + (uint8_t)s) & 31);
                e[c] = v;
                prev_color[c] = v;
            }
            if (grayscale) {
                e[1] = e[0];
                e[2] = e[0];
            }
        }
    }
    htable_free(&m0);
    htable_free(&m1);
    htable_free(&m2);
    htable_free(&mi);
    return st;
}

/* mod.rs:524-583 */
static int lz_decode_selectors(size_t num, const uint8_t *bytes, size_t len, uint8_t *out)
{
    breader r = {bytes, len, 0};
    int global = (int)br_read(&r, 1), hybrid = (int)br_read(&r, 1), raw = (int)br_read(&r, 1);
    if (global || hybrid) return OB_ERR_BASISLZ; /* "... selector codebooks are not supported" */
    if (!raw) {
        htable m = {0};
        int st = read_huffman_table(&r, &m);
        if (st) return st;
        uint8_t prev[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < num; i++) {
            uint8_t rows[4];
            for (int y = 0; y < 4; y++) {
                if (i == 0) {
                    rows[y] = (uint8_t)br_read(&r, 8);
                } else {
                    uint16_t s;
                    st = htable_decode(&m, &r, &s);
                    if (st) {
                        htable_free(&m);
                        return st;
                    }
                    rows[y] = (uint8_t)((uint8_t)s ^ prev[y]);
                }
                prev[y] = rows[y];
            }
            bu_oracle_selector_from_rows(rows, out + 8 * i);
        }
        htable_free(&m);
    } else {
        for (size_t i = 0; i < num; i++) {
            uint8_t rows[4];
            for (int y = 0; y < 4; y++) rows[y] = (uint8_t)br_read(&r, 8);
            bu_oracle_selector_from_rows(rows, out + 8 * i);
        }
    }
    return OB_OK;
}

/* mod.rs:64-95 */
static int lz_new(lz_decoder *d, size_t n_endpoints, size_t n_selectors, const uint8_t *ep, size_t ep_len, const uint8_t *sel,
                  size_t sel_len, const uint8_t *tables, size_t tables_len, int is_video)
{
    memset(d, 0, sizeof *d);
    d->n_endpoints = n_endpoints;
    d->n_selectors = n_selectors;
    d->is_video = is_video;
    d->endpoints = (uint8_t *)calloc(n_endpoints ? n_endpoints : 1, 4);
    d->selectors = (uint8_t *)calloc(n_selectors ? n_selectors : 1, 8);
    int st = lz_decode_endpoints(n_endpoints, ep, ep_len, d->endpoints);
    if (!st) st = lz_decode_selectors(n_selectors, sel, sel_len, d->selectors);
    if (!st) {
        breader r = {tables, tables_len, 0};
        st = read_huffman_table(&r, &d->endpoint_pred);
        if (!st) st = read_huffman_table(&r, &d->delta_endpoint);
        if (!st) st = read_huffman_table(&r, &d->selector);
        if (!st) st = read_huffman_table(&r, &d->history_rle);
        if (!st) d->history_size = br_read(&r, 13);
    }
    if (st) lz_free(d);
    return st;
}

/* mod.rs:585-608 */
static int decode_vlc(breader *r, unsigned chunk_bits, uint32_t *out)
{
    uint32_t chunk_size = 1u << chunk_bits, chunk_mask = chunk_size - 1, v = 0;
    unsigned ofs = 0;
    for (;;) {
        uint32_t s = br_read(r, chunk_bits + 1);
        v |= (s & chunk_mask) << ofs;
        ofs += chunk_bits;
        if ((s & chunk_size) == 0) break;
        if (ofs >= 32) return OB_ERR_BOUNDS; /* panic!() */
    }
    *out = v;
    return OB_OK;
}

/* mod.rs:188-458; idx[2*i] = endpoint index, idx[2*i+1] = selector index, raster order */
static int lz_decode_blocks(const lz_decoder *d, size_t nbx, size_t nby, const uint8_t *data, size_t len, uint16_t *idx)
{
    breader r = {data, len, 0};
    const uint16_t num_endpoints = (uint16_t)d->n_endpoints, num_selectors = (uint16_t)d->n_selectors;
    typedef struct {
        uint16_t endpoint_index;
        uint8_t pred_bits;
    } bpred;
    bpred *preds[2];
    preds[0] = (bpred *)calloc(nbx ? nbx : 1, sizeof(bpred));
    preds[1] = (bpred *)calloc(nbx ? nbx : 1, sizeof(bpred));
    const uint16_t hist_first = num_selectors;
    const uint16_t hist_rle = (uint16_t)((uint16_t)d->history_size + hist_first);
    uint32_t cur_sel_rle = 0;
    uint8_t cur_pred_bits = 0, prev_pred_sym = 0;
    uint32_t pred_repeat = 0;
    uint16_t prev_endpoint = 0;
    uint16_t *prev_frame = (uint16_t *)calloc(nbx * nby ? nbx * nby * 2 : 1, sizeof(uint16_t)); /* re-zeroed per slice: mod.rs:236-237 */
    /* ApproxMoveToFront (mod.rs:610-656) */
    size_t hn = d->history_size;
    uint16_t *hist = (uint16_t *)calloc(hn ? hn : 1, sizeof(uint16_t));
    size_t rover = hn / 2;
    int st = OB_OK;

    for (size_t by = 0; by < nby && !st; by++) {
        const unsigned cur = (unsigned)(by & 1);
        for (size_t bx = 0; bx < nbx && !st; bx++) {
            if ((bx & 1) == 0) {
                if ((by & 1) == 0) {
                    if (pred_repeat != 0) {
                        pred_repeat--;
                        cur_pred_bits = prev_pred_sym;
                    } else {
                        uint16_t s;
                        st = htable_decode(&d->endpoint_pred, &r, &s);
                        if (st) break;
                        if (s == 256) { /* ENDPOINT_PRED_REPEAT_LAST_SYMBOL */
                            uint32_t v;
                            st = decode_vlc(&r, 4, &v);
                            if (st) break;
                            pred_repeat = v + 3 - 1;
                            cur_pred_bits = prev_pred_sym;
                        } else {
                            cur_pred_bits = (uint8_t)s;
                            prev_pred_sym = cur_pred_bits;
                        }
                    }
                    preds[cur ^ 1][bx].pred_bits = cur_pred_bits >> 4;
                } else {
                    cur_pred_bits = preds[cur][bx].pred_bits;
                }
            }
            const uint8_t pred = cur_pred_bits & 3;
            cur_pred_bits >>= 2;
            uint16_t endpoint_index;
            if (pred == 0) {
                if (bx == 0) { st = OB_ERR_BOUNDS; break; } /* assert!(block_x > 0) */
                endpoint_index = prev_endpoint;
            } else if (pred == 1) {
                if (by == 0) { st = OB_ERR_BOUNDS; break; }
                endpoint_index = preds[cur ^ 1][bx].endpoint_index;
            } else if (pred == 2) {
                if (d->is_video) {
                    endpoint_index = prev_frame[2 * (bx + by * nbx)];
                } else {
                    if (bx == 0 || by == 0) { st = OB_ERR_BOUNDS; break; }
                    endpoint_index = preds[cur ^ 1][bx - 1].endpoint_index;
                }
            } else {
                uint16_t s;
                st = htable_decode(&d->delta_endpoint, &r, &s);
                if (st) break;
                endpoint_index = (uint16_t)(s + prev_endpoint); /* release: wrapping u16 add */
                if (endpoint_index >= num_endpoints) endpoint_index = (uint16_t)(endpoint_index - num_endpoints);
            }
            preds[cur][bx].endpoint_index = endpoint_index;
            prev_endpoint = endpoint_index;

            uint16_t selector_index;
            if (!d->is_video || pred != 2) {
                uint16_t sel_sym;
                if (cur_sel_rle > 0) {
                    cur_sel_rle--;
                    sel_sym = num_selectors;
                } else {
                    uint16_t s;
                    st = htable_decode(&d->selector, &r, &s);
                    if (st) break;
                    if (s == hist_rle) {
                        uint16_t run;
                        st = htable_decode(&d->history_rle, &r, &run);
                        if (st) break;
                        if (run == 63) {
                            uint32_t v;
                            st = decode_vlc(&r, 7, &v);
                            if (st) break;
                            cur_sel_rle = 3 + v;
                        } else {
                            cur_sel_rle = 3u + run;
                        }
                        cur_sel_rle--;
                        sel_sym = num_selectors;
                    } else {
                        sel_sym = s;
                    }
                }
                if (sel_sym >= num_selectors) {
                    if (d->history_size == 0) { st = OB_ERR_BOUNDS; break; } /* assert! */
                    size_t hi = (size_t)(sel_sym - num_selectors);
                    if (hi >= hn) { st = OB_ERR_BOUNDS; break; }
                    selector_index = hist[hi];
                    if (hi != 0) { /* use_index: swap with index/2 */
                        uint16_t x = hist[hi / 2], y = hist[hi];
                        hist[hi / 2] = y;
                        hist[hi] = x;
                    }
                } else {
                    if (d->history_size > 0) { /* add */
                        hist[rover] = sel_sym;
                        rover++;
                        if (rover == hn) rover = hn / 2;
                    }
                    selector_index = sel_sym;
                }
            } else {
                selector_index = prev_frame[2 * (bx + by * nbx) + 1];
            }
            if (d->is_video) {
                prev_frame[2 * (bx + nbx * by)] = endpoint_index;
                prev_frame[2 * (bx + nbx * by) + 1] = selector_index;
            }
            if (endpoint_index >= num_endpoints || selector_index >= num_selectors) { st = OB_ERR_BOUNDS; break; } /* asserts :443-445 */
            idx[2 * (by * nbx + bx)] = endpoint_index;
            idx[2 * (by * nbx + bx) + 1] = selector_index;
        }
    }
    free(preds[0]);
    free(preds[1]);
    free(prev_frame);
    free(hist);
    return st;
}

/* ------------------------------------------------------------------ read_to_* drivers (basis.rs:8-260) */
typedef struct {
    uint32_t w, h, stride; /* lib.rs:63-68 */
    uint64_t offset, size; /* data = out + offset */
} ob_image;
enum { RD_RGBA = 0, RD_ETC1 = 1, RD_ETC2 = 2, RD_UASTC = 3, RD_ASTC = 4, RD_BC7 = 5 };

static int in_file(size_t len, size_t ofs, size_t size) { return ofs <= len && size <= len - ofs; }

/* basis.rs:262-298: note total_selectors is passed for BOTH codebook sizes (reference quirk) */
static int make_lz(const uint8_t *buf, size_t len, const uint32_t h[26], lz_decoder *d)
{
    if (!in_file(len, h[H_ENDPOINT_OFS], h[H_ENDPOINT_SIZE]) || !in_file(len, h[H_SELECTOR_OFS], h[H_SELECTOR_SIZE]) ||
        !in_file(len, h[H_TABLES_OFS], h[H_TABLES_SIZE]) || !in_file(len, h[H_EXT_OFS], h[H_EXT_SIZE]))
        return OB_ERR_BOUNDS;
    return lz_new(d, h[H_TOTAL_SELECTORS], h[H_TOTAL_SELECTORS], buf + h[H_ENDPOINT_OFS], h[H_ENDPOINT_SIZE], buf + h[H_SELECTOR_OFS],
                  h[H_SELECTOR_SIZE], buf + h[H_TABLES_OFS], h[H_TABLES_SIZE], h[H_TEX_TYPE] == 3);
}

/* One entry point for the six drivers.  out may be NULL to size the result (n_images, out_bytes). */
int bu_oracle_read_to(int which, const uint8_t *buf, size_t len, uint32_t header_out[26], ob_image *images, size_t max_images,
                      size_t *n_images, uint8_t *out, size_t out_cap, size_t *out_bytes)
{
    uint32_t h[26];
    int st = bu_oracle_read_header(buf, len, h);
    if (st) return st;
    if (header_out) memcpy(header_out, h, sizeof h);
    if (bu_oracle_crc16(buf + 77, len - 77, 0) != h[H_DATA_CRC16]) return OB_ERR_DATA_CRC; /* basis.rs:338-341: to EOF */
    size_t ns = h[H_TOTAL_SLICES];
    ob_slice *sd = (ob_slice *)calloc(ns ? ns : 1, sizeof(ob_slice));
    st = bu_oracle_read_slice_descs(buf, len, h, sd, ns);
    size_t used = 0, ni = 0;
    lz_decoder lz;
    int have_lz = 0;
    if (!st && h[H_TEX_FORMAT] > 1) st = OB_ERR_TEX_FORMAT;
    const int etc1s = h[H_TEX_FORMAT] == 0;
    const int has_alpha = (h[H_FLAGS] & 4) != 0;
    if (!st && etc1s && !(which == RD_RGBA || which == RD_ETC1)) st = OB_ERR_UNSUPPORTED;
    if (!st && etc1s && has_alpha && (ns % 2) != 0) st = OB_ERR_ALPHA_SLICES; /* basis.rs:18-20, 103-105 */
    if (!st && etc1s) {
        st = make_lz(buf, len, h, &lz);
        have_lz = !st;
    }
    for (size_t i = 0; i < ns && !st; i++) {
        const ob_slice *s = &sd[i];
        if (!in_file(len, s->file_ofs, s->file_size)) { st = OB_ERR_BOUNDS; break; }
        const uint8_t *data = buf + s->file_ofs;
        size_t nblk = (size_t)s->num_blocks_x * s->num_blocks_y;
        ob_image im = {s->orig_width, s->orig_height, 0, used, 0};
        if (etc1s) {
            if (which == RD_RGBA && has_alpha) { /* basis.rs:22-50: pairs of slices */
                if (i & 1) continue;
                const ob_slice *a = &sd[i + 1];
                if (!(a->flags & 1)) { st = OB_ERR_ALPHA_SLICES; break; }
                if (a->num_blocks_x != s->num_blocks_x || a->num_blocks_y != s->num_blocks_y) { st = OB_ERR_ALPHA_SLICES; break; }
                if (!in_file(len, a->file_ofs, a->file_size)) { st = OB_ERR_BOUNDS; break; }
            }
            uint16_t *idx = (uint16_t *)malloc((nblk ? nblk : 1) * 4), *aidx = NULL;
            st = lz_decode_blocks(&lz, s->num_blocks_x, s->num_blocks_y, data, s->file_size, idx);
            if (!st && which == RD_RGBA && has_alpha) {
                aidx = (uint16_t *)malloc((nblk ? nblk : 1) * 4);
                st = lz_decode_blocks(&lz, s->num_blocks_x, s->num_blocks_y, buf + sd[i + 1].file_ofs, sd[i + 1].file_size, aidx);
            }
            if (!st) {
                if (which == RD_RGBA) {
                    im.size = nblk * 64;
                    im.stride = 4 * 4 * s->orig_width; /* basis.rs:46,64 then x4 in into_rgba_bytes (lib.rs:75) -- reference quirk */
                    if (out) {
                        if (used + im.size > out_cap) st = OB_ERR_BOUNDS;
                        else bu_oracle_etc1s_to_rgba(idx, aidx, s->num_blocks_x, s->num_blocks_y, lz.endpoints, lz.selectors, out + used);
                    }
                } else {
                    im.size = nblk * 8;
                    im.stride = 8 * s->num_blocks_x;
                    if (out) {
                        if (used + im.size > out_cap) st = OB_ERR_BOUNDS;
                        else bu_oracle_etc1s_to_etc1(idx, nblk, lz.endpoints, lz.selectors, out + used);
                    }
                }
            }
            free(idx);
            free(aidx);
        } else { /* UASTC4x4 */
            size_t fb;
            size_t nb16 = s->file_size / 16;
            if (which == RD_RGBA) {
                im.size = nb16 * 64;
                im.stride = 4 * 4 * s->num_blocks_x; /* basis.rs:81 then x4 */
                if (s->file_size % 16) { st = 3; break; }
                if (out) {
                    if (used + im.size > out_cap) { st = OB_ERR_BOUNDS; break; }
                    if (s->num_blocks_x == 0 && nb16) { st = OB_ERR_BOUNDS; break; } /* i % 0 panics -- only if a block exists (uastc.rs:98-100) */
                    st = bu_oracle_decode_to_rgba(data, s->file_size, s->num_blocks_x, out + used, &fb);
                }
            } else if (which == RD_UASTC) {
                im.size = s->file_size; /* uastc.rs:85-87 */
                im.stride = 16 * s->num_blocks_x;
                if (out) {
                    if (used + im.size > out_cap) { st = OB_ERR_BOUNDS; break; }
                    memcpy(out + used, data, s->file_size);
                }
            } else {
                int target = which == RD_ASTC ? 0 : (which == RD_BC7 ? 1 : (which == RD_ETC1 ? 2 : 3));
                size_t bb = which == RD_ETC1 ? 8 : 16;
                im.size = nb16 * bb;
                im.stride = (uint32_t)bb * s->num_blocks_x;
                if (s->file_size % 16) { st = 3; break; }
                if (out) {
                    if (used + im.size > out_cap) { st = OB_ERR_BOUNDS; break; }
                    st = bu_oracle_transcode(target, data, s->file_size, out + used, &fb);
                }
            }
        }
        if (st) break;
        if (images && ni < max_images) images[ni] = im;
        ni++;
        used += im.size;
    }
    if (have_lz) lz_free(&lz);
    free(sd);
    if (n_images) *n_images = ni;
    if (out_bytes) *out_bytes = used;
    return st;
}

/* standalone BasisLZ pieces for tests: decode the codebooks and one slice's index stream */
int bu_oracle_lz_decode(const uint8_t *ep, size_t ep_len, const uint8_t *sel, size_t sel_len, const uint8_t *tables, size_t tables_len,
                        size_t n_endpoints, size_t n_selectors, int is_video, const uint8_t *slice, size_t slice_len, size_t nbx,
                        size_t nby, uint8_t *endpoints_out, uint8_t *selectors_out, uint16_t *idx_out)
{
    lz_decoder d;
    int st = lz_new(&d, n_endpoints, n_selectors, ep, ep_len, sel, sel_len, tables, tables_len, is_video);
    if (st) return st;
    if (endpoints_out) memcpy(endpoints_out, d.endpoints, 4 * n_endpoints);
    if (selectors_out) memcpy(selectors_out, d.selectors, 8 * n_selectors);
    if (idx_out) st = lz_decode_blocks(&d, nbx, nby, slice, slice_len, idx_out);
    lz_free(&d);
    return st;
}
