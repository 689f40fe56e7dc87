// Host side of the drop-in: `.basis` container reader/writer and the BasisLZ (ETC1S) entropy decoder.
// These stay on the CPU by design -- they are byte-serial / symbol-serial (north star: "the host/container
// parser ... stays [on the host] and calls through a thin C-ABI into HIP") -- and feed the GPU kernels.
//
// Replaces, for this repo's C++ host facade:
//   src/basis.rs:300-372, 417-572   header (77 B), slice descs (23 B), CRC-16/GENIBUS
//   src/basis_lz/huffman.rs:43-199  Huffman table records, canonical decode tables
//   src/basis_lz/mod.rs:64-95, 188-656  codebooks, tables section, the per-slice symbol loop
// Written for speed where it is free: table-driven CRC (8 bits per step), a 64-bit refilling bit reader,
// single-probe Huffman decode.  Reads past the end of a section return zeros like the reference's reader
// (bitreader.rs:45,55); everything the reference would `panic!`/`assert!` on is a BU_ERR_BOUNDS status.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/basisu_hip.h"

namespace bu_host {

// ---- CRC-16/GENIBUS (basis.rs:364-372): poly 0x1021, init 0xFFFF, xorout 0xFFFF, not reflected ----
struct Crc16Table {
    uint16_t t[256];
    Crc16Table()
    {
        for (int b = 0; b < 256; b++) {
            uint16_t crc = (uint16_t)(b << 8);
            for (int k = 0; k < 8; k++) crc = (uint16_t)((crc & 0x8000) ? ((crc << 1) ^ 0x1021) : (crc << 1));
            t[b] = crc;
        }
    }
};
inline uint16_t crc16(const uint8_t* p, size_t n, uint16_t crc)
{
    static const Crc16Table tab;
    crc = (uint16_t)~crc;
    for (size_t i = 0; i < n; i++) crc = (uint16_t)((crc << 8) ^ tab.t[((crc >> 8) ^ p[i]) & 0xFF]);
    return (uint16_t)~crc;
}

inline uint32_t le(const uint8_t* p, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

// basis.rs:475-516
inline void parse_header(const uint8_t* b, bu_basis_header* h)
{
    h->sig = (uint16_t)le(b + 0, 2);
    h->ver = (uint16_t)le(b + 2, 2);
    h->header_size = (uint16_t)le(b + 4, 2);
    h->header_crc16 = (uint16_t)le(b + 6, 2);
    h->data_size = le(b + 8, 4);
    h->data_crc16 = (uint16_t)le(b + 12, 2);
    h->total_slices = le(b + 14, 3);
    h->total_images = le(b + 17, 3);
    h->tex_format = b[20];
    h->flags = (uint16_t)le(b + 21, 2);
    h->tex_type = b[23];
    h->us_per_frame = le(b + 24, 3);
    h->reserved = le(b + 27, 4);
    h->userdata0 = le(b + 31, 4);
    h->userdata1 = le(b + 35, 4);
    h->total_endpoints = (uint16_t)le(b + 39, 2);
    h->endpoint_cb_file_ofs = le(b + 41, 4);
    h->endpoint_cb_file_size = le(b + 45, 3);
    h->total_selectors = (uint16_t)le(b + 48, 2);
    h->selector_cb_file_ofs = le(b + 50, 4);
    h->selector_cb_file_size = le(b + 54, 3);
    h->tables_file_ofs = le(b + 57, 4);
    h->tables_file_size = le(b + 61, 4);
    h->slice_desc_file_ofs = le(b + 65, 4);
    h->extended_file_ofs = le(b + 69, 4);
    h->extended_file_size = le(b + 73, 4);
}

// basis.rs:307-336
inline bu_status read_header(const uint8_t* file, size_t len, bu_basis_header* h)
{
    if (len < 2 || le(file, 2) != 0x4273u) return BU_ERR_SIG;
    if (len < 77) return BU_ERR_HEADER_TRUNCATED;
    parse_header(file, h);
    if (h->header_size != 77) return BU_ERR_HEADER_SIZE;
    if (crc16(file + 8, 77 - 8, 0) != h->header_crc16) return BU_ERR_HEADER_CRC;
    return BU_OK;
}

// basis.rs:343-362, 554-571
inline bu_status read_slice_descs(const uint8_t* file, size_t len, const bu_basis_header* h, std::vector<bu_slice_desc>& out)
{
    out.clear();
    const size_t start = h->slice_desc_file_ofs;
    for (size_t i = 0; i < h->total_slices; i++) {
        const size_t s = start + 23 * i;
        if (s > len) return BU_ERR_BOUNDS;
        if (len - s < 23) return BU_ERR_SLICE_DESC;
        const uint8_t* p = file + s;
        bu_slice_desc d;
        d.image_index = le(p, 3);
        d.level_index = p[3];
        d.flags = p[4];
        d.orig_width = (uint16_t)le(p + 5, 2);
        d.orig_height = (uint16_t)le(p + 7, 2);
        d.num_blocks_x = (uint16_t)le(p + 9, 2);
        d.num_blocks_y = (uint16_t)le(p + 11, 2);
        d.file_ofs = le(p + 13, 4);
        d.file_size = le(p + 17, 4);
        d.slice_data_crc16 = (uint16_t)le(p + 21, 2);
        out.push_back(d);
    }
    return BU_OK;
}

inline bool in_file(size_t len, size_t ofs, size_t size) { return ofs <= len && size <= len - ofs; }

// etc::Selector::set_selector (etc.rs:363-393) for all 16 texels of one codebook entry: ETC1 code = [3,2,0,1][value],
// pixel id = x*4 + y, MSB plane in bytes 4-5 (pixels 8-15, then 0-7), LSB plane in bytes 6-7; bytes 0-3 keep the raw rows
inline void selector_from_rows(const uint8_t rows[4], uint8_t out_entry[8])
{
    static const uint8_t to_etc1[4] = {3, 2, 0, 1};  // etc.rs:433
    uint32_t msb = 0, lsb = 0;
    for (unsigned y = 0; y < 4; y++)
        for (unsigned x = 0; x < 4; x++) {
            const unsigned code = to_etc1[(rows[y] >> (2 * x)) & 3u];
            msb |= (code >> 1) << (x * 4 + y);
            lsb |= (code & 1u) << (x * 4 + y);
        }
    memcpy(out_entry, rows, 4);
    out_entry[4] = (uint8_t)(msb >> 8);
    out_entry[5] = (uint8_t)(msb & 0xFF);
    out_entry[6] = (uint8_t)(lsb >> 8);
    out_entry[7] = (uint8_t)(lsb & 0xFF);
}

// ---- LSB-first bit reader with a 64-bit window ----
class BitReader {
public:
    BitReader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    uint32_t peek(unsigned count)
    {
        if (have_ < count) refill();
        return (uint32_t)(acc_ & ((count >= 32) ? 0xFFFFFFFFull : ((1ull << count) - 1ull)));
    }
    void skip(unsigned count)
    {
        if (have_ < count) refill();
        acc_ >>= count;
        have_ -= count;
    }
    uint32_t read(unsigned count)
    {
        const uint32_t v = peek(count);
        skip(count);
        return v;
    }

private:
    void refill()
    {
        while (have_ <= 56) {
            const uint64_t byte = pos_ < n_ ? p_[pos_] : 0;  // zeros past the end (bitreader.rs:45,55)
            pos_++;
            acc_ |= byte << have_;
            have_ += 8;
        }
    }
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    uint64_t acc_ = 0;
    unsigned have_ = 0;
};

// ---- canonical Huffman, single-probe table (huffman.rs:120-199) ----
class Huffman {
public:
    // code_sizes[sym] in 0..16.  Over-subscribed length sets are accepted exactly like the reference does
    // (huffman.rs:163-170: the code is truncated to its length and later symbols overwrite earlier entries);
    // only a length whose running code count passes 2^16 is an error (huffman.rs:176-178).
    bu_status build(const std::vector<uint8_t>& sizes)
    {
        uint32_t count[17] = {0};
        max_ = 0;
        for (uint8_t s : sizes) {
            if (s > 16) return BU_ERR_BOUNDS;
            count[s]++;
            if (s > max_) max_ = s;
        }
        count[0] = 0;
        uint32_t next[17] = {0}, total = 0;
        for (int bits = 1; bits <= 16; bits++) {
            total = (total + count[bits - 1]) << 1;
            next[bits] = total;
        }
        table_.assign((size_t)1 << max_, 0u);
        for (size_t sym = 0; sym < sizes.size(); sym++) {
            const unsigned size = sizes[sym];
            if (!size) continue;
            const uint32_t code = next[size]++;
            uint32_t rev = 0;
            for (unsigned k = 0; k < size; k++) rev |= ((code >> k) & 1u) << (size - 1 - k);
            const uint32_t entry = ((uint32_t)sym << 5) | size;
            for (uint32_t id = rev; id < ((uint32_t)1 << max_); id += (1u << size)) table_[id] = entry;
        }
        for (int bits = 0; bits <= 16; bits++)
            if (next[bits] > 65536u) return BU_ERR_BASISLZ;
        return BU_OK;
    }
    // returns false on "No matching code found" (huffman.rs:189-197)
    bool decode(BitReader& r, uint32_t* sym) const
    {
        const uint32_t e = table_[r.peek(max_)];
        if ((e & 0x1F) == 0) return false;
        r.skip(e & 0x1F);
        *sym = e >> 5;
        return true;
    }

private:
    std::vector<uint32_t> table_;  // symbol << 5 | code_size, indexed by the next max_ bits (LSB first)
    unsigned max_ = 0;
};

// huffman.rs:43-118
inline bu_status read_huffman_table(BitReader& r, Huffman& out)
{
    const size_t total_used = r.read(14);
    Huffman cl;
    {
        const size_t n = r.read(5);
        static const uint8_t order[21] = {17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16};
        if (n > 21) return BU_ERR_BOUNDS;
        std::vector<uint8_t> sizes(21, 0);
        for (size_t i = 0; i < n; i++) sizes[order[i]] = (uint8_t)r.read(3);
        bu_status st = cl.build(sizes);
        if (st) return st;
    }
    std::vector<uint8_t> sizes;
    sizes.reserve(total_used + 140);
    while (sizes.size() < total_used) {
        uint32_t s;
        if (!cl.decode(r, &s)) return BU_ERR_BASISLZ;
        if (s <= 16) {
            sizes.push_back((uint8_t)s);
        } else if (s <= 18) {
            const size_t count = s == 17 ? 3 + r.read(3) : 11 + r.read(7);
            sizes.insert(sizes.end(), count, 0);
        } else {
            if (sizes.empty() || sizes.back() == 0) return BU_ERR_BASISLZ;  // huffman.rs:82-107
            const size_t count = s == 19 ? 3 + r.read(2) : 7 + r.read(7);
            sizes.insert(sizes.end(), count, sizes.back());
        }
    }
    return out.build(sizes);
}

// ---- BasisLZ decoder state (basis_lz/mod.rs:50-95) ----
struct BasisLz {
    Huffman endpoint_pred, delta_endpoint, selector, history_rle;
    uint32_t history_size = 0;
    bool is_video = false;
    std::vector<uint32_t> endpoints;  // r5 | g5<<8 | b5<<16 | inten<<24
    std::vector<uint8_t> selectors;   // 8 B per entry: rows[4], etc1_bytes[4]

    // mod.rs:461-516
    bu_status decode_endpoints(size_t num, const uint8_t* p, size_t n)
    {
        BitReader r(p, n);
        Huffman model[3], inten;
        for (int m = 0; m < 3; m++) {
            bu_status st = read_huffman_table(r, model[m]);
            if (st) return st;
        }
        bu_status st = read_huffman_table(r, inten);
        if (st) return st;
        const bool grayscale = r.read(1) != 0;
        endpoints.assign(num, 0);
        uint32_t prev[3] = {16, 16, 16}, prev_inten = 0;
        for (size_t i = 0; i < num; i++) {
            uint32_t s;
            if (!inten.decode(r, &s)) return BU_ERR_BASISLZ;
            prev_inten = (s + prev_inten) & 7u;
            uint32_t c[3];
            for (int ch = 0; ch < (grayscale ? 1 : 3); ch++) {
                const Huffman& m = model[prev[ch] <= 9 ? 0 : (prev[ch] <= 21 ? 1 : 2)];  // mod.rs:28-37
                if (!m.decode(r, &s)) return BU_ERR_BASISLZ;
                prev[ch] = (prev[ch] + s) & 31u;  // u8 wrapping_add then & 31
                c[ch] = prev[ch];
            }
            if (grayscale) c[1] = c[2] = c[0];
            endpoints[i] = c[0] | (c[1] << 8) | (c[2] << 16) | (prev_inten << 24);
        }
        return BU_OK;
    }

    // mod.rs:524-583
    bu_status decode_selectors(size_t num, const uint8_t* p, size_t n)
    {
        BitReader r(p, n);
        const bool global = r.read(1), hybrid = r.read(1), raw = r.read(1);
        if (global || hybrid) return BU_ERR_BASISLZ;
        selectors.assign(num * 8, 0);
        Huffman delta;
        if (!raw) {
            bu_status st = read_huffman_table(r, delta);
            if (st) return st;
        }
        uint8_t prev[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < num; i++) {
            uint8_t rows[4];
            for (int y = 0; y < 4; y++) {
                if (raw || i == 0) {
                    rows[y] = (uint8_t)r.read(8);
                } else {
                    uint32_t s;
                    if (!delta.decode(r, &s)) return BU_ERR_BASISLZ;
                    rows[y] = (uint8_t)(s ^ prev[y]);
                }
                prev[y] = rows[y];
            }
            selector_from_rows(rows, &selectors[8 * i]);
        }
        return BU_OK;
    }

    // mod.rs:64-95
    bu_status init(size_t n_endpoints, size_t n_selectors, const uint8_t* ep, size_t ep_len, const uint8_t* sel, size_t sel_len,
                   const uint8_t* tables, size_t tables_len, bool video)
    {
        is_video = video;
        bu_status st = decode_endpoints(n_endpoints, ep, ep_len);
        if (st) return st;
        st = decode_selectors(n_selectors, sel, sel_len);
        if (st) return st;
        BitReader r(tables, tables_len);
        if ((st = read_huffman_table(r, endpoint_pred))) return st;
        if ((st = read_huffman_table(r, delta_endpoint))) return st;
        if ((st = read_huffman_table(r, selector))) return st;
        if ((st = read_huffman_table(r, history_rle))) return st;
        history_size = r.read(13);
        return BU_OK;
    }

    // mod.rs:585-608
    static bool vlc(BitReader& r, unsigned chunk_bits, uint32_t* out)
    {
        uint32_t v = 0;
        for (unsigned ofs = 0;; ofs += chunk_bits) {
            if (ofs >= 32) return false;  // panic!() in the reference
            const uint32_t s = r.read(chunk_bits + 1);
            v |= (s & ((1u << chunk_bits) - 1u)) << ofs;
            if (!(s >> chunk_bits)) break;
        }
        *out = v;
        return true;
    }

    // mod.rs:188-458: the serial symbol loop.  idx[i] = endpoint_index | selector_index << 16, raster order.
    bu_status decode_slice(size_t nbx, size_t nby, const uint8_t* data, size_t len, uint32_t* idx) const
    {
        BitReader r(data, len);
        const uint32_t n_ep = (uint32_t)endpoints.size(), n_sel = (uint32_t)(selectors.size() / 8);
        // two rows of per-column state: endpoint index of the row above / pending 4 predictor bits for the row below
        std::vector<uint16_t> above(nbx, 0), cur_row(nbx, 0);
        std::vector<uint8_t> saved_bits(nbx, 0);
        std::vector<uint16_t> hist(history_size ? history_size : 1, 0);  // ApproxMoveToFront, mod.rs:610-656
        size_t rover = history_size / 2;
        const uint32_t rle_sym = n_sel + history_size;
        uint32_t sel_rle = 0, pred_repeat = 0, cur_bits = 0, prev_pred_sym = 0, prev_ep = 0;
        // texture video: the reference re-zeroes its "previous frame" per slice (mod.rs:236-237), so predictor 2
        // always reads the indices written earlier in this same slice -- i.e. zeros
        for (size_t by = 0; by < nby; by++) {
            for (size_t bx = 0; bx < nbx; bx++) {
                if (!(bx & 1)) {
                    if (!(by & 1)) {
                        if (pred_repeat) {
                            pred_repeat--;
                            cur_bits = prev_pred_sym;
                        } else {
                            uint32_t s;
                            if (!endpoint_pred.decode(r, &s)) return BU_ERR_BASISLZ;
                            if (s == 256) {
                                uint32_t v;
                                if (!vlc(r, 4, &v)) return BU_ERR_BOUNDS;
                                pred_repeat = v + 2;
                                cur_bits = prev_pred_sym;
                            } else {
                                cur_bits = s & 0xFF;
                                prev_pred_sym = cur_bits;
                            }
                        }
                        saved_bits[bx] = (uint8_t)(cur_bits >> 4);
                    } else {
                        cur_bits = saved_bits[bx];
                    }
                }
                const uint32_t pred = cur_bits & 3;
                cur_bits >>= 2;
                uint32_t e;
                switch (pred) {
                case 0:
                    if (bx == 0) return BU_ERR_BOUNDS;
                    e = prev_ep;
                    break;
                case 1:
                    if (by == 0) return BU_ERR_BOUNDS;
                    e = above[bx];
                    break;
                case 2:
                    if (is_video) {
                        e = 0;
                    } else {
                        if (bx == 0 || by == 0) return BU_ERR_BOUNDS;
                        e = above[bx - 1];
                    }
                    break;
                default: {
                    uint32_t s;
                    if (!delta_endpoint.decode(r, &s)) return BU_ERR_BASISLZ;
                    e = (s + prev_ep) & 0xFFFFu;
                    if (e >= n_ep) e = (e - n_ep) & 0xFFFFu;
                }
                }
                cur_row[bx] = (uint16_t)e;
                prev_ep = e;

                uint32_t sel;
                if (!is_video || pred != 2) {
                    uint32_t sym;
                    if (sel_rle) {
                        sel_rle--;
                        sym = n_sel;
                    } else {
                        if (!selector.decode(r, &sym)) return BU_ERR_BASISLZ;
                        if (sym == (rle_sym & 0xFFFFu)) {
                            uint32_t run;
                            if (!history_rle.decode(r, &run)) return BU_ERR_BASISLZ;
                            if (run == 63) {
                                uint32_t v;
                                if (!vlc(r, 7, &v)) return BU_ERR_BOUNDS;
                                sel_rle = 3 + v;
                            } else {
                                sel_rle = 3 + run;
                            }
                            sel_rle--;
                            sym = n_sel;
                        }
                    }
                    if (sym >= n_sel) {
                        const size_t hi = sym - n_sel;
                        if (history_size == 0 || hi >= history_size) return BU_ERR_BOUNDS;
                        sel = hist[hi];
                        if (hi) std::swap(hist[hi / 2], hist[hi]);
                    } else {
                        if (history_size) {
                            hist[rover] = (uint16_t)sym;
                            if (++rover == history_size) rover = history_size / 2;
                        }
                        sel = sym;
                    }
                } else {
                    sel = 0;  // previous frame's selector: zero, see above
                }
                if (e >= n_ep || sel >= n_sel) return BU_ERR_BOUNDS;  // asserts mod.rs:443-445
                idx[by * nbx + bx] = e | (sel << 16);
            }
            // the row just finished becomes "above"; the saved predictor bits written on even rows are read on the next (odd) row
            above.swap(cur_row);
        }
        return BU_OK;
    }
};

// ---- whole-file planning (basis.rs:8-260): every check of read_to_* that needs no block work ----
struct BuFilePlan {
    bu_basis_header h;
    std::vector<bu_slice_desc> slices;
    std::vector<bu_image> images;       // one per output image
    std::vector<size_t> first_slice;    // slice feeding image i (its colour slice for RGBA+alpha)
    size_t out_bytes = 0;
    bool etc1s = false, alpha_pairs = false;
};

// everything of read_to_* that needs no block work: checks in the reference's order, image geometry
inline bu_status bu_plan_file(bu_read_target target, const uint8_t* file, size_t len, BuFilePlan& p)
{
    if (!file) return BU_ERR_ARGUMENT;
    if ((int)target < 0 || (int)target > 5) return BU_ERR_ARGUMENT;
    bu_status st = read_header(file, len, &p.h);
    if (st) return st;
    if (crc16(file + 77, len - 77, 0) != p.h.data_crc16) return BU_ERR_DATA_CRC;  // to EOF, basis.rs:338-341
    st = read_slice_descs(file, len, &p.h, p.slices);
    if (st) return st;
    if (p.h.tex_format > 1) return BU_ERR_TEX_FORMAT;
    p.etc1s = p.h.tex_format == 0;
    const bool has_alpha = (p.h.flags & 4) != 0;
    if (p.etc1s && !(target == BU_READ_RGBA || target == BU_READ_ETC1)) return BU_ERR_UNSUPPORTED;
    if (p.etc1s && has_alpha && (p.slices.size() % 2) != 0) return BU_ERR_ALPHA_SLICES;
    p.alpha_pairs = p.etc1s && has_alpha && target == BU_READ_RGBA;
    for (size_t i = 0; i < p.slices.size(); i++) {
        const bu_slice_desc& s = p.slices[i];
        if (!in_file(len, s.file_ofs, s.file_size)) return BU_ERR_BOUNDS;
        if (p.alpha_pairs) {
            if (i & 1) continue;
            const bu_slice_desc& a = p.slices[i + 1];
            if (!(a.flags & 1)) return BU_ERR_ALPHA_SLICES;
            if (a.num_blocks_x != s.num_blocks_x || a.num_blocks_y != s.num_blocks_y) return BU_ERR_ALPHA_SLICES;
        }
        const size_t nblk = (size_t)s.num_blocks_x * s.num_blocks_y, nb16 = s.file_size / 16;
        bu_image im = {s.orig_width, s.orig_height, 0, 0, p.out_bytes, 0};
        if (p.etc1s) {
            if (target == BU_READ_RGBA) {
                im.size = nblk * 64;
                im.stride = 16u * s.orig_width;  // basis.rs:46,64 x4 (lib.rs:75): reference quirk, rows are 16*nbx apart
            } else {
                im.size = nblk * 8;
                im.stride = 8u * s.num_blocks_x;
            }
        } else {
            if (target != BU_READ_UASTC && s.file_size % 16) return BU_ERR_LENGTH;  // uastc.rs:54-59
            switch (target) {
            case BU_READ_RGBA:
                if (s.num_blocks_x == 0 && nb16) return BU_ERR_BOUNDS;
                if (s.num_blocks_x && nb16 % s.num_blocks_x) return BU_ERR_BOUNDS;  // the reference indexes past its image
                im.size = nb16 * 64;
                im.stride = 16u * s.num_blocks_x;
                break;
            case BU_READ_UASTC: im.size = s.file_size; im.stride = 16u * s.num_blocks_x; break;
            case BU_READ_ETC1: im.size = nb16 * 8; im.stride = 8u * s.num_blocks_x; break;
            default: im.size = nb16 * 16; im.stride = 16u * s.num_blocks_x; break;
            }
        }
        p.images.push_back(im);
        p.first_slice.push_back(i);
        p.out_bytes += im.size;
    }
    return BU_OK;
}

inline bu_status bu_make_lz(const uint8_t* file, size_t len, const bu_basis_header& h, BasisLz& lz)
{
    if (!in_file(len, h.endpoint_cb_file_ofs, h.endpoint_cb_file_size) || !in_file(len, h.selector_cb_file_ofs, h.selector_cb_file_size) ||
        !in_file(len, h.tables_file_ofs, h.tables_file_size) || !in_file(len, h.extended_file_ofs, h.extended_file_size))
        return BU_ERR_BOUNDS;
    // total_selectors for both codebooks: basis.rs:289-291
    return lz.init(h.total_selectors, h.total_selectors, file + h.endpoint_cb_file_ofs, h.endpoint_cb_file_size, file + h.selector_cb_file_ofs,
                   h.selector_cb_file_size, file + h.tables_file_ofs, h.tables_file_size, h.tex_type == 3);
}

}  // namespace bu_host
