// Device-side lookup tables of the UASTC/ETC1S transcode kernels, and the host code that builds them.
//
// One `BuTables` blob is built on the host at context creation (bu_build_tables), uploaded once to
// device memory and copied into LDS by every workgroup (it is ~7 KiB; per-lane divergent lookups are
// LDS reads, never global gathers).  All LUTs are derived here from the format constants of
// bu_tables.h, so the kernels never divide, never take a modulo and never run a search.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "bu_tables.h"

struct BuPart {  // one UASTC partition pattern (layout documented in bu_tables.h / tools/gen_tables.py)
    uint32_t upat, bpat;
    uint16_t seed, uanch, banch;
    uint8_t bpart, perm;
};
static_assert(sizeof(BuPart) == 16, "BuPart must be 16 bytes");

// offsets (bytes) into BuTables::deq of the dequantisation LUT of each BISE range UASTC uses
// (index = trit_or_quint << bits | bits_value); filled from BU_BISE[].lut_ofs_div8
// Members are grouped by who reads them, each group 16-byte aligned, so that a kernel stages only what its target needs
// (bu_table_range below): every workgroup copies its tables from L2 into LDS, and with four workgroups per CU on
// 1024-block tiles the whole 9.3 KiB blob would be more than half of the 16 KiB of payload the workgroup moves.
//   [BC7 only][common front-end][texel unpack: RGBA32, ETC][ASTC only][ETC only]
// Modes by descending length of their code path (BC7 VALU counts, tools/exp/mode_isa.py): the mode-sorted kernel lays the
// runs out in this order so that dynamically scheduled chunks end with the cheap ones.  Entry 19 = invalid mode code.
constexpr uint8_t BU_COST_ORDER[20] = {3, 9, 4, 16, 2, 7, 12, 1, 11, 6, 18, 5, 10, 14, 0, 8, 17, 13, 15, 19};

struct BuTables {
    // ---- BC7 only ----
    alignas(16) uint8_t deq5[56];  // (deq*31+127)/255 for ranges 7 (ofs 0) and 12 (ofs 16): BC7 mode 2 endpoints (bc7.rs:262-264)
    uint32_t w3mask[30][2];  // 2-subset patterns, BC7 subset-1 texel mask at 3 bits/texel (48 bits)
    uint32_t pbit6[256];   // unique p-bit LUT, 6 total bits: q0>>1 | (q1>>1)<<8 | err0<<16 | err1<<24 (bc7.rs:478-553)
    uint32_t pbit7[16];    // shared p-bit LUT, 7 total bits, input = 17*index (bc7.rs:408-475)
    uint16_t m5opt[256];   // BC7 mode 5 solid colour lo | hi<<8 (bc7.rs:734-863)
    uint16_t m6opt[258];   // BC7 mode 6 solid colour lo | hi<<8, index c + !p (bc7.rs:866-1136)
    uint8_t w5to4x2[1024];    // two 5-bit weights (10 bits) -> two 4-bit BC7 weights, x>>1 except 14 -> 6 and 17 -> 9 (bc7.rs:381-384)
    // ---- common front-end (every target) ----
    alignas(16) uint16_t trit5[256];  // 8-bit group -> 5 trits, digit i in bits [2i,2i+2)   (uastc.rs:657-685)
    uint16_t quint3[128];  // 7-bit group -> 3 quints, digit i in bits [3i,3i+3)  (uastc.rs:629-655)
    uint8_t deq[504];      // endpoint dequantisation, ranges 7,8,11,12,13,18,19  (uastc.rs:585-614)
    uint8_t mode_lut[128];    // uastc.rs:560-577
    uint8_t key_lut[128];     // sort key of the mode-sorted kernel: position of the block's mode in BU_COST_ORDER (19 = invalid code)
    BuPart part[61];       // partition records
    // ---- texel unpack (RGBA32, ETC1, ETC2) ----
    alignas(16) uint32_t wpack[64];  // raw weight -> (256 - 4w) | 4w << 16 with w = unquant_weights (uastc.rs:697-719); offset 2^bits - 2
    // ---- ASTC only ----
    alignas(16) uint32_t w3mask_u[30][2];  // 2-subset patterns, UASTC/ASTC subset-1 texel mask at 3 bits/texel (astc.rs:162-170)
    uint8_t astc_trit[244];   // astc.rs:247-264
    uint8_t astc_quint[128];  // astc.rs:208-217
    uint16_t astc_mode13[20]; // astc.rs:333-354
    uint16_t astc_trit_pk[1024];  // five trits packed 2 bits each -> ASTC trit byte T (astc.rs:247-264 re-indexed), its bits already spread to
                                  // their places in a 2-bit-value group, >> 2: T01 @0, T23 @4, T4 @8, T56 @11, T7 @15
    uint16_t astc_quint_pk[512];  // three quints packed 3 bits each -> ASTC quint code Q (astc.rs:208-217 re-indexed), spread for a
                                  // 3-bit-value group: Q012 @3, Q34 @9, Q56 @14
    int16_t pairdiff7[256];      // BISE range 7: index tq_lo | tq_hi << 2 | eb_lo << 4 | eb_hi << 6 -> deq(hi) - deq(lo)  (astc.rs:57-66)
    // ---- ETC1 / ETC2 only ----
    alignas(16) int16_t etc1_mod[32];  // etc.rs:435-445
    int8_t etc2_amod[128];    // etc.rs:450-468
    uint16_t etc1_bias[32];   // apply_etc1_bias deltas (etc.rs:203-234): field (2*subblock*3 + 2*c) = delta + 2
    uint32_t eac_magic[16];   // ceil(2^20 / (2*range)) per EAC modifier table (etc.rs:297-307 as integers)
    int8_t eac_mod_min[16];   // modifier[3] of each table
    uint8_t eac_range[16];    // modifier[7] - modifier[3]
    uint8_t etc1_biasv[256];  // apply_etc1_bias per channel (etc.rs:236-255): index diff << 7 | (delta + 2) << 5 | value
    uint32_t etc1_thrcol[512];  // index diff << 8 | inten << 5 | c: the channel's four modified base values clamp(base + ETC1_MODIFIERS[inten][k]),
                                // k in byte k; base = c*17 (individual, c < 16) or c<<3 | c>>2 (differential)  (etc.rs:165-171, 396-431)
    alignas(16) uint8_t end_marker[16];
};
static_assert(sizeof(BuTables) % 16 == 0, "BuTables is copied to LDS in 16-byte pieces");

// byte ranges [lo, hi) of BuTables a target reads (target ids as in bu_uastc_dispatch.hpp: 0 ASTC, 1 BC7, 2 ETC1, 3 ETC2,
// 4 RGBA32); a second range is empty unless lo2 < hi2.  All bounds are multiples of 16.
struct BuTableRange {
    unsigned lo, hi, lo2, hi2;
};
constexpr BuTableRange bu_table_range(int target)
{
    return target == 1   ? BuTableRange{(unsigned)offsetof(BuTables, deq5), (unsigned)offsetof(BuTables, wpack), 0u, 0u}
           : target == 0 ? BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, etc1_mod), 0u, 0u}
           : target == 4 ? BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, w3mask_u), 0u, 0u}
                         : BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, w3mask_u),
                                        (unsigned)offsetof(BuTables, etc1_mod), (unsigned)offsetof(BuTables, end_marker)};
}

// deq offsets per range, compile-time (must match tools/gen_tables.py's packing order: 7,8,11,12,13,18,19,20)
// sizes: r7 3*4=12->16, r8 16, r11 32, r12 5*8=40, r13 3*16=48, r18 5*32=160, r19 3*64=192 = 504; r20 is the identity
constexpr int bu_deq_ofs(int range)
{
    return range == 7 ? 0 : range == 8 ? 16 : range == 11 ? 32 : range == 12 ? 64 : range == 13 ? 104 : range == 18 ? 152 : range == 19 ? 312 : -1;
}

// host side (plain functions: never emitted for the device)
// exact-rational form of the reference's f32 quantiser (bc7.rs:441-444, 511-514); SURVEY.md appendix A
static inline int bu_quant_p(int x, int S, int p)
{
    int q = ((x * S - 255 * p + 255) / 510) * 2 + p;
    int lo = p, hi = S - 1 + p;
    return q < lo ? lo : (q > hi ? hi : q);
}

// apply_etc1_bias for one channel (etc.rs:236-255); delta in -2..1, v in 0..limit
static inline int bu_etc1_bias1_host(int v, int delta, int limit)
{
    if (v == 0) return delta == -2 ? 3 : delta + 1;
    if (v == limit) return v + delta - 1;
    const int m = v + delta;
    return (m < 0 || m > limit) ? v - delta : m;
}

static inline void bu_build_tables(BuTables* t)
{
    memset(t, 0, sizeof(*t));
    for (int r = 0; r < 256; r++) {
        uint8_t v = (uint8_t)r;
        uint16_t packed = 0;
        for (int i = 0; i < 5; i++) {
            packed |= (uint16_t)((v % 3) << (2 * i));
            v /= 3;
        }
        t->trit5[r] = packed;
    }
    for (int r = 0; r < 128; r++) {
        uint8_t v = (uint8_t)r;
        uint16_t packed = 0;
        for (int i = 0; i < 3; i++) {
            packed |= (uint16_t)((v % 5) << (3 * i));
            v /= 5;
        }
        t->quint3[r] = packed;
    }
    // dequantisation LUT: the generator's BU_ENDPOINT_DEQ holds ranges 7,8,11,12,13,18,19,20 back to back
    static const int used[7] = {7, 8, 11, 12, 13, 18, 19};
    for (int k = 0; k < 7; k++) {
        int r = used[k];
        int n = (BU_BISE[r].trits ? 3 : BU_BISE[r].quints ? 5 : 1) << BU_BISE[r].bits;
        for (int i = 0; i < n; i++) t->deq[bu_deq_ofs(r) + i] = BU_ENDPOINT_DEQ[8 * BU_BISE[r].lut_ofs_div8 + i];
    }
    for (int i = 0; i < 12; i++) t->deq5[i] = (uint8_t)((t->deq[bu_deq_ofs(7) + i] * 31 + 127) / 255);
    for (int i = 0; i < 40; i++) t->deq5[16 + i] = (uint8_t)((t->deq[bu_deq_ofs(12) + i] * 31 + 127) / 255);
    for (int i = 0; i < 61; i++) {
        t->part[i].upat = BU_PART[i].upat;
        t->part[i].bpat = BU_PART[i].bpat;
        t->part[i].seed = BU_PART[i].seed;
        t->part[i].uanch = BU_PART[i].uanch;
        t->part[i].banch = BU_PART[i].banch;
        t->part[i].bpart = BU_PART[i].bpart;
        t->part[i].perm = BU_PART[i].perm;
    }
    for (int i = 0; i < 30; i++) {
        uint64_t m = 0;
        for (int tx = 0; tx < 16; tx++)
            if ((BU_PART[i].bpat >> (2 * tx)) & 1) m |= 7ull << (3 * tx);
        t->w3mask[i][0] = (uint32_t)m;
        t->w3mask[i][1] = (uint32_t)(m >> 32);
    }
    for (int x = 0; x < 256; x++) {  // 6 total bits (BC7 mode 7)
        int q0 = bu_quant_p(x, 63, 0), q1 = bu_quant_p(x, 63, 1);
        int s0 = ((q0 << 2) | (q0 >> 4)) & 255, s1 = ((q1 << 2) | (q1 >> 4)) & 255;
        int e0 = (s0 - x) * (s0 - x), e1 = (s1 - x) * (s1 - x);
        t->pbit6[x] = (uint32_t)(q0 >> 1) | (uint32_t)(q1 >> 1) << 8 | (uint32_t)e0 << 16 | (uint32_t)e1 << 24;
    }
    for (int i = 0; i < 16; i++) {  // 7 total bits (BC7 mode 1), inputs are multiples of 17
        int x = 17 * i;
        int q0 = bu_quant_p(x, 127, 0), q1 = bu_quant_p(x, 127, 1);
        int s0 = ((q0 << 1) | (q0 >> 6)) & 255, s1 = ((q1 << 1) | (q1 >> 6)) & 255;
        int e0 = (s0 - x) * (s0 - x), e1 = (s1 - x) * (s1 - x);
        t->pbit7[i] = (uint32_t)(q0 >> 1) | (uint32_t)(q1 >> 1) << 8 | (uint32_t)e0 << 16 | (uint32_t)e1 << 24;
    }
    for (int i = 0; i < 256; i++) t->m5opt[i] = BU_BC7_M5_OPT[i];
    for (int i = 0; i < 257; i++) t->m6opt[i] = BU_BC7_M6_OPT[i];
    for (int i = 0; i < 243; i++) t->astc_trit[i] = BU_ASTC_TRIT_ENC[i];
    for (int i = 0; i < 125; i++) t->astc_quint[i] = BU_ASTC_QUINT_ENC[i];
    for (int i = 0; i < 20; i++) t->astc_mode13[i] = BU_ASTC_BLOCK_MODE13[i];
    for (int i = 0; i < 32; i++) t->etc1_mod[i] = BU_ETC1_MOD[i];
    for (int i = 0; i < 128; i++) t->etc2_amod[i] = BU_ETC2_ALPHA_MOD[i];
    for (int i = 0; i < 128; i++) t->mode_lut[i] = BU_MODE_LUT[i];
    for (int i = 0; i < 128; i++) {
        t->key_lut[i] = 19;
        for (int k = 0; k < 20; k++)
            if (BU_COST_ORDER[k] == BU_MODE_LUT[i]) t->key_lut[i] = (uint8_t)k;
    }
    for (int i = 0; i < 30; i++) {
        uint64_t m = 0;
        for (int tx = 0; tx < 16; tx++)
            if ((BU_PART[i].upat >> (2 * tx)) & 1) m |= 7ull << (3 * tx);
        t->w3mask_u[i][0] = (uint32_t)m;
        t->w3mask_u[i][1] = (uint32_t)(m >> 32);
    }
    for (int bias = 0; bias < 32; bias++) {  // etc.rs:203-234 tabulated
        uint16_t packed = 0;
        for (int sb = 0; sb < 2; sb++)
            for (int c = 0; c < 3; c++) {
                int delta;
                static const int divs[3] = {1, 3, 9};
                switch (bias) {
                case 2: delta = sb ? 0 : (c == 0 ? -1 : 0); break;
                case 5: delta = sb ? 0 : (c == 1 ? -1 : 0); break;
                case 6: delta = sb ? 0 : (c == 2 ? -1 : 0); break;
                case 7: delta = sb ? 0 : (c == 0 ? 1 : 0); break;
                case 11: delta = sb ? 0 : (c == 1 ? 1 : 0); break;
                case 15: delta = sb ? 0 : (c == 2 ? 1 : 0); break;
                case 18: delta = sb ? (c == 0 ? -1 : 0) : 0; break;
                case 19: delta = sb ? (c == 1 ? -1 : 0) : 0; break;
                case 20: delta = sb ? (c == 2 ? -1 : 0) : 0; break;
                case 21: delta = sb ? (c == 0 ? 1 : 0) : 0; break;
                case 24: delta = sb ? (c == 1 ? 1 : 0) : 0; break;
                case 8: delta = sb ? (c == 2 ? 1 : 0) : 0; break;
                case 10: delta = -2; break;
                case 27: delta = sb ? 0 : -1; break;
                case 28: delta = sb ? -1 : 1; break;
                case 29: delta = sb ? 1 : 0; break;
                case 30: delta = sb ? -1 : 0; break;
                case 31: delta = sb ? 0 : 1; break;
                default: delta = (bias / divs[c]) % 3 - 1; break;
                }
                packed |= (uint16_t)((delta + 2) << (2 * (sb * 3 + c)));
            }
        t->etc1_bias[bias] = packed;
    }
    for (int d = 0; d < 2; d++)
        for (int dc = 0; dc < 4; dc++)
            for (int v = 0; v < 32; v++) {
                const int limit = d ? 31 : 15;
                t->etc1_biasv[(d << 7) | (dc << 5) | v] = v <= limit ? (uint8_t)bu_etc1_bias1_host(v, dc - 2, limit) : 0;
            }
    for (int i = 0; i < 1024; i++) {
        int id = 0, mul = 1, okd = 1;
        for (int k = 0; k < 5; k++) {
            const int d = (i >> (2 * k)) & 3;
            if (d == 3) okd = 0;
            id += d * mul;
            mul *= 3;
        }
        const uint32_t tb = okd ? BU_ASTC_TRIT_ENC[id] : 0;
        t->astc_trit_pk[i] = (uint16_t)((tb & 3u) | ((tb & 0xCu) << 2) | ((tb & 0x10u) << 4) | ((tb & 0x60u) << 6) | ((tb & 0x80u) << 8));
    }
    for (int i = 0; i < 512; i++) {
        const int d0 = i & 7, d1 = (i >> 3) & 7, d2 = i >> 6;
        const uint32_t qb = (d0 < 5 && d1 < 5 && d2 < 5) ? BU_ASTC_QUINT_ENC[(d2 * 5 + d1) * 5 + d0] : 0;
        t->astc_quint_pk[i] = (uint16_t)(((qb & 7u) << 3) | ((qb & 0x18u) << 6) | ((qb & 0x60u) << 9));
    }
    for (int i = 0; i < 256; i++) {
        const int tl = i & 3, th = (i >> 2) & 3, el = (i >> 4) & 3, eh = (i >> 6) & 3;
        // range 7 sits at offset 0 of deq[] with index tq << 2 | eb (bu_deq_ofs)
        t->pairdiff7[i] = (tl < 3 && th < 3) ? (int16_t)((int)t->deq[(th << 2) | eh] - (int)t->deq[(tl << 2) | el]) : 0;
    }
    for (int i = 0; i < 1024; i++) {
        const int a = i & 31, b = i >> 5;
        const int va = (a >> 1) - (a == 14) + (a == 17), vb = (b >> 1) - (b == 14) + (b == 17);
        t->w5to4x2[i] = (uint8_t)(va | (vb << 4));
    }
    for (int bits = 1; bits <= 5; bits++)
        for (int r = 0; r < (1 << bits); r++) {
            // uastc.rs:697-719 (LUT1..LUT5) as arithmetic, then the operand form of the v_dot2 interpolation
            const int w = bits == 1 ? r << 6 : bits == 2 ? r * 21 + (r >> 1) : bits == 3 ? r * 9 + (r >> 2) : bits == 4 ? r * 4 + (r >> 2) + (r >> 3) : r * 2 + ((r >> 4) << 1);
            t->wpack[(1 << bits) - 2 + r] = (uint32_t)w * 0x3FFFCu + 256u;
        }
    for (int d = 0; d < 2; d++)
        for (int inten = 0; inten < 8; inten++)
            for (int c = 0; c < 32; c++) {
                const int base = d ? ((c << 3) | (c >> 2)) : ((c & 15) * 17);
                uint32_t v = 0;
                for (int k = 0; k < 4; k++) {
                    int x = base + BU_ETC1_MOD[inten * 4 + k];
                    x = x < 0 ? 0 : (x > 255 ? 255 : x);
                    v |= (uint32_t)x << (8 * k);
                }
                t->etc1_thrcol[(d << 8) | (inten << 5) | c] = v;
            }
    for (int i = 0; i < 16; i++) {
        int mn = BU_ETC2_ALPHA_MOD[8 * i + 3], mx = BU_ETC2_ALPHA_MOD[8 * i + 7];
        int range = mx - mn;
        t->eac_mod_min[i] = (int8_t)mn;
        t->eac_range[i] = (uint8_t)range;
        t->eac_magic[i] = (uint32_t)(((1u << 20) + 2 * range - 1) / (2 * range));  // exact for num*2*range < 2^20
    }
}
