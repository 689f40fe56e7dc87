// Device-side lookup tables of the UASTC/ETC1S transcode kernels, and the host code that builds them.
//
// One `BuTables` blob is built on the host at context creation (bu_build_tables), uploaded once to
// device memory and copied into LDS by every workgroup -- the part its target reads, 4 to 27 KiB (bu_table_range);
// per-lane divergent lookups are LDS reads, never global gathers.  All LUTs are derived here from the format constants of
// bu_tables.h, so the kernels never divide, never take a modulo and never run a search.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define BU_TABLE static constexpr
#include "bu_tables.h"

// Orderings the ETC code relies on, checked where the tables are compiled in:
//  - every EAC modifier table is monotone in the entry order 3,2,1,0,4,5,6,7 (bu_eac_block counts thresholds instead of
//    searching the eight values; etc.rs:450-468)
//  - every ETC1 modifier row rises (the three luma thresholds of a sub-block are monotone; etc.rs:435-445)
constexpr bool bu_etc_tables_ordered()
{
    for (int t = 0; t < 16; t++) {
        const int by_rank[8] = {3, 2, 1, 0, 4, 5, 6, 7};
        for (int r = 1; r < 8; r++)
            if (BU_ETC2_ALPHA_MOD[8 * t + by_rank[r - 1]] >= BU_ETC2_ALPHA_MOD[8 * t + by_rank[r]]) return false;
    }
    for (int i = 0; i < 8; i++)
        for (int k = 1; k < 4; k++)
            if (BU_ETC1_MOD[4 * i + k - 1] >= BU_ETC1_MOD[4 * i + k]) return false;
    return true;
}
static_assert(bu_etc_tables_ordered(), "EAC / ETC1 modifier tables are not in the order the threshold forms assume");

struct alignas(16) BuPart {  // one UASTC partition pattern (layout documented in bu_tables.h / tools/gen_tables.py)
    // (upat, seed and uanch are what every target but BC7 reads: adjacent, so that they arrive in ONE ds_read_b64 -- a ds_read2_b32
    // of two separate dwords costs 79 clocks per SIMD against 24, tools/exp/ldsbench.hip)
    uint32_t upat;
    uint16_t seed, uanch;
    uint32_t bpat;
    uint16_t banch;
    uint8_t bpart, perm;
};
static_assert(sizeof(BuPart) == 16, "BuPart must be 16 bytes");
// The BC7 packer's own view of a partition pattern: everything bu_block_bc7 derives from a BuPart at run time (anchor bit
// positions, sorted deletion order, placement shifts), precomputed on the host.  "Label" = which format's subset numbering
// the record speaks: BC7's for the 2-subset patterns [0,30), the mode-7 family [41,60) and the mode-1 record 60; UASTC's for
// the 3-subset family [30,41) (UASTC mode 3 places whole per-subset endpoint fields with a variable shift, so its subsets keep
// their UASTC numbers and only the field position knows about BC7's order).  All bit positions are for 2-bit weights
// (2 * texel + 1 = the weight's MSB) unless stated.
struct alignas(16) BuPart7 {
    uint32_t pat;  // subset of every texel in the record's label, 2 bits per texel, texel 0 low
    uint32_t pos;  // 5-bit fields: [0] uq_lo, [5] uq_hi: MSB positions of the UASTC anchors other than texel 0, ascending (31 = none)
                   //   [10] dq_hi, [15] dq_lo: MSB positions of the BC7 anchors other than texel 0, DEscending (dq_lo 0 = none)
                   //   [20] aq_0, [25] aq_1: MSB position of the BC7 anchor of subset 0 / 1 of the label
    uint32_t aux;  // [0:5) aq_2, [5:11) BC7 partition id, [11:16) sh_0, [16:21) sh_1, [21:26) sh_2: 10 x the BC7 slot of UASTC
                   //   subset s (family [30,41) only), [26:32) UASTC source subset of BC7 subset j, 2 bits each (bc7.rs:144)
    uint32_t w3;   // 3-bit weights (UASTC mode 2, patterns [0,30)): [0:6) 3 * ua + 2 (the UASTC anchor's MSB), [6:12) 3 * a1 + 2 (BC7's)
};
static_assert(sizeof(BuPart7) == 16, "BuPart7 must be 16 bytes");
struct alignas(8) BuU2 {
    uint32_t x, y;
};
struct alignas(16) BuU4 {
    uint32_t x, y, z, w;
};
// (sum * limit + 1020) / 2040 = (sum * (limit * BU_Q_M) + 1020 * BU_Q_M) >> 26 for every sum <= 2040, limit <= 31 (bu_build_tables checks)
constexpr uint32_t BU_Q_M = 32897u;  // ceil(2^26 / 2040)
constexpr bool bu_q_exact()
{
    for (uint32_t limit = 15; limit <= 31; limit += 16)
        for (uint32_t sum = 0; sum <= 2040; sum++)
            if ((sum * (limit * BU_Q_M) + 1020u * BU_Q_M) >> 26 != (sum * limit + 1020u) / 2040u) return false;
    return true;
}
static_assert(bu_q_exact(), "multiply-shift form of the ETC1 base colour quantiser");
static_assert(31u * BU_Q_M < (1u << 24), "the quantiser runs on v_mad_u32_u24");

// offsets (bytes) into BuTables::deq of the dequantisation LUT of each BISE range UASTC uses
// (index = trit_or_quint << bits | bits_value); filled from BU_BISE[].lut_ofs_div8
// Members are grouped by who reads them, each group 16-byte aligned, so that a kernel stages only what its target needs
// (bu_table_range below): every workgroup copies its tables from L2 into LDS, and with four workgroups per CU on
// 1024-block tiles the whole 9.3 KiB blob would be more than half of the 16 KiB of payload the workgroup moves.
//   [BC7 only][common front-end][texel unpack: RGBA32, ETC][ASTC only][ETC only]
// Modes by descending cost of their code path, per target (tools/exp/mode_isa.py): the mode-sorted kernel lays the runs out in
// this order so that dynamically scheduled chunks end with the cheap ones.  Entry 19 = invalid mode code.
constexpr uint8_t BU_COST_ORDER[5][20] = {  // [target: ASTC, BC7, ETC1, ETC2, RGBA32], est. SIMD clocks of round 3's paths
    {3, 4, 7, 11, 12, 10, 9, 0, 2, 6, 18, 13, 14, 5, 1, 16, 17, 15, 8, 19},
    {4, 7, 9, 11, 12, 3, 16, 6, 18, 2, 5, 10, 14, 1, 0, 17, 13, 8, 15, 19},
    {3, 2, 4, 7, 18, 6, 9, 0, 11, 12, 10, 5, 16, 14, 1, 13, 15, 17, 8, 19},
    {12, 10, 9, 15, 11, 16, 14, 13, 3, 2, 17, 4, 7, 6, 18, 0, 5, 1, 8, 19},
    {2, 3, 12, 10, 11, 9, 18, 4, 7, 0, 6, 5, 14, 13, 16, 17, 15, 1, 8, 19},
};

// The BC7 packer's own tables (bu_uastc_bc7.hpp).  They sit IN FRONT of the common blob, in device memory and in the LDS of the
// BC7 kernels alike (BuTablesAll), so that only BC7 reserves and stages them; bu_bc7_tables() steps back from the common blob.
struct BuBc7Tables {
    // ---- BC7 only ----
    alignas(16) uint8_t deq5[56];  // (deq*31+127)/255 for ranges 7 (ofs 0) and 12 (ofs 16): BC7 mode 2 endpoints (bc7.rs:262-264)
    uint32_t w3mask[30][2];  // 2-subset patterns, BC7 subset-1 texel mask at 3 bits/texel (48 bits)
    uint32_t pbit6[256];   // unique p-bit LUT, 6 total bits: q0>>1 | (q1>>1)<<8 | err0<<16 | err1<<24 (bc7.rs:478-553)
    uint32_t pbit7[16];    // shared p-bit LUT, 7 total bits, input = 17*index (bc7.rs:408-475)
    uint16_t m5opt[256];   // BC7 mode 5 solid colour lo | hi<<8 (bc7.rs:734-863)
    uint16_t m6opt[258];   // BC7 mode 6 solid colour lo | hi<<8, index c + !p (bc7.rs:866-1136)
    uint8_t w5to4x2[1024];    // two 5-bit weights (10 bits) -> two 4-bit BC7 weights, x>>1 except 14 -> 6 and 17 -> 9 (bc7.rs:381-384)
    BuPart7 part7[61];        // the BC7 packer's partition records (same indexing as `part`)
    // BISE range 7 endpoint PAIR (lo, hi) of one channel -> BC7 mode 2's two 5-bit fields (UASTC mode 3): index = eb_lo | eb_hi << 2 |
    // tq_lo << 4 | tq_hi << 6; value = (lo5 | hi5 << 5) | (hi5 | lo5 << 5) << 16, the swapped pair for an inverted subset
    uint32_t pair7x[256];
    // 4-bit endpoint PAIR (lo, hi) of one channel -> BC7 mode 1 (6 bits + p-bit shared by the subset's six values; UASTC mode 2):
    // index = lo | hi << 4 (+ 256: swapped), value = q0_lo | q0_hi << 6 | q1_lo << 12 | q1_hi << 18 (the 6-bit fields for p = 0 / 1)
    // | (8 + err1 - err0) << 24, err_p = the pair's summed squared error of 17 * v under p-bit p (at most 4 per value): the subset takes
    // p = 1 iff the three channels' bytes sum to less than 24 (bc7.rs:408-475: strictly smaller error, ties -> 0)
    uint32_t pair4m1[512];
    // 4-bit endpoint PAIR (lo, hi) of one channel -> BC7 mode 7 (5 bits + one p-bit per endpoint; UASTC mode 9): index = lo | hi << 4,
    // x = q0_lo | q1_lo << 5 | q0_hi << 10 | q1_hi << 15, y = err0_lo | err1_lo << 8 | err0_hi << 16 | err1_hi << 24
    BuU2 pair4m7[256];
};
static_assert(sizeof(BuBc7Tables) % 16 == 0, "staged in 16-byte pieces, and the common blob behind it stays 16-byte aligned");

struct BuTables {
    // ---- common front-end (every target) ----
    alignas(16) uint16_t trit5[256];  // 8-bit group -> 5 trits, digit i in bits [2i,2i+2)   (uastc.rs:657-685)
    uint16_t quint3[128];  // 7-bit group -> 3 quints, digit i in bits [3i,3i+3)  (uastc.rs:629-655)
    uint8_t deq[504];      // endpoint dequantisation, ranges 7,8,11,12,13,18,19  (uastc.rs:585-614)
    uint8_t mode_lut[128];    // uastc.rs:560-577
    uint8_t key_lut[5][128];  // sort key of the mode-sorted kernel, per target: position of the block's mode in BU_COST_ORDER[target] (19 = invalid code)
    alignas(16) BuPart part[61];  // partition records (every target but BC7, which reads part7: last of the group, outside BC7's range)
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
    alignas(16) int16_t etc1_mod[32];  // etc.rs:435-445 (the ETC1S kernels)
    int8_t eac_mods[128];     // etc.rs:450-468, every table in ascending order of the modifier: entries 3,2,1,0,4,5,6,7
    uint32_t eac_magic[16];   // ceil(2^20 / (2*range)) per EAC modifier table (etc.rs:297-307 as integers)
    int8_t eac_mod_min[16];   // modifier[3] of each table
    uint8_t eac_range[16];    // modifier[7] - modifier[3]
    // the eight ETC1 flag bits as stored (flip, diff, inten0:3, inten1:3; uastc.rs:411-436) -> everything derived from them:
    //   x = header byte 3 << 24 | (diff << 3 | inten1) << 16 | (diff << 3 | inten0) << 8   (the two fields are etc1_thr row offsets)
    //   y = diff << 7 in bytes 0..2 (etc1_biasv index bit), z = byte offset of the diff half of etc1_hdr,
    //   w = limit * BU_Q_M (limit = 15 or 31: the quantiser's multiplier, bu_uastc_etc.hpp)
    alignas(16) BuU4 etc1_flags[256];
    BuU2 etc1_bias2[32];      // apply_etc1_bias deltas (etc.rs:203-234): x = sub-block 0, y = sub-block 1; byte c = (delta + 2) << 5
    uint8_t etc1_biasv0[256]; // apply_etc1_bias per channel (etc.rs:236-255): index diff << 7 | (delta + 2) << 5 | value -> result << 3
    uint8_t etc1_biasv1[256]; // the same -> result << 1
    // diff << 10 | c0 << 5 | c1 (the biased 4/5-bit base colours of one channel) -> header byte (etc.rs:113-149) | (the value the
    // second half really decodes with: c1, or (c0 + clamped delta) & 31) << 11
    uint16_t etc1_hdr[2048];
    // [channel][diff << 8 | inten << 5 | c]: with v0..v3 = clamp(base + ETC1_MODIFIERS[inten][k]) (etc.rs:165-171, 396-431) and
    // w = the channel's luma weight 54 / 183 / 19: x = -w (v0 + v1), y = w (v2 - v0) | w (v3 - v1) << 16.  Summed over the
    // channels these are -(L0 + L1), L2 - L0 and L3 - L1, from which the three luma thresholds follow by two subtractions.
    alignas(16) BuU2 etc1_thr[3][512];
    alignas(16) uint8_t end_marker[16];
    // ---- ETC1S kernels only (behind every UASTC target's ranges: read from device memory / staged by those kernels themselves) ----
    // the four colours of one channel of an ETC1S endpoint as a byte palette: [inten << 5 | c5] -> clamp(extend5(c5) + modifier[inten][k])
    // in byte k (etc.rs:396-431)
    alignas(16) uint32_t etc1s_pal[256];
};
static_assert(sizeof(BuTables) % 16 == 0, "BuTables is copied to LDS in 16-byte pieces");
struct BuTablesAll {  // the device blob: one allocation, one kernel argument
    BuBc7Tables b7;
    BuTables t;
};
static_assert(offsetof(BuTablesAll, t) == sizeof(BuBc7Tables), "bu_bc7_tables() steps back by sizeof(BuBc7Tables)");

// byte ranges [lo, hi) of BuTables a target reads (target ids as in bu_uastc_dispatch.hpp: 0 ASTC, 1 BC7, 2 ETC1, 3 ETC2,
// 4 RGBA32); a second range is empty unless lo2 < hi2.  All bounds are multiples of 16.
struct BuTableRange {
    unsigned lo, hi, lo2, hi2;
};
constexpr BuTableRange bu_table_range(int target)
{
    return target == 1   ? BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, part), 0u, 0u}
           : target == 0 ? BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, etc1_mod), 0u, 0u}
           : target == 4 ? BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, w3mask_u), 0u, 0u}
                         : BuTableRange{(unsigned)offsetof(BuTables, trit5), (unsigned)offsetof(BuTables, w3mask_u),
                                        (unsigned)offsetof(BuTables, etc1_mod), (unsigned)offsetof(BuTables, end_marker)};
}

// bytes of LDS a target's kernels reserve for the blob (its staged ranges keep their offsets: the front of the struct)
constexpr unsigned bu_table_bytes(int target) { return bu_table_range(target).lo2 < bu_table_range(target).hi2 ? bu_table_range(target).hi2 : bu_table_range(target).hi; }

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

static inline void bu_build_tables(BuTablesAll* all)
{
    memset(all, 0, sizeof(*all));
    BuTables* t = &all->t;
    BuBc7Tables* b7 = &all->b7;
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
    for (int i = 0; i < 12; i++) b7->deq5[i] = (uint8_t)((t->deq[bu_deq_ofs(7) + i] * 31 + 127) / 255);
    for (int i = 0; i < 40; i++) b7->deq5[16 + i] = (uint8_t)((t->deq[bu_deq_ofs(12) + i] * 31 + 127) / 255);
    for (int i = 0; i < 61; i++) {
        t->part[i].upat = BU_PART[i].upat;
        t->part[i].bpat = BU_PART[i].bpat;
        t->part[i].seed = BU_PART[i].seed;
        t->part[i].uanch = BU_PART[i].uanch;
        t->part[i].banch = BU_PART[i].banch;
        t->part[i].bpart = BU_PART[i].bpart;
        t->part[i].perm = BU_PART[i].perm;
    }
    for (int i = 0; i < 61; i++) {
        const bool three = i >= BU_PART_BASE3 && i < BU_PART_BASE23;  // UASTC label
        const int nb = (i < BU_PART_BASE3 || i == BU_PART_MODE1) ? 2 : 3;       // BC7 subsets
        const int nu = i == BU_PART_MODE1 ? 1 : (i < BU_PART_BASE3 || i >= BU_PART_BASE23) ? 2 : 3;  // UASTC subsets
        int ua[3], ba[3], src[3], slot[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            ua[k] = (BU_PART[i].uanch >> (4 * k)) & 15;
            ba[k] = (BU_PART[i].banch >> (4 * k)) & 15;
            src[k] = (BU_PART[i].perm >> (2 * k)) & 3;
        }
        for (int j = 0; j < nb; j++)
            if (src[j] < 3) slot[src[j]] = j;  // (the 3-subset family's perm is a bijection)
        int un[2] = {31, 31}, n = 0;  // UASTC anchors other than texel 0, ascending
        for (int k = 0; k < nu; k++)
            if (ua[k]) un[n++] = 2 * ua[k] + 1;
        if (n == 2 && un[0] > un[1]) { const int x = un[0]; un[0] = un[1]; un[1] = x; }
        int dn[2] = {0, 0};  // BC7 anchors other than texel 0, descending
        n = 0;
        for (int j = 1; j < nb; j++) dn[n++] = 2 * ba[j] + 1;
        if (n == 2 && dn[0] < dn[1]) { const int x = dn[0]; dn[0] = dn[1]; dn[1] = x; }
        int aq[3];
        for (int k = 0; k < 3; k++) aq[k] = three ? 2 * ba[slot[k]] + 1 : 2 * ba[k] + 1;
        BuPart7& r = b7->part7[i];
        r.pat = three ? BU_PART[i].upat : BU_PART[i].bpat;
        r.pos = (uint32_t)un[0] | (uint32_t)un[1] << 5 | (uint32_t)dn[0] << 10 | (uint32_t)dn[1] << 15 | (uint32_t)aq[0] << 20 | (uint32_t)aq[1] << 25;
        r.aux = (uint32_t)aq[2] | (uint32_t)BU_PART[i].bpart << 5 | (uint32_t)(10 * slot[0]) << 11 | (uint32_t)(10 * slot[1]) << 16 |
                (uint32_t)(10 * slot[2]) << 21 | (uint32_t)(BU_PART[i].perm & 63u) << 26;
        r.w3 = i < BU_PART_BASE3 ? (uint32_t)(3 * (ua[0] | ua[1]) + 2) | (uint32_t)(3 * ba[1] + 2) << 6 : 0u;
    }
    for (int i = 0; i < 256; i++) {
        const int el = i & 3, eh = (i >> 2) & 3, tl = (i >> 4) & 3, th = (i >> 6) & 3;
        const uint32_t lo5 = b7->deq5[(tl << 2) | el], hi5 = b7->deq5[(th << 2) | eh];
        b7->pair7x[i] = (tl < 3 && th < 3) ? ((lo5 | hi5 << 5) | (hi5 | lo5 << 5) << 16) : 0u;
    }
    for (int i = 0; i < 30; i++) {
        uint64_t m = 0;
        for (int tx = 0; tx < 16; tx++)
            if ((BU_PART[i].bpat >> (2 * tx)) & 1) m |= 7ull << (3 * tx);
        b7->w3mask[i][0] = (uint32_t)m;
        b7->w3mask[i][1] = (uint32_t)(m >> 32);
    }
    for (int x = 0; x < 256; x++) {  // 6 total bits (BC7 mode 7)
        int q0 = bu_quant_p(x, 63, 0), q1 = bu_quant_p(x, 63, 1);
        int s0 = ((q0 << 2) | (q0 >> 4)) & 255, s1 = ((q1 << 2) | (q1 >> 4)) & 255;
        int e0 = (s0 - x) * (s0 - x), e1 = (s1 - x) * (s1 - x);
        b7->pbit6[x] = (uint32_t)(q0 >> 1) | (uint32_t)(q1 >> 1) << 8 | (uint32_t)e0 << 16 | (uint32_t)e1 << 24;
    }
    for (int i = 0; i < 16; i++) {  // 7 total bits (BC7 mode 1), inputs are multiples of 17
        int x = 17 * i;
        int q0 = bu_quant_p(x, 127, 0), q1 = bu_quant_p(x, 127, 1);
        int s0 = ((q0 << 1) | (q0 >> 6)) & 255, s1 = ((q1 << 1) | (q1 >> 6)) & 255;
        int e0 = (s0 - x) * (s0 - x), e1 = (s1 - x) * (s1 - x);
        b7->pbit7[i] = (uint32_t)(q0 >> 1) | (uint32_t)(q1 >> 1) << 8 | (uint32_t)e0 << 16 | (uint32_t)e1 << 24;
    }
    for (int i = 0; i < 512; i++) {
        const int a = (i & 256) ? (i >> 4) & 15 : i & 15, b = (i & 256) ? i & 15 : (i >> 4) & 15;  // (lo, hi), swapped above 256
        const uint32_t ea = b7->pbit7[a], eb = b7->pbit7[b];
        const uint32_t e0 = ((ea >> 16) & 255u) + ((eb >> 16) & 255u), e1 = (ea >> 24) + (eb >> 24);
        b7->pair4m1[i] = (ea & 63u) | (eb & 63u) << 6 | ((ea >> 8) & 63u) << 12 | ((eb >> 8) & 63u) << 18 | (8u + e1 - e0) << 24;
    }
    for (int i = 0; i < 256; i++) {
        const uint32_t ea = b7->pbit6[17 * (i & 15)], eb = b7->pbit6[17 * (i >> 4)];
        b7->pair4m7[i].x = (ea & 31u) | ((ea >> 8) & 31u) << 5 | (eb & 31u) << 10 | ((eb >> 8) & 31u) << 15;
        b7->pair4m7[i].y = ((ea >> 16) & 255u) | (ea >> 24) << 8 | ((eb >> 16) & 255u) << 16 | (eb >> 24) << 24;
    }
    for (int i = 0; i < 256; i++) b7->m5opt[i] = BU_BC7_M5_OPT[i];
    for (int i = 0; i < 257; i++) b7->m6opt[i] = BU_BC7_M6_OPT[i];
    for (int i = 0; i < 243; i++) t->astc_trit[i] = BU_ASTC_TRIT_ENC[i];
    for (int i = 0; i < 125; i++) t->astc_quint[i] = BU_ASTC_QUINT_ENC[i];
    for (int i = 0; i < 20; i++) t->astc_mode13[i] = BU_ASTC_BLOCK_MODE13[i];
    for (int i = 0; i < 32; i++) t->etc1_mod[i] = BU_ETC1_MOD[i];
    for (int inten = 0; inten < 8; inten++)
        for (int c5 = 0; c5 < 32; c5++) {
            uint32_t pal = 0;
            for (int k = 0; k < 4; k++) {
                const int v = ((c5 << 3) | (c5 >> 2)) + BU_ETC1_MOD[inten * 4 + k];
                pal |= (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * k);
            }
            t->etc1s_pal[(inten << 5) | c5] = pal;
        }
    for (int i = 0; i < 128; i++) {
        static const int by_rank[8] = {3, 2, 1, 0, 4, 5, 6, 7};
        t->eac_mods[i] = BU_ETC2_ALPHA_MOD[(i & ~7) + by_rank[i & 7]];
    }
    for (int i = 0; i < 128; i++) t->mode_lut[i] = BU_MODE_LUT[i];
    for (int i = 0; i < 128; i++) {
        for (int tg = 0; tg < 5; tg++) {
            t->key_lut[tg][i] = 19;
            for (int k = 0; k < 20; k++)
                if (BU_COST_ORDER[tg][k] == BU_MODE_LUT[i]) t->key_lut[tg][i] = (uint8_t)k;
        }
    }
    for (int i = 0; i < 30; i++) {
        uint64_t m = 0;
        for (int tx = 0; tx < 16; tx++)
            if ((BU_PART[i].upat >> (2 * tx)) & 1) m |= 7ull << (3 * tx);
        t->w3mask_u[i][0] = (uint32_t)m;
        t->w3mask_u[i][1] = (uint32_t)(m >> 32);
    }
    for (int bias = 0; bias < 32; bias++) {  // etc.rs:203-234 tabulated
        uint32_t packed[2] = {0, 0};
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
                packed[sb] |= (uint32_t)((delta + 2) << 5) << (8 * c);
            }
        t->etc1_bias2[bias].x = packed[0];
        t->etc1_bias2[bias].y = packed[1];
    }
    for (int d = 0; d < 2; d++)
        for (int dc = 0; dc < 4; dc++)
            for (int v = 0; v < 32; v++) {
                const int limit = d ? 31 : 15;
                const int r = v <= limit ? bu_etc1_bias1_host(v, dc - 2, limit) : 0;
                t->etc1_biasv0[(d << 7) | (dc << 5) | v] = (uint8_t)(r << 3);
                t->etc1_biasv1[(d << 7) | (dc << 5) | v] = (uint8_t)(r << 1);
            }
    for (int raw = 0; raw < 256; raw++) {
        const uint32_t f = raw & 1, d = (raw >> 1) & 1, i0 = (raw >> 2) & 7, i1 = (raw >> 5) & 7;
        const uint32_t hdr3 = ((i0 << 5) | (i1 << 2) | (d << 1) | f) & 0xFFu;  // etc.rs:151-158
        t->etc1_flags[raw].x = hdr3 << 24 | ((d << 3) | i1) << 16 | ((d << 3) | i0) << 8;
        t->etc1_flags[raw].y = d ? 0x808080u : 0u;
        t->etc1_flags[raw].z = d * 2048u;
        t->etc1_flags[raw].w = (d ? 31u : 15u) * BU_Q_M;
    }
    for (int d = 0; d < 2; d++)
        for (int c0 = 0; c0 < 32; c0++)
            for (int c1 = 0; c1 < 32; c1++) {
                uint32_t byte, cq1;
                if (!d) {  // individual: 4 + 4 bits (etc.rs:122-129)
                    byte = ((uint32_t)(c0 << 4) | (uint32_t)c1) & 0xFFu;
                    cq1 = (uint32_t)c1;
                } else {  // differential: 5 bits + clamped 3-bit delta; the second half decodes from c0 + delta (etc.rs:130-149)
                    int dl = c1 - c0;
                    dl = dl < -4 ? -4 : (dl > 3 ? 3 : dl);
                    byte = ((uint32_t)(c0 << 3) | ((uint32_t)dl & 7u)) & 0xFFu;
                    cq1 = (uint32_t)((c0 + dl) & 31);
                }
                t->etc1_hdr[(d << 10) | (c0 << 5) | c1] = (uint16_t)(byte | cq1 << 11);
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
        b7->w5to4x2[i] = (uint8_t)(va | (vb << 4));
    }
    for (int bits = 1; bits <= 5; bits++)
        for (int r = 0; r < (1 << bits); r++) {
            // uastc.rs:697-719 (LUT1..LUT5) as arithmetic, then the operand form of the v_dot2 interpolation
            const int w = bits == 1 ? r << 6 : bits == 2 ? r * 21 + (r >> 1) : bits == 3 ? r * 9 + (r >> 2) : bits == 4 ? r * 4 + (r >> 2) + (r >> 3) : r * 2 + ((r >> 4) << 1);
            t->wpack[(1 << bits) - 2 + r] = (uint32_t)w * 0x3FFFCu + 256u;
        }
    for (int ch = 0; ch < 3; ch++)
        for (int d = 0; d < 2; d++)
            for (int inten = 0; inten < 8; inten++)
                for (int c = 0; c < 32; c++) {
                    static const int lw[3] = {54, 183, 19};  // half the reference's luma weights 108, 366, 38 (etc.rs:165-177)
                    const int base = d ? ((c << 3) | (c >> 2)) : ((c & 15) * 17);
                    int v[4];
                    for (int k = 0; k < 4; k++) {
                        int x = base + BU_ETC1_MOD[inten * 4 + k];
                        v[k] = x < 0 ? 0 : (x > 255 ? 255 : x);
                    }
                    BuU2& e = t->etc1_thr[ch][(d << 8) | (inten << 5) | c];
                    e.x = 0u - (uint32_t)(lw[ch] * (v[0] + v[1]));
                    e.y = (uint32_t)(lw[ch] * (v[2] - v[0])) | (uint32_t)(lw[ch] * (v[3] - v[1])) << 16;
                }
    for (int i = 0; i < 16; i++) {
        int mn = BU_ETC2_ALPHA_MOD[8 * i + 3], mx = BU_ETC2_ALPHA_MOD[8 * i + 7];
        int range = mx - mn;
        t->eac_mod_min[i] = (int8_t)mn;
        t->eac_range[i] = (uint8_t)range;
        t->eac_magic[i] = (uint32_t)(((1u << 20) + 2 * range - 1) / (2 * range));  // exact for num*2*range < 2^20
    }
}
