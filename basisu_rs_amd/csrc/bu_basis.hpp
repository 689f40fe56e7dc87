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

#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#if defined(__linux__)
#include <sched.h>
#endif
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/basisu_hip.h"

namespace bu_host {

// ---- host worker pool --------------------------------------------------------------------------------------------
// The reference is single-threaded; the host work that stays serial per item here (one slice's symbol stream, one piece
// of a CRC) is spread over items.  Threads are created once (thread start costs ~50 us, as much as half a small slice)
// and parked on a condition variable.  A forked child finds no threads and runs inline.
class Pool {
    std::mutex run_m, m;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    const std::function<void()>* job = nullptr;
    unsigned epoch = 0, want = 0, started = 0, running = 0, active_helpers = 0;
    bool stop = false;
    int owner_pid = 0;

    void worker()
    {
        unsigned seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || (epoch != seen && started < want); });
            if (stop) return;
            seen = epoch;
            started++;
            running++;
            const std::function<void()>* f = job;
            lk.unlock();
            (*f)();
            lk.lock();
            if (--running == 0) cv_done.notify_all();
        }
    }

public:
    // helper threads worth starting: the CPUs this process may actually run on (its affinity mask -- a container pinned to a few
    // cores reports the machine's count through hardware_concurrency(), and helpers that only spin on each other's progress then compete
    // with the one thread that makes it), at most 32
    static unsigned capacity()
    {
        unsigned hw = std::thread::hardware_concurrency();
#if defined(__linux__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int n = CPU_COUNT(&set);
            if (n > 0 && (hw == 0 || (unsigned)n < hw)) hw = (unsigned)n;
        }
#endif
        if (hw == 0) hw = 1;
        return hw < 32u ? hw : 32u;
    }
    // runs f on the calling thread and on up to `helpers` pool threads at once; returns when every copy has returned
    void run(unsigned helpers, const std::function<void()>& f)
    {
        begin(helpers, f);
        try {
            f();
        } catch (...) {
            end();
            throw;
        }
        end();
    }
    // the two halves of run() for a caller that has something else to do meanwhile (bu_read_to feeds the GPU while pool threads
    // decode): begin() starts up to `helpers` copies of f on pool threads and returns how many it started (0: no thread could
    // be had -- the caller must run f itself); end() waits for them.  The pool is reserved from begin() to end() -- ONE job at a time
    // per process: a begin() on another thread WAITS here until the current job's end() (concurrent whole-file calls on different
    // contexts therefore take turns on the host side of their work).  f must stay alive until end() returns.
    unsigned begin(unsigned helpers, const std::function<void()>& f)
    {
        run_m.lock();
        const int pid = (int)getpid();
        if (owner_pid != pid) {  // first use, or a forked child (threads do not survive fork: forget them)
            for (std::thread& t : th) t.detach();
            th.clear();
            owner_pid = pid;
        }
        if (helpers > capacity()) helpers = capacity();
        try {
            while (th.size() < helpers) th.emplace_back([this] { worker(); });
        } catch (...) {  // no more threads to be had (std::system_error): run with the helpers that exist -- every job
            helpers = (unsigned)th.size();  // pulls its items from a shared counter, so fewer copies only means less overlap
        }
        active_helpers = helpers;
        if (helpers) {
            std::lock_guard<std::mutex> lk(m);
            job = &f;
            want = helpers;
            started = 0;
            running = 0;  // counts copies that have actually started (worker()): end() waits for those, not for wake-ups that never came
            epoch++;
            // One wake-up per helper wanted (a pool that once ran 32 copies has 32 parked threads: waking them all for a 3-thread job
            // queues 29 of them on the mutex in front of the three that have work) -- issued UNDER the mutex: a woken thread cannot
            // run, finish and park again while the loop is still notifying, so every notification reaches a different parked thread.
            // (Notified outside the lock, a copy that finished at once re-entered the wait queue and swallowed the next
            // notification: the job then waited for a copy that was never going to start.)
            for (unsigned i = 0; i < helpers; i++) cv.notify_one();
        }
        return helpers;
    }
    void end()
    {
        if (active_helpers) {
            std::unique_lock<std::mutex> lk(m);
            want = 0;  // admissions are closed: a helper that has not started by now sits this job out (the items come from a shared counter)
            cv_done.wait(lk, [&] { return running == 0; });
            job = nullptr;
        }
        active_helpers = 0;
        run_m.unlock();
    }
    ~Pool()
    {
        if (owner_pid != (int)getpid()) {
            for (std::thread& t : th) t.detach();
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (std::thread& t : th) t.join();
    }
};
inline Pool& pool()
{
    static Pool p;
    return p;
}

// ---- CRC-16/GENIBUS (basis.rs:364-372): poly 0x1021, init 0xFFFF, xorout 0xFFFF, not reflected ----
// Slicing-by-8: t[k][b] = register after byte b followed by k zero bytes, so 8 message bytes cost 8 independent lookups.
struct Crc16Table {
    uint16_t t[8][256];
    Crc16Table()
    {
        for (int b = 0; b < 256; b++) {
            uint16_t crc = (uint16_t)(b << 8);
            for (int k = 0; k < 8; k++) crc = (uint16_t)((crc & 0x8000) ? ((crc << 1) ^ 0x1021) : (crc << 1));
            t[0][b] = crc;
        }
        for (int k = 1; k < 8; k++)
            for (int b = 0; b < 256; b++) t[k][b] = (uint16_t)((t[k - 1][b] << 8) ^ t[0][t[k - 1][b] >> 8]);
    }
};
// raw register update (no init / final complement)
inline uint16_t crc16_raw(const uint8_t* p, size_t n, uint16_t s)
{
    static const Crc16Table tab;
    while (n >= 8) {
        s = (uint16_t)(tab.t[7][p[0] ^ (s >> 8)] ^ tab.t[6][p[1] ^ (s & 0xFF)] ^ tab.t[5][p[2]] ^ tab.t[4][p[3]] ^ tab.t[3][p[4]] ^ tab.t[2][p[5]] ^
                       tab.t[1][p[6]] ^ tab.t[0][p[7]]);
        p += 8;
        n -= 8;
    }
    for (size_t i = 0; i < n; i++) s = (uint16_t)((s << 8) ^ tab.t[0][((s >> 8) ^ p[i]) & 0xFF]);
    return s;
}
// a * b in GF(2)[x] / (x^16 + x^12 + x^5 + 1)
inline uint16_t crc16_gf_mul(uint16_t a, uint16_t b)
{
    uint32_t r = 0;
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        if (r & 0x10000u) r ^= 0x11021u;
        if ((b >> i) & 1u) r ^= a;
    }
    return (uint16_t)r;
}
// register s after n zero bytes = s * x^(8n)
inline uint16_t crc16_shift(uint16_t s, size_t n)
{
    uint16_t result = 1, base = 0x0100;
    for (; n; n >>= 1) {
        if (n & 1) result = crc16_gf_mul(result, base);
        base = crc16_gf_mul(base, base);
    }
    return crc16_gf_mul(s, result);
}
// The CRC is linear: register(s, A || B) = register(s, A) * x^(8|B|) + register(0, B).  Large payloads (a 4096^2 UASTC
// file is 16 MiB; config 5 is 512 MiB) are cut into pieces whose registers are computed concurrently and then folded.
inline uint16_t crc16(const uint8_t* p, size_t n, uint16_t crc)
{
    uint16_t s = (uint16_t)~crc;
    constexpr size_t PIECE = (size_t)1 << 18;
    const size_t pieces = (n + PIECE - 1) / PIECE;
    if (pieces < 4) return (uint16_t)~crc16_raw(p, n, s);
    std::vector<uint16_t> part(pieces, 0);
    std::atomic<size_t> next{0};
    const std::function<void()> work = [&] {
        for (size_t k; (k = next.fetch_add(1)) < pieces;) {
            const size_t lo = k * PIECE, len = (n - lo < PIECE) ? n - lo : PIECE;
            part[k] = crc16_raw(p + lo, len, 0);
        }
    };
    pool().run((unsigned)(pieces - 1 < Pool::capacity() ? pieces - 1 : Pool::capacity()), work);
    for (size_t k = 0; k < pieces; k++) {
        const size_t lo = k * PIECE, len = (n - lo < PIECE) ? n - lo : PIECE;
        s = (uint16_t)(crc16_shift(s, len) ^ part[k]);
    }
    return (uint16_t)~s;
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
        // Kraft sum in units of 2^-max_: above 2^max_ the length set is over-subscribed and entries collide
        uint64_t kraft = 0;
        for (int bits = 1; bits <= 16; bits++) kraft += (uint64_t)count[bits] << (max_ >= (unsigned)bits ? max_ - bits : 0);
        auto rev_code = [](uint32_t code, unsigned size) {  // the low `size` bits of code, reversed
            uint32_t v = code;
            v = ((v >> 1) & 0x5555u) | ((v & 0x5555u) << 1);
            v = ((v >> 2) & 0x3333u) | ((v & 0x3333u) << 2);
            v = ((v >> 4) & 0x0F0Fu) | ((v & 0x0F0Fu) << 4);
            v = ((v >> 8) & 0x00FFu) | ((v & 0x00FFu) << 8);
            return (v & 0xFFFFu) >> (16 - size);
        };
        if (kraft > ((uint64_t)1 << max_)) {
            // over-subscribed (a damaged file): the reference's symbol-order fill, later symbols overwriting earlier ones
            for (size_t sym = 0; sym < sizes.size(); sym++) {
                const unsigned size = sizes[sym];
                if (!size) continue;
                const uint32_t code = next[size]++;
                uint32_t rev = 0;
                for (unsigned k = 0; k < size; k++) rev |= ((code >> k) & 1u) << (size - 1 - k);
                const uint32_t entry = ((uint32_t)sym << 5) | size;
                for (uint32_t id = rev; id < ((uint32_t)1 << max_); id += (1u << size)) table_[id] = entry;
            }
        } else {
            // a prefix code: no two symbols share an entry, so the fill order is free.  Symbol by symbol the copies of a code lie
            // 2^size entries apart -- 32 768 cache-missing stores for a 15-bit selector table (0.1 ms: on the critical path of a
            // single-slice file).  Grouped by length and walked copy by copy instead, every pass stays inside one window of
            // 2^size entries.
            std::vector<uint32_t> ent(sizes.size()), revs(sizes.size());
            uint32_t first[18] = {0};
            for (int bits = 1; bits <= 16; bits++) first[bits + 1] = first[bits] + count[bits];
            uint32_t fill[18];
            memcpy(fill, first, sizeof(fill));
            for (size_t sym = 0; sym < sizes.size(); sym++) {
                const unsigned size = sizes[sym];
                if (!size) continue;
                const uint32_t code = next[size]++;
                const uint32_t at = fill[size]++;
                revs[at] = rev_code(code, size);
                ent[at] = ((uint32_t)sym << 5) | size;
            }
            for (unsigned size = 1; size <= max_; size++) {
                const uint32_t a = first[size], b = first[size + 1];
                if (a == b) continue;
                for (uint32_t hi = 0; hi < ((uint32_t)1 << (max_ - size)); hi++) {
                    uint32_t* const win = table_.data() + ((size_t)hi << size);
                    for (uint32_t k = a; k < b; k++) win[revs[k]] = ent[k];
                }
            }
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

    // the fast slice loop reads the table directly: entry = symbol << 5 | code_size (0 = no code), index = the next bits() bits
    const uint32_t* table() const { return table_.data(); }
    unsigned bits() const { return max_; }

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
    // Codebook sizes as the slice loop checks them (mod.rs:443-445).  Kept beside the vectors because bu_read_to decodes the
    // codebooks on another thread WHILE the slices' symbol streams are decoded: the slice loop needs the sizes and the four
    // Huffman tables (init_tables), never the codebook entries.
    uint32_t n_endpoints = 0, n_selectors = 0;

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

    // mod.rs:77-83: the four tables of the slice loop and the history size
    bu_status init_tables(size_t n_ep, size_t n_sel, const uint8_t* tables, size_t tables_len, bool video)
    {
        is_video = video;
        n_endpoints = (uint32_t)n_ep;
        n_selectors = (uint32_t)n_sel;
        BitReader r(tables, tables_len);
        bu_status st;
        if ((st = read_huffman_table(r, endpoint_pred))) return st;
        if ((st = read_huffman_table(r, delta_endpoint))) return st;
        if ((st = read_huffman_table(r, selector))) return st;
        if ((st = read_huffman_table(r, history_rle))) return st;
        history_size = r.read(13);
        return BU_OK;
    }
    // mod.rs:69-76: both codebooks (errors of the endpoint codebook come first, as in the reference)
    bu_status init_codebooks(const uint8_t* ep, size_t ep_len, const uint8_t* sel, size_t sel_len)
    {
        bu_status st = decode_endpoints(n_endpoints, ep, ep_len);
        if (st) return st;
        return decode_selectors(n_selectors, sel, sel_len);
    }
    // mod.rs:64-95 in the reference's order: codebooks, then tables
    bu_status init(size_t n_ep, size_t n_sel, const uint8_t* ep, size_t ep_len, const uint8_t* sel, size_t sel_len,
                   const uint8_t* tables, size_t tables_len, bool video)
    {
        is_video = video;
        n_endpoints = (uint32_t)n_ep;
        n_selectors = (uint32_t)n_sel;
        bu_status st = init_codebooks(ep, ep_len, sel, sel_len);
        if (st) return st;
        return init_tables(n_ep, n_sel, tables, tables_len, video);
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
    // A slice's stream is serial, so this loop IS the end-to-end time of a single-slice ETC1S file (BASELINE config 4: 1.86 of
    // 2.6 ms).  decode_slice_fast is the same state machine written for the host core's pipeline -- one 8-byte window refill per
    // block, the delta-endpoint code looked up speculatively and consumed by a conditional move (the 4-way predictor switch
    // mispredicts three times in four on real streams), edge violations and missing codes folded into one sticky flag -- and
    // decides nothing about errors: whenever the flag is set the exact loop below (decode_slice_exact, the round-1 code, status
    // for status the oracle's) decodes the slice again from its first bit.
    // rows_done (optional): the number of complete block rows in idx[], published row by row (release) for a consumer that
    // works on finished rows while the rest is decoded; only rows free of any irregularity are ever published, and those are
    // final.  abort (optional): checked between rows; a set flag ends the call with BU_ERR_ARGUMENT (the caller discards it).
    bu_status decode_slice(size_t nbx, size_t nby, const uint8_t* data, size_t len, uint32_t* idx, std::atomic<uint32_t>* rows_done = nullptr,
                           const std::atomic<bool>* abort = nullptr) const
    {
        if (nbx && nby && decode_slice_fast(nbx, nby, data, len, idx, rows_done, abort)) return BU_OK;
        if (abort && abort->load(std::memory_order_relaxed)) return BU_ERR_ARGUMENT;
        return decode_slice_exact(nbx, nby, data, len, idx);
    }

    // true = the slice decoded without any irregularity and idx[] is complete; false = anything else (idx[] partly written)
    bool decode_slice_fast(size_t nbx, size_t nby, const uint8_t* data, size_t len, uint32_t* idx, std::atomic<uint32_t>* rows_done = nullptr,
                           const std::atomic<bool>* abort = nullptr) const
    {
        const uint32_t n_ep = n_endpoints, n_sel = n_selectors;
        const uint32_t hs = history_size;
        if (n_ep == 0 || n_sel == 0 || n_ep > 65536u || n_sel > 65536u) return false;
        if (!endpoint_pred.bits() || !delta_endpoint.bits() || !selector.bits() || !history_rle.bits()) return false;
        const uint32_t *tp = endpoint_pred.table(), *td = delta_endpoint.table(), *ts = selector.table(), *tr = history_rle.table();
        const uint32_t mp = (1u << endpoint_pred.bits()) - 1u, md = (1u << delta_endpoint.bits()) - 1u, msel = (1u << selector.bits()) - 1u,
                       mr = (1u << history_rle.bits()) - 1u;
        // bit window: `have` valid bits in `acc`; refilled to >= 56 once per block (a block consumes at most 16 + 16 + 16 bits outside
        // the rare escape paths, which refill for themselves); bytes past the end read as zeros (bitreader.rs:45,55)
        uint64_t acc = 0;
        unsigned have = 0;
        size_t pos = 0;
        auto refill = [&]() __attribute__((always_inline)) {
            if (pos + 8 <= len) {
                uint64_t w;
                memcpy(&w, data + pos, 8);
                acc |= w << have;
                pos += (63u - have) >> 3;
                have |= 56u;
            } else {
                while (have <= 56) {
                    const uint64_t byte = pos < len ? data[pos] : 0;
                    pos++;
                    acc |= byte << have;
                    have += 8;
                }
            }
        };
        auto take = [&](unsigned n) {
            const uint32_t v = (uint32_t)(acc & ((1ull << n) - 1ull));
            acc >>= n;
            have -= n;
            return v;
        };
        auto vlc_fast = [&](unsigned chunk_bits, uint32_t* out) {
            uint32_t v = 0;
            for (unsigned ofs = 0;; ofs += chunk_bits) {
                if (ofs >= 32) return false;
                if (have < 16) refill();
                const uint32_t sy = take(chunk_bits + 1);
                v |= (sy & ((1u << chunk_bits) - 1u)) << ofs;
                if (!(sy >> chunk_bits)) break;
            }
            *out = v;
            return true;
        };
        std::vector<uint16_t> above_v(nbx + 1, 0), cur_v(nbx + 1, 0);  // [0] = the column left of the slice (never a valid source)
        std::vector<uint8_t> saved_bits((nbx + 1) / 2, 0);                // per 2-block group: the predictors of the odd row below
        std::vector<uint16_t> hist(hs ? hs : 1, 0);
        uint16_t *above = above_v.data() + 1, *cur_row = cur_v.data() + 1;
        uint32_t rover = hs / 2;
        const uint32_t rle_sym = (n_sel + hs) & 0xFFFFu;
        uint32_t sel_rle = 0, pred_repeat = 0, prev_pred_sym = 0, prev_ep = 0, bad = 0;
        const bool video = is_video;
        uint32_t* out_row = idx;
        // one block: predictor `pred`; edge = bit p set where predictor p has no source (checked only on the slice's first row and
        // column: EDGE = false compiles the test away for the interior).  Returns false where the fast path gives up.
        auto block = [&](size_t bx, uint32_t pred, uint32_t edge, auto edge_tag) __attribute__((always_inline)) -> bool {
            if constexpr (decltype(edge_tag)::value) bad |= (edge >> pred) & 1u;
            // the delta-endpoint code at the window's head is looked up whether or not this block uses it, and consumed by a
            // conditional move: the predictor is a coin toss to the branch predictor
            const uint32_t ed = td[(uint32_t)acc & md];
            const bool is3 = pred == 3;
            const uint32_t dl = is3 ? (ed & 31u) : 0u;
            acc >>= dl;
            have -= dl;
            uint32_t e3 = ((ed >> 5) + prev_ep) & 0xFFFFu;
            e3 = e3 >= n_ep ? (e3 - n_ep) & 0xFFFFu : e3;
            bad |= is3 & (((ed & 31u) == 0u) | (e3 >= n_ep));
            const uint32_t e01 = (pred & 1) ? above[bx] : prev_ep;
            const uint32_t e2 = video ? 0u : above[(ptrdiff_t)bx - 1];
            const uint32_t e = is3 ? e3 : ((pred & 2) ? e2 : e01);  // (copies of earlier, checked indices: < n_ep)
            cur_row[bx] = (uint16_t)e;
            prev_ep = e;
            if (video && pred == 2) {  // previous frame's selector: zero (mod.rs:236-237, 428-431)
                out_row[bx] = e;
                return true;
            }
            if (sel_rle) {  // inside a run of history entry 0 (mod.rs:380-396): use_index(0) swaps the entry with itself
                sel_rle--;
                out_row[bx] = e | ((uint32_t)hist[0] << 16);
                return true;
            }
            const uint32_t es = ts[(uint32_t)acc & msel];
            bad |= (es & 31u) == 0u;
            acc >>= (es & 31u);
            have -= (es & 31u);
            uint32_t sym = es >> 5, sel;
            if (sym == rle_sym) {
                if (hs == 0) return false;
                if (have < 16) refill();
                const uint32_t er = tr[(uint32_t)acc & mr];
                if ((er & 31u) == 0u) return false;
                acc >>= (er & 31u);
                have -= (er & 31u);
                uint32_t run = er >> 5;
                if (run == 63) {
                    uint32_t v;
                    if (!vlc_fast(7, &v)) return false;
                    run = v;
                }
                sel_rle = 3 + run - 1;
                sym = n_sel;
            }
            if (sym >= n_sel) {
                const uint32_t hi = sym - n_sel;
                if (hi >= hs) return false;
                sel = hist[hi];
                const uint16_t t = hist[hi / 2];  // (hi == 0: swaps entry 0 with itself)
                hist[hi / 2] = (uint16_t)sel;
                hist[hi] = t;
            } else {
                if (hs) {
                    hist[rover] = (uint16_t)sym;
                    rover = rover + 1 == hs ? hs / 2 : rover + 1;
                }
                sel = sym;
            }
            out_row[bx] = e | (sel << 16);  // (sel: a symbol below n_sel or a history entry, i.e. an earlier such symbol or the initial 0)
            return true;
        };
        // the 8 predictor bits of the 2 x 2 group at an even row: a symbol, or the previous symbol again inside a repeat run
        auto group_bits = [&](uint32_t* bits) __attribute__((always_inline)) -> bool {
            if (pred_repeat) {
                pred_repeat--;
                *bits = prev_pred_sym;
                return true;
            }
            const uint32_t e0 = tp[(uint32_t)acc & mp];
            bad |= (e0 & 31u) == 0u;
            acc >>= (e0 & 31u);
            have -= (e0 & 31u);
            const uint32_t sy = e0 >> 5;
            if (sy == 256) {
                uint32_t v;
                if (!vlc_fast(4, &v)) return false;
                pred_repeat = v + 2;
                *bits = prev_pred_sym;
                refill();
            } else {
                *bits = sy & 0xFF;
                prev_pred_sym = sy & 0xFF;
            }
            return true;
        };
        using Edge = std::integral_constant<bool, true>;
        using Inner = std::integral_constant<bool, false>;
        for (size_t by = 0; by < nby; by++) {
            // predictor p is invalid where bit p of the edge mask is set: 0 (left) in column 0, 1 (above) in row 0, 2 (above-left) in
            // both -- except in texture video, where 2 means "previous frame" (mod.rs:301-355)
            const uint32_t row_edge = by == 0 ? (video ? 2u : 6u) : 0u, col_edge = video ? 1u : 5u;
            const bool even = !(by & 1);
            out_row = idx + by * nbx;
            for (size_t bx = 0; bx < nbx; bx += 2) {
                refill();
                uint32_t bits;
                if (even) {
                    if (!group_bits(&bits)) return false;
                    saved_bits[bx >> 1] = (uint8_t)(bits >> 4);
                } else {
                    bits = saved_bits[bx >> 1];
                }
                if (bx == 0 || by == 0) {
                    if (!block(bx, bits & 3, row_edge | (bx == 0 ? col_edge : 0u), Edge())) return false;
                    if (bx + 1 < nbx) {
                        refill();
                        if (!block(bx + 1, (bits >> 2) & 3, row_edge, Edge())) return false;
                    }
                } else {
                    if (!block(bx, bits & 3, 0u, Inner())) return false;
                    if (bx + 1 < nbx) {
                        refill();
                        if (!block(bx + 1, (bits >> 2) & 3, 0u, Inner())) return false;
                    }
                }
            }
            if (bad) return false;
            std::swap(above, cur_row);
            if (rows_done) {
                rows_done->store((uint32_t)(by + 1), std::memory_order_release);
                if (abort && abort->load(std::memory_order_relaxed)) return false;
            }
        }
        return true;
    }

    // ---- the slice loop on TWO threads (round 4) -----------------------------------------------------------------------------------
    // The loop above is bound by instruction throughput on one host core (~21 clocks per block), and half of its instructions do
    // not touch the bit stream: endpoint prediction, the selector history, the index store.  slice_lex is the bit-serial half -- it
    // walks the stream with exactly the state that decides WHICH table reads next (predictor bits, predictor-repeat and selector-run
    // counters) and leaves one token pair per block: tok_ep = predictor | delta symbol << 2, tok_sel = selector symbol (a run block:
    // n_selectors, i.e. history entry 0; a texture-video block that keeps the previous frame's selector: SPLIT_SKIP).  slice_resolve,
    // on another thread, turns tokens into indices one row behind.  Rows are handed over through `rows_lexed` (release / acquire);
    // the token arrays cover the whole slice, so the lexer never waits for the resolver.  Like decode_slice_fast both halves only
    // handle regular streams: any irregularity makes them give up (`failed`), and the caller decodes the slice again with the exact loop.
    static constexpr uint16_t SPLIT_SKIP = 0xFFFF;
    bool split_ok() const
    {
        return n_endpoints != 0 && n_selectors != 0 && n_endpoints <= 65536u && (uint64_t)n_selectors + history_size < 0xFFFFu && endpoint_pred.bits() &&
               delta_endpoint.bits() && selector.bits() && history_rle.bits();
    }
    bool slice_lex(size_t nbx, size_t nby, const uint8_t* data, size_t len, uint32_t* tok_ep, uint16_t* tok_sel, std::atomic<uint32_t>& rows_lexed,
                   std::atomic<bool>& failed) const
    {
        const uint32_t n_sel = n_selectors, hs = history_size;
        const uint32_t *tp = endpoint_pred.table(), *td = delta_endpoint.table(), *ts = selector.table(), *tr = history_rle.table();
        const uint32_t mp = (1u << endpoint_pred.bits()) - 1u, md = (1u << delta_endpoint.bits()) - 1u, msel = (1u << selector.bits()) - 1u,
                       mr = (1u << history_rle.bits()) - 1u;
        uint64_t acc = 0;
        unsigned have = 0;
        size_t pos = 0;
        auto refill = [&]() __attribute__((always_inline)) {
            if (pos + 8 <= len) {
                uint64_t w;
                memcpy(&w, data + pos, 8);
                acc |= w << have;
                pos += (63u - have) >> 3;
                have |= 56u;
            } else {
                while (have <= 56) {
                    const uint64_t byte = pos < len ? data[pos] : 0;
                    pos++;
                    acc |= byte << have;
                    have += 8;
                }
            }
        };
        auto vlc_fast = [&](unsigned chunk_bits, uint32_t* out) {
            uint32_t v = 0;
            for (unsigned ofs = 0;; ofs += chunk_bits) {
                if (ofs >= 32) return false;
                if (have < 16) refill();
                const uint32_t sy = (uint32_t)(acc & ((1ull << (chunk_bits + 1)) - 1ull));
                acc >>= chunk_bits + 1;
                have -= chunk_bits + 1;
                v |= (sy & ((1u << chunk_bits) - 1u)) << ofs;
                if (!(sy >> chunk_bits)) break;
            }
            *out = v;
            return true;
        };
        std::vector<uint8_t> saved_bits((nbx + 1) / 2, 0);
        const uint32_t rle_sym = (n_sel + hs) & 0xFFFFu;
        uint32_t sel_rle = 0, pred_repeat = 0, prev_pred_sym = 0, bad = 0;
        const bool video = is_video;
        auto block = [&](size_t i, uint32_t pred) __attribute__((always_inline)) -> bool {
            const uint32_t ed = td[(uint32_t)acc & md];
            const bool is3 = pred == 3;
            const uint32_t dl = is3 ? (ed & 31u) : 0u;
            acc >>= dl;
            have -= dl;
            bad |= is3 & ((ed & 31u) == 0u);
            tok_ep[i] = pred | ((ed >> 5) << 2);
            if (video && pred == 2) {
                tok_sel[i] = SPLIT_SKIP;
                return true;
            }
            if (sel_rle) {
                sel_rle--;
                tok_sel[i] = (uint16_t)n_sel;
                return true;
            }
            const uint32_t es = ts[(uint32_t)acc & msel];
            bad |= (es & 31u) == 0u;
            acc >>= (es & 31u);
            have -= (es & 31u);
            uint32_t sym = es >> 5;
            if (sym == rle_sym) {
                if (hs == 0) return false;
                if (have < 16) refill();
                const uint32_t er = tr[(uint32_t)acc & mr];
                if ((er & 31u) == 0u) return false;
                acc >>= (er & 31u);
                have -= (er & 31u);
                uint32_t run = er >> 5;
                if (run == 63) {
                    uint32_t v;
                    if (!vlc_fast(7, &v)) return false;
                    run = v;
                }
                sel_rle = 3 + run - 1;
                sym = n_sel;
            }
            tok_sel[i] = (uint16_t)sym;
            return true;
        };
        for (size_t by = 0; by < nby; by++) {
            const bool even = !(by & 1);
            const size_t row = by * nbx;
            for (size_t bx = 0; bx < nbx; bx += 2) {
                refill();
                uint32_t bits;
                if (even) {
                    if (pred_repeat) {
                        pred_repeat--;
                        bits = prev_pred_sym;
                    } else {
                        const uint32_t e0 = tp[(uint32_t)acc & mp];
                        bad |= (e0 & 31u) == 0u;
                        acc >>= (e0 & 31u);
                        have -= (e0 & 31u);
                        const uint32_t sy = e0 >> 5;
                        if (sy == 256) {
                            uint32_t v;
                            if (!vlc_fast(4, &v)) {
                                failed.store(true, std::memory_order_release);
                                return false;
                            }
                            pred_repeat = v + 2;
                            bits = prev_pred_sym;
                            refill();
                        } else {
                            bits = sy & 0xFF;
                            prev_pred_sym = sy & 0xFF;
                        }
                    }
                    saved_bits[bx >> 1] = (uint8_t)(bits >> 4);
                } else {
                    bits = saved_bits[bx >> 1];
                }
                bool ok = block(row + bx, bits & 3);
                if (ok && bx + 1 < nbx) {
                    refill();
                    ok = block(row + bx + 1, (bits >> 2) & 3);
                }
                if (!ok) {
                    failed.store(true, std::memory_order_release);
                    return false;
                }
            }
            if (bad || failed.load(std::memory_order_relaxed)) {
                failed.store(true, std::memory_order_release);
                return false;
            }
            rows_lexed.store((uint32_t)(by + 1), std::memory_order_release);
        }
        return true;
    }
    bool slice_resolve(size_t nbx, size_t nby, const uint32_t* tok_ep, const uint16_t* tok_sel, uint32_t* idx, const std::atomic<uint32_t>& rows_lexed,
                       std::atomic<bool>& failed, std::atomic<uint32_t>* rows_done, const std::atomic<bool>* abort) const
    {
        const uint32_t n_ep = n_endpoints, n_sel = n_selectors, hs = history_size;
        std::vector<uint16_t> above_v(nbx + 1, 0), cur_v(nbx + 1, 0);  // [0] = the column left of the slice (never a valid source)
        std::vector<uint16_t> hist(hs ? hs : 1, 0);
        uint16_t *above = above_v.data() + 1, *cur_row = cur_v.data() + 1;
        uint32_t rover = hs / 2, prev_ep = 0, bad = 0;
        const bool video = is_video;
        for (size_t by = 0; by < nby; by++) {
            for (unsigned spin = 0; rows_lexed.load(std::memory_order_acquire) <= by; spin++) {
                if (failed.load(std::memory_order_acquire) || (abort && abort->load(std::memory_order_relaxed))) return false;
                if (spin > 256) std::this_thread::yield();
            }
            const uint32_t row_edge = by == 0 ? (video ? 2u : 6u) : 0u, col_edge = video ? 1u : 5u;
            const uint32_t* te = tok_ep + by * nbx;
            const uint16_t* tsl = tok_sel + by * nbx;
            uint32_t* out_row = idx + by * nbx;
            for (size_t bx = 0; bx < nbx; bx++) {
                const uint32_t t = te[bx], pred = t & 3u;
                if (bx == 0 || by == 0) bad |= ((row_edge | (bx == 0 ? col_edge : 0u)) >> pred) & 1u;
                const bool is3 = pred == 3;
                uint32_t e3 = ((t >> 2) + prev_ep) & 0xFFFFu;
                e3 = e3 >= n_ep ? (e3 - n_ep) & 0xFFFFu : e3;
                bad |= is3 & (e3 >= n_ep);
                const uint32_t e01 = (pred & 1) ? above[bx] : prev_ep;
                const uint32_t e2 = video ? 0u : above[(ptrdiff_t)bx - 1];
                const uint32_t e = is3 ? e3 : ((pred & 2) ? e2 : e01);
                cur_row[bx] = (uint16_t)e;
                prev_ep = e;
                const uint32_t sym = tsl[bx];
                uint32_t sel;
                if (sym == SPLIT_SKIP) {
                    sel = 0;
                } else if (sym >= n_sel) {
                    const uint32_t hi = sym - n_sel;
                    if (hi >= hs) {
                        failed.store(true, std::memory_order_release);
                        return false;
                    }
                    sel = hist[hi];
                    const uint16_t x = hist[hi / 2];  // (hi == 0: swaps entry 0 with itself)
                    hist[hi / 2] = (uint16_t)sel;
                    hist[hi] = x;
                } else {
                    if (hs) {
                        hist[rover] = (uint16_t)sym;
                        rover = rover + 1 == hs ? hs / 2 : rover + 1;
                    }
                    sel = sym;
                }
                out_row[bx] = e | (sel << 16);
            }
            if (bad) {
                failed.store(true, std::memory_order_release);
                return false;
            }
            std::swap(above, cur_row);
            if (rows_done) rows_done->store((uint32_t)(by + 1), std::memory_order_release);
        }
        return true;
    }

    bu_status decode_slice_exact(size_t nbx, size_t nby, const uint8_t* data, size_t len, uint32_t* idx) const
    {
        BitReader r(data, len);
        const uint32_t n_ep = n_endpoints, n_sel = n_selectors;
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

// ---- all slices of a file ----
// The reference decodes slice after slice (basis.rs:42-58, 103-123).  The symbol stream is serial WITHIN a slice, but
// slices share nothing except the read-only codebooks and Huffman tables (the "previous frame" of texture video is
// re-zeroed per slice, mod.rs:236-237), so they decode concurrently on the host cores.  The visible result is that of
// the sequential loop: the status of the first failing slice in file order.
struct SliceJob {
    size_t nbx, nby;
    const uint8_t* data;
    size_t len;
    uint32_t* idx;
    bu_status st;
};

inline bu_status decode_slices(const BasisLz& lz, std::vector<SliceJob>& jobs, unsigned max_threads = 0, size_t min_parallel_blocks = 16384)
{
    size_t total = 0;
    for (const SliceJob& j : jobs) total += j.nbx * j.nby;
    unsigned nt = max_threads ? max_threads : Pool::capacity() + 1;
    if (nt > jobs.size()) nt = (unsigned)jobs.size();
    if (nt <= 1 || total < min_parallel_blocks) {
        for (SliceJob& j : jobs) {
            j.st = lz.decode_slice(j.nbx, j.nby, j.data, j.len, j.idx);
            if (j.st) return j.st;
        }
        return BU_OK;
    }
    std::atomic<size_t> next{0};
    const std::function<void()> work = [&] {
        for (size_t k; (k = next.fetch_add(1)) < jobs.size();) {
            try {  // (a copy of this runs on pool threads: an exception must become a status there)
                jobs[k].st = lz.decode_slice(jobs[k].nbx, jobs[k].nby, jobs[k].data, jobs[k].len, jobs[k].idx);
            } catch (...) {
                jobs[k].st = BU_ERR_BOUNDS;
            }
        }
    };
    pool().run(nt - 1, work);
    for (const SliceJob& j : jobs)
        if (j.st) return j.st;
    return BU_OK;
}

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
// check_data_crc = false: the caller verifies the payload CRC itself (bu_read_to overlaps it with the upload) and gives
// a CRC failure precedence over any later error, as the reference's order of checks would
inline bu_status bu_plan_file(bu_read_target target, const uint8_t* file, size_t len, BuFilePlan& p, bool check_data_crc = true)
{
    if (!file) return BU_ERR_ARGUMENT;
    if ((int)target < 0 || (int)target > 5) return BU_ERR_ARGUMENT;
    bu_status st = read_header(file, len, &p.h);
    if (st) return st;
    if (check_data_crc && crc16(file + 77, len - 77, 0) != p.h.data_crc16) return BU_ERR_DATA_CRC;  // to EOF, basis.rs:338-341
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
