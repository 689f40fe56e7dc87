// TEST-ONLY: the product's host-side container parser + BasisLZ decoder (csrc/bu_basis.hpp, the exact code that ships
// inside libbasisu_hip.so) compiled with AddressSanitizer + UBSan, plus a corrupt-file fuzz loop.  GPU ASan is not
// available on the pool, and this code is what faces untrusted `.basis` bytes, so it gets the sanitizer treatment.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>

#include <thread>
#include <vector>

#include "bu_basis.hpp"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}

// plan + (for ETC1S) full host decode of every slice, as bu_read_to does before any upload
static int process(const uint8_t* f, size_t len, int target)
{
    bu_host::BuFilePlan p;
    bu_status st = bu_host::bu_plan_file((bu_read_target)target, f, len, p);
    if (st) return st;
    if (!p.etc1s) return 0;
    bu_host::BasisLz lz;
    st = bu_host::bu_make_lz(f, len, p.h, lz);
    if (st) return st;
    // sequential, as the reference does it ...
    std::vector<std::vector<uint32_t>> seq(p.slices.size()), par(p.slices.size());
    bu_status seq_st = BU_OK;
    size_t seq_decoded = 0;  // slices the sequential loop got through without an error
    for (size_t k = 0; k < p.slices.size() && !seq_st; k++) {
        const bu_slice_desc& s = p.slices[k];
        seq[k].assign((size_t)s.num_blocks_x * s.num_blocks_y + 1, 0);
        seq_st = lz.decode_slice(s.num_blocks_x, s.num_blocks_y, f + s.file_ofs, s.file_size, seq[k].data());
        if (!seq_st) seq_decoded = k + 1;
        // decode_slice is the fast loop with the exact loop (round 1's, status for status the oracle's) as its fallback: compare the two
        // DIRECTLY on every slice -- the exact loop's status is the verdict, and wherever the fast loop claims a regular stream its indices
        // are the exact loop's; rows it published before giving up are final and must match too
        {
            const size_t n = (size_t)s.num_blocks_x * s.num_blocks_y;
            std::vector<uint32_t> ex(n + 1, 0), fa(n + 1, 0);
            const bu_status ex_st = lz.decode_slice_exact(s.num_blocks_x, s.num_blocks_y, f + s.file_ofs, s.file_size, ex.data());
            if (ex_st != seq_st) {
                fprintf(stderr, "decode_slice status %d, decode_slice_exact %d in slice %zu\n", (int)seq_st, (int)ex_st, k);
                abort();
            }
            if (!ex_st && ex != seq[k]) {
                fprintf(stderr, "decode_slice differs from decode_slice_exact in slice %zu\n", k);
                abort();
            }
            if (s.num_blocks_x && s.num_blocks_y) {
                std::atomic<uint32_t> rows{0};
                const bool fast_ok = lz.decode_slice_fast(s.num_blocks_x, s.num_blocks_y, f + s.file_ofs, s.file_size, fa.data(), &rows);
                if (fast_ok && (ex_st != BU_OK || fa != ex)) {
                    fprintf(stderr, "decode_slice_fast accepts slice %zu and disagrees with the exact loop (status %d)\n", k, (int)ex_st);
                    abort();
                }
                if (!ex_st) {  // rows published before a give-up (or all of them) are the exact loop's
                    const size_t pub = (size_t)rows.load() * s.num_blocks_x;
                    if (pub > n || !std::equal(fa.begin(), fa.begin() + (long)pub, ex.begin())) {
                        fprintf(stderr, "rows published by decode_slice_fast differ from the exact loop in slice %zu\n", k);
                        abort();
                    }
                }
            }
        }
    }
    // ... and on 4 threads, as bu_read_to does it (threading forced: these files are tiny): same status, same indices
    std::vector<bu_host::SliceJob> jobs;
    for (size_t k = 0; k < p.slices.size(); k++) {
        const bu_slice_desc& s = p.slices[k];
        par[k].assign((size_t)s.num_blocks_x * s.num_blocks_y + 1, 0);
        jobs.push_back({s.num_blocks_x, s.num_blocks_y, f + s.file_ofs, s.file_size, par[k].data(), BU_OK});
    }
    // ... and every slice on two threads (slice_lex + slice_resolve, the streamed front door's form for the first slice): where both
    // halves report a regular stream the indices are the sequential loop's; they may only give up where that loop is the judge
    if (lz.split_ok()) {
        for (size_t k = 0; k < p.slices.size(); k++) {
            const bu_slice_desc& s = p.slices[k];
            const size_t n = (size_t)s.num_blocks_x * s.num_blocks_y;
            std::vector<uint32_t> tok_ep(n + 1), out(n + 1, 0);
            std::vector<uint16_t> tok_sel(n + 1);
            std::atomic<uint32_t> rows_lexed{0}, rows_done{0};
            std::atomic<bool> failed{false};
            bool res_ok = false;
            std::thread th([&] { res_ok = lz.slice_resolve(s.num_blocks_x, s.num_blocks_y, tok_ep.data(), tok_sel.data(), out.data(), rows_lexed, failed, &rows_done, nullptr); });
            const bool lex_ok = lz.slice_lex(s.num_blocks_x, s.num_blocks_y, f + s.file_ofs, s.file_size, tok_ep.data(), tok_sel.data(), rows_lexed, failed);
            th.join();
            if (lex_ok && res_ok) {
                // (slices behind the first failing one were never decoded by the sequential loop: nothing to compare with)
                const bool have_seq = seq_decoded > k;
                if (have_seq && out != seq[k]) {
                    fprintf(stderr, "two-thread decode differs in slice %zu\n", k);
                    abort();
                }
            } else if (seq_st == BU_OK && s.num_blocks_x && s.num_blocks_y) {
                fprintf(stderr, "two-thread decode gave up on slice %zu of a file the sequential loop accepts\n", k);
                abort();
            }
        }
    }
    const bu_status par_st = bu_host::decode_slices(lz, jobs, 4, 0);
    if (par_st != seq_st) {
        fprintf(stderr, "parallel decode status %d != sequential %d\n", (int)par_st, (int)seq_st);
        abort();
    }
    if (!seq_st)
        for (size_t k = 0; k < p.slices.size(); k++)
            if (seq[k] != par[k]) {
                fprintf(stderr, "parallel decode differs in slice %zu\n", k);
                abort();
            }
    return seq_st;
}

// The worker pool under the pattern that once deadlocked it: many parked threads (a wide job first), then thousands of narrow jobs
// whose copies finish at once -- a copy that parked again used to swallow the wake-up meant for the next helper, and the job
// waited for a copy that never started.  Also the split begin() / end() form with the caller doing something else meanwhile.
static int pool_stress(int rounds)
{
    std::atomic<long> sum{0};
    const std::function<void()> wide = [&] { sum.fetch_add(1); };
    bu_host::pool().run(bu_host::Pool::capacity(), wide);
    for (int r = 0; r < rounds; r++) {
        std::atomic<int> next{0}, done{0};
        const std::function<void()> narrow = [&] {
            for (int k; (k = next.fetch_add(1)) < 5;) done.fetch_add(1);
        };
        if (r & 1) {
            bu_host::pool().run(3, narrow);
        } else {
            bu_host::pool().begin(3, narrow);
            narrow();
            bu_host::pool().end();
        }
        if (done.load() != 5) return 1;
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 3 && !strcmp(argv[1], "--pool-stress")) {
        const int st = pool_stress(atoi(argv[2]));
        printf("pool stress %s\n", st ? "FAILED" : "ok");
        return st;
    }
    if (argc < 3) return 2;
    const int iters = atoi(argv[2]);
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    std::vector<uint8_t> base;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) base.insert(base.end(), buf, buf + n);
    fclose(fp);
    int ok = 0, bad = 0;
    if (process(base.data(), base.size(), BU_READ_RGBA) != 0) return 3;  // the pristine file must parse
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> g = base;
        const int kind = rnd() % 4;
        const int flips = 1 + rnd() % 4;
        for (int k = 0; k < flips; k++) {
            size_t pos = kind == 0 ? rnd() % 77 : (kind == 1 ? 77 + rnd() % (g.size() > 200 ? 123 : 1) : rnd() % g.size());
            if (pos >= g.size()) pos = g.size() - 1;
            g[pos] ^= (uint8_t)(1u << (rnd() % 8));
        }
        if (kind == 3 && g.size() > 100) g.resize(77 + rnd() % (g.size() - 77));  // truncation
        // re-seal both CRCs so the mutation reaches the parsers behind them
        if (g.size() >= 77) {
            const uint16_t dc = bu_host::crc16(g.data() + 77, g.size() - 77, 0);
            g[12] = (uint8_t)dc;
            g[13] = (uint8_t)(dc >> 8);
            const uint16_t hc = bu_host::crc16(g.data() + 8, 69, 0);
            g[6] = (uint8_t)hc;
            g[7] = (uint8_t)(hc >> 8);
        }
        // heap copy of exact size so ASan sees any over-read of the file buffer
        uint8_t* exact = (uint8_t*)malloc(g.size() ? g.size() : 1);
        memcpy(exact, g.data(), g.size());
        const int st = process(exact, g.size(), (it & 1) ? BU_READ_RGBA : BU_READ_ETC1);
        free(exact);
        st ? bad++ : ok++;
    }
    printf("fuzz done: %d parsed, %d rejected\n", ok, bad);
    return 0;
}
