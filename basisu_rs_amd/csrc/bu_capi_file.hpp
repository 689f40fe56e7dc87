// C ABI, whole-file level: the crate's public read_to_* API (src/lib.rs:20-22, src/basis.rs) -- container parse, CRC and BasisLZ
// decode on the host (bu_basis.hpp), block work through the kernels; the UASTC file writer.
// Part of the single translation unit bu_hip.hip.
#pragma once
extern "C" {

// ---- whole-file level (basis.rs) --------------------------------------------------------------------
bu_status bu_basis_read_header(const uint8_t* file, size_t len, bu_basis_header* out)
{
    if (!file || !out) return BU_ERR_ARGUMENT;
    return bu_host::read_header(file, len, out);
}

static bu_status bu_basis_read_slice_descs_impl(const uint8_t* file, size_t len, const bu_basis_header* header, bu_slice_desc* out, size_t max_descs,
                                              size_t* n_descs)
{
    if (!file || !header) return BU_ERR_ARGUMENT;
    std::vector<bu_slice_desc> v;
    bu_status st = bu_host::read_slice_descs(file, len, header, v);
    if (st) return st;
    if (n_descs) *n_descs = v.size();
    if (out) {
        if (v.size() > max_descs) return BU_ERR_OUTPUT_SIZE;
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    }
    return BU_OK;
}

uint16_t bu_basis_crc16(const uint8_t* data, size_t len, uint16_t crc) { return bu_host::crc16(data, len, crc); }

using bu_host::BuFilePlan;
using bu_host::bu_plan_file;
using bu_host::bu_make_lz;

static bu_status bu_read_query_impl(bu_read_target target, const uint8_t* file, size_t len, size_t* n_images, size_t* out_bytes)
{
    BuFilePlan p;
    bu_status st = bu_plan_file(target, file, len, p);
    if (st) return st;
    if (n_images) *n_images = p.images.size();
    if (out_bytes) *out_bytes = p.out_bytes;
    return BU_OK;
}

static bu_status bu_basislz_decode_impl(const uint8_t* file, size_t len, uint32_t slice_index, uint32_t* endpoints_out, uint8_t* selectors_out,
                                       uint32_t* idx_out)
{
    if (!file) return BU_ERR_ARGUMENT;
    bu_basis_header h;
    bu_status st = bu_host::read_header(file, len, &h);
    if (st) return st;
    if (h.tex_format != 0) return BU_ERR_UNSUPPORTED;
    std::vector<bu_slice_desc> slices;
    st = bu_host::read_slice_descs(file, len, &h, slices);
    if (st) return st;
    bu_host::BasisLz lz;
    st = bu_make_lz(file, len, h, lz);
    if (st) return st;
    if (endpoints_out) memcpy(endpoints_out, lz.endpoints.data(), lz.endpoints.size() * 4);
    if (selectors_out) memcpy(selectors_out, lz.selectors.data(), lz.selectors.size());
    if (idx_out) {
        if (slice_index >= slices.size()) return BU_ERR_ARGUMENT;
        const bu_slice_desc& s = slices[slice_index];
        if (!bu_host::in_file(len, s.file_ofs, s.file_size)) return BU_ERR_BOUNDS;
        st = lz.decode_slice(s.num_blocks_x, s.num_blocks_y, file + s.file_ofs, s.file_size, idx_out);
    }
    return st;
}


// ---- streamed ETC1S front door -------------------------------------------------------------------------------------------------
// A slice's symbol stream is serial (one host core: 1.86 ms of BASELINE config 4's 2.6), and everything else used to queue up in
// front of and behind it: payload CRC 0.12 ms, codebooks 0.19 ms, upload + kernel + download 0.4 ms.  Here they run BESIDE it:
//   * the slice loop needs the four Huffman tables and the codebook SIZES, never the codebook entries: the tables are parsed first
//     (tens of microseconds), then pool threads decode the codebooks, run the payload CRC and decode the slices, all at once;
//   * the decoders write their indices into a page-locked buffer the kernels read directly over PCIe (no upload), and publish
//     finished block rows; the calling thread launches the whole-file kernel over each band of finished 64-block units, so the
//     band's results travel to the (page-locked) output while the next rows are decoded.
// Errors keep the reference's order: CRC (basis.rs:338-341) before codebooks (mod.rs:69-76) before tables (:77-83) before slices
// (first failing slice in file order); work already launched for a file that fails is drained and its output discarded.
constexpr size_t BU_ETC1S_STREAM_MIN_BLOCKS = 32768, BU_ETC1S_BAND_BLOCKS = 16384;

static bu_status bu_read_etc1s_streamed(bu_context* ctx, bu_read_target target, const uint8_t* file, size_t len, const BuFilePlan& p, uint8_t* out,
                                        bool crc_pending, uint16_t crc_want, bool trace)
{
    using bu_host::in_file;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t = now();
        fprintf(stderr, "[bu_read_to] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const bu_basis_header& h = p.h;
    if (!in_file(len, h.endpoint_cb_file_ofs, h.endpoint_cb_file_size) || !in_file(len, h.selector_cb_file_ofs, h.selector_cb_file_size) ||
        !in_file(len, h.tables_file_ofs, h.tables_file_size) || !in_file(len, h.extended_file_ofs, h.extended_file_size))
        return (crc_pending && bu_host::crc16(file + 77, len - 77, 0) != crc_want) ? BU_ERR_DATA_CRC : BU_ERR_BOUNDS;  // (pool not taken yet)
    bu_host::BasisLz lz;
    // total_selectors for both codebooks: basis.rs:289-291
    const bu_status st_tables = lz.init_tables(h.total_selectors, h.total_selectors, file + h.tables_file_ofs, h.tables_file_size, h.tex_type == 3);
    lap("tables");
    const size_t n_img = p.images.size();
    auto align_up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // layout of the index buffer and the kernel's slice table, as in the one-launch path
    std::vector<size_t> in_off(n_img, 0), ain_off(n_img, 0);
    size_t words = 0;
    for (size_t k = 0; k < n_img; k++) {
        const bu_slice_desc& s = p.slices[p.first_slice[k]];
        const size_t nblk = (size_t)s.num_blocks_x * s.num_blocks_y;
        in_off[k] = words;
        words += (nblk + 63) & ~(size_t)63;
        if (p.alpha_pairs) {
            ain_off[k] = words;
            words += (nblk + 63) & ~(size_t)63;
        }
    }
    std::vector<BuEtc1sSlice> descs;
    std::vector<uint32_t> unit0(n_img, 0), units_of(n_img, 0);
    uint32_t n_units = 0;
    for (size_t k = 0; k < n_img; k++) {
        const bu_slice_desc& sl = p.slices[p.first_slice[k]];
        const size_t nblk = (size_t)sl.num_blocks_x * sl.num_blocks_y;
        unit0[k] = n_units;
        if (p.images[k].size == 0 || nblk == 0) continue;
        BuEtc1sSlice d;
        d.unit0 = n_units;
        d.n_blocks = (uint32_t)nblk;
        d.nbx = sl.num_blocks_x;
        d.idx_ofs = (uint32_t)in_off[k];
        d.aidx_ofs = (p.alpha_pairs && target == BU_READ_RGBA) ? (uint32_t)ain_off[k] : 0xFFFFFFFFu;
        d.image = (uint32_t)k;
        d.out_ofs = p.images[k].offset;
        descs.push_back(d);
        units_of[k] = (uint32_t)((nblk + 63) / 64);
        n_units += units_of[k];
    }
    BuEtc1sSlice end = {};
    end.unit0 = n_units;
    descs.push_back(end);

    std::lock_guard<std::mutex> g(ctx->lock);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> status_words(n_img, 0);  // landing area of the status download: outlives the drain
    BuDrain drain(ctx);
    const size_t idx_bytes = (words ? words : 16) * 4;
    if (idx_bytes > ctx->h_idx_cap) {
        if (ctx->h_idx) BU_HIP(ctx, hipHostFree(ctx->h_idx));
        ctx->h_idx = nullptr;
        ctx->h_idx_cap = 0;
        const size_t cap = idx_bytes < ((size_t)4 << 20) ? ((size_t)4 << 20) : idx_bytes + idx_bytes / 4;
        BU_HIP(ctx, hipHostMalloc(&ctx->h_idx, cap, hipHostMallocDefault));
        ctx->h_idx_cap = cap;
    }
    uint32_t* const h_idx = static_cast<uint32_t*>(ctx->h_idx);
    void* d_idx_view = nullptr;
    if (!bu_device_view(h_idx, &d_idx_view)) return BU_ERR_HIP;
    void* zout = nullptr;
    const bool direct_out = bu_device_view(out, &zout);
    bu_status st;
    if (!direct_out && (st = bu_reserve(ctx, &ctx->d_out, &ctx->out_cap, p.out_bytes ? p.out_bytes : 16))) return st;
    const size_t n_cb = h.total_selectors;
    const size_t ep_bytes = align_up(n_cb * 4), sel_bytes = align_up(n_cb * 8), status_bytes = align_up(8 * n_img),
                 desc_bytes = align_up(descs.size() * sizeof(BuEtc1sSlice));
    if ((st = bu_reserve(ctx, &ctx->d_aux, &ctx->aux_cap, ep_bytes + sel_bytes + status_bytes + desc_bytes + 256))) return st;
    uint8_t* const aux = static_cast<uint8_t*>(ctx->d_aux);
    uint8_t* const d_out = direct_out ? static_cast<uint8_t*>(zout) : static_cast<uint8_t*>(ctx->d_out);
    uint64_t* const d_status = reinterpret_cast<uint64_t*>(aux + ep_bytes + sel_bytes);
    const BuEtc1sSlice* const d_descs = reinterpret_cast<const BuEtc1sSlice*>(aux + ep_bytes + sel_bytes + status_bytes);
    BU_HIP(ctx, hipMemsetAsync(d_status, 0xFF, 8 * n_img, ctx->stream));
    BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes + sel_bytes + status_bytes, descs.data(), descs.size() * sizeof(BuEtc1sSlice), hipMemcpyHostToDevice, ctx->stream));
    lap("reserve");

    // ---- the jobs: [codebooks] [payload CRC] [slices ...] pulled from one counter by the pool threads ----
    // A slice of BU_ETC1S_STREAM_MIN_BLOCKS blocks and more is decoded on TWO threads (bu_host::BasisLz::slice_lex / slice_resolve):
    // one walks the bit stream and leaves a token pair per block, the other turns tokens into indices one row behind (endpoint
    // prediction, selector history, the stores the GPU reads) -- the symbol loop of a slice is bound by one host core's instruction
    // throughput, and this takes a quarter off it (1.55 -> 1.2 ms end to end for BASELINE config 4).  Token buffers live in the
    // context.  Any irregularity ends both halves, and whichever of the two finishes second decodes the slice again with the ordinary
    // loops, which decide the status.
    struct Job {
        size_t nbx = 0, nby = 0;
        const uint8_t* data = nullptr;
        size_t len = 0;
        uint32_t* idx = nullptr;
        std::atomic<uint32_t> rows{0};
        std::atomic<int> done{0};
        bu_status st = BU_OK;
        // the two-thread form
        bool split = false;
        uint32_t* tok_ep = nullptr;
        uint16_t* tok_sel = nullptr;
        std::atomic<uint32_t> rows_lexed{0};
        std::atomic<bool> split_failed{false};
        std::atomic<int> halves_done{0};
        bool lex_ok = false, res_ok = false;
    };
    const size_t per_img = p.alpha_pairs ? 2 : 1;
    std::vector<Job> jobs(n_img * per_img);
    const bool may_split = st_tables == BU_OK && lz.split_ok() && !getenv("BU_ETC1S_ONE_THREAD");
    size_t tok_bytes = 0;
    for (size_t k = 0; k < n_img; k++)
        for (size_t a = 0; a < per_img; a++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k] + a];
            Job& j = jobs[k * per_img + a];
            j.nbx = s.num_blocks_x;
            j.nby = s.num_blocks_y;
            j.data = file + s.file_ofs;
            j.len = s.file_size;
            j.idx = h_idx + (a ? ain_off[k] : in_off[k]);
            if (may_split && j.nbx * j.nby >= BU_ETC1S_STREAM_MIN_BLOCKS) {
                j.split = true;
                tok_bytes += (j.nbx * j.nby * 6 + 127) & ~(size_t)63;
            }
        }
    if (tok_bytes > ctx->lex_cap) {
        free(ctx->lex_buf);
        ctx->lex_cap = 0;
        ctx->lex_buf = malloc(tok_bytes + tok_bytes / 4);
        if (ctx->lex_buf) ctx->lex_cap = tok_bytes + tok_bytes / 4;
    }
    {
        uint8_t* t = static_cast<uint8_t*>(ctx->lex_buf);
        for (Job& j : jobs) {
            if (!j.split) continue;
            if (!t) {  // (no memory for the tokens: the ordinary loop)
                j.split = false;
                continue;
            }
            const size_t nblk = j.nbx * j.nby;
            j.tok_ep = reinterpret_cast<uint32_t*>(t);
            j.tok_sel = reinterpret_cast<uint16_t*>(t + nblk * 4);
            t += (nblk * 6 + 127) & ~(size_t)63;
        }
    }
    // abort: nothing that is still running can matter any more (a codebook / table / device error, or this function is on its way
    // out): decoders stop at the next row.  slice_failed: some slice failed -- the feeder stops launching, but the OTHER slices
    // decode on: the call reports the first failing slice in file order, and an earlier slice may still fail too.
    std::atomic<bool> abort{false}, slice_failed{false}, feeder_taken{false};
    std::atomic<int> cb_done{0};
    bu_status st_cb = BU_OK, st_feed = BU_OK;
    bool crc_ok = true;
    // Who does what: the CALLING thread starts on the first slice at once (all of it, or its bit-serial half) -- the symbol stream
    // of the (first) slice is the critical path, and a parked pool thread takes tens of microseconds to wake up.  The pool threads
    // pull work items from one counter, in this order: the first slice's resolver half (its lexer is already running), the codebooks,
    // the feeder role (it launches bands of finished units on the context stream until everything is launched), the payload
    // CRC, then the other slices -- a two-thread slice as two items, lexer first, so that whoever pulls a resolver item knows its
    // lexer has been claimed by a thread that never waits for anything.  The calling thread joins the queue when its own slice is
    // done; if no pool thread ever took the feeder role it runs it last, when everything is decoded.
    const uint32_t n_cb0 = (uint32_t)n_cb;
    auto launch = [&](uint32_t u0, uint32_t u1) -> bu_status {
        const unsigned grid = bu_grid_for((size_t)(u1 - u0) * 64, ctx->cu_count);
        if (target == BU_READ_RGBA)
            hipLaunchKernelGGL(bu_etc1s_file_kernel<true>, dim3(grid), dim3(BU_WG), 0, ctx->stream, static_cast<const uint32_t*>(d_idx_view), d_descs,
                               (uint32_t)(descs.size() - 1), u0, u1, reinterpret_cast<const uint32_t*>(aux), n_cb0,
                               reinterpret_cast<const uint2*>(aux + ep_bytes), n_cb0, d_out, reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
        else
            hipLaunchKernelGGL(bu_etc1s_file_kernel<false>, dim3(grid), dim3(BU_WG), 0, ctx->stream, static_cast<const uint32_t*>(d_idx_view), d_descs,
                               (uint32_t)(descs.size() - 1), u0, u1, reinterpret_cast<const uint32_t*>(aux), n_cb0,
                               reinterpret_cast<const uint2*>(aux + ep_bytes), n_cb0, d_out, reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
        BU_HIP(ctx, hipGetLastError());
        return BU_OK;
    };
    auto feeder = [&]() -> bu_status {
        BU_HIP(ctx, hipSetDevice(ctx->device));  // (a pool thread has no current device yet)
        for (unsigned spin = 0; !cb_done.load(std::memory_order_acquire); spin++) {
            if (abort.load(std::memory_order_relaxed)) return BU_OK;
            if (spin > 64) std::this_thread::yield();
        }
        if (st_cb != BU_OK || st_tables != BU_OK) return BU_OK;  // (reported by the caller, in the reference's order)
        if (!lz.endpoints.empty()) BU_HIP(ctx, hipMemcpyAsync(aux, lz.endpoints.data(), lz.endpoints.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!lz.selectors.empty()) BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes, lz.selectors.data(), lz.selectors.size(), hipMemcpyHostToDevice, ctx->stream));
        std::vector<uint32_t> launched(n_img, 0);  // units of image k already handed to the GPU
        for (unsigned spin = 0;; spin++) {
            bool all = true, progressed = false;
            for (size_t k = 0; k < n_img; k++) {
                if (units_of[k] == 0 || launched[k] == units_of[k]) continue;
                // finished units of image k: whole 64-block units inside the rows BOTH of its slices have finished
                uint32_t avail = units_of[k];
                bool finished = true;
                for (size_t a2 = 0; a2 < per_img; a2++) {
                    Job& j = jobs[k * per_img + a2];
                    if (j.done.load(std::memory_order_acquire)) continue;
                    finished = false;
                    const size_t blocks = (size_t)j.rows.load(std::memory_order_acquire) * j.nbx;
                    avail = std::min<uint32_t>(avail, (uint32_t)(blocks / 64));
                }
                if (avail > launched[k] && (finished || (size_t)(avail - launched[k]) * 64 >= BU_ETC1S_BAND_BLOCKS)) {
                    const bu_status ls = launch(unit0[k] + launched[k], unit0[k] + avail);
                    if (ls) return ls;
                    launched[k] = avail;
                    progressed = true;
                }
                if (launched[k] != units_of[k]) all = false;
            }
            if (all || abort.load(std::memory_order_relaxed) || slice_failed.load(std::memory_order_relaxed)) return BU_OK;
            if (!progressed && spin > 16) std::this_thread::yield();
        }
    };
    // (a job runs on a pool thread: an exception -- std::bad_alloc from a vector sized by a hostile file -- must become a status there,
    // it cannot unwind through the pool)
    // publish_rows = false: the slice is decoded AGAIN after a two-thread attempt gave up (settle_split).  The resolver has already published
    // rows -- final ones, which this pass writes again with the same values -- and the feeder may have launched bands over them: the row
    // counter must not fall back to 1 and climb a second time, so this pass publishes nothing and the feeder picks the rest up from `done`.
    auto run_slice = [&](Job& j, bool publish_rows = true) {
        if (st_tables == BU_OK && !abort.load(std::memory_order_relaxed)) {
            try {
                j.st = lz.decode_slice(j.nbx, j.nby, j.data, j.len, j.idx, publish_rows ? &j.rows : nullptr, &abort);
            } catch (...) {
                j.st = BU_ERR_BOUNDS;
            }
        } else
            j.st = BU_ERR_ARGUMENT;  // never reported: an earlier error decides
        if (j.st) slice_failed.store(true, std::memory_order_relaxed);
        j.done.store(1, std::memory_order_release);
    };
    // the halves of a two-thread slice; the one that finishes second settles the slice
    auto settle_split = [&](Job& j) {
        if (j.halves_done.fetch_add(1, std::memory_order_acq_rel) != 1) return;
        if (j.lex_ok && j.res_ok) {
            j.st = BU_OK;
            j.done.store(1, std::memory_order_release);
        } else {
            run_slice(j, false);  // an irregular stream (or an abort): the ordinary loops, from the slice's first bit
        }
    };
    auto run_lex = [&](Job& j) {
        try {
            j.lex_ok = lz.slice_lex(j.nbx, j.nby, j.data, j.len, j.tok_ep, j.tok_sel, j.rows_lexed, j.split_failed);
        } catch (...) {
            j.lex_ok = false;
        }
        if (!j.lex_ok) j.split_failed.store(true, std::memory_order_release);
        settle_split(j);
    };
    auto run_resolve = [&](Job& j) {
        try {
            j.res_ok = lz.slice_resolve(j.nbx, j.nby, j.tok_ep, j.tok_sel, j.idx, j.rows_lexed, j.split_failed, &j.rows, &abort);
        } catch (...) {
            j.res_ok = false;
        }
        if (!j.res_ok) j.split_failed.store(true, std::memory_order_release);
        settle_split(j);
    };
    // the work items behind the calling thread's own start on slice 0
    enum ItemKind { I_RESOLVE, I_FEED, I_CODEBOOKS, I_CRC, I_SLICE, I_LEX };
    struct Item {
        ItemKind kind;
        size_t job;
    };
    std::vector<Item> items;
    if (!jobs.empty() && jobs[0].split) items.push_back({I_RESOLVE, 0});
    items.push_back({I_CODEBOOKS, 0});
    items.push_back({I_FEED, 0});
    items.push_back({I_CRC, 0});
    for (size_t k = 1; k < jobs.size(); k++) {
        if (jobs[k].split) {
            items.push_back({I_LEX, k});
            items.push_back({I_RESOLVE, k});
        } else
            items.push_back({I_SLICE, k});
    }
    std::atomic<size_t> next{0};
    const std::thread::id caller = std::this_thread::get_id();
    // BU_TRACE: when every item started and ended, relative to this point
    struct ItemLog {
        double t0 = 0, t1 = 0;
    };
    std::vector<ItemLog> item_log(trace ? items.size() + 1 : 0);
    const auto t_items = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_items).count(); };
    const std::function<void()> work = [&] {
        for (size_t i; (i = next.fetch_add(1)) < items.size();) {
            const Item it = items[i];
            struct Stamp {
                ItemLog* l;
                decltype(since)& now;
                Stamp(ItemLog* l_, decltype(since)& n) : l(l_), now(n)
                {
                    if (l) l->t0 = now();
                }
                ~Stamp()
                {
                    if (l) l->t1 = now();
                }
            } stamp(trace ? &item_log[i] : nullptr, since);
            switch (it.kind) {
            case I_FEED: {
                // (the calling thread leaves the role to a pool thread: it would stop pulling items for as long as bands are pending;
                // it runs the feeder itself at the very end if nobody else did)
                if (std::this_thread::get_id() == caller || feeder_taken.exchange(true)) break;
                bu_status fs;
                try {
                    fs = feeder();
                } catch (...) {
                    fs = BU_ERR_HIP;
                }
                if (fs) {
                    st_feed = fs;
                    abort.store(true, std::memory_order_relaxed);
                }
                return;  // everything is launched (or stopped): nothing is left that this thread should start on
            }
            case I_CODEBOOKS:  // the first kernel launch waits for them
                try {
                    st_cb = lz.init_codebooks(file + h.endpoint_cb_file_ofs, h.endpoint_cb_file_size, file + h.selector_cb_file_ofs, h.selector_cb_file_size);
                } catch (...) {
                    st_cb = BU_ERR_BOUNDS;
                }
                if (st_cb) abort.store(true, std::memory_order_relaxed);
                cb_done.store(1, std::memory_order_release);
                break;
            case I_CRC:
                // (the serial form: bu_host::crc16 cuts large inputs into pieces for the pool -- which this call is holding)
                if (crc_pending) crc_ok = (uint16_t)~bu_host::crc16_raw(file + 77, len - 77, 0xFFFF) == crc_want;
                break;
            case I_SLICE: run_slice(jobs[it.job]); break;
            case I_LEX: run_lex(jobs[it.job]); break;
            case I_RESOLVE: run_resolve(jobs[it.job]); break;
            }
        }
    };
    const unsigned helpers = bu_host::pool().begin((unsigned)std::min<size_t>(items.size(), bu_host::Pool::capacity()), work);
    (void)helpers;
    struct PoolEnd {  // the pool is released on every path out of this function, after the jobs have seen the abort flag
        std::atomic<bool>& abort;
        bool armed = true;
        ~PoolEnd()
        {
            if (armed) {
                abort.store(true);
                bu_host::pool().end();
            }
        }
    } pool_end{abort};
    if (!jobs.empty()) {
        if (jobs[0].split) run_lex(jobs[0]);
        else run_slice(jobs[0]);
    }
    lap(!jobs.empty() && jobs[0].split ? "slice 0 lexed (resolver on a pool thread)" : "slice 0 decoded");
    work();  // whatever is still in the queue (the feeder item is skipped by this thread)
    pool_end.armed = false;
    bu_host::pool().end();  // every item has finished; the feeder, if a pool thread took the role, has launched everything (or seen a flag)
    if (!feeder_taken.exchange(true)) {  // nobody fed the GPU meanwhile: everything is decoded, launch it now
        try {
            st_feed = feeder();
        } catch (...) {
            st_feed = BU_ERR_HIP;
        }
    }
    lap("pool joined, bands launched");
    if (trace) {
        static const char* const names[] = {"resolve", "feed", "codebooks", "crc", "slice", "lex"};
        for (size_t i = 0; i < items.size(); i++)
            fprintf(stderr, "[bu_read_to]   item %2zu %-9s job %2zu (%5zu x %5zu)  %7.3f .. %7.3f ms\n", i, names[items[i].kind], items[i].job,
                    items[i].kind >= I_SLICE || items[i].kind == I_RESOLVE ? jobs[items[i].job].nbx : (size_t)0,
                    items[i].kind >= I_SLICE || items[i].kind == I_RESOLVE ? jobs[items[i].job].nby : (size_t)0, item_log[i].t0, item_log[i].t1);
    }
    if (crc_pending && !crc_ok) return BU_ERR_DATA_CRC;
    if (st_cb) return st_cb;
    if (st_tables) return st_tables;
    if (st_feed) return st_feed;  // (a device error: it stopped the decoders, whose statuses mean nothing then)
    for (const Job& j : jobs)
        if (j.st) return j.st;
    BU_HIP(ctx, hipMemcpyAsync(status_words.data(), d_status, 8 * n_img, hipMemcpyDeviceToHost, ctx->stream));
    if (p.out_bytes && !direct_out) BU_HIP(ctx, hipMemcpyAsync(out, d_out, p.out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain.armed = false;
    lap("last band + synchronise");
    for (size_t k = 0; k < n_img; k++) {
        st = bu_status_word_decode(status_words[k], nullptr);
        if (st) return st;
    }
    return BU_OK;
}

static bu_status bu_read_to_impl(bu_context* ctx, bu_read_target target, const uint8_t* file, size_t len, bu_basis_header* header_out, bu_image* images,
                                size_t max_images, size_t* n_images, uint8_t* out, size_t out_bytes)
{
    if (!ctx || !out) return BU_ERR_ARGUMENT;
    const bool trace = getenv("BU_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t = now();
        fprintf(stderr, "[bu_read_to] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    BuFilePlan p;
    // Large UASTC files: the payload CRC (0.7 ms per 16 MiB on the host cores -- as long as upload, kernel and download
    // together) is computed ON THE DEVICE over the slice bytes the call uploads anyway (bu_crc16_pieces_kernel, one
    // register per 64 KiB piece); the host only runs the bytes the device never sees (slice table, gaps, the tail of a run
    // behind its last whole piece) and folds everything in file order.  The reference checks the CRC before anything else
    // behind the header (basis.rs:338-341), so a CRC failure takes precedence over every later error, and nothing is
    // reported as success before it is known; an early error return settles it on the host.
    bool crc_deferred = false;
    uint16_t crc_want = 0;
    {
        bu_basis_header h0;
        if (file && len >= ((size_t)1 << 20) && bu_host::read_header(file, len, &h0) == BU_OK && h0.tex_format == 1 && target != BU_READ_UASTC) {
            crc_deferred = true;
            crc_want = h0.data_crc16;
        }
        // ETC1S files of some size: the payload CRC runs on a pool thread beside the entropy decode (bu_read_etc1s_streamed)
        if (file && len >= ((size_t)64 << 10) && !getenv("BU_ETC1S_ONE_LAUNCH") && bu_host::read_header(file, len, &h0) == BU_OK && h0.tex_format == 0) {
            crc_deferred = true;
            crc_want = h0.data_crc16;
        }
    }
    auto settle = [&](bu_status s) {  // the status to report once the deferred CRC is known
        if (crc_deferred) {
            crc_deferred = false;
            if (bu_host::crc16(file + 77, len - 77, 0) != crc_want) return BU_ERR_DATA_CRC;
        }
        return s;
    };
    bu_status st = bu_plan_file(target, file, len, p, !crc_deferred);
    lap("plan (parse + CRCs)");
    if (st) return settle(st);
    const auto rest = [&]() -> bu_status {
    if (header_out) *header_out = p.h;
    if (n_images) *n_images = p.images.size();
    if (out_bytes < p.out_bytes) return BU_ERR_OUTPUT_SIZE;
    if (images) {
        if (p.images.size() > max_images) return BU_ERR_OUTPUT_SIZE;
        for (size_t i = 0; i < p.images.size(); i++) images[i] = p.images[i];
    }
    if (p.images.empty()) return BU_OK;
    if (!p.etc1s && target == BU_READ_UASTC) {  // uastc.rs:85-87: plain copies, no device work
        for (size_t k = 0; k < p.images.size(); k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            if (s.file_size) memcpy(out + p.images[k].offset, file + s.file_ofs, s.file_size);
        }
        return BU_OK;
    }
    // Batched front door: every slice's input is staged at an aligned offset of one device buffer, the device output
    // buffer mirrors `out`, all launches go to the context stream back to back (one status word per image) and a
    // single synchronisation ends the call.  The host-side BasisLZ decode of all slices happens before any upload.
    if (p.etc1s && !getenv("BU_ETC1S_ONE_LAUNCH")) {
        size_t total_blocks = 0;
        for (size_t k = 0; k < p.images.size(); k++) total_blocks += (size_t)p.slices[p.first_slice[k]].num_blocks_x * p.slices[p.first_slice[k]].num_blocks_y;
        if (total_blocks >= BU_ETC1S_STREAM_MIN_BLOCKS) {
            const bool pending = crc_deferred;
            crc_deferred = false;  // settled inside (on a pool thread); nothing is left for settle()
            return bu_read_etc1s_streamed(ctx, target, file, len, p, out, pending, crc_want, trace);
        }
    }
    bu_host::BasisLz lz;
    const size_t n_img = p.images.size();
    std::vector<size_t> in_off(n_img, 0), ain_off(n_img, 0), run_of(n_img, 0);  // run_of[k]: first image of k's run
    std::vector<uint32_t> idx_all;
    size_t total_in = 0;
    auto align_up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    if (p.etc1s) {
        st = bu_make_lz(file, len, p.h, lz);
        if (st) return st;
        lap("codebooks + tables");
        size_t words = 0;
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            const size_t nblk = (size_t)s.num_blocks_x * s.num_blocks_y;
            in_off[k] = words * 4;
            words += (nblk + 63) & ~(size_t)63;
            if (p.alpha_pairs) {
                ain_off[k] = words * 4;
                words += (nblk + 63) & ~(size_t)63;
            }
        }
        idx_all.assign(words ? words : 1, 0);
        std::vector<bu_host::SliceJob> jobs;  // file order: colour slice, then its alpha slice
        jobs.reserve(n_img * (p.alpha_pairs ? 2 : 1));
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            jobs.push_back({s.num_blocks_x, s.num_blocks_y, file + s.file_ofs, s.file_size, idx_all.data() + in_off[k] / 4, BU_OK});
            if (p.alpha_pairs) {
                const bu_slice_desc& a = p.slices[p.first_slice[k] + 1];
                jobs.push_back({a.num_blocks_x, a.num_blocks_y, file + a.file_ofs, a.file_size, idx_all.data() + ain_off[k] / 4, BU_OK});
            }
        }
        st = bu_host::decode_slices(lz, jobs);  // host cores in parallel; the symbol stream is serial only within a slice
        if (st) return st;
        lap("slice symbol streams");
        total_in = words * 4;
    } else {
        // Runs: consecutive slices that sit back to back in the file (the usual layout of a mip chain or a texture array)
        // are staged back to back with ONE upload, and -- for the block-linear targets, whose outputs are then contiguous
        // too -- transcoded with ONE launch over the whole run: 512 slices of 65 536 blocks are one 33 M-block launch
        // (0.3 ms) instead of 512 latency-bound ones (3.9 ms).  The lowest failing block of a run lies in its first
        // failing slice, so the reported error is the sequential loop's.
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& s = p.slices[p.first_slice[k]];
            const bool joins = k > 0 && run_of[k - 1] != SIZE_MAX && s.file_size % 16 == 0 && s.file_size != 0 &&
                               p.slices[p.first_slice[k - 1]].file_size % 16 == 0 && p.slices[p.first_slice[k - 1]].file_size != 0 &&
                               (size_t)p.slices[p.first_slice[k - 1]].file_ofs + p.slices[p.first_slice[k - 1]].file_size == s.file_ofs;
            if (joins) {
                run_of[k] = run_of[k - 1];
                in_off[k] = in_off[k - 1] + p.slices[p.first_slice[k - 1]].file_size;
                total_in = in_off[k] + s.file_size;
            } else {
                total_in = align_up(total_in);
                run_of[k] = k;
                in_off[k] = total_in;
                total_in += s.file_size;
            }
        }
        total_in = align_up(total_in);
    }
    std::lock_guard<std::mutex> g(ctx->lock);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> words(n_img, 0);  // status landing area: declared before anything is queued, outlives the drain
    std::vector<BuEtc1sSlice> descs;        // (likewise: source of an upload)
    struct CrcSeg {
        size_t file_ofs, len, first_piece, n_pieces;
    };
    std::vector<CrcSeg> crc_segs;           // uploaded file ranges whose 64 KiB pieces the device registers
    std::vector<uint16_t> crc_parts((crc_deferred && !p.etc1s) ? total_in / BU_CRC_PIECE + 1 : 1, 0);  // (landing area too)
    size_t crc_pieces = 0;
    BuDrain drain(ctx);
    if ((st = bu_reserve(ctx, &ctx->d_in, &ctx->in_cap, total_in ? total_in : 16))) return st;
    // a page-locked `out` (bu_host_alloc) receives the kernels' stores directly over PCIe: no device output buffer, no download
    void* zout = nullptr;
    const bool direct_out = bu_device_view(out, &zout);
    if (!direct_out && (st = bu_reserve(ctx, &ctx->d_out, &ctx->out_cap, p.out_bytes ? p.out_bytes : 16))) return st;
    const size_t ep_bytes = p.etc1s ? align_up(lz.endpoints.size() * 4) : 0, sel_bytes = p.etc1s ? align_up(lz.selectors.size()) : 0;
    // ETC1S: one descriptor per image (+ sentinel) behind the codebooks and the status words
    uint32_t n_units = 0;
    if (p.etc1s) {
        descs.reserve(n_img + 1);
        for (size_t k = 0; k < n_img; k++) {
            const bu_slice_desc& sl = p.slices[p.first_slice[k]];
            const size_t nblk = (size_t)sl.num_blocks_x * sl.num_blocks_y;
            if (p.images[k].size == 0 || nblk == 0) continue;
            BuEtc1sSlice d;
            d.unit0 = n_units;
            d.n_blocks = (uint32_t)nblk;
            d.nbx = sl.num_blocks_x;
            d.idx_ofs = (uint32_t)(in_off[k] / 4);
            d.aidx_ofs = (p.alpha_pairs && target == BU_READ_RGBA) ? (uint32_t)(ain_off[k] / 4) : 0xFFFFFFFFu;
            d.image = (uint32_t)k;
            d.out_ofs = p.images[k].offset;
            descs.push_back(d);
            n_units += (uint32_t)((nblk + 63) / 64);
        }
        BuEtc1sSlice end = {};
        end.unit0 = n_units;
        descs.push_back(end);
    }
    const size_t desc_bytes = align_up(descs.size() * sizeof(BuEtc1sSlice)), status_bytes = align_up(8 * n_img);
    const size_t crc_bytes = align_up(2 * crc_parts.size());
    if ((st = bu_reserve(ctx, &ctx->d_aux, &ctx->aux_cap, ep_bytes + sel_bytes + status_bytes + desc_bytes + crc_bytes + 256))) return st;
    uint8_t* d_in = static_cast<uint8_t*>(ctx->d_in);
    uint8_t* d_out = direct_out ? static_cast<uint8_t*>(zout) : static_cast<uint8_t*>(ctx->d_out);
    uint8_t* aux = static_cast<uint8_t*>(ctx->d_aux);
    uint64_t* d_status = reinterpret_cast<uint64_t*>(aux + ep_bytes + sel_bytes);
    uint16_t* d_crc = reinterpret_cast<uint16_t*>(aux + ep_bytes + sel_bytes + status_bytes + desc_bytes);
    // registers of the whole 64 KiB pieces of an uploaded range, on the stream that carries its upload
    auto crc_enqueue = [&](const uint8_t* d_range, size_t file_ofs, size_t nbytes, hipStream_t ps) {
        if (!crc_deferred) return;
        const size_t np = nbytes / BU_CRC_PIECE;
        if (np) hipLaunchKernelGGL(bu_crc16_pieces_kernel, dim3((unsigned)np), dim3(256), 0, ps, reinterpret_cast<const uint4*>(d_range), d_crc + crc_pieces, ctx->d_crc_tables);
        crc_segs.push_back({file_ofs, nbytes, crc_pieces, np});
        crc_pieces += np;
    };
    BU_HIP(ctx, hipMemsetAsync(d_status, 0xFF, 8 * n_img, ctx->stream));
    if (p.etc1s) {
        BU_HIP(ctx, hipMemcpyAsync(d_in, idx_all.data(), total_in, hipMemcpyHostToDevice, ctx->stream));
        if (!lz.endpoints.empty()) BU_HIP(ctx, hipMemcpyAsync(aux, lz.endpoints.data(), lz.endpoints.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!lz.selectors.empty()) BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes, lz.selectors.data(), lz.selectors.size(), hipMemcpyHostToDevice, ctx->stream));
        BU_HIP(ctx, hipMemcpyAsync(aux + ep_bytes + sel_bytes + status_bytes, descs.data(), descs.size() * sizeof(BuEtc1sSlice), hipMemcpyHostToDevice, ctx->stream));
        // ONE launch for the whole file (basis.rs:42-58 / 103-123 walk the slices one by one)
        if (n_units) {
            const uint32_t n_cb0 = (uint32_t)lz.endpoints.size();
            const unsigned grid = bu_grid_for((size_t)n_units * 64, ctx->cu_count);
            const BuEtc1sSlice* d_descs = reinterpret_cast<const BuEtc1sSlice*>(aux + ep_bytes + sel_bytes + status_bytes);
            if (target == BU_READ_RGBA)
                hipLaunchKernelGGL(bu_etc1s_file_kernel<true>, dim3(grid), dim3(BU_WG), 0, ctx->stream, reinterpret_cast<const uint32_t*>(d_in), d_descs,
                                   (uint32_t)(descs.size() - 1), 0u, n_units, reinterpret_cast<const uint32_t*>(aux), n_cb0,
                                   reinterpret_cast<const uint2*>(aux + ep_bytes), n_cb0, d_out, reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
            else
                hipLaunchKernelGGL(bu_etc1s_file_kernel<false>, dim3(grid), dim3(BU_WG), 0, ctx->stream, reinterpret_cast<const uint32_t*>(d_in), d_descs,
                                   (uint32_t)(descs.size() - 1), 0u, n_units, reinterpret_cast<const uint32_t*>(aux), n_cb0,
                                   reinterpret_cast<const uint2*>(aux + ep_bytes), n_cb0, d_out, reinterpret_cast<unsigned long long*>(d_status), ctx->d_tables);
            BU_HIP(ctx, hipGetLastError());
        }
    }
    int used_extra = 0;  // own streams 0..used_extra-1 carry pieces (forked behind the status reset, joined on the host)
    size_t run_piece_bytes = (size_t)16 << 20;  // upper bound; a run is cut into >= 4 pieces of >= 4 MiB (below)
    bool piece_fixed = false;
    if (const char* e = getenv("BU_RUN_PIECE_MIB")) {  // 0 disables the pieced pipeline
        run_piece_bytes = (size_t)atoll(e) << 20;
        piece_fixed = true;
    }
    for (size_t k = 0; k < n_img; k++) {
        const bu_slice_desc& s = p.slices[p.first_slice[k]];
        const bu_image& im = p.images[k];
        if (im.size == 0) continue;
        if (p.etc1s) {
            // (launched once for the whole file above)
        } else {
            size_t run_end = k;  // last image of the run starting at k (only evaluated for run leaders)
            bool pieced = false;
            if (run_of[k] == k) {
                while (run_end + 1 < n_img && run_of[run_end + 1] == k) run_end++;
                const size_t run_bytes = in_off[run_end] + p.slices[p.first_slice[run_end]].file_size - in_off[k];
                // A large block-linear run with a mapped (page-locked) output: upload and transcode in pieces on two
                // streams, so that piece i's results cross PCIe upstream while piece i+1 comes down.
                // piece size: a quarter of the run, between 4 and 16 MiB (a 16 MiB file: 0.672 ms in one piece, 0.585 in four,
                // 0.645 in eight -- tools/exp/file_time.py)
                size_t piece_bytes = run_piece_bytes;
                if (!piece_fixed) {
                    piece_bytes = (run_bytes / 4) & ~(((size_t)1 << 20) - 1);
                    if (piece_bytes < ((size_t)4 << 20)) piece_bytes = (size_t)4 << 20;
                    if (piece_bytes > run_piece_bytes) piece_bytes = run_piece_bytes;
                }
                // (Round 6 tried the same pieces for a PAGEABLE output too -- four streams, upload and launch of a piece on one stream, shared launch shapes -- from
                // runs of 64 MiB on: 2.64 against 2.52 ms for a 64 MiB file, 5.17 against 4.97 for 128 MiB, profiles/r06_read_to_pieces_pageable_output.txt.  The call is
                // two PCIe copies long and the kernels are 1 % of it; with a pageable output the run stays one upload, one launch -- which draws its tiles by
                // ticket from 2^24 blocks on -- and one download.)
                if (target != BU_READ_RGBA && direct_out && piece_bytes && run_bytes >= 2 * piece_bytes) {
                    pieced = true;
                    constexpr int n_ps = 2;  // streams that carry pieces: the context's internal one and one of its own
                    {
                        const bu_status sst = bu_ctx_streams(ctx, n_ps - 1);
                        if (sst) return sst;
                    }
                    if (used_extra < n_ps - 1) {
                        BU_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));  // the status words are reset on the context stream
                        for (; used_extra < n_ps - 1; used_extra++) BU_HIP(ctx, hipStreamWaitEvent(ctx->extra_streams[used_extra], ctx->ev0, 0));
                    }
                    const bu_target pbt = target == BU_READ_ASTC ? BU_TARGET_ASTC : target == BU_READ_BC7 ? BU_TARGET_BC7
                                          : target == BU_READ_ETC1 ? BU_TARGET_ETC1 : BU_TARGET_ETC2;
                    const size_t obytes = bu_target_block_bytes(pbt);
                    size_t piece_no = 0;
                    for (size_t done = 0; done < run_bytes; done += piece_bytes, piece_no++) {
                        const size_t nbytes = run_bytes - done < piece_bytes ? run_bytes - done : piece_bytes;
                        const size_t lane = piece_no % (size_t)n_ps;
                        hipStream_t ps = lane ? ctx->extra_streams[lane - 1].load(std::memory_order_acquire) : ctx->stream;
                        BU_HIP(ctx, hipMemcpyAsync(d_in + in_off[k] + done, file + s.file_ofs + done, nbytes, hipMemcpyHostToDevice, ps));
                        crc_enqueue(d_in + in_off[k] + done, (size_t)s.file_ofs + done, nbytes, ps);
                        st = bu_launch_uastc(ctx, pbt, d_in + in_off[k] + done, nbytes / 16, d_out + im.offset + (done / 16) * obytes, 1, done / 16, d_status + k, ps,
                                             BU_ZEROCOPY_GRID);
                        if (st) return st;
                    }
                } else {
                    BU_HIP(ctx, hipMemcpyAsync(d_in + in_off[k], file + s.file_ofs, run_bytes, hipMemcpyHostToDevice, ctx->stream));
                    crc_enqueue(d_in + in_off[k], s.file_ofs, run_bytes, ctx->stream);
                }
            }
            const bu_target bt = target == BU_READ_RGBA ? BU_TARGET_RGBA32
                                 : target == BU_READ_ASTC ? BU_TARGET_ASTC
                                 : target == BU_READ_BC7  ? BU_TARGET_BC7
                                 : target == BU_READ_ETC1 ? BU_TARGET_ETC1
                                                          : BU_TARGET_ETC2;
            if (target == BU_READ_RGBA) {  // image geometry differs per slice: one launch each
                st = bu_launch_uastc(ctx, bt, d_in + in_off[k], s.file_size / 16, d_out + im.offset, s.num_blocks_x ? s.num_blocks_x : 1, 0, d_status + k,
                                     ctx->stream, direct_out ? BU_ZEROCOPY_GRID : 0);
            } else if (run_of[k] == k && !pieced) {  // block-linear: the run's outputs are contiguous from im.offset on
                const size_t run_bytes = in_off[run_end] + p.slices[p.first_slice[run_end]].file_size - in_off[k];
                st = bu_launch_uastc(ctx, bt, d_in + in_off[k], run_bytes / 16, d_out + im.offset, 1, 0, d_status + k, ctx->stream,
                                     direct_out ? BU_ZEROCOPY_GRID : 0);
            }
        }
        if (st) return st;
    }
    lap("reserve + enqueue");
    // the status words (and the CRC registers of the pieces) are read on the context stream: it must see the other streams' kernels.  Joined on the
    // HOST -- a cross-stream event wait costs 70-80 us on this runtime, a wait for streams that are nearly done costs nothing
    for (int i = 0; i < used_extra; i++) BU_HIP(ctx, hipStreamSynchronize(ctx->extra_streams[i]));
    BU_HIP(ctx, hipGetLastError());
    BU_HIP(ctx, hipMemcpyAsync(words.data(), d_status, 8 * n_img, hipMemcpyDeviceToHost, ctx->stream));
    if (crc_pieces) BU_HIP(ctx, hipMemcpyAsync(crc_parts.data(), d_crc, 2 * crc_pieces, hipMemcpyDeviceToHost, ctx->stream));
    if (p.out_bytes && !direct_out) BU_HIP(ctx, hipMemcpyAsync(out, d_out, p.out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain.armed = false;
    lap("download + synchronise");
    if (crc_deferred) {  // fold the device's piece registers with the bytes only the host has seen, in file order
        crc_deferred = false;
        std::sort(crc_segs.begin(), crc_segs.end(), [](const CrcSeg& a, const CrcSeg& b) { return a.file_ofs < b.file_ofs; });
        bool foldable = true;
        size_t pos = 77;
        for (const CrcSeg& g2 : crc_segs) {  // overlapping slices (a hostile slice table): the plain host CRC decides
            if (g2.file_ofs < pos || g2.file_ofs + g2.len > len) foldable = false;
            pos = g2.file_ofs + g2.len;
        }
        bool crc_ok;
        if (!foldable) {
            crc_ok = bu_host::crc16(file + 77, len - 77, 0) == crc_want;
        } else {
            const uint16_t shift_piece = bu_host::crc16_shift(1, BU_CRC_PIECE);  // x^(8 * 65536)
            uint16_t reg = 0xFFFF;  // register of CRC-16/GENIBUS before the first payload byte
            pos = 77;
            for (const CrcSeg& g2 : crc_segs) {
                reg = bu_host::crc16_raw(file + pos, g2.file_ofs - pos, reg);
                for (size_t i = 0; i < g2.n_pieces; i++) reg = (uint16_t)(bu_host::crc16_gf_mul(reg, shift_piece) ^ crc_parts[g2.first_piece + i]);
                const size_t covered = g2.n_pieces * BU_CRC_PIECE;
                reg = bu_host::crc16_raw(file + g2.file_ofs + covered, g2.len - covered, reg);
                pos = g2.file_ofs + g2.len;
            }
            reg = bu_host::crc16_raw(file + pos, len - pos, reg);
            crc_ok = (uint16_t)~reg == crc_want;
        }
        lap("data CRC fold");
        if (!crc_ok) return BU_ERR_DATA_CRC;
    }
    for (size_t k = 0; k < n_img; k++) {  // first Err (in slice order) aborts the whole call, like the `?` in the reference drivers
        st = bu_status_word_decode(words[k], nullptr);
        if (st) return st;
    }
    return BU_OK;
    };
    return settle(rest());
}


// C++ exceptions must not cross the C ABI (a ctypes or Rust caller would be terminated): vectors sized from untrusted
// file fields can throw std::bad_alloc, thread creation std::system_error.  A file that asks for more memory than
// exists is reported like any other out-of-bounds field.
#define BU_GUARDED(call)                 \
    try {                                \
        return call;                     \
    } catch (const std::bad_alloc&) {    \
        return BU_ERR_BOUNDS;            \
    } catch (...) {                      \
        return BU_ERR_HIP;               \
    }
bu_status bu_basis_read_slice_descs(const uint8_t* file, size_t len, const bu_basis_header* header, bu_slice_desc* out, size_t max_descs,
                                    size_t* n_descs)
{
    BU_GUARDED(bu_basis_read_slice_descs_impl(file, len, header, out, max_descs, n_descs))
}
bu_status bu_read_query(bu_read_target target, const uint8_t* file, size_t len, size_t* n_images, size_t* out_bytes)
{
    BU_GUARDED(bu_read_query_impl(target, file, len, n_images, out_bytes))
}
bu_status bu_basislz_decode(const uint8_t* file, size_t len, uint32_t slice_index, uint32_t* endpoints_out, uint8_t* selectors_out,
                            uint32_t* idx_out)
{
    BU_GUARDED(bu_basislz_decode_impl(file, len, slice_index, endpoints_out, selectors_out, idx_out))
}
bu_status bu_read_to(bu_context* ctx, bu_read_target target, const uint8_t* file, size_t len, bu_basis_header* header_out, bu_image* images,
                     size_t max_images, size_t* n_images, uint8_t* out, size_t out_bytes)
{
    BU_GUARDED(bu_read_to_impl(ctx, target, file, len, header_out, images, max_images, n_images, out, out_bytes))
}
#undef BU_GUARDED

bu_status bu_basis_write_uastc(const bu_slice_desc* descs, const uint8_t* const* slice_data, const size_t* slice_bytes, size_t n_slices,
                               uint16_t header_flags, uint8_t tex_type, uint8_t* out, size_t out_cap, size_t* out_len)
{
    if ((n_slices && (!descs || !slice_data || !slice_bytes)) || n_slices >= (1u << 24)) return BU_ERR_ARGUMENT;
    size_t total = 77 + 23 * n_slices;
    for (size_t i = 0; i < n_slices; i++) total += slice_bytes[i];
    if (out_len) *out_len = total;
    if (!out) return BU_OK;
    if (out_cap < total || total > 0xFFFFFFFFull) return BU_ERR_OUTPUT_SIZE;
    auto put = [&](size_t pos, uint32_t v, int n) { for (int k = 0; k < n; k++) out[pos + k] = (uint8_t)(v >> (8 * k)); };
    memset(out, 0, 77 + 23 * n_slices);
    size_t ofs = 77 + 23 * n_slices;
    uint32_t n_images = 0;
    for (size_t i = 0; i < n_slices; i++) {
        const size_t d = 77 + 23 * i;
        put(d, descs[i].image_index, 3);
        out[d + 3] = descs[i].level_index;
        out[d + 4] = descs[i].flags;
        put(d + 5, descs[i].orig_width, 2);
        put(d + 7, descs[i].orig_height, 2);
        put(d + 9, descs[i].num_blocks_x, 2);
        put(d + 11, descs[i].num_blocks_y, 2);
        put(d + 13, (uint32_t)ofs, 4);
        put(d + 17, (uint32_t)slice_bytes[i], 4);
        put(d + 21, bu_host::crc16(slice_data[i], slice_bytes[i], 0), 2);
        if (slice_bytes[i]) memcpy(out + ofs, slice_data[i], slice_bytes[i]);
        ofs += slice_bytes[i];
        if (descs[i].image_index + 1 > n_images) n_images = descs[i].image_index + 1;
    }
    put(0, 0x4273, 2);   // sig
    put(2, 0x13, 2);     // ver
    put(4, 77, 2);       // header_size
    put(8, (uint32_t)(total - 77), 4);
    put(12, bu_host::crc16(out + 77, total - 77, 0), 2);
    put(14, (uint32_t)n_slices, 3);
    put(17, n_images, 3);
    out[20] = 1;         // UASTC4x4
    put(21, header_flags, 2);
    out[23] = tex_type;
    put(65, 77, 4);      // slice_desc_file_ofs
    put(6, bu_host::crc16(out + 8, 77 - 8, 0), 2);
    return BU_OK;
}


}  // extern "C"
