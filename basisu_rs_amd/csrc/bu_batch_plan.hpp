// Host-side planning of the batch entry points (bu_capi_slice.hpp): slices -> runs -> launches.  No HIP in here: the test-only host
// build (tests/host_emul) compiles it as it is and tests/test_batch_plan.py checks its invariants without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

// a run = slices that are contiguous in input, output and block numbering
struct BuRun {
    const uint8_t* in;
    uint8_t* out;
    size_t n;
    uint64_t base;
};

// one launch of the pipelined batch call: `count` consecutive runs from `first` -- an index into the run list or, from runs.size() on,
// into the list of pieces cut out of large runs
struct BuLaunchGroup {
    size_t first, count, blocks;
};

// slices merged into runs: slice i = n_blocks[i] blocks at d_in[i] -> d_out[i], numbered from index_base[i] (NULL: back to back from 0);
// empty slices vanish, a slice that continues its predecessor in input, output (block_bytes per block) and numbering extends its run
inline void bu_merge_runs(size_t n_slices, const void* const* d_in, const size_t* n_blocks, void* const* d_out, size_t block_bytes,
                          const uint64_t* index_base, std::vector<BuRun>& runs)
{
    uint64_t next_base = 0;
    for (size_t i = 0; i < n_slices; i++) {
        const uint64_t base = index_base ? index_base[i] : next_base;
        next_base = base + n_blocks[i];
        if (n_blocks[i] == 0) continue;
        const uint8_t* in = static_cast<const uint8_t*>(d_in[i]);
        uint8_t* out = static_cast<uint8_t*>(d_out[i]);
        if (!runs.empty()) {
            BuRun& r = runs.back();
            if (r.in + r.n * 16 == in && r.out + r.n * block_bytes == out && r.base + r.n == base) {
                r.n += n_blocks[i];
                continue;
            }
        }
        runs.push_back(BuRun{in, out, n_blocks[i], base});
    }
}

// The launches of bu_uastc_transcode_batch_in_flight.
// 1. Consecutive runs are grouped into launches of up to `group_blocks` blocks: a group is closed when it holds that many blocks or `max_runs` runs,
//    a run of `group_blocks` or more is a group of its own.  Below 2^20 blocks a launch is bound by the host's ~4 us.  BC7 / ASTC group up to 2^23 blocks --
//    their multi-run kernel (a persistent grid over the tiles of all runs, bu_launch_runs) runs a group at the rate of the plain launches or better, with
//    one enqueue per group: 64 / 512 slices of 65 536 blocks in separate allocations 36-48 / 226-303 us -> 30-31 / 197-204 per batch, 128 of 262 144
//    227-248 -> 190, 64 atlases of 2^20 blocks 388-392 -> 373-377 on the host's clock with a single enqueueing thread (groups of 2^20 / 2^22 / 2^23 blocks, queue
//    pool and CU-mask streams: tools/exp/group_size.sh, group_size_few.sh).  ETC1 / ETC2 / RGBA32 keep 2^20: the ETC multi-run kernel is 10 % behind
//    their plain launches in flight (64 atlases 797 -> 886 / 973 -> 1122 us if grouped, tools/exp/group_size_targets.sh).
// 2. With more than one stream, a single-run group of more than 2^23 blocks is cut into equal pieces of at most 2^23: launches of 2^22-2^23 blocks
//    are what a pipeline of four runs best on (per 2^25-block array, four in flight: 2^20-block launches 179 us, 2^22 175, 2^23 175-176, 2^25
//    178-184 -- a launch's 512 persistent workgroups walk fixed shares, and the longer the walk the longer its uneven tail;
//    profiles/r06_in_flight_launch_size.txt).
// 3. A batch that still makes fewer launches than streams has its largest single-run groups cut further, as long as a piece keeps at least 2^20
//    blocks.  Pieces end on tile boundaries: 1024 blocks, and with a pitch 16 block rows of it (the rectangular tiles of the kernels; RGBA32
//    needs whole block rows) -- lcm(16 * blocks_per_row, 1024).
// Every block of every run is in exactly one launch, in order; pieces carry their share of the run's block numbering.
inline void bu_plan_in_flight(const std::vector<BuRun>& runs, int n_streams, size_t blocks_per_row, size_t block_bytes, size_t max_runs, size_t group_blocks,
                              std::vector<BuLaunchGroup>& groups, std::vector<BuRun>& pieces)
{
    constexpr size_t GROUP_BLOCKS = (size_t)1 << 20, MAX_LAUNCH_BLOCKS = (size_t)1 << 23;
    if (group_blocks < GROUP_BLOCKS) group_blocks = GROUP_BLOCKS;
    if (group_blocks > MAX_LAUNCH_BLOCKS) group_blocks = MAX_LAUNCH_BLOCKS;
    groups.clear();
    pieces.clear();
    for (size_t i = 0; i < runs.size();) {
        BuLaunchGroup g{i, 0, 0};
        while (i < runs.size() && g.count < max_runs && (g.count == 0 || g.blocks + runs[i].n <= group_blocks)) {
            g.blocks += runs[i].n;
            g.count++;
            i++;
            if (g.blocks >= group_blocks) break;
        }
        groups.push_back(g);
    }
    if (groups.empty() || n_streams <= 1) return;
    // pieces per group: by size first ...
    std::vector<size_t> want(groups.size(), 1);
    size_t launches = 0;
    bool any_cut = false;
    for (size_t k = 0; k < groups.size(); k++) {
        if (groups[k].count == 1 && groups[k].blocks > MAX_LAUNCH_BLOCKS) {
            want[k] = (groups[k].blocks + MAX_LAUNCH_BLOCKS - 1) / MAX_LAUNCH_BLOCKS;
            any_cut = true;
        }
        launches += want[k];
    }
    size_t align = 1024;
    if (blocks_per_row) {
        size_t a = blocks_per_row * 16, b = 1024;  // gcd
        while (b) {
            const size_t t = a % b;
            a = b;
            b = t;
        }
        align = blocks_per_row * 16 / a * 1024;
    }
    // ... then, if the batch still has fewer launches than streams, more pieces for the single-run groups while a piece keeps 2^20 blocks
    if (launches < (size_t)n_streams) {
        const size_t spare = (size_t)n_streams - launches, n_groups = groups.size();
        for (size_t k = 0; k < groups.size() && launches < (size_t)n_streams; k++) {
            const BuLaunchGroup& g = groups[k];
            if (g.count != 1) continue;
            size_t n_pieces = want[k] + (spare + n_groups - 1) / n_groups;
            while (n_pieces > want[k] && ((g.blocks / n_pieces) / align) * align < GROUP_BLOCKS) n_pieces--;
            if (n_pieces > want[k]) {
                launches += n_pieces - want[k];
                want[k] = n_pieces;
                any_cut = true;
            }
        }
    }
    if (!any_cut) return;
    std::vector<BuLaunchGroup> cut;
    for (size_t k = 0; k < groups.size(); k++) {
        const BuLaunchGroup& g = groups[k];
        if (want[k] <= 1) {
            cut.push_back(g);
            continue;
        }
        const BuRun r = runs[g.first];
        size_t per = ((r.n + want[k] - 1) / want[k] + align - 1) / align * align;
        // (rounding a piece up to the alignment must not carry it past 2^23 again: one piece more then)
        for (size_t w = want[k] + 1; want[k] > 1 && per > MAX_LAUNCH_BLOCKS && align < MAX_LAUNCH_BLOCKS / 2 && r.n > MAX_LAUNCH_BLOCKS; w++) per = ((r.n + w - 1) / w + align - 1) / align * align;
        for (size_t done = 0; done < r.n; done += per) {
            const size_t n = r.n - done < per ? r.n - done : per;
            pieces.push_back(BuRun{r.in + done * 16, r.out + done * block_bytes, n, r.base + done});
            cut.push_back(BuLaunchGroup{runs.size() + pieces.size() - 1, 1, n});
        }
    }
    groups.swap(cut);
}
