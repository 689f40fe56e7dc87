// TEST-ONLY host build of the kernels' per-block code (the same headers the HIP kernels include,
// compiled as plain C++).  It lets the CPU test-suite compare the exact device logic with the
// oracle on millions of blocks in a container that has no GPU.  It is never part of the product
// library and nothing under basisu_rs_amd/ loads it.
#include "bu_uastc_dispatch.hpp"
#include "bu_batch_plan.hpp"

static BuTablesAll g_tables;
static bool g_init = false;
static const BuTables& tables()
{
    if (!g_init) {
        bu_build_tables(&g_tables);
        g_init = true;
    }
    return g_tables.t;
}

extern "C" {
// returns 0 ok / 1 bad mode / 2 bad pattern; out sized 16/16/8/16/64 bytes
int bu_emul_block(int target, const uint8_t* in, uint8_t* out)
{
    const BuTables& T = tables();
    BuBlk b;
    memcpy(b.w, in, 16);
    const uint32_t mode = T.mode_lut[b.w[0] & 127u];
    uint32_t o[16] = {0};
    int st;
    switch (target) {
    case BU_TGT_ASTC: st = bu_block_any<BU_TGT_ASTC>(T, mode, b, o); break;
    case BU_TGT_BC7: st = bu_block_any<BU_TGT_BC7>(T, mode, b, o); break;
    case BU_TGT_ETC1: st = bu_block_any<BU_TGT_ETC1>(T, mode, b, o); break;
    case BU_TGT_ETC2: st = bu_block_any<BU_TGT_ETC2>(T, mode, b, o); break;
    default: st = bu_block_any<BU_TGT_RGBA>(T, mode, b, o); break;
    }
    const int n = target == BU_TGT_ETC1 ? 8 : (target == BU_TGT_RGBA ? 64 : 16);
    memcpy(out, o, n);
    return st;
}
// block-linear batch (RGBA = 64 bytes per block); stops at nothing, statuses in st[]
void bu_emul_batch(int target, const uint8_t* in, size_t n_blocks, uint8_t* out, uint8_t* st)
{
    const size_t obs = target == BU_TGT_ETC1 ? 8 : (target == BU_TGT_RGBA ? 64 : 16);
    for (size_t i = 0; i < n_blocks; i++) st[i] = (uint8_t)bu_emul_block(target, in + 16 * i, out + obs * i);
}
size_t bu_emul_tables_size(void) { return sizeof(BuTablesAll); }
// the launch plan of bu_uastc_transcode_batch_in_flight for a slice table given as ADDRESSES (nothing is dereferenced): rows of
// (launch, in, out, n_blocks, index base), one per run or piece, in launch order; returns the number of rows (or the number needed
// if `cap` is too small) and the number of launches in *out_launches
size_t bu_emul_plan_in_flight(size_t n_slices, const uint64_t* in_addr, const size_t* n_blocks, const uint64_t* out_addr, size_t block_bytes,
                              const uint64_t* index_base, int n_streams, size_t blocks_per_row, size_t max_runs, size_t group_blocks, uint64_t* rows,
                              size_t cap, size_t* out_launches)
{
    std::vector<const void*> in(n_slices);
    std::vector<void*> out(n_slices);
    for (size_t i = 0; i < n_slices; i++) {
        in[i] = reinterpret_cast<const void*>(in_addr[i]);
        out[i] = reinterpret_cast<void*>(out_addr[i]);
    }
    std::vector<BuRun> runs, pieces;
    bu_merge_runs(n_slices, in.data(), n_blocks, out.data(), block_bytes, index_base, runs);
    std::vector<BuLaunchGroup> groups;
    bu_plan_in_flight(runs, n_streams, blocks_per_row, block_bytes, max_runs, group_blocks, groups, pieces);
    size_t n = 0;
    for (size_t j = 0; j < groups.size(); j++)
        for (size_t k = 0; k < groups[j].count; k++, n++) {
            const size_t idx = groups[j].first + k;
            const BuRun& r = idx >= runs.size() ? pieces[idx - runs.size()] : runs[idx];
            if (n < cap) {
                rows[5 * n + 0] = j;
                rows[5 * n + 1] = reinterpret_cast<uint64_t>(r.in);
                rows[5 * n + 2] = reinterpret_cast<uint64_t>(r.out);
                rows[5 * n + 3] = r.n;
                rows[5 * n + 4] = r.base;
            }
        }
    if (out_launches) *out_launches = groups.size();
    return n;
}
}
