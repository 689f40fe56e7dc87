// Multi-GPU half of the C ABI (include/basisu_hip.h, "texture-array shards"): the one exchange step of the
// north-star configuration.  Slices of a texture array are independent contiguous block ranges
// (basis.rs:531-552; the per-slice loop of read_to_bc7, basis.rs:246-257), so every device transcodes its own
// contiguous range with no data-path collective; afterwards one all-gather leaves the whole array on every device.
//
//   bu_comm_* / bu_allgather_inplace     one process per GPU: RCCL in-place all-gather over xGMI
//                                        (send = own shard inside the full buffer, recv = the full buffer)
//   bu_ipc_* / bu_allgather_peer         one process per GPU: direct peer pulls, world-1 concurrent copies
//                                        (xGMI is point to point: 7 links carry 7 copies at once, no ring)
//   bu_array_transcode_sharded           one process driving n devices: transcode + peer pulls
//
// RCCL is resolved at run time (dlsym on what the process already maps -- PyTorch-ROCm brings its own librccl --
// then librccl.so.1 from the ROCm installation), so the library has no link-time dependency on it.
// Included by bu_hip.hip after bu_context and the BU_HIP macro are defined.
#pragma once
#include <dlfcn.h>

#include <vector>

struct bu_comm {
    bu_context* ctx = nullptr;
    void* nccl = nullptr;  // ncclComm_t
    int world = 1, rank = 0;
};

namespace bu_multi {

// the slice of rccl.h this file uses (ncclResult_t is an int enum, ncclUint8 == 1, ncclUniqueId is 128 opaque bytes)
struct NcclId {
    char internal[128];
};
struct Rccl {
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;     // optional: how many ranks the communicator itself reports
    int (*CommUserRank)(void*, int*) = nullptr;  // optional
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

inline const Rccl& rccl()
{
    static const Rccl r = [] {
        Rccl t;
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclAllGather")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return t;
        }
        t.GetUniqueId = reinterpret_cast<decltype(t.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        t.CommInitRank = reinterpret_cast<decltype(t.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        t.AllGather = reinterpret_cast<decltype(t.AllGather)>(dlsym(h, "ncclAllGather"));
        t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        t.CommCount = reinterpret_cast<decltype(t.CommCount)>(dlsym(h, "ncclCommCount"));
        t.CommUserRank = reinterpret_cast<decltype(t.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.AllGather;
        return t;
    }();
    return r;
}

inline bu_status nccl_fail(bu_context* ctx, int rc, const char* what)
{
    const Rccl& r = rccl();
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return BU_ERR_HIP;
}

// contiguous range of rank r out of n items over `world` ranks: [r*n/world, (r+1)*n/world)  
inline void partition(size_t n, int world, int r, size_t* lo, size_t* hi)
{
    *lo = n * (size_t)r / (size_t)world;
    *hi = n * (size_t)(r + 1) / (size_t)world;
}

}  // namespace bu_multi

extern "C" {

bu_status bu_comm_unique_id(uint8_t id[BU_COMM_ID_BYTES])
{
    if (!id) return BU_ERR_ARGUMENT;
    const bu_multi::Rccl& r = bu_multi::rccl();
    if (!r.ok) return BU_ERR_UNSUPPORTED;
    bu_multi::NcclId nid;
    static_assert(sizeof(nid) == BU_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (r.GetUniqueId(&nid) != 0) return BU_ERR_HIP;
    memcpy(id, nid.internal, BU_COMM_ID_BYTES);
    return BU_OK;
}

bu_status bu_comm_create(bu_context* ctx, int world, int rank, const uint8_t id[BU_COMM_ID_BYTES], bu_comm** out_comm)
{
    if (!ctx || !id || !out_comm || world < 1 || rank < 0 || rank >= world) return BU_ERR_ARGUMENT;
    *out_comm = nullptr;
    const bu_multi::Rccl& r = bu_multi::rccl();
    if (!r.ok) {
        snprintf(ctx->err, sizeof(ctx->err), "RCCL (librccl.so) could not be resolved");
        return BU_ERR_UNSUPPORTED;
    }
    BU_HIP(ctx, hipSetDevice(ctx->device));
    bu_comm* c = new (std::nothrow) bu_comm();
    if (!c) return BU_ERR_HIP;
    c->ctx = ctx;
    c->world = world;
    c->rank = rank;
    bu_multi::NcclId nid;
    memcpy(nid.internal, id, BU_COMM_ID_BYTES);
    const int rc = r.CommInitRank(&c->nccl, world, nid, rank);
    if (rc != 0) {
        delete c;
        return bu_multi::nccl_fail(ctx, rc, "ncclCommInitRank");
    }
    *out_comm = c;
    return BU_OK;
}

void bu_comm_destroy(bu_comm* comm)
{
    if (!comm) return;
    if (comm->nccl) {
        (void)hipSetDevice(comm->ctx->device);
        (void)bu_multi::rccl().CommDestroy(comm->nccl);
    }
    delete comm;
}

bu_status bu_comm_query(bu_comm* comm, int* out_ranks, int* out_rank)
{
    if (!comm || !comm->nccl) return BU_ERR_ARGUMENT;
    const bu_multi::Rccl& r = bu_multi::rccl();
    if (!r.CommCount || !r.CommUserRank) return BU_ERR_UNSUPPORTED;
    int n = 0, me = 0;
    int rc = r.CommCount(comm->nccl, &n);
    if (rc == 0) rc = r.CommUserRank(comm->nccl, &me);
    if (rc != 0) return bu_multi::nccl_fail(comm->ctx, rc, "ncclCommCount");
    if (out_ranks) *out_ranks = n;
    if (out_rank) *out_rank = me;
    return BU_OK;
}

bu_status bu_allgather_inplace(bu_comm* comm, void* d_full, size_t shard_bytes, void* stream)
{
    if (!comm || (shard_bytes && !d_full)) return BU_ERR_ARGUMENT;
    if (shard_bytes == 0) return BU_OK;
    // in place: the send buffer is this rank's shard inside the receive buffer (rccl.h, ncclAllGather "in-place" note)
    const uint8_t* send = static_cast<const uint8_t*>(d_full) + (size_t)comm->rank * shard_bytes;
    const int rc = bu_multi::rccl().AllGather(send, d_full, shard_bytes, /*ncclUint8*/ 1, comm->nccl, static_cast<hipStream_t>(stream));
    if (rc != 0) return bu_multi::nccl_fail(comm->ctx, rc, "ncclAllGather");
    return BU_OK;
}

// ---- direct peer pulls between processes (one process per GPU) ---------------------------------------
bu_status bu_ipc_export(bu_context* ctx, void* d_ptr, uint8_t handle[BU_IPC_HANDLE_BYTES])
{
    if (!ctx || !d_ptr || !handle) return BU_ERR_ARGUMENT;
    static_assert(sizeof(hipIpcMemHandle_t) <= BU_IPC_HANDLE_BYTES, "hipIpcMemHandle_t grew");
    BU_HIP(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    BU_HIP(ctx, hipIpcGetMemHandle(&h, d_ptr));
    memset(handle, 0, BU_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return BU_OK;
}

bu_status bu_ipc_open(bu_context* ctx, const uint8_t handle[BU_IPC_HANDLE_BYTES], void** d_peer)
{
    if (!ctx || !handle || !d_peer) return BU_ERR_ARGUMENT;
    *d_peer = nullptr;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    BU_HIP(ctx, hipIpcOpenMemHandle(d_peer, h, hipIpcMemLazyEnablePeerAccess));
    return BU_OK;
}

bu_status bu_ipc_close(bu_context* ctx, void* d_peer)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    if (!d_peer) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BU_HIP(ctx, hipIpcCloseMemHandle(d_peer));
    return BU_OK;
}

// streams of the concurrent pulls: extra_streams[0..6] of the context, created on first use
static bu_status bu_peer_streams(bu_context* ctx, int n) { return bu_ctx_streams(ctx, n < 8 ? n : 8); }

bu_status bu_allgather_peer(bu_context* ctx, void* d_full, void* const* d_peer_full, int world, int rank, size_t shard_bytes,
                            void* stream)
{
    if (!ctx || world < 1 || rank < 0 || rank >= world || (shard_bytes && (!d_full || !d_peer_full))) return BU_ERR_ARGUMENT;
    if (world == 1 || shard_bytes == 0) return BU_OK;
    for (int p = 0; p < world; p++)  // every pointer is checked before the first copy is queued
        if (p != rank && !d_peer_full[p]) return BU_ERR_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BU_HIP(ctx, hipSetDevice(ctx->device));
    const int lanes = world - 1 < 7 ? world - 1 : 7;
    bu_status st = bu_peer_streams(ctx, lanes);
    if (st) return st;
    // the pulls start once `stream` (which carries this rank's transcode) reaches this point and are joined back into it.
    // The caller guarantees the PEERS' shards are complete (a barrier between the transcode and this call).
    BuDrain drain(ctx);  // an error below must not leave pulls in flight on the side streams, un-joined
    BU_HIP(ctx, hipEventRecord(ctx->ev0, s));
    int k = 0;
    for (int p = 0; p < world; p++) {
        if (p == rank) continue;
        hipStream_t ps = ctx->extra_streams[k % lanes];
        if (k < lanes) BU_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev0, 0));
        const size_t ofs = (size_t)p * shard_bytes;
        BU_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t*>(d_full) + ofs, static_cast<const uint8_t*>(d_peer_full[p]) + ofs, shard_bytes,
                                   hipMemcpyDeviceToDevice, ps));
        k++;
    }
    for (int i = 0; i < lanes; i++) {
        BU_HIP(ctx, hipEventRecord(ctx->ev1, ctx->extra_streams[i]));
        BU_HIP(ctx, hipStreamWaitEvent(s, ctx->ev1, 0));
    }
    drain.armed = false;
    return BU_OK;
}

// ---- one process, n devices ---------------------------------------------------------------------------
bu_status bu_array_transcode_sharded(bu_context* const* ctxs, int n_ctx, bu_target target, const void* const* d_in_shard, size_t n_slices,
                                     size_t blocks_per_slice, void* const* d_full, int gather, uint64_t* first_bad_block)
{
    if (!ctxs || n_ctx < 1 || !d_in_shard || !d_full) return BU_ERR_ARGUMENT;
    const size_t bb = bu_target_block_bytes(target);
    if (bb == 0 || target == BU_TARGET_RGBA32) return BU_ERR_ARGUMENT;  // block-linear targets: shards are contiguous byte ranges
    for (int i = 0; i < n_ctx; i++)
        if (!ctxs[i]) return BU_ERR_ARGUMENT;
    std::vector<size_t> lo(n_ctx), hi(n_ctx);
    // every argument is checked before the first launch: nothing may be queued on devices 0..i-1 when device i is refused
    for (int i = 0; i < n_ctx; i++) {
        bu_multi::partition(n_slices, n_ctx, i, &lo[i], &hi[i]);
        if (hi[i] > lo[i] && blocks_per_slice && (!d_in_shard[i] || !d_full[i])) return BU_ERR_ARGUMENT;
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return BU_ERR_ARGUMENT;  // one status word and one stream per context: contexts must be distinct
    }
    // the call owns every context's status word and stream until it returns.  The locks are taken in ONE canonical order -- by
    // context address, whatever order the caller listed the contexts in -- so two threads that pass overlapping sets in different
    // orders cannot deadlock each other (A-then-B against B-then-A); released by the guard
    struct Locks {
        std::vector<bu_context*> c;
        size_t n = 0;
        ~Locks()
        {
            for (size_t i = n; i-- > 0;) c[i]->lock.unlock();
        }
    } locks;
    locks.c.assign(ctxs, ctxs + n_ctx);
    std::sort(locks.c.begin(), locks.c.end(), [](const bu_context* a, const bu_context* b) { return std::less<const bu_context*>()(a, b); });
    for (size_t i = 0; i < locks.c.size(); i++) {
        locks.c[i]->lock.lock();
        locks.n = i + 1;
    }
    // an early error return must not leave kernels or copies in flight on any of the contexts
    struct DrainAll {
        bu_context* const* c;
        int n;
        bool armed = true;
        ~DrainAll()
        {
            if (!armed) return;
            // callers read bu_last_error from the first context: bring the failing context's message there
            for (int i = 1; i < n && c[0]->err[0] == 0; i++)
                if (c[i]->err[0]) memcpy(c[0]->err, c[i]->err, sizeof(c[0]->err));
            for (int i = 0; i < n; i++) {
                (void)hipSetDevice(c[i]->device);
                if (c[i]->stream) (void)hipStreamSynchronize(c[i]->stream);
                for (hipStream_t es : c[i]->extra_streams)
                    if (es) (void)hipStreamSynchronize(es);
            }
        }
    } drain{ctxs, n_ctx};
    for (int i = 0; i < n_ctx; i++) ctxs[i]->err[0] = 0;
    // 1. every device transcodes its contiguous slice range into its own full buffer, at the range's final position: one exclusive launch per
    //    device, tile tickets on long walks (bu_range_begin: 0.77 of the roofline for a 2^25-block range against 0.71 with the fixed walk); every
    //    device is started before the first one is waited for
    std::vector<BuRangeJob> jobs((size_t)n_ctx);
    for (int i = 0; i < n_ctx; i++) {
        bu_context* c = ctxs[i];
        BU_HIP(c, hipSetDevice(c->device));
        const size_t nb = (hi[i] - lo[i]) * blocks_per_slice;
        if (nb == 0) continue;
        const bu_status st = bu_range_begin(c, target, d_in_shard[i], nb, static_cast<uint8_t*>(d_full[i]) + lo[i] * blocks_per_slice * bb, 1,
                                                      lo[i] * blocks_per_slice, &jobs[(size_t)i]);
        if (st) return st;
    }
    // 2. block status of every shard (the host-side join of the launches); the lowest failing block of the whole array is the sequential loop's error
    uint64_t best = BU_STATUS_WORD_CLEAR;
    for (int i = 0; i < n_ctx; i++) {
        bu_context* c = ctxs[i];
        BU_HIP(c, hipSetDevice(c->device));
        uint64_t word = BU_STATUS_WORD_CLEAR;
        const bu_status st = bu_range_end(c, jobs[(size_t)i], &word);
        if (st) return st;
        if (word < best) best = word;
    }
    bu_status st = bu_status_word_decode(best, first_bad_block);
    if (st) {  // (every stream that carried work was waited for above: nothing left to drain)
        drain.armed = false;
        return st;
    }
    // 3. all-gather by direct peer pulls: device i copies range j from device j's buffer, all pairs in flight together
    if (gather && n_ctx > 1) {
        for (int i = 0; i < n_ctx; i++) {
            bu_context* c = ctxs[i];
            BU_HIP(c, hipSetDevice(c->device));
            const int lanes = n_ctx - 1 < 7 ? n_ctx - 1 : 7;
            if ((st = bu_peer_streams(c, lanes))) return st;
            int k = 0;
            for (int j = 0; j < n_ctx; j++) {
                if (j == i || hi[j] == lo[j]) continue;
                if (c->device != ctxs[j]->device) {
                    int can = 0;
                    BU_HIP(c, hipDeviceCanAccessPeer(&can, c->device, ctxs[j]->device));
                    if (can) {
                        const hipError_t e = hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bu_fail(c, e, "hipDeviceEnablePeerAccess");
                        (void)hipGetLastError();
                    }
                }
                const size_t ofs = lo[j] * blocks_per_slice * bb, nbytes = (hi[j] - lo[j]) * blocks_per_slice * bb;
                BU_HIP(c, hipMemcpyPeerAsync(static_cast<uint8_t*>(d_full[i]) + ofs, c->device, static_cast<const uint8_t*>(d_full[j]) + ofs,
                                             ctxs[j]->device, nbytes, c->extra_streams[k % lanes]));
                k++;
            }
        }
        for (int i = 0; i < n_ctx; i++) {
            bu_context* c = ctxs[i];
            BU_HIP(c, hipSetDevice(c->device));
            const int lanes = n_ctx - 1 < 7 ? n_ctx - 1 : 7;
            for (int l = 0; l < lanes; l++) BU_HIP(c, hipStreamSynchronize(c->extra_streams[l]));
        }
    }
    drain.armed = false;
    return BU_OK;
}

bu_status bu_device_alloc(bu_context* ctx, size_t bytes, void** out_ptr)
{
    if (!ctx || !out_ptr) return BU_ERR_ARGUMENT;
    *out_ptr = nullptr;
    if (bytes == 0) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BU_HIP(ctx, hipMalloc(out_ptr, bytes));
    return BU_OK;
}

bu_status bu_device_free(bu_context* ctx, void* ptr)
{
    if (!ctx) return BU_ERR_ARGUMENT;
    if (!ptr) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BU_HIP(ctx, hipFree(ptr));
    return BU_OK;
}

bu_status bu_memcpy(bu_context* ctx, void* dst, const void* src, size_t bytes, int to_device)
{
    if (!ctx || (bytes && (!dst || !src))) return BU_ERR_ARGUMENT;
    if (bytes == 0) return BU_OK;
    BU_HIP(ctx, hipSetDevice(ctx->device));
    BU_HIP(ctx, hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
    BU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BU_OK;
}

}  // extern "C"
