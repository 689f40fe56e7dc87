/*
 * basisu_hip.h -- C ABI of the MI355X (gfx950) block-transcode path.
 *
 * Drop-in boundary for the per-4x4-block hot path of the Rust crate JakubValtar/basisu_rs.  The
 * reference has no FFI layer of its own; each entry point below names the reference item it
 * replaces (file:line relative to the reference checkout) and is what a Rust `extern "C"` block
 * in that crate would bind (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - plain pointers and sizes only; no ownership is transferred; outputs are caller-allocated.
 *   - `*_device` entry points take device pointers and a hipStream_t (passed as void*), never
 *     synchronise, never allocate, and are graph-capturable.  The other entry points take host
 *     pointers -- pageable memory is staged through the context's device buffers, page-locked memory
 *     (bu_host_alloc) is read and written by the kernels directly -- and return after the result is in `out`.
 *   - errors of the reference (`Result<_, String>`, lib.rs:26-27) become bu_status codes;
 *     bu_status_string() returns the reference's message text for the hot-path errors.
 *   - "first failing block aborts the call" (uastc.rs:157-165): on a block error the returned status
 *     and *first_bad_block describe the LOWEST failing block index, as the reference's sequential
 *     loop would; the output buffer contents are then unspecified (the reference drops its Vec).
 *   - the library has no CPU fallback: without a usable HIP device every call fails with
 *     BU_ERR_NO_DEVICE / BU_ERR_HIP.
 */
#ifndef BASISU_HIP_H
#define BASISU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* uastc::TargetTextureFormat (src/uastc.rs:41-47) + the RGBA32 unpack (src/uastc.rs:89-110) */
typedef enum bu_target {
    BU_TARGET_ASTC = 0,   /* 16 B/block  target_formats/astc.rs:8   */
    BU_TARGET_BC7 = 1,    /* 16 B/block  target_formats/bc7.rs:9    */
    BU_TARGET_ETC1 = 2,   /*  8 B/block  target_formats/etc.rs:11   */
    BU_TARGET_ETC2 = 3,   /* 16 B/block  target_formats/etc.rs:19   */
    BU_TARGET_RGBA32 = 4  /* 64 B/block  uastc.rs:237 (row-major image for the slice entry points) */
} bu_target;

typedef enum bu_status {
    BU_OK = 0,
    BU_ERR_INVALID_MODE = 1,    /* "invalid mode index"                                    uastc.rs:336 */
    BU_ERR_INVALID_PATTERN = 2, /* "block pattern is not valid"                            uastc.rs:364 */
    BU_ERR_LENGTH = 3,          /* "data length is not divisible by UASTC block size (16)" uastc.rs:56  */
    BU_ERR_OUTPUT_SIZE = 4,     /* caller's output buffer is too small (the reference allocates its own) */
    BU_ERR_ARGUMENT = 5,        /* null pointer / unknown target / blocks_per_row == 0 */
    BU_ERR_INDEX_RANGE = 6,     /* ETC1S endpoint/selector index outside the codebook (reference: assert!, basis_lz/mod.rs:443-445) */
    BU_ERR_NO_DEVICE = 7,       /* no HIP device / not gfx950 */
    BU_ERR_HIP = 8,             /* a HIP runtime call failed; see bu_last_error() */
    /* container / BasisLZ (host side) */
    BU_ERR_SIG = 9,               /* "Sig mismatch, not a Basis Universal file"            basis.rs:309 */
    BU_ERR_HEADER_TRUNCATED = 10, /* "Expected at least 77 byte header, got N bytes"       basis.rs:313 */
    BU_ERR_HEADER_SIZE = 11,      /* "File specified unexpected header size, ..."          basis.rs:323 */
    BU_ERR_HEADER_CRC = 12,       /* "Header CRC16 failed"                                 basis.rs:332 */
    BU_ERR_DATA_CRC = 13,         /* "Data CRC16 failed"                                   basis.rs:12  */
    BU_ERR_TEX_FORMAT = 14,       /* "Unknown texture format"                              basis.rs:404 */
    BU_ERR_SLICE_DESC = 15,       /* "Expected 23 byte slice desc at pos ..."              basis.rs:350 */
    BU_ERR_ALPHA_SLICES = 16,     /* odd slice count / "Expected slice with alpha" / size mismatch  basis.rs:19,29,34 */
    BU_ERR_UNSUPPORTED = 17,      /* format/target pair the reference answers with unimplemented!() basis.rs:88,141,171,200,229,258 */
    BU_ERR_BASISLZ = 18,          /* any Err of basis_lz (bad Huffman table / code, global or hybrid selector codebook) */
    BU_ERR_BOUNDS = 19            /* offsets outside the file, or a condition the reference asserts/panics on */
} bu_status;

typedef struct bu_context bu_context;

/* Create a context on HIP device `device`: uploads the lookup tables, creates a stream and the
 * staging buffers used by the host-pointer entry points.  (Reference: uastc::Decoder::new,
 * uastc.rs:79-83, is stateless; the context only holds device resources.) */
bu_status bu_context_create(int device, bu_context** out_ctx);
void bu_context_destroy(bu_context* ctx);
/* text of the reference error for hot-path statuses, a short description otherwise */
const char* bu_status_string(bu_status st);
/* detail of the last BU_ERR_HIP on this context (never NULL) */
const char* bu_last_error(const bu_context* ctx);
/* bytes per output block: 16, 16, 8, 16, 64 (uastc.rs:13-17); 0 for an unknown target */
size_t bu_target_block_bytes(bu_target target);

/* ---- launch policy -------------------------------------------------------------------------------
 * How much of the chip ONE large launch of the slice-level transcode takes (read by bu_uastc_transcode_device / _batch_device -- the
 * entry points that take a stream -- at the moment they enqueue; the host-pointer entry points issue one launch at a time and always use
 * the exclusive shapes; results never depend on it).
 * The reference's slice loop (uastc.rs:112-165, basis.rs:246-257) runs slices one after another; on the GPU a launch over one
 * 4096 x 4096 slice spends its first ~3.4 us waiting for HBM with the ALUs idle and the rest computing with HBM idle, and launches
 * queued on ONE stream never overlap (the queue drains between two dispatches).  A caller with several independent slices gets
 * both resources busy by issuing them round-robin on 2-4 of the context's streams (bu_context_stream):
 *   BU_LAUNCH_EXCLUSIVE            a launch is shaped to fill the chip by itself: lowest latency for a single slice (UASTC->BC7,
 *                                  2^20 blocks: 8.4 us); 6.2-6.7 us per slice with 2-4 streams (head / tail overlap only)
 *   BU_LAUNCH_SHARED               a launch keeps at most half of every CU's wave slots, registers and LDS, so launches from different
 *                                  streams run side by side on each CU: 5.45-5.6 us per slice with 4 streams (BC7; ASTC 5.4), ETC1 17.5 -> 12.1,
 *                                  ETC2 20.6 -> 15.0, RGBA32 14.9 -> 13.1; alone on the chip such a launch is 15-40 % SLOWER than an exclusive one.
 *   BU_LAUNCH_AUTO (default)       chosen PER CALL by how many OTHER streams of the context have work that has not completed (enqueued there within
 *                                  the last 40 us of host time; if none was, one hipStreamQuery per stream ever used): none -> exclusive (a lone slice:
 *                                  8.4 us; also every launch on a stream of the caller's own, which the library cannot see beside); three or more ->
 *                                  shared (5.5-5.7); one or two -> the shared kernels on one-tile workgroups dealt by the hardware (BC7 / ASTC: 6.1 / 5.7
 *                                  us per slice with two / three in flight, where exclusive gives 6.85 / 6.3 and shared 6.95 / 6.1).  The first launches
 *                                  of a pipeline go out in the shapes of a shallower one, the rest shared.
 * Figures: profiles/r05_ab_bc7_two_launches_in_flight.txt, r05_ab_wave_priorities_with_launches_in_flight.txt, r05_ab_etc_shared_shapes_x_streams.txt,
 * r06_auto_policy_matrix.txt.  Small launches (at most one 1024-block tile per CU) are the same under all three. */
typedef enum bu_launch_policy { BU_LAUNCH_EXCLUSIVE = 0, BU_LAUNCH_SHARED = 1, BU_LAUNCH_AUTO = 2 } bu_launch_policy;
bu_status bu_context_set_launch_policy(bu_context* ctx, bu_launch_policy policy);
bu_status bu_context_get_launch_policy(const bu_context* ctx, bu_launch_policy* out_policy);
/* The context's own streams, index 0..7 (hipStream_t, created on first use in groups of four, destroyed with the context): what a caller
 * that wants several launches in flight issues them on.  Launches only overlap when their streams sit on DIFFERENT hardware queues, and
 * the HIP runtime multiplexes all ordinary streams of a process over GPU_MAX_HW_QUEUES (default 4) queues per priority level -- two
 * streams that share one run their kernels strictly one after the other.  The library does not depend on that variable: a new group
 * is created as ordinary non-blocking streams, checked (a 200 us sleeping wave on every stream behind a common event: streams with a queue each
 * sleep together), and if two streams share a queue the group is re-created with full CU masks (hipExtStreamCreateWithCUMask: a hardware
 * queue of its own per stream whatever the pool size) and checked again, before anybody sees a handle (~1 ms, once per group).  CU-mask streams
 * have default-stream semantics towards the process's NULL stream (they wait for work queued there and it waits for them); a process that keeps
 * GPU_MAX_HW_QUEUES >= its number of streams (bench.py: 8, 16 beside an RCCL communicator; set before the first HIP call) keeps ordinary
 * non-blocking streams.  bu_context_query_in_flight says which kind the context got and whether the check passed. */
bu_status bu_context_stream(bu_context* ctx, int index, void** out_stream);
/* waits until everything enqueued on the context's own streams (bu_context_stream) and on its internal stream has completed: the
 * host-side join for a caller without a HIP binding of its own (examples/slices_in_flight.c) */
bu_status bu_context_synchronize(bu_context* ctx);
/* Do the context's streams 0..n_streams-1 (1..8) really run side by side in THIS process, NOW?  The creation-time check repeated on demand
 * (other libraries' streams -- an RCCL communicator's -- created later take queues of the pool too): *out_max_sharing = the largest number
 * of the probed streams found on one queue -- 1: every stream has its own.  Blocks for about 0.4 ms. */
bu_status bu_context_probe_streams(bu_context* ctx, int n_streams, int* out_max_sharing);
/* What a pipeline over the context's streams 0..n_streams-1 gets in this process, by the creation-time check (creates the streams if need be):
 *   *out_effective_streams  how many launches the streams really keep in flight: n_streams when every stream has a hardware queue of its own,
 *                           n_streams / (streams per queue) when neither kind of stream got one (not seen on ROCm 7.2; BU_STREAM_MODE=plain in the
 *                           environment forces ordinary streams and reproduces it).  bu_uastc_transcode_batch_in_flight then degrades by itself:
 *                           with <= 2 effective streams BC7 / ASTC / RGBA32 batches go out as the stream-ordered batch launch on stream 0 (6.4 us
 *                           per 2^20-block slice from eight slices on, against 7.3 for a pipeline two deep); results are the same either way.
 *   *out_stream_mode        BU_STREAM_QUEUE_POOL: ordinary non-blocking streams on the runtime's queue pool; BU_STREAM_QUEUE_CU_MASK: CU-mask streams
 * Either pointer may be NULL. */
#define BU_STREAM_QUEUE_POOL 0
#define BU_STREAM_QUEUE_CU_MASK 1
bu_status bu_context_query_in_flight(bu_context* ctx, int n_streams, int* out_effective_streams, int* out_stream_mode);

/* ---- UASTC slice level, host pointers ------------------------------------------------------- */

/* uastc::Decoder::transcode / _transcode_into (uastc.rs:112-146): n = in_bytes/16 blocks in
 * raster order -> n * bu_target_block_bytes(target) bytes, block i at i * block_bytes.
 * target must be ASTC, BC7, ETC1 or ETC2. */
bu_status bu_uastc_transcode(bu_context* ctx, bu_target target, const uint8_t* in, size_t in_bytes, uint8_t* out,
                             size_t out_bytes, uint64_t* first_bad_block);

/* uastc::Decoder::decode_to_rgba (uastc.rs:89-110): row-major RGBA8 image, 4*blocks_per_row pixels
 * per row, bytes R,G,B,A; out_bytes >= 64 * n_blocks. */
bu_status bu_uastc_decode_to_rgba(bu_context* ctx, const uint8_t* in, size_t in_bytes, size_t blocks_per_row,
                                  uint8_t* out, size_t out_bytes, uint64_t* first_bad_block);

/* ---- per-block API of lib.rs:29-53 -------------------------------------------------------------
 * One 16-byte block per call, the way the reference's callers and its benchmark use them (benches/benchmark.rs:66-98:
 * 32 blocks x 1000 calls).  By default a call runs on the calling thread: the library compiles its own per-block
 * code (the mode-templated front-end and packers the gfx950 kernels are built from) for the host as well, and a block is
 * not worth an upload + launch + download (well under 1 us against tens of us).  A context is still required -- the
 * library has no CPU mode -- and is only read.  bu_block_api_on_device(ctx, 1) sends these calls through a one-block
 * kernel launch instead (same results; kept so that the device path of a single block stays testable), 0 switches back. */
bu_status bu_block_api_on_device(bu_context* ctx, int enable);
bu_status bu_unpack_uastc_block_to_rgba(bu_context* ctx, const uint8_t in[16], uint32_t out[16]); /* lib.rs:29 */
bu_status bu_transcode_uastc_block_to_astc(bu_context* ctx, const uint8_t in[16], uint8_t out[16]); /* lib.rs:33 */
bu_status bu_transcode_uastc_block_to_bc7(bu_context* ctx, const uint8_t in[16], uint8_t out[16]);  /* lib.rs:39 */
bu_status bu_transcode_uastc_block_to_etc1(bu_context* ctx, const uint8_t in[16], uint8_t out[8]);  /* lib.rs:43 */
bu_status bu_transcode_uastc_block_to_etc2(bu_context* ctx, const uint8_t in[16], uint8_t out[16]); /* lib.rs:49 */

/* ---- UASTC slice level, device pointers, asynchronous ----------------------------------------- */

/* Same work as the two slice entry points above on memory already resident in HBM.
 *   d_in            n_blocks * 16 bytes, 16-byte aligned
 *   d_out           n_blocks * block_bytes, 16-byte aligned (8 for ETC1)
 *   blocks_per_row  the width of the slice's block grid; 0 = unknown (block-linear targets only).  REQUIRED for BU_TARGET_RGBA32
 *                   (image pitch).  For the block-linear targets it never changes a byte of the result; a multiple of 64 (at
 *                   least 128) over whole rows of 64 x 16-block tiles lets the kernels sort rectangles of the image instead of
 *                   strips of 1024 consecutive blocks (faster on real, mode-coherent textures).  For RGBA32 n_blocks must be a whole number of
 *                   block rows (n_blocks % blocks_per_row == 0, else BU_ERR_ARGUMENT): the four pixel rows of a
 *                   block row are stored at the full image pitch, so a ragged last row would be written past
 *                   64 * n_blocks bytes (uastc.rs:95 sizes the image the same way and would panic there).
 *   d_status        optional device uint64_t initialised with bu_status_word_reset(): receives
 *                   min over failing blocks of (block_index << 8 | status); may be shared by several
 *                   launches whose block indices are disjoint (`block_index_base` is added)
 *   stream          hipStream_t
 * Returns launch/argument errors only; block errors arrive through d_status.
 * This call, the batch call below and bu_status_word_reset only enqueue work on `stream` (nothing is allocated, copied from the host or
 * synchronised behind them), so they may be recorded into a HIP graph by stream capture and replayed (tests/test_gpu_round4.py). */
bu_status bu_uastc_transcode_device(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out,
                                    size_t blocks_per_row, uint64_t block_index_base, uint64_t* d_status, void* stream);
/* A loop over independent slices (the per-slice loops of basis.rs:246-257) as ONE call: slice i = n_blocks[i] blocks at d_in[i]
 * -> d_out[i], block indices numbered from index_base[i] (NULL: slices numbered back to back from 0).  The slices must not
 * depend on each other (no output aliasing an input); the call starts when `stream` reaches it and `stream` continues when
 * every slice is done.  Slices that are contiguous in memory, in order, are merged into runs; one run is the plain launch, and
 * up to 96 runs at unrelated addresses are ONE kernel launch with the run table in its kernel arguments (nothing is allocated,
 * copied or freed behind the call; longer batches go out as one launch per 96 runs) -- 64 slices of 65 536 blocks in separate
 * allocations take ~46 us call + synchronize where 64 launches take 290 us on one stream and 230-260 us on two to eight: a loop
 * of launches is bound by the host's ~4 us per launch, not by the GPU.  A batch of LARGE slices at unrelated addresses (BC7, ASTC, RGBA32)
 * is one launch of a persistent grid that walks all the runs' tiles with the next tile's loads in flight.  Since round 6 that launch is as
 * fast as a pipeline of launches and is what bench.py's headline times: the run table is copied to LDS once per workgroup, every BC7 / ASTC
 * run that is whole 64 x 16-block rectangles of a power-of-two grid (blocks_per_row if it is one, else a virtual pitch) is tiled that way,
 * and from 16 tiles per workgroup on the workgroups draw their tiles by ticket -- SIXTY-FOUR 4096 x 4096 slices in 64 separate allocations:
 * ONE launch, 351-357 us = 5.5-5.6 us per slice = 0.75-0.765 of the HBM roofline, by HIP events and by rocprofv3's kernel average alike
 * (eight slices: 7.3 us each -- short walks keep their fixed shares; one slice: what bu_uastc_transcode_device costs).
 * RGBA32: every slice uses the same blocks_per_row. */
bu_status bu_uastc_transcode_batch_device(bu_context* ctx, bu_target target, size_t n_slices, const void* const* d_in,
                                          const size_t* n_blocks, void* const* d_out, size_t blocks_per_row,
                                          const uint64_t* index_base, uint64_t* d_status, void* stream);
/* The same loop over independent slices at the rate of a PIPELINE of launches -- what bench.py's headline measures, as one call.
 * Launches queued on one stream never overlap, and a launch over a slice waits for HBM with the ALUs idle and then computes with HBM idle;
 * launches on several streams fill each other's gaps (UASTC -> BC7 over 4096 x 4096 slices: 8.4 us per slice one at a time, 5.6-6.0 us in a
 * pipeline).  The call merges the slices into runs as above, groups the runs into multi-run launches (one launch per group, the run table in its
 * kernel arguments; BC7 / ASTC: up to 2^23 blocks per launch, so that eight 4096 x 4096 atlases in separate allocations are one enqueue; ETC1 / ETC2 /
 * RGBA32: small runs up to 2^20 blocks, a larger run is a launch of its own -- profiles/r06_in_flight_group_size.txt), cuts runs of more than 2^23 blocks and the largest runs of a batch that would make fewer launches than streams into equal pieces (a 512-slice
 * texture array in one allocation becomes n_streams launches: 0.76-0.78 of the HBM roofline against 0.70 as one launch), and issues launch j on
 * the context's own stream j % n_streams (1..8; bu_context_stream; FOUR is the depth to use -- the chip runs four queues' dispatches side by side,
 * and with five to eight streams BC7 / ASTC fall from 5.5 to 7-10 us per 2^20-block slice: profiles/r06_more_than_four_in_flight.txt) under the
 * shared launch policy, whatever the context's policy is.
 * It only ENQUEUES: bu_context_synchronize(ctx), or synchronising those streams, waits for the results, and nothing the caller enqueued on
 * other streams is waited for -- inputs and the status word must be ready before the call (bu_status_word_reset + a synchronise).  The streams
 * need a hardware queue each; the library sees to that itself and reports it (bu_context_stream, bu_context_query_in_flight).  A batch of 64
 * launches or more is enqueued by one host thread per stream (the calling thread among them; all joined before the call returns), so that
 * the host's ~4.7 us per enqueue stays off the pipeline's critical path.  Results, status reporting and argument rules are those of the call above. */
bu_status bu_uastc_transcode_batch_in_flight(bu_context* ctx, bu_target target, size_t n_slices, const void* const* d_in,
                                             const size_t* n_blocks, void* const* d_out, size_t blocks_per_row,
                                             const uint64_t* index_base, uint64_t* d_status, int n_streams);

/* bu_uastc_transcode_device that WAITS: the per-slice loop of basis.rs:246-257 over one contiguous device-resident range (a slice, or the
 * slices a rank owns of a texture array: basis.rs:531-552), as a blocking call.  ONE launch on the context's internal stream in the exclusive
 * shape; from 16 tiles per workgroup on (BC7 / ASTC: 2^24 blocks, RGBA32: 2^23) the workgroups draw their tiles by ticket instead of walking
 * fixed shares, so the launch ends when the tiles do and not when the slowest share does: a 2^25-block array 174 us = 0.77 of the HBM
 * roofline (fixed walk 188.5; profiles/r06_ab_tile_tickets.txt) -- the rate round 5 needed four launches in flight for, here for a single array.
 * The status word lives in page-locked memory of the context: no reset launch, no copy back.  *out_status_word (optional) =
 * BU_STATUS_WORD_CLEAR or the LOWEST (block_index_base + block) << 8 | status over the range -- bu_status_word_decode turns it into the
 * reference's error; ranks of a multi-process job reduce it with MIN before anyone raises (basisu_rs_amd/sharded.py).  The call takes the
 * context's lock (one blocking call per context at a time).  bu_array_transcode_sharded runs every device's range through the same code.
 * (Tile tickets are drawn from counters the CONTEXT owns, one set per STREAM -- launches of one stream run one after the other, so a set never
 * serves two launches at once: the context's own streams have fixed sets, a stream of the caller's gets the next free one of 32 on its first large
 * launch, so bu_uastc_transcode_device draws tickets on any stream.  Excluded: hipStreamPerThread, a stream that is being captured into a graph, and
 * a context's 33rd caller stream -- those keep the fixed walk.) */
bu_status bu_uastc_transcode_device_sync(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out,
                                         size_t blocks_per_row, uint64_t block_index_base, uint64_t* out_status_word);

/* value a status word must hold before the launches that report into it (all ones) */
#define BU_STATUS_WORD_CLEAR 0xFFFFFFFFFFFFFFFFull
/* enqueue a reset of *d_status on `stream` */
bu_status bu_status_word_reset(bu_context* ctx, uint64_t* d_status, void* stream);
/* decode a status word copied back to the host */
bu_status bu_status_word_decode(uint64_t word, uint64_t* first_bad_block);

/* ---- page-locked host buffers ---------------------------------------------------------------------
 * The reference's entry points take and return ordinary slices (`&[u8]` in, `Vec<u8>` out, uastc.rs:112);
 * over PCIe the two copies are the whole cost of a call.  When BOTH buffers handed to
 * bu_uastc_transcode / bu_uastc_decode_to_rgba come from bu_host_alloc() (or were page-locked by the
 * caller through the HIP runtime), the kernels read the slice and write the result directly over PCIe --
 * no staging copies, reads and writes overlapped.  Ordinary pageable memory keeps working through
 * device staging buffers. */
bu_status bu_host_alloc(bu_context* ctx, size_t bytes, void** out_ptr);
bu_status bu_host_free(bu_context* ctx, void* ptr);

/* ---- ETC1S block back-end (basis_lz/mod.rs:97-186) ---------------------------------------------
 * The serial BasisLZ entropy decode stays on the host and produces, per block in raster order,
 *   idx[i] = endpoint_index | selector_index << 16          (DecodedBlock, basis_lz/mod.rs:43-48)
 * and the two codebooks
 *   endpoints[k] = r5 | g5 << 8 | b5 << 16 | inten << 24     (Endpoint, basis_lz/mod.rs:518-522)
 *   selectors[k] = {rows[4] (2 bits/texel, x = 0 low), etc1_bytes[4]}  8 bytes (etc::Selector, etc.rs:343-350)
 * bu_etc1s_selector_from_rows() builds a selector entry from its 4 raw row bytes exactly as
 * Selector::set_selector does (etc.rs:363-393). */
void bu_etc1s_selector_from_rows(const uint8_t rows[4], uint8_t out_entry[8]);

/* Decoder::transcode_to_etc1 back-end, closure block_to_etc1 (basis_lz/mod.rs:163-181): 8 B per block.
 * Any 4-byte-aligned d_idx and 8-byte-aligned d_out work; from 2^19 blocks the codebooks are staged in LDS (when both fit
 * 152 KiB) and an 8-byte-aligned d_idx with a 16-byte-aligned d_out takes the four-blocks-per-lane path (2^22 blocks: 13 us). */
bu_status bu_etc1s_transcode_etc1_device(bu_context* ctx, const uint32_t* d_idx, size_t n_blocks,
                                         const uint32_t* d_endpoints, uint32_t n_endpoints, const void* d_selectors,
                                         uint32_t n_selectors, void* d_out, uint64_t* d_status, void* stream);
/* Decoder::decode_to_rgba back-end, closure block_to_rgba (basis_lz/mod.rs:122-146): row-major RGBA8
 * image of (4*nbx) x (4*nby) pixels; d_alpha_idx (may be NULL) is the paired alpha slice whose
 * green channel becomes .a (basis_lz/mod.rs:139-143). */
bu_status bu_etc1s_decode_rgba_device(bu_context* ctx, const uint32_t* d_idx, const uint32_t* d_alpha_idx, size_t nbx,
                                      size_t nby, const uint32_t* d_endpoints, uint32_t n_endpoints,
                                      const void* d_selectors, uint32_t n_selectors, void* d_out, uint64_t* d_status,
                                      void* stream);
/* host-pointer forms (stage, launch, copy back, synchronise) */
bu_status bu_etc1s_transcode_etc1(bu_context* ctx, const uint32_t* idx, size_t n_blocks, const uint32_t* endpoints,
                                  uint32_t n_endpoints, const uint8_t* selectors, uint32_t n_selectors, uint8_t* out,
                                  size_t out_bytes, uint64_t* first_bad_block);
bu_status bu_etc1s_decode_rgba(bu_context* ctx, const uint32_t* idx, const uint32_t* alpha_idx, size_t nbx, size_t nby,
                               const uint32_t* endpoints, uint32_t n_endpoints, const uint8_t* selectors,
                               uint32_t n_selectors, uint8_t* out, size_t out_bytes, uint64_t* first_bad_block);

/* ---- whole-file level: the crate's public read_to_* API (src/lib.rs:20-22, src/basis.rs) -------------
 * Host side in C++ (container parse, CRC-16, BasisLZ entropy decode -- byte/symbol serial, stays on the CPU),
 * block work on the GPU through the entry points above. */

/* basis::Header (basis.rs:417-454), same 26 public fields */
typedef struct bu_basis_header {
    uint16_t sig, ver, header_size, header_crc16;
    uint32_t data_size;
    uint16_t data_crc16;
    uint32_t total_slices, total_images; /* u24 in the file */
    uint8_t tex_format;                  /* 0 ETC1S, 1 UASTC4x4 (basis.rs:389-407) */
    uint16_t flags;                      /* 1 ETC1S, 2 YFlipped, 4 HasAlphaSlices (basis.rs:409-415) */
    uint8_t tex_type;                    /* 3 = video frames (basis.rs:374-381) */
    uint32_t us_per_frame, reserved, userdata0, userdata1;
    uint16_t total_endpoints;
    uint32_t endpoint_cb_file_ofs, endpoint_cb_file_size;
    uint16_t total_selectors;
    uint32_t selector_cb_file_ofs, selector_cb_file_size;
    uint32_t tables_file_ofs, tables_file_size, slice_desc_file_ofs, extended_file_ofs, extended_file_size;
} bu_basis_header;

/* basis::SliceDesc (basis.rs:519-535) */
typedef struct bu_slice_desc {
    uint32_t image_index; /* u24 */
    uint8_t level_index, flags; /* flags: 1 HasAlpha, 2 FrameIsIFrame */
    uint16_t orig_width, orig_height, num_blocks_x, num_blocks_y;
    uint32_t file_ofs, file_size;
    uint16_t slice_data_crc16;
} bu_slice_desc;

/* Image<u8> (lib.rs:63-68): data = out + offset, `size` bytes; stride as the reference reports it */
typedef struct bu_image {
    uint32_t w, h, stride, reserved;
    uint64_t offset, size;
} bu_image;

typedef enum bu_read_target {
    BU_READ_RGBA = 0, /* read_to_rgba  basis.rs:8   */
    BU_READ_ETC1 = 1, /* read_to_etc1  basis.rs:92  */
    BU_READ_ETC2 = 2, /* read_to_etc2  basis.rs:145 */
    BU_READ_UASTC = 3, /* read_to_uastc basis.rs:175 */
    BU_READ_ASTC = 4, /* read_to_astc  basis.rs:204 */
    BU_READ_BC7 = 5   /* read_to_bc7   basis.rs:233 */
} bu_read_target;

/* read_header (basis.rs:307-336): signature, size, header_size == 77, header CRC */
bu_status bu_basis_read_header(const uint8_t* file, size_t len, bu_basis_header* out);
/* read_slice_descs (basis.rs:343-362) */
bu_status bu_basis_read_slice_descs(const uint8_t* file, size_t len, const bu_basis_header* header, bu_slice_desc* out,
                                    size_t max_descs, size_t* n_descs);
/* crc16 (basis.rs:364-372): CRC-16/GENIBUS continued from `crc` */
uint16_t bu_basis_crc16(const uint8_t* data, size_t len, uint16_t crc);
/* number of images and output bytes bu_read_to() will produce for this file (no GPU work) */
bu_status bu_read_query(bu_read_target target, const uint8_t* file, size_t len, size_t* n_images, size_t* out_bytes);
/* read_to_{rgba,etc1,etc2,uastc,astc,bc7}: every slice of the file (every colour/alpha pair for RGBA from an
 * ETC1S file with alpha) becomes one bu_image whose bytes are written to `out`.  header_out may be NULL.
 * Reference quirks are reproduced: the data CRC covers bytes[77..EOF]; total_selectors sizes BOTH ETC1S
 * codebooks (basis.rs:289-291); ETC1S RGBA images report stride 16*orig_width (basis.rs:46,64).
 * On an error return the contents of `out` are undefined (parts of it may already have been written). */
bu_status bu_read_to(bu_context* ctx, bu_read_target target, const uint8_t* file, size_t len, bu_basis_header* header_out,
                     bu_image* images, size_t max_images, size_t* n_images, uint8_t* out, size_t out_bytes);
/* Host-only BasisLZ decode of an ETC1S file (basis_lz/mod.rs:64-95, 188-458): the two codebooks in the layouts
 * the bu_etc1s_* entry points take, and the per-block indices of slice `slice_index`.  Any pointer may be NULL. */
bu_status bu_basislz_decode(const uint8_t* file, size_t len, uint32_t slice_index, uint32_t* endpoints_out,
                            uint8_t* selectors_out, uint32_t* idx_out);
/* Writer: assemble a UASTC `.basis` file (header + slice descs + slice data, CRCs filled).  descs[i] supplies
 * image_index, level_index, flags, orig_*, num_blocks_*; file_ofs/file_size/crc are computed.  With out == NULL
 * only *out_len is set. */
bu_status bu_basis_write_uastc(const bu_slice_desc* descs, const uint8_t* const* slice_data, const size_t* slice_bytes,
                               size_t n_slices, uint16_t header_flags, uint8_t tex_type, uint8_t* out, size_t out_cap,
                               size_t* out_len);

/* ---- multi-GPU: shards of one texture array -----------------------------------------------------
 * Slices of a texture array are independent contiguous byte ranges (SliceDesc.file_ofs/file_size, basis.rs:531-552;
 * the reference walks them in the loop of read_to_bc7, basis.rs:246-257).  Rank / device g of P owns the slices
 * [g*n/P, (g+1)*n/P): it transcodes them with bu_uastc_transcode_device straight into its copy of the full output
 * buffer, at the range's final offset -- no data-path collective.  One all-gather then leaves the whole array on every
 * device.  Two transports:
 *   RCCL        ncclAllGather in place (send = own shard inside the receive buffer), one process per GPU
 *   peer pulls  world-1 concurrent device-to-device copies per rank (xGMI is point to point: every link carries
 *               one copy, no ring hops); between processes through HIP IPC handles, inside one process directly */
typedef struct bu_comm bu_comm;
#define BU_COMM_ID_BYTES 128
#define BU_IPC_HANDLE_BYTES 64
/* rank 0 creates the id (ncclGetUniqueId) and hands it to every rank by any out-of-band channel.
 * BU_ERR_UNSUPPORTED when no RCCL library can be resolved in this process. */
bu_status bu_comm_unique_id(uint8_t id[BU_COMM_ID_BYTES]);
/* collective over all `world` ranks (ncclCommInitRank) on ctx's device */
bu_status bu_comm_create(bu_context* ctx, int world, int rank, const uint8_t id[BU_COMM_ID_BYTES], bu_comm** out_comm);
void bu_comm_destroy(bu_comm* comm);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank): evidence in a benchmark record that RCCL saw
 * `world` ranks.  BU_ERR_UNSUPPORTED if the resolved RCCL lacks the two entry points. */
bu_status bu_comm_query(bu_comm* comm, int* out_ranks, int* out_rank);
/* d_full: world * shard_bytes bytes on this rank's device, this rank's shard already at rank * shard_bytes (written by
 * work enqueued on `stream` before this call).  Asynchronous on `stream`; afterwards every rank holds every shard. */
bu_status bu_allgather_inplace(bu_comm* comm, void* d_full, size_t shard_bytes, void* stream);
/* HIP IPC view of a peer process's full buffer (export on the owner, open on every other rank, close before the
 * owner frees it) */
bu_status bu_ipc_export(bu_context* ctx, void* d_ptr, uint8_t handle[BU_IPC_HANDLE_BYTES]);
bu_status bu_ipc_open(bu_context* ctx, const uint8_t handle[BU_IPC_HANDLE_BYTES], void** d_peer);
bu_status bu_ipc_close(bu_context* ctx, void* d_peer);
/* Pull every other rank's shard out of its full buffer: copies [p*shard_bytes, (p+1)*shard_bytes) of d_peer_full[p]
 * into d_full at the same offset for every p != rank (d_peer_full[rank] is ignored), all copies in flight together on
 * context-owned streams that wait for `stream` and are joined back into it.  The caller orders this call after the
 * PEERS' transcodes (a barrier across ranks); asynchronous on `stream`. */
bu_status bu_allgather_peer(bu_context* ctx, void* d_full, void* const* d_peer_full, int world, int rank, size_t shard_bytes,
                            void* stream);
/* One process driving n_ctx devices (one context each; several contexts on one device are allowed and behave as
 * virtual ranks).  d_in_shard[i]: device i's slice range only, resident on device i; d_full[i]: n_slices *
 * blocks_per_slice * block_bytes bytes on device i.  Transcodes every range in place, then (gather != 0) every device
 * pulls the other ranges from its peers.  Block-linear targets only.  Synchronous; on a block error returns the
 * status and index (array-wide) of the LOWEST failing block, like the sequential loop over slices.
 * Threading: the call locks every context it is given for its whole duration (other calls on those contexts wait), in one
 * canonical order whatever the order of `ctxs`, so concurrent calls over overlapping context sets cannot deadlock. */
bu_status bu_array_transcode_sharded(bu_context* const* ctxs, int n_ctx, bu_target target, const void* const* d_in_shard,
                                     size_t n_slices, size_t blocks_per_slice, void* const* d_full, int gather,
                                     uint64_t* first_bad_block);
/* device memory on ctx's device for callers without a HIP binding of their own (hipMalloc / hipFree / synchronous copy) */
bu_status bu_device_alloc(bu_context* ctx, size_t bytes, void** out_ptr);
bu_status bu_device_free(bu_context* ctx, void* ptr);
bu_status bu_memcpy(bu_context* ctx, void* dst, const void* src, size_t bytes, int to_device);

/* ---- measurement helpers (bench.py) ------------------------------------------------------------
 * uint4 -> uint4 copy kernel, 16 B in / 16 B out like the transcoders, in the fastest shape measured for a 4096^2 atlas
 * (512 threads x 4 elements per thread, nontemporal: profiles/r02_copy_shapes_32MiB.txt): the practical HBM ceiling the
 * roofline fraction is reported next to. */
bu_status bu_copy_ceiling_device(bu_context* ctx, const void* d_in, size_t n_blocks, void* d_out, void* stream);
/* Times `launches` back-to-back launches of one transcode with hipEvents recorded on `stream`
 * around the whole batch (the stream the kernels are launched on).  d_in/d_out are arrays of
 * `n_buffers` device pointers rotated round-robin so that consecutive launches touch different
 * HBM (cold-cache protocol): launch i uses buffer (first_buffer + i) % n_buffers, so a caller that carries
 * first_buffer across calls never lets a timed launch re-touch what its warm-up just touched.
 * Writes the elapsed milliseconds for the whole batch. */
bu_status bu_time_uastc_launches(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out,
                                 size_t n_buffers, size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int launches,
                                 uint64_t* d_status, void* stream, float* out_ms);
/* The timed region of bench.py on a GPU that never went idle: `lead` untimed launches, event 0, exactly `launches` timed
 * launches, event 1, enqueued back to back on `stream` with no host synchronisation in between (buffer rotation continues
 * through both parts).  *out_event_ms = hipEventElapsedTime(event 0, event 1); *out_host_ms = host steady-clock time from the
 * moment event 0 is first seen complete to the moment event 1 is (hipEventQuery polling): a wall-clock bracket around exactly
 * the timed launches.  *out_late (optional) = 1 if event 0 had already completed when the host finished enqueueing -- the
 * host bracket then started late and the caller must take max(host, event). */
bu_status bu_time_uastc_launches_window(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out,
                                        size_t n_buffers, size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int lead,
                                        int launches, uint64_t* d_status, void* stream, float* out_event_ms, float* out_host_ms,
                                        int* out_late);
/* Same launches with one hipEvent between every two of them: out_us[i] = microseconds from the event before launch i to
 * the event after it (includes the event's own packet; the batch form above is the one the mean is taken from, this one
 * gives the distribution: median, min, max). */
bu_status bu_time_uastc_launches_each(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out,
                                      size_t n_buffers, size_t first_buffer, size_t n_blocks, size_t blocks_per_row, int launches,
                                      uint64_t* d_status, void* stream, float* out_us);
/* Same launches spread round-robin over `n_streams` context-owned streams (independent slices in flight together);
 * wall-clock milliseconds between two device synchronisations.  Steady-state throughput row, not the roofline row. */
bu_status bu_time_uastc_launches_streams(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out,
                                         size_t n_buffers, size_t n_blocks, size_t blocks_per_row, int launches, int n_streams,
                                         float* out_ms);
/* bu_time_uastc_launches_window with SEVERAL launches in flight: launch i (lead, timed, tail) goes to context-owned stream
 * i % n_streams (1..8), everything enqueued up front.  n launches in flight are a pipeline (launch j starts one period after launch
 * j - 1 and is under way for about n periods), so throughput is counted in completions: every stream gets a start event behind its
 * last lead launch (in front of its first timed one) and an end event behind its last timed launch; *out_event_ms = latest end event -
 * LATEST start event on the device clock, i.e. from "the last lead launch has completed" to "the last timed launch has completed" --
 * exactly `launches` launches complete in between.  With lead >= n_streams and tail >= n_streams (untimed launches behind the end
 * events) the pipeline is full at both instants and *out_event_ms / launches is the steady-state launch-to-launch period (without tail
 * launches the head start of the first timed launches is credited and nothing debited: too short -- use the strict bracket).
 * *out_fill_drain_ms (optional) = latest end event - EARLIEST start event, the strict bracket: first instruction of the first timed
 * launch to last instruction of the last one (launches + n_streams - 1 periods in a full pipeline; with lead = tail = 0 the whole run
 * from an idle chip to an idle chip).  *out_host_ms = host steady clock from "every start event seen complete" to "every end event seen complete"; *out_late
 * (optional) = 1 if the first start event had already completed when the host finished enqueueing. */
/* on != 0: the two streams windows below enqueue from one host thread per stream (the order inside a stream unchanged, between streams
 * free) instead of from the calling thread alone -- how a caller with a thread per stream drives the context, and what keeps the host
 * from setting the pace when one enqueue costs more host time than a period (under rocprofv3 --kernel-trace: 6-8 us). */
bu_status bu_time_set_enqueue_threads(bu_context* ctx, int on);
/* on == 0: the context's persistent launches walk fixed shares of the tiles (rounds 1-5); on != 0 (default): long walks draw their tiles by ticket
 * (bu_uastc_transcode_device_sync).  Measurement only: bench.py shows both forms of the 2^25-block launch in one process; results never depend on it. */
bu_status bu_time_set_tile_tickets(bu_context* ctx, int on);
/* what BU_LAUNCH_AUTO has chosen for this context's large launches so far: out[0] exclusive, out[1] the shared kernels on one-tile workgroups, out[2] shared */
bu_status bu_time_auto_policy_counts(bu_context* ctx, unsigned long long out[3]);
/* the per-stream events of the context's LAST streams window, ms from the head of that call: out_start_ms[i] / out_end_ms[i] (8 floats each)
 * = stream i's start event (behind its last lead launch) / end event (behind its last timed launch), -1 for a stream without timed launches.
 * Streams running in step start and end within a few periods of each other; a stream that shares a hardware queue falls behind, and the
 * window "latest start to latest end" then no longer brackets `launches` completions -- callers check the spread. */
bu_status bu_time_last_window_streams(bu_context* ctx, float* out_start_ms, float* out_end_ms, int* out_n_streams);
/* host time the context's LAST streams window spent enqueueing (launches and events, before it began to wait) and the number of launches:
 * *out_ms / *out_launches = an UPPER bound of what one enqueue costs this host (waits for space in a full hardware queue are inside); at or
 * above the pipeline's period the host, not the chip, may be setting the pace */
bu_status bu_time_last_window_enqueue(bu_context* ctx, float* out_ms, int* out_launches);
/* The same window around PRODUCT calls instead of the helper's own launches (bench.py --config array512 and the *_through_product_api rows time
 * bu_uastc_transcode_batch_in_flight this way): bu_time_mark_streams(ctx, n, 0) records a timing-only start event, (ctx, n, 1) an end event, on the
 * context's own streams 0..n-1 behind what has been enqueued there -- call that enqueues lead work, mark 0, call that enqueues the timed work,
 * mark 1, call that enqueues tail work -- and bu_time_marks_elapsed waits for the marks: *out_event_ms = latest end - LATEST start (the timed
 * work's completions in a full pipeline), *out_strict_ms = latest end - EARLIEST start, *out_host_ms = the host clock between "all starts
 * seen complete" and "all ends seen complete"; the per-stream times are read with bu_time_last_window_streams. */
bu_status bu_time_mark_streams(bu_context* ctx, int n_streams, int which);
bu_status bu_time_marks_elapsed(bu_context* ctx, int n_streams, float* out_event_ms, float* out_strict_ms, float* out_host_ms);
#define BU_TIME_COPY_CEILING 100 /* as `target` of the call below: the uint4 -> uint4 copy kernel (bu_copy_ceiling_device) in place of a transcode */
bu_status bu_time_uastc_launches_streams_window(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out,
                                                size_t n_buffers, size_t first_buffer, size_t n_blocks, size_t blocks_per_row,
                                                int lead, int launches, int tail, int n_streams, uint64_t* d_status,
                                                float* out_event_ms, float* out_host_ms, float* out_fill_drain_ms, int* out_late);
/* the same window over the ETC1S codebook-lookup kernels (parity unpinned): launch i decodes index array d_idx[(first_buffer + i) % n_buffers]
 * (nbx x nby blocks) against one pair of device codebooks into d_out[...]; rgba = 0: ETC1 blocks, 1: the RGBA8 image (no alpha slice) */
bu_status bu_time_etc1s_launches_streams_window(bu_context* ctx, int rgba, const uint32_t* const* d_idx, void* const* d_out, size_t n_buffers,
                                                size_t first_buffer, size_t nbx, size_t nby, const uint32_t* d_endpoints, uint32_t n_endpoints,
                                                const void* d_selectors, uint32_t n_selectors, int lead, int launches, int tail, int n_streams,
                                                float* out_event_ms, float* out_host_ms);
/* The reference's micro-benchmark shape (benches/benchmark.rs:66-98): `reps` passes over `n_blocks` blocks, one per-block API
 * call per block (RGBA32: bu_unpack_uastc_block_to_rgba), host steady clock around the loop; nanoseconds per call. */
bu_status bu_time_block_api(bu_context* ctx, bu_target target, const uint8_t* blocks, size_t n_blocks, int reps, uint8_t* out,
                            float* out_ns_per_call);
bu_status bu_time_copy_launches(bu_context* ctx, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                size_t first_buffer, size_t n_blocks, int launches, void* stream, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* BASISU_HIP_H */
