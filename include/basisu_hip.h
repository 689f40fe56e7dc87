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
 *     pointers, stage through the context's device buffers and return after the result is in `out`.
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
    BU_ERR_HIP = 8              /* a HIP runtime call failed; see bu_last_error() */
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

/* ---- per-block API of lib.rs:29-53 (each call is one 1-block launch; present for drop-in
 *      completeness and the known-answer tests, not for throughput) -------------------------- */
bu_status bu_unpack_uastc_block_to_rgba(bu_context* ctx, const uint8_t in[16], uint32_t out[16]); /* lib.rs:29 */
bu_status bu_transcode_uastc_block_to_astc(bu_context* ctx, const uint8_t in[16], uint8_t out[16]); /* lib.rs:33 */
bu_status bu_transcode_uastc_block_to_bc7(bu_context* ctx, const uint8_t in[16], uint8_t out[16]);  /* lib.rs:39 */
bu_status bu_transcode_uastc_block_to_etc1(bu_context* ctx, const uint8_t in[16], uint8_t out[8]);  /* lib.rs:43 */
bu_status bu_transcode_uastc_block_to_etc2(bu_context* ctx, const uint8_t in[16], uint8_t out[16]); /* lib.rs:49 */

/* ---- UASTC slice level, device pointers, asynchronous ----------------------------------------- */

/* Same work as the two slice entry points above on memory already resident in HBM.
 *   d_in            n_blocks * 16 bytes, 16-byte aligned
 *   d_out           n_blocks * block_bytes, 16-byte aligned (8 for ETC1)
 *   blocks_per_row  only read for BU_TARGET_RGBA32 (image pitch); n_blocks need not be a multiple
 *   d_status        optional device uint64_t initialised with bu_status_word_reset(): receives
 *                   min over failing blocks of (block_index << 8 | status); may be shared by several
 *                   launches whose block indices are disjoint (`block_index_base` is added)
 *   stream          hipStream_t
 * Returns launch/argument errors only; block errors arrive through d_status. */
bu_status bu_uastc_transcode_device(bu_context* ctx, bu_target target, const void* d_in, size_t n_blocks, void* d_out,
                                    size_t blocks_per_row, uint64_t block_index_base, uint64_t* d_status, void* stream);

/* value a status word must hold before the launches that report into it (all ones) */
#define BU_STATUS_WORD_CLEAR 0xFFFFFFFFFFFFFFFFull
/* enqueue a reset of *d_status on `stream` */
bu_status bu_status_word_reset(bu_context* ctx, uint64_t* d_status, void* stream);
/* decode a status word copied back to the host */
bu_status bu_status_word_decode(uint64_t word, uint64_t* first_bad_block);

/* ---- ETC1S block back-end (basis_lz/mod.rs:97-186) ---------------------------------------------
 * The serial BasisLZ entropy decode stays on the host and produces, per block in raster order,
 *   idx[i] = endpoint_index | selector_index << 16          (DecodedBlock, basis_lz/mod.rs:43-48)
 * and the two codebooks
 *   endpoints[k] = r5 | g5 << 8 | b5 << 16 | inten << 24     (Endpoint, basis_lz/mod.rs:518-522)
 *   selectors[k] = {rows[4] (2 bits/texel, x = 0 low), etc1_bytes[4]}  8 bytes (etc::Selector, etc.rs:343-350)
 * bu_etc1s_selector_from_rows() builds a selector entry from its 4 raw row bytes exactly as
 * Selector::set_selector does (etc.rs:363-393). */
void bu_etc1s_selector_from_rows(const uint8_t rows[4], uint8_t out_entry[8]);

/* Decoder::transcode_to_etc1 back-end, closure block_to_etc1 (basis_lz/mod.rs:163-181): 8 B per block */
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

/* ---- measurement helpers (bench.py) ------------------------------------------------------------
 * uint4 -> uint4 copy kernel of the same launch shape as the 16 B -> 16 B transcoders: the practical
 * HBM ceiling the roofline fraction is reported next to. */
bu_status bu_copy_ceiling_device(bu_context* ctx, const void* d_in, size_t n_blocks, void* d_out, void* stream);
/* Times `launches` back-to-back launches of one transcode with hipEvents recorded on `stream`
 * around the whole batch (the stream the kernels are launched on).  d_in/d_out are arrays of
 * `n_buffers` device pointers rotated round-robin so that consecutive launches touch different
 * HBM (cold-cache protocol).  Writes the elapsed milliseconds for the whole batch. */
bu_status bu_time_uastc_launches(bu_context* ctx, bu_target target, const void* const* d_in, void* const* d_out,
                                 size_t n_buffers, size_t n_blocks, size_t blocks_per_row, int launches,
                                 uint64_t* d_status, void* stream, float* out_ms);
bu_status bu_time_copy_launches(bu_context* ctx, const void* const* d_in, void* const* d_out, size_t n_buffers,
                                size_t n_blocks, int launches, void* stream, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* BASISU_HIP_H */
