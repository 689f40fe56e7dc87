//! `extern "C"` view of include/basisu_hip.h.  NOT COMPILED HERE (no Rust toolchain in the build image).
//! Every function below is exported by libbasisu_hip.so; tests/test_capi_symbols.py checks the names.
#![allow(non_camel_case_types, dead_code)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct bu_context {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bu_comm {
    _private: [u8; 0],
}

// bu_target
pub const BU_TARGET_ASTC: c_int = 0;
pub const BU_TARGET_BC7: c_int = 1;
pub const BU_TARGET_ETC1: c_int = 2;
pub const BU_TARGET_ETC2: c_int = 3;
pub const BU_TARGET_RGBA32: c_int = 4;
// bu_read_target
pub const BU_READ_RGBA: c_int = 0;
pub const BU_READ_ETC1: c_int = 1;
pub const BU_READ_ETC2: c_int = 2;
pub const BU_READ_UASTC: c_int = 3;
pub const BU_READ_ASTC: c_int = 4;
pub const BU_READ_BC7: c_int = 5;
pub const BU_OK: c_int = 0;
pub const BU_COMM_ID_BYTES: usize = 128;
pub const BU_IPC_HANDLE_BYTES: usize = 64;

/// bu_basis_header == basis::Header (basis.rs:417-454): same 26 fields, C layout
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bu_basis_header {
    pub sig: u16,
    pub ver: u16,
    pub header_size: u16,
    pub header_crc16: u16,
    pub data_size: u32,
    pub data_crc16: u16,
    pub total_slices: u32,
    pub total_images: u32,
    pub tex_format: u8,
    pub flags: u16,
    pub tex_type: u8,
    pub us_per_frame: u32,
    pub reserved: u32,
    pub userdata0: u32,
    pub userdata1: u32,
    pub total_endpoints: u16,
    pub endpoint_cb_file_ofs: u32,
    pub endpoint_cb_file_size: u32,
    pub total_selectors: u16,
    pub selector_cb_file_ofs: u32,
    pub selector_cb_file_size: u32,
    pub tables_file_ofs: u32,
    pub tables_file_size: u32,
    pub slice_desc_file_ofs: u32,
    pub extended_file_ofs: u32,
    pub extended_file_size: u32,
}

/// bu_slice_desc == basis::SliceDesc (basis.rs:519-535)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bu_slice_desc {
    pub image_index: u32,
    pub level_index: u8,
    pub flags: u8,
    pub orig_width: u16,
    pub orig_height: u16,
    pub num_blocks_x: u16,
    pub num_blocks_y: u16,
    pub file_ofs: u32,
    pub file_size: u32,
    pub slice_data_crc16: u16,
}

/// bu_image: where one Image<u8> of a read_to_* call lives inside the caller's output buffer
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bu_image {
    pub w: u32,
    pub h: u32,
    pub stride: u32,
    pub reserved: u32,
    pub offset: u64,
    pub size: u64,
}

extern "C" {
    pub fn bu_context_create(device: c_int, out_ctx: *mut *mut bu_context) -> c_int;
    pub fn bu_context_destroy(ctx: *mut bu_context);
    pub fn bu_status_string(st: c_int) -> *const c_char;
    pub fn bu_last_error(ctx: *const bu_context) -> *const c_char;
    pub fn bu_target_block_bytes(target: c_int) -> usize;
    // launch policy of the slice-level device entry points: 0 = BU_LAUNCH_EXCLUSIVE, 1 = BU_LAUNCH_SHARED, 2 = BU_LAUNCH_AUTO (default: chosen per call)
    pub fn bu_context_set_launch_policy(ctx: *mut bu_context, policy: c_int) -> c_int;
    pub fn bu_context_get_launch_policy(ctx: *const bu_context, out_policy: *mut c_int) -> c_int;
    // the context's own streams (hipStream_t), index 0..7, for several launches in flight.  The library checks at creation that each has a hardware
    // queue of its own and re-creates them with CU masks when the runtime's pool (GPU_MAX_HW_QUEUES) is too small: bu_context_query_in_flight reports
    pub fn bu_context_stream(ctx: *mut bu_context, index: c_int, out_stream: *mut *mut c_void) -> c_int;
    pub fn bu_context_synchronize(ctx: *mut bu_context) -> c_int;
    // the largest number of the context's streams 0..n_streams-1 that share one hardware queue in this process (1 = none do)
    pub fn bu_context_probe_streams(ctx: *mut bu_context, n_streams: c_int, out_max_sharing: *mut c_int) -> c_int;
    // how many launches a pipeline over streams 0..n_streams-1 really keeps in flight in this process; stream mode 0 = queue pool, 1 = CU-mask streams
    pub fn bu_context_query_in_flight(ctx: *mut bu_context, n_streams: c_int, out_effective_streams: *mut c_int, out_stream_mode: *mut c_int) -> c_int;
    // slice level, host pointers (uastc.rs:89-146)
    pub fn bu_uastc_transcode(ctx: *mut bu_context, target: c_int, input: *const u8, in_bytes: usize, out: *mut u8, out_bytes: usize,
                              first_bad_block: *mut u64) -> c_int;
    pub fn bu_uastc_decode_to_rgba(ctx: *mut bu_context, input: *const u8, in_bytes: usize, blocks_per_row: usize, out: *mut u8,
                                   out_bytes: usize, first_bad_block: *mut u64) -> c_int;
    // per-block API (lib.rs:29-53): on the calling thread by default, through a one-block launch after bu_block_api_on_device(ctx, 1)
    pub fn bu_block_api_on_device(ctx: *mut bu_context, enable: c_int) -> c_int;
    pub fn bu_unpack_uastc_block_to_rgba(ctx: *mut bu_context, input: *const u8, out: *mut u32) -> c_int;
    pub fn bu_transcode_uastc_block_to_astc(ctx: *mut bu_context, input: *const u8, out: *mut u8) -> c_int;
    pub fn bu_transcode_uastc_block_to_bc7(ctx: *mut bu_context, input: *const u8, out: *mut u8) -> c_int;
    pub fn bu_transcode_uastc_block_to_etc1(ctx: *mut bu_context, input: *const u8, out: *mut u8) -> c_int;
    pub fn bu_transcode_uastc_block_to_etc2(ctx: *mut bu_context, input: *const u8, out: *mut u8) -> c_int;
    // slice level, device pointers, asynchronous
    pub fn bu_uastc_transcode_device(ctx: *mut bu_context, target: c_int, d_in: *const c_void, n_blocks: usize, d_out: *mut c_void,
                                     blocks_per_row: usize, block_index_base: u64, d_status: *mut u64, stream: *mut c_void) -> c_int;
    pub fn bu_uastc_transcode_batch_device(ctx: *mut bu_context, target: c_int, n_slices: usize, d_in: *const *const c_void, n_blocks: *const usize,
                                           d_out: *const *mut c_void, blocks_per_row: usize, index_base: *const u64, d_status: *mut u64,
                                           stream: *mut c_void) -> c_int;
    // the same loop as a pipeline of launches on the context's own streams 0..n_streams-1 (shared launch policy); only enqueues:
    // bu_context_synchronize waits
    pub fn bu_uastc_transcode_batch_in_flight(ctx: *mut bu_context, target: c_int, n_slices: usize, d_in: *const *const c_void, n_blocks: *const usize,
                                              d_out: *const *mut c_void, blocks_per_row: usize, index_base: *const u64, d_status: *mut u64,
                                              n_streams: c_int) -> c_int;
    // bu_uastc_transcode_device that waits: one exclusive launch (tile tickets on long walks: 0.77 of the roofline for a 2^25-block array), status word returned
    pub fn bu_uastc_transcode_device_sync(ctx: *mut bu_context, target: c_int, d_in: *const c_void, n_blocks: usize, d_out: *mut c_void,
                                          blocks_per_row: usize, block_index_base: u64, out_status_word: *mut u64) -> c_int;
    pub fn bu_status_word_reset(ctx: *mut bu_context, d_status: *mut u64, stream: *mut c_void) -> c_int;
    pub fn bu_status_word_decode(word: u64, first_bad_block: *mut u64) -> c_int;
    pub fn bu_host_alloc(ctx: *mut bu_context, bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn bu_host_free(ctx: *mut bu_context, ptr: *mut c_void) -> c_int;
    // ETC1S back-end (basis_lz/mod.rs:97-186)
    pub fn bu_etc1s_selector_from_rows(rows: *const u8, out_entry: *mut u8);
    pub fn bu_etc1s_transcode_etc1(ctx: *mut bu_context, idx: *const u32, n_blocks: usize, endpoints: *const u32, n_endpoints: u32,
                                   selectors: *const u8, n_selectors: u32, out: *mut u8, out_bytes: usize, first_bad_block: *mut u64) -> c_int;
    pub fn bu_etc1s_decode_rgba(ctx: *mut bu_context, idx: *const u32, alpha_idx: *const u32, nbx: usize, nby: usize,
                                endpoints: *const u32, n_endpoints: u32, selectors: *const u8, n_selectors: u32, out: *mut u8,
                                out_bytes: usize, first_bad_block: *mut u64) -> c_int;
    // whole-file level (basis.rs)
    pub fn bu_basis_read_header(file: *const u8, len: usize, out: *mut bu_basis_header) -> c_int;
    pub fn bu_basis_read_slice_descs(file: *const u8, len: usize, header: *const bu_basis_header, out: *mut bu_slice_desc,
                                     max_descs: usize, n_descs: *mut usize) -> c_int;
    pub fn bu_basis_crc16(data: *const u8, len: usize, crc: u16) -> u16;
    pub fn bu_read_query(target: c_int, file: *const u8, len: usize, n_images: *mut usize, out_bytes: *mut usize) -> c_int;
    pub fn bu_read_to(ctx: *mut bu_context, target: c_int, file: *const u8, len: usize, header_out: *mut bu_basis_header,
                      images: *mut bu_image, max_images: usize, n_images: *mut usize, out: *mut u8, out_bytes: usize) -> c_int;
    // multi-GPU: shards of one texture array
    pub fn bu_comm_unique_id(id: *mut u8) -> c_int;
    pub fn bu_comm_create(ctx: *mut bu_context, world: c_int, rank: c_int, id: *const u8, out_comm: *mut *mut bu_comm) -> c_int;
    pub fn bu_comm_destroy(comm: *mut bu_comm);
    pub fn bu_comm_query(comm: *mut bu_comm, out_ranks: *mut c_int, out_rank: *mut c_int) -> c_int;
    pub fn bu_allgather_inplace(comm: *mut bu_comm, d_full: *mut c_void, shard_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn bu_ipc_export(ctx: *mut bu_context, d_ptr: *mut c_void, handle: *mut u8) -> c_int;
    pub fn bu_ipc_open(ctx: *mut bu_context, handle: *const u8, d_peer: *mut *mut c_void) -> c_int;
    pub fn bu_ipc_close(ctx: *mut bu_context, d_peer: *mut c_void) -> c_int;
    pub fn bu_allgather_peer(ctx: *mut bu_context, d_full: *mut c_void, d_peer_full: *const *mut c_void, world: c_int, rank: c_int,
                             shard_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn bu_array_transcode_sharded(ctxs: *const *mut bu_context, n_ctx: c_int, target: c_int, d_in_shard: *const *const c_void,
                                      n_slices: usize, blocks_per_slice: usize, d_full: *const *mut c_void, gather: c_int,
                                      first_bad_block: *mut u64) -> c_int;
    pub fn bu_device_alloc(ctx: *mut bu_context, bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn bu_device_free(ctx: *mut bu_context, ptr: *mut c_void) -> c_int;
    pub fn bu_memcpy(ctx: *mut bu_context, dst: *mut c_void, src: *const c_void, bytes: usize, to_device: c_int) -> c_int;
}
