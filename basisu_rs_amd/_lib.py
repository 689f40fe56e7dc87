"""ctypes binding of the C ABI (include/basisu_hip.h).  There is no fallback: if the HIP shared
library is missing this module raises, and every call needs a gfx950 device."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BASISU_HIP_LIB") or os.path.join(HERE, "libbasisu_hip.so")  # the override exists for A/B kernel experiments (tools/exp)

# bu_target
ASTC, BC7, ETC1, ETC2, RGBA32 = 0, 1, 2, 3, 4
BLOCK_BYTES = {ASTC: 16, BC7: 16, ETC1: 8, ETC2: 16, RGBA32: 64}
# bu_status
OK, ERR_INVALID_MODE, ERR_INVALID_PATTERN, ERR_LENGTH, ERR_OUTPUT_SIZE, ERR_ARGUMENT, ERR_INDEX_RANGE, ERR_NO_DEVICE, ERR_HIP = range(9)
ERR_UNSUPPORTED, ERR_BOUNDS = 17, 19
STATUS_WORD_CLEAR = 0xFFFFFFFFFFFFFFFF

# every symbol include/basisu_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "bu_context_create", "bu_context_destroy", "bu_status_string", "bu_last_error", "bu_target_block_bytes",
    "bu_uastc_transcode", "bu_uastc_decode_to_rgba",
    "bu_unpack_uastc_block_to_rgba", "bu_transcode_uastc_block_to_astc", "bu_transcode_uastc_block_to_bc7",
    "bu_transcode_uastc_block_to_etc1", "bu_transcode_uastc_block_to_etc2", "bu_block_api_on_device", "bu_context_set_launch_policy", "bu_context_get_launch_policy", "bu_context_stream", "bu_context_synchronize", "bu_context_probe_streams", "bu_context_query_in_flight", "bu_uastc_transcode_device_sync",
    "bu_uastc_transcode_device", "bu_uastc_transcode_batch_device", "bu_uastc_transcode_batch_in_flight", "bu_status_word_reset", "bu_status_word_decode", "bu_host_alloc", "bu_host_free",
    "bu_etc1s_selector_from_rows", "bu_etc1s_transcode_etc1_device", "bu_etc1s_decode_rgba_device",
    "bu_etc1s_transcode_etc1", "bu_etc1s_decode_rgba",
    "bu_basis_read_header", "bu_basis_read_slice_descs", "bu_basis_crc16", "bu_read_query", "bu_read_to", "bu_basislz_decode",
    "bu_basis_write_uastc",
    "bu_comm_unique_id", "bu_comm_create", "bu_comm_destroy", "bu_comm_query", "bu_allgather_inplace",
    "bu_ipc_export", "bu_ipc_open", "bu_ipc_close", "bu_allgather_peer", "bu_array_transcode_sharded",
    "bu_device_alloc", "bu_device_free", "bu_memcpy",
    "bu_copy_ceiling_device", "bu_time_uastc_launches", "bu_time_uastc_launches_window", "bu_time_uastc_launches_each", "bu_time_uastc_launches_streams", "bu_time_uastc_launches_streams_window", "bu_time_set_enqueue_threads", "bu_time_set_tile_tickets", "bu_time_auto_policy_counts", "bu_time_last_window_streams", "bu_time_last_window_enqueue", "bu_time_mark_streams", "bu_time_marks_elapsed", "bu_time_etc1s_launches_streams_window", "bu_time_copy_launches", "bu_time_block_api",
]
COMM_ID_BYTES, IPC_HANDLE_BYTES = 128, 64

# bu_read_target
READ_RGBA, READ_ETC1, READ_ETC2, READ_UASTC, READ_ASTC, READ_BC7 = range(6)


class BasisHeader(ctypes.Structure):  # bu_basis_header == basis::Header (basis.rs:417-454)
    _fields_ = [("sig", ctypes.c_uint16), ("ver", ctypes.c_uint16), ("header_size", ctypes.c_uint16), ("header_crc16", ctypes.c_uint16),
                ("data_size", ctypes.c_uint32), ("data_crc16", ctypes.c_uint16), ("total_slices", ctypes.c_uint32), ("total_images", ctypes.c_uint32),
                ("tex_format", ctypes.c_uint8), ("flags", ctypes.c_uint16), ("tex_type", ctypes.c_uint8), ("us_per_frame", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("userdata0", ctypes.c_uint32), ("userdata1", ctypes.c_uint32), ("total_endpoints", ctypes.c_uint16),
                ("endpoint_cb_file_ofs", ctypes.c_uint32), ("endpoint_cb_file_size", ctypes.c_uint32), ("total_selectors", ctypes.c_uint16),
                ("selector_cb_file_ofs", ctypes.c_uint32), ("selector_cb_file_size", ctypes.c_uint32), ("tables_file_ofs", ctypes.c_uint32),
                ("tables_file_size", ctypes.c_uint32), ("slice_desc_file_ofs", ctypes.c_uint32), ("extended_file_ofs", ctypes.c_uint32),
                ("extended_file_size", ctypes.c_uint32)]

    def as_list(self):
        return [getattr(self, f) for f, _ in self._fields_]


class SliceDesc(ctypes.Structure):  # bu_slice_desc == basis::SliceDesc (basis.rs:519-535)
    _fields_ = [("image_index", ctypes.c_uint32), ("level_index", ctypes.c_uint8), ("flags", ctypes.c_uint8), ("orig_width", ctypes.c_uint16),
                ("orig_height", ctypes.c_uint16), ("num_blocks_x", ctypes.c_uint16), ("num_blocks_y", ctypes.c_uint16), ("file_ofs", ctypes.c_uint32),
                ("file_size", ctypes.c_uint32), ("slice_data_crc16", ctypes.c_uint16)]


class ImageDesc(ctypes.Structure):  # bu_image
    _fields_ = [("w", ctypes.c_uint32), ("h", ctypes.c_uint32), ("stride", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("offset", ctypes.c_uint64), ("size", ctypes.c_uint64)]


_lib = None


def load():
    """Load libbasisu_hip.so (built by basisu_rs_amd/build.py).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "basisu_rs_amd: %s is missing -- build it with `python -m basisu_rs_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    try:
        # PyTorch-ROCm ships its own libamdhip64.  Whichever HIP runtime is mapped first serves the whole process, and
        # torch.cuda sees no devices when it is not torch's own -- so in a process that can import torch, import it first.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, sz, u64p, u32 = c.c_void_p, c.c_size_t, c.POINTER(c.c_uint64), c.c_uint32
    lib.bu_context_create.argtypes = [c.c_int, c.POINTER(vp)]
    lib.bu_context_create.restype = c.c_int
    lib.bu_context_destroy.argtypes = [vp]
    lib.bu_context_destroy.restype = None
    lib.bu_status_string.argtypes = [c.c_int]
    lib.bu_status_string.restype = c.c_char_p
    lib.bu_last_error.argtypes = [vp]
    lib.bu_last_error.restype = c.c_char_p
    lib.bu_target_block_bytes.argtypes = [c.c_int]
    lib.bu_target_block_bytes.restype = sz
    lib.bu_uastc_transcode.argtypes = [vp, c.c_int, vp, sz, vp, sz, u64p]
    lib.bu_uastc_transcode.restype = c.c_int
    lib.bu_uastc_decode_to_rgba.argtypes = [vp, vp, sz, sz, vp, sz, u64p]
    lib.bu_uastc_decode_to_rgba.restype = c.c_int
    for name in ("bu_unpack_uastc_block_to_rgba", "bu_transcode_uastc_block_to_astc", "bu_transcode_uastc_block_to_bc7",
                 "bu_transcode_uastc_block_to_etc1", "bu_transcode_uastc_block_to_etc2"):
        getattr(lib, name).argtypes = [vp, vp, vp]
        getattr(lib, name).restype = c.c_int
    lib.bu_time_block_api.argtypes = [vp, c.c_int, vp, sz, c.c_int, vp, c.POINTER(c.c_float)]
    lib.bu_time_block_api.restype = c.c_int
    lib.bu_context_set_launch_policy.argtypes = [vp, c.c_int]
    lib.bu_context_set_launch_policy.restype = c.c_int
    lib.bu_context_get_launch_policy.argtypes = [vp, c.POINTER(c.c_int)]
    lib.bu_context_get_launch_policy.restype = c.c_int
    lib.bu_context_stream.argtypes = [vp, c.c_int, c.POINTER(vp)]
    lib.bu_context_stream.restype = c.c_int
    lib.bu_context_synchronize.argtypes = [vp]
    lib.bu_context_synchronize.restype = c.c_int
    lib.bu_context_probe_streams.argtypes = [vp, c.c_int, c.POINTER(c.c_int)]
    lib.bu_context_probe_streams.restype = c.c_int
    lib.bu_context_query_in_flight.argtypes = [vp, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_int)]
    lib.bu_context_query_in_flight.restype = c.c_int
    lib.bu_uastc_transcode_device_sync.argtypes = [vp, c.c_int, vp, sz, vp, sz, c.c_uint64, u64p]
    lib.bu_uastc_transcode_device_sync.restype = c.c_int
    lib.bu_block_api_on_device.argtypes = [vp, c.c_int]
    lib.bu_block_api_on_device.restype = c.c_int
    lib.bu_uastc_transcode_device.argtypes = [vp, c.c_int, vp, sz, vp, sz, c.c_uint64, vp, vp]
    lib.bu_uastc_transcode_device.restype = c.c_int
    lib.bu_status_word_reset.argtypes = [vp, vp, vp]
    lib.bu_status_word_reset.restype = c.c_int
    lib.bu_status_word_decode.argtypes = [c.c_uint64, u64p]
    lib.bu_status_word_decode.restype = c.c_int
    lib.bu_host_alloc.argtypes = [vp, sz, c.POINTER(vp)]
    lib.bu_host_alloc.restype = c.c_int
    lib.bu_host_free.argtypes = [vp, vp]
    lib.bu_host_free.restype = c.c_int
    lib.bu_etc1s_selector_from_rows.argtypes = [vp, vp]
    lib.bu_etc1s_selector_from_rows.restype = None
    lib.bu_etc1s_transcode_etc1_device.argtypes = [vp, vp, sz, vp, u32, vp, u32, vp, vp, vp]
    lib.bu_etc1s_transcode_etc1_device.restype = c.c_int
    lib.bu_etc1s_decode_rgba_device.argtypes = [vp, vp, vp, sz, sz, vp, u32, vp, u32, vp, vp, vp]
    lib.bu_etc1s_decode_rgba_device.restype = c.c_int
    lib.bu_etc1s_transcode_etc1.argtypes = [vp, vp, sz, vp, u32, vp, u32, vp, sz, u64p]
    lib.bu_etc1s_transcode_etc1.restype = c.c_int
    lib.bu_etc1s_decode_rgba.argtypes = [vp, vp, vp, sz, sz, vp, u32, vp, u32, vp, sz, u64p]
    lib.bu_etc1s_decode_rgba.restype = c.c_int
    szp = c.POINTER(sz)
    lib.bu_basis_read_header.argtypes = [vp, sz, c.POINTER(BasisHeader)]
    lib.bu_basis_read_header.restype = c.c_int
    lib.bu_basis_read_slice_descs.argtypes = [vp, sz, c.POINTER(BasisHeader), c.POINTER(SliceDesc), sz, szp]
    lib.bu_basis_read_slice_descs.restype = c.c_int
    lib.bu_basis_crc16.argtypes = [vp, sz, c.c_uint16]
    lib.bu_basis_crc16.restype = c.c_uint16
    lib.bu_read_query.argtypes = [c.c_int, vp, sz, szp, szp]
    lib.bu_read_query.restype = c.c_int
    lib.bu_read_to.argtypes = [vp, c.c_int, vp, sz, c.POINTER(BasisHeader), c.POINTER(ImageDesc), sz, szp, vp, sz]
    lib.bu_read_to.restype = c.c_int
    lib.bu_basislz_decode.argtypes = [vp, sz, u32, vp, vp, vp]
    lib.bu_basislz_decode.restype = c.c_int
    lib.bu_basis_write_uastc.argtypes = [c.POINTER(SliceDesc), c.POINTER(vp), szp, sz, c.c_uint16, c.c_uint8, vp, sz, szp]
    lib.bu_basis_write_uastc.restype = c.c_int
    lib.bu_copy_ceiling_device.argtypes = [vp, vp, sz, vp, vp]
    lib.bu_copy_ceiling_device.restype = c.c_int
    lib.bu_time_uastc_launches.argtypes = [vp, c.c_int, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, sz, c.c_int, vp, vp, c.POINTER(c.c_float)]
    lib.bu_time_uastc_launches.restype = c.c_int
    lib.bu_uastc_transcode_batch_device.argtypes = [vp, c.c_int, sz, c.POINTER(vp), c.POINTER(sz), c.POINTER(vp), sz, c.POINTER(c.c_uint64), vp, vp]
    lib.bu_uastc_transcode_batch_device.restype = c.c_int
    lib.bu_uastc_transcode_batch_in_flight.argtypes = [vp, c.c_int, sz, c.POINTER(vp), c.POINTER(sz), c.POINTER(vp), sz, c.POINTER(c.c_uint64), vp, c.c_int]
    lib.bu_uastc_transcode_batch_in_flight.restype = c.c_int
    lib.bu_time_uastc_launches_window.argtypes = [vp, c.c_int, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, sz, c.c_int, c.c_int, vp, vp,
                                                  c.POINTER(c.c_float), c.POINTER(c.c_float), c.POINTER(c.c_int)]
    lib.bu_time_uastc_launches_window.restype = c.c_int
    lib.bu_time_uastc_launches_each.argtypes = [vp, c.c_int, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, sz, c.c_int, vp, vp, c.POINTER(c.c_float)]
    lib.bu_time_uastc_launches_each.restype = c.c_int
    # multi-GPU
    lib.bu_comm_unique_id.argtypes = [vp]
    lib.bu_comm_unique_id.restype = c.c_int
    lib.bu_comm_create.argtypes = [vp, c.c_int, c.c_int, vp, c.POINTER(vp)]
    lib.bu_comm_create.restype = c.c_int
    lib.bu_comm_destroy.argtypes = [vp]
    lib.bu_comm_destroy.restype = None
    lib.bu_comm_query.argtypes = [vp, c.POINTER(c.c_int), c.POINTER(c.c_int)]
    lib.bu_comm_query.restype = c.c_int
    lib.bu_allgather_inplace.argtypes = [vp, vp, sz, vp]
    lib.bu_allgather_inplace.restype = c.c_int
    lib.bu_ipc_export.argtypes = [vp, vp, vp]
    lib.bu_ipc_export.restype = c.c_int
    lib.bu_ipc_open.argtypes = [vp, vp, c.POINTER(vp)]
    lib.bu_ipc_open.restype = c.c_int
    lib.bu_ipc_close.argtypes = [vp, vp]
    lib.bu_ipc_close.restype = c.c_int
    lib.bu_allgather_peer.argtypes = [vp, vp, c.POINTER(vp), c.c_int, c.c_int, sz, vp]
    lib.bu_allgather_peer.restype = c.c_int
    lib.bu_array_transcode_sharded.argtypes = [c.POINTER(vp), c.c_int, c.c_int, c.POINTER(vp), sz, sz, c.POINTER(vp), c.c_int, u64p]
    lib.bu_array_transcode_sharded.restype = c.c_int
    lib.bu_device_alloc.argtypes = [vp, sz, c.POINTER(vp)]
    lib.bu_device_alloc.restype = c.c_int
    lib.bu_device_free.argtypes = [vp, vp]
    lib.bu_device_free.restype = c.c_int
    lib.bu_memcpy.argtypes = [vp, vp, vp, sz, c.c_int]
    lib.bu_memcpy.restype = c.c_int
    lib.bu_time_uastc_launches_streams.argtypes = [vp, c.c_int, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, c.c_int, c.c_int, c.POINTER(c.c_float)]
    lib.bu_time_uastc_launches_streams.restype = c.c_int
    lib.bu_time_uastc_launches_streams_window.argtypes = [vp, c.c_int, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, sz, c.c_int, c.c_int, c.c_int, c.c_int, vp,
                                                          c.POINTER(c.c_float), c.POINTER(c.c_float), c.POINTER(c.c_float), c.POINTER(c.c_int)]
    lib.bu_time_uastc_launches_streams_window.restype = c.c_int
    lib.bu_time_set_enqueue_threads.argtypes = [vp, c.c_int]
    lib.bu_time_set_enqueue_threads.restype = c.c_int
    lib.bu_time_set_tile_tickets.argtypes = [vp, c.c_int]
    lib.bu_time_set_tile_tickets.restype = c.c_int
    lib.bu_time_auto_policy_counts.argtypes = [vp, c.POINTER(c.c_ulonglong)]
    lib.bu_time_auto_policy_counts.restype = c.c_int
    lib.bu_time_last_window_streams.argtypes = [vp, c.POINTER(c.c_float), c.POINTER(c.c_float), c.POINTER(c.c_int)]
    lib.bu_time_last_window_streams.restype = c.c_int
    lib.bu_time_last_window_enqueue.argtypes = [vp, c.POINTER(c.c_float), c.POINTER(c.c_int)]
    lib.bu_time_last_window_enqueue.restype = c.c_int
    lib.bu_time_mark_streams.argtypes = [vp, c.c_int, c.c_int]
    lib.bu_time_mark_streams.restype = c.c_int
    lib.bu_time_marks_elapsed.argtypes = [vp, c.c_int, c.POINTER(c.c_float), c.POINTER(c.c_float), c.POINTER(c.c_float)]
    lib.bu_time_marks_elapsed.restype = c.c_int
    lib.bu_time_etc1s_launches_streams_window.argtypes = [vp, c.c_int, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, sz, vp, c.c_uint32, vp, c.c_uint32, c.c_int, c.c_int, c.c_int, c.c_int,
                                                          c.POINTER(c.c_float), c.POINTER(c.c_float)]
    lib.bu_time_etc1s_launches_streams_window.restype = c.c_int
    lib.bu_time_copy_launches.argtypes = [vp, c.POINTER(vp), c.POINTER(vp), sz, sz, sz, c.c_int, vp, c.POINTER(c.c_float)]
    lib.bu_time_copy_launches.restype = c.c_int
    _lib = lib
    return lib
