"""basisu_rs_amd -- MI355X (gfx950) block-transcode hot path of a Basis Universal transcoder.

The product is the C ABI in include/basisu_hip.h (built from csrc/ into libbasisu_hip.so); this
package is the thin Python driver used by the tests and bench.py.  Nothing here imports oracle/.
"""
from .api import (  # noqa: F401
    BasisuError,
    Context,
    Decoder,
    TargetTextureFormat,
    default_context,
    Image,
    basislz_decode,
    crc16,
    etc1s_selector_from_rows,
    read_header,
    read_query,
    read_slice_descs,
    read_to_astc,
    read_to_bc7,
    read_to_etc1,
    read_to_etc2,
    read_to_rgba,
    read_to_uastc,
    write_uastc_file,
    transcode_uastc_block_to_astc,
    transcode_uastc_block_to_bc7,
    transcode_uastc_block_to_etc1,
    transcode_uastc_block_to_etc2,
    unpack_uastc_block_to_rgba,
)
