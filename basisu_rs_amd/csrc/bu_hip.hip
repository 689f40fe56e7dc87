// gfx950 kernels and the C ABI (include/basisu_hip.h) of the UASTC / ETC1S block-transcode path.
//
// Data layout in HBM
//   UASTC slice      n x 16-byte blocks, raster order (uastc.rs:49-75)            -> read once as uint4
//   ASTC/BC7/ETC2    n x 16-byte blocks, same order                                -> written once as uint4
//   ETC1             n x  8-byte blocks                                            -> uint2
//   RGBA32           row-major image, pitch 16*blocks_per_row bytes (uastc.rs:96) -> 4 x uint4 per block
//   tables           one BuTables blob (9.5 KiB); every workgroup copies the ranges its target reads to LDS
// Mapping: one lane = one block.  A wave reads 64 x 16 B = 1 KiB contiguous and writes 1 KiB (512 B
// for ETC1; for RGBA32 four 1 KiB row segments when the row has >= 64 blocks).  No MFMA: the work is
// bit-field surgery on 128-bit values.  The roofline is HBM (bytes per block); what limits the kernels
// in practice is vector-ALU instruction issue.
//
// One translation unit; the pieces, in inclusion order, are listed at the bottom of this file.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/basisu_hip.h"
#include "bu_basis.hpp"
#include "bu_batch_plan.hpp"   // slices -> runs -> launches of the batch entry points (host only)
#include "bu_uastc_dispatch.hpp"

#include "bu_kernels.hpp"        // device code
#include "bu_context.hpp"        // bu_context, launcher, host-pointer driver
#include "bu_streams.hpp"        // the context's own streams (hardware-queue check), BU_LAUNCH_AUTO
#include "bu_capi_slice.hpp"     // extern "C": slice level
#include "bu_capi_file.hpp"      // extern "C": whole-file level
#include "bu_capi_measure.hpp"   // extern "C": measurement helpers
#include "bu_multi.hpp"          // extern "C": multi-GPU
