#!/bin/bash
# tools/exp/kres.sh LIB.so PATTERN : VGPRs / SGPRs / LDS / scratch of the gfx950 kernels whose name matches PATTERN
lib=$1; pat=${2:-sorted_kernelILi1E}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$lib --output=$tmp/dev.o 2>/dev/null || { 
  # shared libraries embed the bundle in .hip_fatbin
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $lib $tmp/fat.bin && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.o; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.o | awk -v pat="$pat" '
/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2} /\.agpr_count:/ {a=$2}
/\.wavefront_size:/ { if (name ~ pat) printf "%s vgpr %s agpr %s sgpr %s lds %s scratch %s\n", name, v, a, s, l, p }'
rm -rf $tmp
