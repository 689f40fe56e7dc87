#!/bin/bash
# tools/exp/astc_multi_shape.sh : the ASTC multi-run launch as 256 x 4 five per CU (lib_astcnow; under the shared policy three per CU, lib_astch2 / h4: two / four) against
# 512 x 2 four per CU (lib_grp23all: the build before) -- in-flight call over small slices and atlases in separate allocations
cd $GRAFT_REPO_ROOT/tools/exp
for shape in "64 65536" "512 65536" "128 262144" "64 1048576"; do
  for l in lib_grp23all.so lib_astcnow.so lib_astch2.so lib_astch4.so; do
    python3 slices_in_flight_ab.py $l astc $shape 2>&1 | grep -v amdgpu.ids
  done
done
