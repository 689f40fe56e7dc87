#!/bin/bash
# tools/exp/bc7_multi_shape.sh : the BC7 multi-run launch as 256 x 4 five per CU (lib_bc7m256g5: three per CU under the shared policy; ...h4: four) against 512 x 2 four
# per CU / two under the shared policy (lib_astcnow) -- in-flight call and stream-ordered batch call over slices in separate allocations
cd $GRAFT_REPO_ROOT/tools/exp
for shape in "64 65536" "512 65536" "128 262144" "64 1048576"; do
  for l in lib_astcnow.so lib_bc7m256g5.so lib_bc7m256g5h4.so; do
    python3 slices_in_flight_ab.py $l bc7 $shape 2>&1 | grep -v amdgpu.ids
  done
done
