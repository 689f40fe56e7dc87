#!/bin/bash
# tools/exp/group_size_few.sh : FEW atlases of 2^20 blocks per in-flight call (4 / 8 / 16 / 32): one launch per atlas against grouped launches of 2^22 / 2^23 blocks
cd $GRAFT_REPO_ROOT/tools/exp
for shape in "4 1048576" "8 1048576" "16 1048576" "32 1048576"; do
  for l in lib_grp20.so lib_grp22all.so lib_grp23all.so; do
    python3 slices_in_flight_ab.py $l bc7 $shape 2>&1 | grep -v amdgpu.ids
  done
done
