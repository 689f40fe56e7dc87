#!/bin/bash
# tools/exp/group_size_targets.sh : as group_size.sh for the other targets (grouped launches of 2^20 / 2^23 blocks / 2^23 with the 2^20-block runs grouped too)
cd $GRAFT_REPO_ROOT/tools/exp
for t in astc etc1 etc2; do
  for shape in "64 65536" "512 65536" "64 1048576"; do
    for l in lib_grp20.so lib_grp23.so lib_grp23all.so; do
      python3 slices_in_flight_ab.py $l $t $shape 2>&1 | grep -v amdgpu.ids
    done
  done
done
