#!/bin/bash
# tools/exp/group_size.sh : bu_uastc_transcode_batch_in_flight over small slices in separate allocations -- blocks per grouped (multi-run) launch 2^20 (round 5) / 2^22 / 2^23,
# and 2^23 with 2^20-block runs grouped too (lib_grp23all); with the runtime's queue pool and with CU-mask streams
cd $GRAFT_REPO_ROOT/tools/exp
for q in "" 8; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  for shape in "64 65536" "512 65536" "128 262144" "64 1048576"; do
    for l in lib_grp20.so lib_grp22.so lib_grp23.so lib_grp23all.so; do
      python3 slices_in_flight_ab.py $l bc7 $shape 2>&1 | grep -v amdgpu.ids
    done
  done
done
