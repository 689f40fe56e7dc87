#!/bin/bash
# A/B of kernel variants inside ONE gpurun call (box-to-box variance is ~4 %): tools/exp/ab.sh libA.so libB.so ...
# each variant is a full libbasisu_hip.so (tools/exp/build_variant.sh); TARGETS="bc7 etc1" limits the targets timed
for round in 1 2 3; do
  for lib in "$@"; do
    BASISU_HIP_LIB=$PWD/$lib timeout 300 python3 tools/exp/ab_all.py ${TARGETS:-} 2>/dev/null
  done
done
