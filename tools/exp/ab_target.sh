#!/bin/bash
# A/B of one target's kernel at 2^20 blocks: TARGET=<0 astc|1 bc7|2 etc1|3 etc2> tools/exp/ab_target.sh libA.so libB.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    BASISU_HIP_LIB=$PWD/$lib timeout 200 python tools/exp/size_sweep.py 2>/dev/null | grep "2^20" | sed "s|^|$lib |"
  done
done
