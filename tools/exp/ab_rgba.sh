#!/bin/bash
# A/B of RGBA32 kernel variants inside one gpurun call: tools/exp/ab_rgba.sh libA.so libB.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    BASISU_HIP_LIB=$PWD/$lib TARGET=4 timeout 200 python tools/exp/rgba_time.py 2>/dev/null | sed "s|^|$lib |"
  done
done
