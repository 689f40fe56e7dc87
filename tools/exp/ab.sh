#!/bin/bash
# A/B of kernel variants inside ONE gpurun call (box-to-box variance is ~4 %): tools/exp/ab.sh libA.so libB.so ...
# each variant is a full libbasisu_hip.so built with different -D flags (tools/exp/build_variant.sh)
for round in 1 2 3; do
  for lib in "$@"; do
    BASISU_HIP_LIB=$PWD/$lib timeout 200 python bench.py --steps 512 --warmup 64 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['us_per_launch'])"
  done
done
