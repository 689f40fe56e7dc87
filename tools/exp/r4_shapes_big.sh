#!/bin/bash
# BC7 tile shapes at large slices (BASELINE config 5 is 2^25 blocks in one launch)
L="tools/exp/lib_base.so tools/exp/lib_x1024_2_2_1.so tools/exp/lib_x512_4_2_1.so tools/exp/lib_x512_4_3_1.so tools/exp/lib_x1024_4_1_1.so tools/exp/lib_x256_4_4_1.so"
for lg in 22 23 25; do
  echo "=== 2^$lg blocks"
  timeout 600 python tools/exp/ab_multi.py --targets bc7 --rounds 3 --n $((1<<lg)) --launches 48 $L 2>&1 | grep -v amdgpu.ids
done
