#!/bin/bash
# round 4, GPU call 1: BC7 big-shape A/B (tile size, waves per CU, tiles loaded up front) + phase stamps of the interesting ones
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/../.."
echo "== ab_multi" 
timeout 900 python3 tools/exp/ab_multi.py --targets bc7,copy --rounds 3 tools/exp/lib_base.so tools/exp/lib_x1024_1_2_2.so tools/exp/lib_x1024_2_2_1.so \
   tools/exp/lib_x1024_4_1_1.so tools/exp/lib_x512_4_2_1.so tools/exp/lib_x1024_2_1_2.so tools/exp/lib_x512_2_2_2.so tools/exp/lib_x512_1_4_2.so \
   tools/exp/lib_x1024_1_1_4.so tools/exp/lib_x256_4_4_1.so 2>&1 | tee gpurun_out/r4_ab1.txt
for v in "60 8 1024" "61 16 2048" "62 16 2048" "63 16 4096"; do
  set -- $v
  echo "== stamps variant $1"; timeout 300 python3 tools/exp/stamps_run.py $1 20 $2 $3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_stamps_$1.txt
done
echo "== stamps variant 61 second pass"; STAMP_BLOCK=1 timeout 300 python3 tools/exp/stamps_run.py 61 20 16 2048 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_stamps_61_pass2.txt
