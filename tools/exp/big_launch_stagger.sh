#!/bin/bash
# tools/exp/big_launch_stagger.sh : one exclusive BC7 launch of 2^24 / 2^25 blocks at a time; workgroup generations started 0 / 0.85 / 1.7 / 3.4 us apart
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
for lg in 22 23 25; do
  echo "== 2^$lg blocks per launch, us per launch"
  python3 ab_streams.py --streams 1 --policy 0 --n $((1<<lg)) --rounds 3 --launches 48 --lead 8 --prewarm_ms 30 lib_x512.so lib_nopri.so lib_nopri256p5.so 2>&1 | grep -v amdgpu.ids
done
