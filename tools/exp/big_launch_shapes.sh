#!/bin/bash
# tools/exp/big_launch_shapes.sh : ONE exclusive BC7 launch at a time, sizes 2^20..2^25, over variants of the exclusive big shape (tools/exp/lib_x*.so)
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
for lg in 20 21 22 23 25; do
  n=$((1<<lg)); k=$(( (1<<27) >> lg )); [ $k -lt 16 ] && k=16
  echo "== 2^$lg blocks per launch, $k timed launches, us per launch"
  python3 ab_streams.py --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 20 lib_x512.so lib_x256p4.so lib_x256p5.so lib_x512p5.so 2>&1 | grep -v amdgpu.ids
done
