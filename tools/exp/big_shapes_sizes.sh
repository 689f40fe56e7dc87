#!/bin/bash
# tools/exp/big_shapes_sizes.sh : ONE exclusive launch at a time, 2^21 .. 2^25 blocks: the shipped large shape (512 x 2, four per CU) against 256 x 4 five / six per CU, BC7 and ASTC
cd $GRAFT_REPO_ROOT/tools/exp
for lg in 21 22 23 24 25; do
  n=$((1<<lg)); k=$(( (1<<28) >> lg )); [ $k -lt 16 ] && k=16
  echo "== bc7 2^$lg blocks per launch, $k timed launches, us per launch"
  python3 ab_streams.py --target bc7 --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 40 lib_now.so lib_bc7x256g5.so 2>&1 | grep -v amdgpu.ids
  echo "== astc 2^$lg blocks per launch, $k timed launches, us per launch"
  python3 ab_streams.py --target astc --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 40 lib_now.so lib_astcx256g5.so lib_astcx256g6.so 2>&1 | grep -v amdgpu.ids
done
