#!/bin/bash
# tools/exp/one_shot_big.sh : large exclusive BC7 launches as ONE-TILE workgroups dealt by the hardware dispatcher (no persistent grid, no prefetch, tables staged per tile) --
# the memory behaviour of the one-pass copies of tools/exp/copy_big.hip -- against the shipped persistent ticketed grid
cd $GRAFT_REPO_ROOT/tools/exp
for lg in 20 22 23 24 25; do
  n=$((1<<lg)); k=$(( (1<<28) >> lg )); [ $k -lt 16 ] && k=16
  echo "== bc7 2^$lg blocks per launch, one at a time, us per launch"
  python3 ab_streams.py --target bc7 --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 40 lib_now.so lib_os5122.so lib_os2564.so lib_os10241.so 2>&1 | grep -v amdgpu.ids
done
