#!/bin/bash
# tools/exp/one_shot_big2.sh : as one_shot_big.sh, the sizes in between (blocks per launch in units of 2^20)
cd $GRAFT_REPO_ROOT/tools/exp
for m in 1.25 1.5 2 3 4 6 8 12 16 24; do
  n=$(python3 -c "print(int($m * (1 << 20)))"); k=$(python3 -c "print(max(16, int((1 << 28) / $n)))")
  echo "== bc7 $m x 2^20 blocks per launch, one at a time, us per launch"
  python3 ab_streams.py --target bc7 --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 40 lib_now.so lib_os5122.so 2>&1 | grep -v amdgpu.ids
done
