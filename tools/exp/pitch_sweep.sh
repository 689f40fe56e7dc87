#!/bin/bash
# tools/exp/pitch_sweep.sh : blocks_per_row of a block-linear target (BC7) decides the tile layout -- 0: strips of 1024 consecutive blocks; a multiple of 64 (>= 128): rectangles
# of 64 x 16 blocks, i.e. 16 segments of 1 KiB at a pitch of 16 x blocks_per_row bytes.  The bytes never depend on it; the rate does.
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
L=../../basisu_rs_amd/libbasisu_hip.so
for bpr in 0 128 256 512 1024 2048 4096; do
  echo "== blocks_per_row $bpr: 2^20 blocks per launch (p1 shared S4 / p0 exclusive S1), then 2^25 blocks (p0 S1)"
  python3 ab_streams.py --target bc7 --streams 1,4 --policy 0,1 --bpr $bpr --n $((1<<20)) --rounds 1 --launches 256 --lead 64 --prewarm_ms 30 $L 2>&1 | grep -v amdgpu.ids
  python3 ab_streams.py --target bc7 --streams 1 --policy 0 --bpr $bpr --n $((1<<25)) --rounds 1 --launches 24 --lead 8 --prewarm_ms 100 $L 2>&1 | grep -v amdgpu.ids
done
