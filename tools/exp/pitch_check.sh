#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
python tools/exp/r6_host.py multi2 2>&1 | grep -v amdgpu
cd tools/exp
L=../../basisu_rs_amd/libbasisu_hip.so
for bpr in 0 1024; do
  echo "== blocks_per_row $bpr"
  python3 ab_streams.py --target bc7 --streams 1,4 --policy 0,1 --bpr $bpr --n $((1<<20)) --rounds 1 --launches 256 --lead 64 --prewarm_ms 30 $L 2>&1 | grep -v amdgpu.ids
  python3 ab_streams.py --target bc7 --streams 1 --policy 0 --bpr $bpr --n $((1<<25)) --rounds 1 --launches 24 --lead 8 --prewarm_ms 100 $L 2>&1 | grep -v amdgpu.ids
done
