#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
L=../../basisu_rs_amd/libbasisu_hip.so
for bpr in 1024 256; do
  for tk in 0 1; do
    echo "== bc7 2^25 blocks per launch, bpr=$bpr tickets=$tk"
    BU_TILE_TICKETS=$tk python3 ab_streams.py --target bc7 --streams 1,4 --policy 0,1 --bpr $bpr --n $((1<<25)) --rounds 2 --launches 24 --lead 8 --prewarm_ms 30 $L 2>&1 | grep -v amdgpu.ids
  done
done
echo "== 2^23 pieces, bpr 256, shared, 4 in flight"
python3 ab_streams.py --target bc7 --streams 4 --policy 1 --bpr 256 --n $((1<<23)) --rounds 2 --launches 96 --lead 16 --prewarm_ms 30 $L 2>&1 | grep -v amdgpu.ids
