#!/bin/bash
# tools/exp/big_shapes_tickets.sh : ONE exclusive 2^25-block BC7 launch WITH tile tickets over shapes of the persistent grid (round 5's A/B of these shapes had fixed shares)
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do for l in lib_e512x2p4.so lib_e256x4p5.so lib_e256x4p4.so lib_e1024x1p2.so; do
python3 ab_streams.py --target bc7 --streams 1 --policy 0 --n $((1<<25)) --rounds 1 --launches 24 --lead 8 --prewarm_ms 100 $l 2>&1 | grep -v amdgpu.ids
done; done
