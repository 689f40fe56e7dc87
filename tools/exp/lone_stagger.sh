#!/bin/bash
# tools/exp/lone_stagger.sh : a LONE 2^20-block launch (exclusive shape, 4 workgroups per CU): the second half of every CU's workgroups starts its loads 0.4 / 0.9 / 1.7 us late
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
for l in lib_lone0.so lib_lone1.so lib_lone2.so lib_lone4.so; do
python3 ab_streams.py --target bc7 --streams 1 --policy 0 --rounds 3 --launches 256 --lead 64 --prewarm_ms 30 $l 2>&1 | grep -v amdgpu.ids
done
