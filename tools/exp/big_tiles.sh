#!/bin/bash
# tools/exp/big_tiles.sh : ONE exclusive 2^25-block BC7 launch (tile tickets): 1024-block tiles (512 x 2, four workgroups per CU) against 2048-block tiles (512 x 4, three per CU; 1024 x 2, two per CU)
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
for l in lib_t1024.so lib_t2048p3.so lib_t2048w1024.so; do
python3 ab_streams.py --target bc7 --streams 1 --policy 0 --n $((1<<25)) --rounds 2 --launches 24 --lead 8 --prewarm_ms 100 $l 2>&1 | grep -v amdgpu.ids
done
