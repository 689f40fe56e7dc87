#!/bin/bash
# tools/exp/shared_grid.sh : shared BC7 shape, 2^20 blocks per launch: two persistent workgroups per CU walking two tiles each (shipped, A) against
# one-tile workgroups dealt by the hardware dispatcher (B: 256 x 4, 1024 workgroups; C: 512 x 2, 1024 workgroups; D: 256 x 4, three per CU), 1-4 launches
# in flight; one library per process (a process with four contexts has more streams than hardware queues)
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
for l in lib_A_cur.so lib_B_256x4_p4.so lib_C_512x2_p4.so lib_D_256x4_p3.so; do
python3 ab_streams.py --target bc7 --streams 1,2,3,4 --policy 1 --rounds 2 --launches 256 --lead 64 --prewarm_ms 30 $l 2>&1 | grep -v amdgpu.ids
done; done
