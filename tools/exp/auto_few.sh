#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
python tools/exp/r6_host.py matrix 2>&1 | grep -v amdgpu
cd tools/exp
for t in astc bc7; do echo "== $t: p0 exclusive, p1 shared, p2 auto"; python3 ab_streams.py --target $t --streams 1,2,3,4 --policy 0,1,2 --rounds 2 --launches 256 --lead 64 --prewarm_ms 30 ../../basisu_rs_amd/libbasisu_hip.so 2>&1 | grep -v amdgpu.ids; done
