#!/bin/bash
# tools/exp/deep_pipeline.sh : more than four launches in flight (CU-mask streams: a hardware queue each, whatever the pool), shared and auto policy
cd $GRAFT_REPO_ROOT/tools/exp
export BU_STREAM_MODE=cumask
for t in bc7 astc etc1; do
echo "== $t (p1 shared, p2 auto)"
python3 ab_streams.py --target $t --streams 4,5,6,8 --policy 1,2 --rounds 2 --launches 256 --lead 64 --prewarm_ms 30 ../../basisu_rs_amd/libbasisu_hip.so 2>&1 | grep -v amdgpu.ids
done
