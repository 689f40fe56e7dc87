#!/bin/bash
# tools/exp/etc_multi_persist.sh : ETC1 / ETC2 through ONE bu_uastc_transcode_batch_device launch over 64 atlases in separate allocations: shipped multi-run shape
# against a persistent two-per-CU grid with the one-tile prefetch (lib_etcpersist.so)
cd $GRAFT_REPO_ROOT/tools/exp
export SEP=1
for t in etc1 etc2; do for l in lib_exbase.so lib_etcpersist.so; do
  python3 etc_long_walk.py $l $t 0 2>&1 | grep -v amdgpu.ids
done; done
