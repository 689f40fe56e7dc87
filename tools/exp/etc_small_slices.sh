#!/bin/bash
# tools/exp/etc_small_slices.sh : 64 slices of 65 536 blocks in separate allocations, ETC1 / ETC2: round-5 multi-run shape (lib_exbase), persistent two per CU (lib_etcnow),
# and one per CU under the shared policy (lib_etchalf1)
cd $GRAFT_REPO_ROOT/tools/exp
for t in etc1 etc2; do for l in lib_exbase.so lib_etcnow.so lib_etchalf1.so; do
  python3 slices_in_flight_ab.py $l $t 2>&1 | grep -v amdgpu.ids
done; done
python3 slices_in_flight_ab.py lib_etcnow.so bc7 2>&1 | grep -v amdgpu.ids
