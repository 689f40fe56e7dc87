#!/bin/bash
# tools/exp/tickets_ab.sh : tile tickets (dynamic tile assignment inside a persistent launch) on / off: ONE exclusive launch at a time, every target, 2^23..2^25 blocks
cd $GRAFT_REPO_ROOT/tools/exp
export GPU_MAX_HW_QUEUES=8
L=../../basisu_rs_amd/libbasisu_hip.so
for tgt in ${TARGETS:-bc7 astc etc1 etc2 rgba}; do
for lg in ${SIZES:-23 24 25}; do
  [ $tgt = rgba ] && [ $lg = 25 ] && continue   # (8 x 2 GiB of output buffers)
  for tk in 0 1; do
    echo "== $tgt 2^$lg blocks per launch, tickets=$tk, exclusive policy, one launch at a time"
    BU_TILE_TICKETS=$tk python3 ab_streams.py --target $tgt --streams 1 --policy 0 --n $((1<<lg)) --rounds 2 --launches 24 --lead 8 --prewarm_ms 30 $L 2>&1 | grep -v amdgpu.ids
  done
done
done
