# launches in flight x launch size: us per launch (and per 2^20 blocks) of the shipped library, shared and exclusive policy
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
cd tools/exp
for n in 19 20 21 22 23; do
  L=$(( (1<<29) >> n )); [ $L -gt 1024 ] && L=1024
  echo "== 2^$n blocks per launch, $L timed launches"
  python3 ab_streams.py --n $((1<<n)) --streams 1,2,4 --policy 0,1 --rounds 2 --launches $L --lead 256 --prewarm_ms 40 ../../basisu_rs_amd/libbasisu_hip.so 2>&1 | grep -v amdgpu.ids
done
