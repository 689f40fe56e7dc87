# shared-policy BC7 shapes with larger tiles (better filled chunks), 1 and 4 launches in flight; one library per process
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
cd tools/exp
for r in 1 2; do
for lib in "$@"; do
  python3 ab_streams.py --streams 1,4 --policy 1 --rounds 2 --launches 512 --lead 512 --prewarm_ms 40 $lib 2>&1 | grep -v amdgpu.ids
done
done
