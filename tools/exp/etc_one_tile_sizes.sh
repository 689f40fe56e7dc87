#!/bin/bash
# tools/exp/etc_one_tile_sizes.sh : large exclusive ETC1 / ETC2 launches as one-tile workgroups of the shared shape (512 x 4, 2048-block tiles, two resident per CU) dealt by the
# hardware dispatcher (lib_etcsemi1all) against the shipped 1024 x 4 one-per-CU persistent grid (lib_new), by size, one launch at a time
cd $GRAFT_REPO_ROOT/tools/exp
for t in etc1 etc2; do
for m in 0.8 1 1.25 1.5 2 3 4 8 16 32; do
  n=$(python3 -c "print(int($m * (1 << 20)) // 1024 * 1024)"); k=$(python3 -c "print(max(16, int((1 << 27) / $n)))")
  echo "== $t $m x 2^20 blocks per launch, one at a time, us per launch"
  python3 ab_streams.py --target $t --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 40 lib_new.so lib_etcsemi1all.so 2>&1 | grep -v amdgpu.ids
done; done
