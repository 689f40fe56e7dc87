#!/bin/bash
# tools/exp/etc_long_walk.sh : ETC1 / ETC2 exclusive-shape candidates on long walks (see etc_long_walk.py)
cd $GRAFT_REPO_ROOT/tools/exp
for t in etc1 etc2; do for l in lib_exbase.so lib_ex512g2.so; do for p in 0 1; do
  python3 etc_long_walk.py $l $t $p 2>&1 | grep -v amdgpu.ids
done; done; done
