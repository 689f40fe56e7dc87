#!/bin/bash
# tools/exp/ticket_min_walk.sh : tile tickets from 16 (shipped) / 8 / 4 tiles per workgroup on -- stream-ordered batch calls over few atlases or many small slices, and the
# plain launch of 2^22 .. 2^24 blocks
cd $GRAFT_REPO_ROOT/tools/exp
for shape in "8 1048576" "16 1048576" "128 262144" "256 65536"; do
  for l in lib_now.so lib_walk8.so lib_walk4.so; do
    python3 slices_in_flight_ab.py $l bc7 $shape 2>&1 | grep -v amdgpu.ids | sed 's/   in_flight_4.*//'
  done
done
for lg in 22 23 24; do
  n=$((1<<lg)); k=$(( (1<<28) >> lg ))
  echo "== bc7 2^$lg blocks per launch, one at a time, us per launch"
  python3 ab_streams.py --target bc7 --streams 1 --policy 0 --n $n --rounds 2 --launches $k --lead 8 --prewarm_ms 40 lib_now.so lib_walk8.so lib_walk4.so 2>&1 | grep -v amdgpu.ids
done
