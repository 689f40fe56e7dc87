#!/bin/bash
# tools/exp/lead_sweep.sh LIB: the multi-stream timing window at K = 20 and K = 240 timed launches against the number of lead launches
cd $GRAFT_REPO_ROOT/tools/exp
for ll in "0 20" "4 20" "8 20" "16 20" "32 20" "64 20" "256 20" "0 240" "8 240" "64 240" "0 1000"; do
  set -- $ll
  echo "lead $1 K $2: $(python3 ab_streams.py --streams 4 --rounds 3 --policy 1 --lead $1 --launches $2 $LIBS 2>&1 | grep -v amdgpu.ids | sed -e 's/.*S4/S4/' -e 's/ (ev[^)]*)//' | paste -sd' ')"
done
