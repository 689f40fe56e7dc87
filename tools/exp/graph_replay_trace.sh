#!/bin/bash
# tools/exp/graph_replay_trace.sh : the headline pipeline (a) through the product call, (b) replayed from a HIP graph -- unprofiled and under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== (a) product call, unprofiled"; python3 $R/tools/exp/product_call_trace.py 512 30 2>&1 | grep -v amdgpu.ids
for thr in 1 0; do
  rm -rf /tmp/pc_$thr
  echo "== (a) product call under rocprofv3 --kernel-trace, BU_ENQUEUE_THREADS=$thr"
  BU_ENQUEUE_THREADS=$thr rocprofv3 --kernel-trace --output-format csv -d /tmp/pc_$thr -- python3 $R/tools/exp/product_call_trace.py 512 30 2>&1 | grep "calls of"
  python3 $R/tools/exp/trace_periods.py $(find /tmp/pc_$thr -name "*kernel_trace.csv" | head -1) sorted_kernel
done
echo "== (b) graph replay, unprofiled"; python3 $R/tools/exp/graph_replay.py 256 60 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/gr
echo "== (b) graph replay under rocprofv3 --kernel-trace (graph of 256 launches x 60 replays, then the same launches one by one)"
rocprofv3 --kernel-trace --output-format csv -d /tmp/gr -- python3 $R/tools/exp/graph_replay.py 256 60 2>&1 | grep "graph of\|one by one"
python3 $R/tools/exp/trace_periods.py $(find /tmp/gr -name "*kernel_trace.csv" | head -1) sorted_kernel
