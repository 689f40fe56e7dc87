#!/bin/bash
# tools/exp/stream_modes.sh LIB : how the context's streams are created (BU_TEST_STREAMS of an experiment build) x HIP queue environment:
# multi-stream timing windows of three shapes, then a rocprofv3 kernel trace of one run (hardware queues per stream, overlap, drift)
lib=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/exp
timing() {
  for ll in "64 240" "256 20" "1024 20"; do
    set -- $ll
    echo "  lead $1 K $2: $(python3 ab_streams.py --streams 4 --rounds 2 --policy 1 --lead $1 --launches $2 $lib 2>&1 | grep -v amdgpu.ids | sed -e 's/.*S4/S4/' | paste -sd' ')"
  done
}
trace() {
  rm -rf /tmp/tr_$1
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$1 -- python3 ab_streams.py --streams 4 --rounds 1 --policy 1 --lead 64 --launches 240 $lib > /dev/null 2>&1
  python3 trace_streams.py $(find /tmp/tr_$1 -name "*kernel_trace.csv" | head -1) 304
}
echo "=== normal priority, default env"; export BU_TEST_STREAMS=0; timing; trace a
echo "=== normal priority, GPU_MAX_HW_QUEUES=8"; export GPU_MAX_HW_QUEUES=8; timing; trace b; unset GPU_MAX_HW_QUEUES
echo "=== priority pairs, default env"; export BU_TEST_STREAMS=1; timing; trace c
echo "=== CU-mask streams, default env"; export BU_TEST_STREAMS=2; timing; trace d
