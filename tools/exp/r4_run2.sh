#!/bin/bash
# round 4, run 2: GPU suite on the build with the two-thread slice decode + 3-bit palette unpack; A/B of that unpack; config-4 phase trace
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_pytest.log
timeout 600 python tools/exp/ab_multi.py --targets etc1,etc2,rgba --rounds 4 tools/exp/lib_pre3bit.so tools/exp/lib_w3pal.so > gpurun_out/r2_ab_w3pal.txt 2>&1
timeout 300 python tools/exp/cfg4_trace.py > gpurun_out/r2_cfg4_trace.txt 2>&1
tail -5 gpurun_out/r2_pytest.log; cat gpurun_out/r2_ab_w3pal.txt | tail -30; tail -30 gpurun_out/r2_cfg4_trace.txt
