#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_pytest.log
tail -4 gpurun_out/r3_pytest.log
timeout 300 python tools/exp/cfg4_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_cfg4_trace.txt | head -12
timeout 900 python tools/exp/streamed_stress.py 2>&1 | tail -6 | tee gpurun_out/r3_stress.txt
