#!/bin/bash
# round-2 probe: phase timeline of the shipped BC7 shape + counters for every kernel
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > gpurun_out/rocminfo.txt
echo "== stamps 24 (512x2, four per CU)"; timeout 300 python3 tools/exp/stamps_run.py 24 20 8 1024 2>&1 | tail -14 | tee gpurun_out/stamps24.log
echo "== stamps 22 (1024x2)"; timeout 300 python3 tools/exp/stamps_run.py 22 20 16 2048 2>&1 | tail -14 | tee gpurun_out/stamps22.log
echo "== pmc"; bash tools/gpu_pmc.sh > gpurun_out/pmc_stdout.log 2>&1; tail -5 gpurun_out/pmc_stdout.log
ls gpurun_out/pmc
