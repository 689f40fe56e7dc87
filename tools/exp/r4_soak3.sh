#!/bin/bash
mkdir -p gpurun_out
{
echo "round 4, final kernels: long soak (tests/soak/gpu_bigfuzz.py; valid + high-contrast + raw random blocks, five targets, statuses and first-error index against the oracle)"
echo "FUZZ_KINDS=valid,contrast,raw FUZZ_SEED0=12000 FUZZ_SEEDS=128 (strip kernels, host entry points)"
FUZZ_KINDS=valid,contrast,raw FUZZ_SEED0=12000 FUZZ_SEEDS=128 timeout 3000 python tests/soak/gpu_bigfuzz.py 2>&1 | tail -2
echo "FUZZ_RECT=1 FUZZ_KINDS=valid,contrast FUZZ_SEED0=13000 FUZZ_SEEDS=96 (rectangular-tile kernels, device entry point)"
FUZZ_RECT=1 FUZZ_KINDS=valid,contrast FUZZ_SEED0=13000 FUZZ_SEEDS=96 timeout 2400 python tests/soak/gpu_bigfuzz.py 2>&1 | tail -2
} > gpurun_out/r4_soak3.txt 2>&1
cat gpurun_out/r4_soak3.txt
