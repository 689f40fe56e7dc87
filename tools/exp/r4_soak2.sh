#!/bin/bash
# soaks on the final build (three-bit palette unpack in the ETC / RGBA32 paths)
mkdir -p gpurun_out
{
echo "round 4 soaks on the final build (three-bit weights through a byte palette in the ETC1 / ETC2 / RGBA32 unpack; two-thread slice decode)"
echo
echo "tests/soak/block_api_fuzz.py FUZZ_BLOCKS=300000"
FUZZ_BLOCKS=300000 timeout 900 python tests/soak/block_api_fuzz.py 2>&1 | tail -4
echo
echo "tests/soak/gpu_bigfuzz.py FUZZ_SEED0=9000 FUZZ_SEEDS=32 (strip kernels, host entry points)"
FUZZ_SEED0=9000 FUZZ_SEEDS=32 timeout 1200 python tests/soak/gpu_bigfuzz.py 2>&1 | tail -2
echo
echo "FUZZ_RECT=1 FUZZ_SEED0=9500 FUZZ_SEEDS=32 (rectangular-tile kernels, device entry point)"
FUZZ_RECT=1 FUZZ_SEED0=9500 FUZZ_SEEDS=32 timeout 1200 python tests/soak/gpu_bigfuzz.py 2>&1 | tail -2
echo
echo "tests/soak/odd_sizes.py"
timeout 900 python tests/soak/odd_sizes.py 2>&1 | tail -3
echo
echo "tools/exp/streamed_stress.py"
timeout 900 python tools/exp/streamed_stress.py 2>&1 | tail -3
} > gpurun_out/r4_soak2.txt 2>&1
cat gpurun_out/r4_soak2.txt
