# final-build soak bundle, every step under its own timeout
set -u
export TMPDIR=/tmp
O=gpurun_out/final_soaks; mkdir -p $O
( timeout 200 python tests/soak/odd_sizes.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/odd_sizes exclusive: /'
( FUZZ_POLICY=shared timeout 200 python tests/soak/odd_sizes.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/odd_sizes shared: /'
( FUZZ_SEED0=5000 FUZZ_SEEDS=12 FUZZ_KINDS=valid,raw,contrast timeout 400 python tests/soak/gpu_bigfuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/bigfuzz exclusive: /'
( timeout 200 python tests/soak/block_api_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/block_api_fuzz: /'
( FUZZ_SECONDS=90 timeout 300 python tests/soak/streamed_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/streamed_fuzz: /'
( FUZZ_SEED=11 FUZZ_SECONDS=90 timeout 200 python tests/soak/batch_in_flight_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -1 ) | sed 's/^/batch_in_flight_fuzz: /'
