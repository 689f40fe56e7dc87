# round 6, final build: soak bundle, every step under its own timeout
set -u
export TMPDIR=/tmp
( FUZZ_POLICY=auto timeout 300 python tests/soak/odd_sizes.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/odd_sizes auto: /'
( FUZZ_POLICY=shared timeout 300 python tests/soak/odd_sizes.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/odd_sizes shared: /'
( FUZZ_POLICY=auto FUZZ_CONCURRENT=1 FUZZ_SEED0=6000 FUZZ_SEEDS=12 FUZZ_KINDS=valid,raw,contrast timeout 500 python tests/soak/gpu_bigfuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/bigfuzz auto, concurrent targets: /'
( FUZZ_SEED0=6100 FUZZ_SEEDS=8 FUZZ_KINDS=valid,raw timeout 400 python tests/soak/gpu_bigfuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | sed 's/^/bigfuzz exclusive: /'
( TICKET_SEEDS=4 timeout 600 python tests/soak/gpu_tickets.py 2>&1 | grep -v amdgpu.ids | tail -1 ) | sed 's/^/gpu_tickets: /'
( FUZZ_SEED=71 FUZZ_SECONDS=240 timeout 500 python tests/soak/batch_in_flight_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -1 ) | sed 's/^/batch_in_flight_fuzz: /'
( FUZZ_SEED=72 FUZZ_SECONDS=120 timeout 300 python tests/soak/batch_in_flight_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -1 ) | sed 's/^/batch_in_flight_fuzz: /'
( timeout 200 python tests/soak/block_api_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -1 ) | sed 's/^/block_api_fuzz: /'
( FUZZ_SECONDS=60 timeout 300 python tests/soak/streamed_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -1 ) | sed 's/^/streamed_fuzz: /'
