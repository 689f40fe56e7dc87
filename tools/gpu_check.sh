#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench (both configs), rocprofv3 kernel trace of the headline.
# Everything judged is copied from gpurun_out/ into profiles/ afterwards (in the build container).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > gpurun_out/rocminfo.txt
echo "== pytest -m gpu" ; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py "$@" 2> gpurun_out/bench.err | tee gpurun_out/bench.json ; tail -5 gpurun_out/bench.err
echo "== bench array512 (forced dist, one rank)" ; BENCH_FORCE_DIST=1 timeout 900 python bench.py --config array512 --steps 20 --warmup 3 2> gpurun_out/bench_array512.err | tee gpurun_out/bench_array512.json ; tail -5 gpurun_out/bench_array512.err
echo "== rocprofv3 kernel trace"
rm -rf gpurun_out/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 512 --warmup 64 --headline-only > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
find gpurun_out/prof -name "*kernel_stats*.csv" | head -2 | while read f; do echo "-- $f"; head -12 "$f" | cut -c1-300; done
