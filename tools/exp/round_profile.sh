set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/v12
python -m pytest tests -m gpu -x -q > gpurun_out/v12/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/v12/pytest_gpu.log | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/v12/bench_steps20.json 2> gpurun_out/v12/bench_steps20.err
python bench.py --steps 512 --warmup 64 --no-cpu > gpurun_out/v12/bench_steps512.json 2> gpurun_out/v12/bench_steps512.err
rm -rf gpurun_out/v12/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v12/prof -- python3 bench.py --headline-only --steps 512 --warmup 64 > gpurun_out/v12/bench_profiled.json 2> gpurun_out/v12/bench_profiled.err
find gpurun_out/v12/prof -name "*kernel_stats*.csv" | head -1 | while read f; do cp "$f" gpurun_out/v12/rocprofv3_kernel_stats.csv; done
rm -rf gpurun_out/v12/prof
bash tools/gpu_pmc.sh > gpurun_out/v12/gpu_pmc.log 2>&1
cp gpurun_out/pmc/kernel_stats.csv gpurun_out/v12/rocprofv3_kernel_stats_all_kernels.csv
cp gpurun_out/pmc/pmc_all.json gpurun_out/pmc/pmc_bc7.json gpurun_out/pmc/pmc_summary.txt gpurun_out/v12/
rm -rf gpurun_out/pmc/trace gpurun_out/pmc/lds gpurun_out/pmc/sq gpurun_out/pmc/sq2 gpurun_out/pmc/fetch gpurun_out/pmc/write gpurun_out/pmc/grbm
LG_LO=10 LG_HI=23 python tools/exp/size_sweep_all.py > gpurun_out/v12/size_sweep_all_targets.txt 2>&1
tail -3 gpurun_out/v12/size_sweep_all_targets.txt
