# round-end evidence, part B: the rocprofv3 kernel trace of the headline bench, kernel trace + counter passes over every kernel, size sweeps
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # (bench.py sets it itself, but under rocprofv3 the HIP runtime is initialised before python starts)
TAG=${TAG:-v}
O=gpurun_out/$TAG
mkdir -p $O
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --headline-only --steps 512 --warmup 64 > $O/bench_profiled.json 2> $O/bench_profiled.err
find $O/prof -name "*kernel_stats*.csv" | head -1 | while read f; do cp "$f" $O/rocprofv3_kernel_stats.csv; done
# what the same trace says about the pipeline: span of a dispatch, completion period UNDER THE PROFILER (its own cost per dispatch is 6-8 us:
# profiles/r05_rocprofv3_dispatch_floor_*), dispatches running, hardware queues
find $O/prof -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" "sorted_kernel<1, 256" 1 > $O/rocprofv3_headline_trace_summary.txt; done
# the headline's own kernel (round 6, --method batch): one launch over 64 atlases -- span of a dispatch = the step
find $O/prof -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" "multi_kernel<1" 1 > $O/rocprofv3_headline_batch_kernel_summary.txt; done
rm -rf $O/prof
# the same with one enqueue thread per stream (bu_time_set_enqueue_threads): the profiler's 6-8 us of host time per enqueue no longer set the pace.  The
# streams are then fed out of step, so bench.py's own figure in this mode is the strict bracket; the profiler's clock over the steady stretches is the evidence
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --method pipeline --headline-only --steps 512 --warmup 64 --enqueue-threads 1 > $O/bench_profiled_enqueue_threads.json 2> $O/bench_profiled_enqueue_threads.err
find $O/prof -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" "sorted_kernel<1, 256" 1 > $O/rocprofv3_headline_trace_summary_enqueue_threads.txt; done
rm -rf $O/prof
# BASELINE config 5 with four launches in flight: at 2^23 blocks per launch the profiler's cost is small beside the period, its clock and the events agree
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof5 -- python3 bench.py --config array512 --steps 200 --warmup 3 > $O/bench_array512_profiled.json 2> $O/bench_array512_profiled.err
find $O/prof5 -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" sorted_kernel 4 > $O/rocprofv3_array512_four_launches_in_flight.txt; done
rm -rf $O/prof5
head -5 $O/rocprofv3_kernel_stats.csv | cut -c1-200
timeout 1500 bash tools/gpu_pmc.sh > $O/gpu_pmc.log 2>&1
cp gpurun_out/pmc/kernel_stats.csv $O/rocprofv3_kernel_stats_all_kernels.csv
cp gpurun_out/pmc/pmc_all.json gpurun_out/pmc/pmc_bc7.json gpurun_out/pmc/pmc_bc7_array512.json gpurun_out/pmc/pmc_summary.txt gpurun_out/pmc/pmc_summary_array512.txt gpurun_out/pmc/kernel_stats_array512.csv $O/
rm -rf gpurun_out/pmc/trace gpurun_out/pmc/lds gpurun_out/pmc/sq gpurun_out/pmc/sq2 gpurun_out/pmc/fetch gpurun_out/pmc/write gpurun_out/pmc/grbm
LG_LO=10 LG_HI=23 timeout 600 python tools/exp/size_sweep_all.py > $O/size_sweep_all_targets.txt 2>&1
tail -3 $O/size_sweep_all_targets.txt
timeout 600 python tools/exp/copy_sweep.py > $O/copy_vs_bc7_size_sweep.txt 2>&1
tail -4 $O/copy_vs_bc7_size_sweep.txt
