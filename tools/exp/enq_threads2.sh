set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/enq2
mkdir -p $O
python -m pytest tests/test_gpu_round5.py -q -m gpu -x 2>&1 | tail -3
python3 bench.py --steps 20 > $O/bench_steps20.json 2> $O/bench_steps20.err
python3 tools/exp/bench_summary.py $O/bench_steps20.json | head -4
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --headline-only --steps 512 --warmup 64 --enqueue-threads 1 > $O/bench_profiled.json 2> $O/bench_profiled.err
python3 -c "import json,sys; d=json.load(open('$O/bench_profiled.json')); print('profiled threads=1  %.3f us  frac %.4f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"
find $O/prof -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" "sorted_kernel<1, 256" 1 | tee $O/rocprofv3_headline_trace_summary.txt; done
find $O/prof -name "*kernel_stats*.csv" | head -1 | while read f; do cp "$f" $O/rocprofv3_kernel_stats.csv; done
rm -rf $O/prof
