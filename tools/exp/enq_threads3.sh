set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/enq4
mkdir -p $O
python3 bench.py --steps 20 > $O/bench_steps20.json 2> $O/bench_steps20.err
python3 tools/exp/bench_summary.py $O/bench_steps20.json | head -4
for t in 0 1; do
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --headline-only --steps 512 --warmup 64 --enqueue-threads $t > $O/bench_profiled_t$t.json 2> $O/bench_profiled_t$t.err
python3 -c "import json,sys; d=json.load(open('$O/bench_profiled_t$t.json')); print('profiled threads=$t  %.3f us  frac %.4f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"
find $O/prof -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" "sorted_kernel<1, 256" 1 | tee $O/trace_summary_t$t.txt; done
rm -rf $O/prof
done
