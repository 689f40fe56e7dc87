# one enqueue thread per stream against one thread for everything: unprofiled, and under rocprofv3 --kernel-trace
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/enq
mkdir -p $O
for t in 0 1 0 1; do
  python3 bench.py --headline-only --steps 512 --warmup 64 --enqueue-threads $t > $O/plain_t$t.json 2> $O/plain_t$t.err
  python3 -c "import json,sys; d=json.load(open('$O/plain_t$t.json')); print('plain threads=$t  %.3f us  frac %.4f  strict %.1f' % (d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline'].get('strict_bracket_ns_per_step', 0)))"
done
for t in 0 1; do
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --headline-only --steps 512 --warmup 64 --enqueue-threads $t > $O/prof_t$t.json 2> $O/prof_t$t.err
  python3 -c "import json,sys; d=json.load(open('$O/prof_t$t.json')); print('profiled threads=$t  %.3f us  frac %.4f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"
  find $O/prof -name "*kernel_trace.csv" | head -1 | while read f; do python3 tools/exp/trace_periods.py "$f" "sorted_kernel<1, 256" 1 | tee $O/trace_t$t.txt; done
  rm -rf $O/prof
done
