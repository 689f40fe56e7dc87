# the N > 1 branch of bench.py with one rank (RCCL communicator alive): do the context's four streams still get a hardware queue each?
set -u
export TMPDIR=/tmp
O=gpurun_out/skew; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.load(open('$1')); t=d['config']['timed_region']
print('$2', 'us/step %.3f' % (d['ms_per_step']*1e3), 'strict', d['roofline']['strict_bracket_ns_per_step'], 'windows', t['windows_us_per_step'])
print('   median window', t['streams_of_median_window'])
"; }
for q in 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$q BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/dist_q$q.json 2> $O/dist_q$q.err; show $O/dist_q$q.json "dist GPU_MAX_HW_QUEUES=$q"
done
DEBUG_HIP_DYNAMIC_QUEUES=1 BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/dist_dyn.json 2> $O/dist_dyn.err; show $O/dist_dyn.json "dist DEBUG_HIP_DYNAMIC_QUEUES=1 (GPU_MAX_HW_QUEUES=8)"
GPU_MAX_HW_QUEUES=16 python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/plain_q16.json 2> $O/plain_q16.err; show $O/plain_q16.json "plain GPU_MAX_HW_QUEUES=16"
