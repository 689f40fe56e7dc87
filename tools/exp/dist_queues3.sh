set -u
export TMPDIR=/tmp
O=gpurun_out/skew3; mkdir -p $O
show() { python3 -c "
import json
d=json.load(open('$1')); t=d['config']['timed_region']
print('$2', 'us/step %.3f' % (d['ms_per_step']*1e3), 'frac', d['roofline']['frac'], 'in step', t.get('streams_in_step'), 'spread', t.get('start_event_spread_us'), t.get('streams'))
"; }
for i in 1 2; do
GPU_MAX_HW_QUEUES=16 python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "array512 plain q16"
GPU_MAX_HW_QUEUES=8 python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "array512 plain q8"
GPU_MAX_HW_QUEUES=12 BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "array512 dist q12"
GPU_MAX_HW_QUEUES=16 BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "array512 dist q16"
GPU_MAX_HW_QUEUES=16 BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 --in-flight 1 > $O/a.json 2> $O/a.err; show $O/a.json "array512 dist q16 in-flight 1"
python bench.py --config array512 --steps 20 --warmup 3 --in-flight 1 > $O/a.json 2> $O/a.err; show $O/a.json "array512 plain in-flight 1"
done
