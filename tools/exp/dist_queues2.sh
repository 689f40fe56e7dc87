# the N > 1 branch with one rank: default environment (bench.py picks 16 queues) and a forced 8 (the in-step guard has to fire)
set -u
export TMPDIR=/tmp
O=gpurun_out/skew2; mkdir -p $O
show() { python3 -c "
import json
d=json.load(open('$1')); t=d['config']['timed_region']
print('$2', 'us/step %.3f' % (d['ms_per_step']*1e3), 'frac', d['roofline']['frac'], 'in flight', d['roofline'].get('launches_in_flight', d['config'].get('launches_in_flight')), 'in step', t.get('streams_in_step'), 'spread', t.get('start_event_spread_us'), 'queues', d['config'].get('hip_runtime_env'))
"; }
BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/atlas_default.json 2> $O/atlas_default.err; show $O/atlas_default.json "atlas4096 dist default"
GPU_MAX_HW_QUEUES=8 BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/atlas_q8.json 2> $O/atlas_q8.err; show $O/atlas_q8.json "atlas4096 dist q8"
BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 > $O/a512_default.json 2> $O/a512_default.err; show $O/a512_default.json "array512 dist default"
GPU_MAX_HW_QUEUES=8 BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 > $O/a512_q8.json 2> $O/a512_q8.err; show $O/a512_q8.json "array512 dist q8"
python bench.py --config array512 --steps 20 --warmup 3 > $O/a512_plain.json 2> $O/a512_plain.err; show $O/a512_plain.json "array512 plain"
tail -2 $O/*.err | head -40
