set -u
export TMPDIR=/tmp
O=gpurun_out/probe; mkdir -p $O
show() { python3 -c "
import json
d=json.load(open('$1')); t=d['config']['timed_region']
print('$2', 'us/step %.3f' % (d['ms_per_step']*1e3), 'in step', t.get('streams_in_step'), 'spread', t.get('start_event_spread_us'), d['config'].get('hip_runtime_env', {}).get('streams_on_one_hardware_queue_max', t.get('streams_on_one_hardware_queue_max')))
"; }
python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "probe or dist_branch" 2>&1 | tail -3
for q in 4 8; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/a.json 2> $O/a.err; show $O/a.json "plain q$q"; done
for q in 8 16; do GPU_MAX_HW_QUEUES=$q BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/a.json 2> $O/a.err; show $O/a.json "dist q$q"; done
for q in 8 16; do GPU_MAX_HW_QUEUES=$q BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "a512 dist q$q"; done
