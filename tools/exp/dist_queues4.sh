set -u
export TMPDIR=/tmp
O=gpurun_out/skew4; mkdir -p $O
show() { python3 -c "
import json
d=json.load(open('$1')); t=d['config']['timed_region']
print('$2', 'us/step %.3f' % (d['ms_per_step']*1e3), 'frac', d['roofline']['frac'], 'in step', t.get('streams_in_step'), 'spread', t.get('start_event_spread_us'))
"; }
for L in 16 64 128 256; do
BENCH_ARRAY_LEAD=$L BENCH_FORCE_DIST=1 python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "array512 dist lead $L"
BENCH_ARRAY_LEAD=$L python bench.py --config array512 --steps 20 --warmup 3 > $O/a.json 2> $O/a.err; show $O/a.json "array512 plain lead $L"
done
