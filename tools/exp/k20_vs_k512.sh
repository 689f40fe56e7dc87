set -u
export TMPDIR=/tmp
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu --headline-only --repeats 9 > gpurun_out/k.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/k.json')); t=d['config']['timed_region']; print('K=20  median %.3f  windows %s  strict %.0f rates %s' % (d['ms_per_step']*1e3, t['windows_us_per_step'], d['roofline']['strict_bracket_ns_per_step'], t['us_per_step_by_stream_rates']))"
  python bench.py --steps 512 --warmup 64 --no-cpu --headline-only > gpurun_out/k.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/k.json')); t=d['config']['timed_region']; print('K=512 median %.3f  windows %s' % (d['ms_per_step']*1e3, t['windows_us_per_step']))"
done
