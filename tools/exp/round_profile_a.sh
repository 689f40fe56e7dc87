# round-end evidence, part A: tests, smoke, both bench protocols, the N > 1 branch with one rank (both configs)
set -u
export TMPDIR=/tmp
TAG=${TAG:-v}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; cut -c1-300 $O/bench_steps20.json
timeout 600 python bench.py --steps 512 --warmup 64 --no-cpu > $O/bench_steps512.json 2> $O/bench_steps512.err; cut -c1-200 $O/bench_steps512.json
BENCH_FORCE_DIST=1 timeout 600 python bench.py --config array512 --steps 20 --warmup 3 > $O/bench_array512_forced_dist_one_rank.json 2> $O/bench_array512.err; cut -c1-300 $O/bench_array512_forced_dist_one_rank.json
timeout 600 python bench.py --method pipeline --steps 20 --warmup 5 --no-cpu --no-live-traffic > $O/bench_steps20_method_pipeline.json 2> $O/bench_steps20_method_pipeline.err; cut -c1-200 $O/bench_steps20_method_pipeline.json
BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_atlas4096_forced_dist_one_rank.json 2> $O/bench_atlas4096_dist.err; cut -c1-200 $O/bench_atlas4096_forced_dist_one_rank.json
