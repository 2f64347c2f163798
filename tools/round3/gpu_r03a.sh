#!/bin/bash
# round 3, call a: full-size R = 8 tests on today's code, whole suite, per-rank baseline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03a
mkdir -p $O
cd $ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_fullsize_distributed.py -m gpu -x -q --durations=8 > $O/fullsize_dist.log 2>&1; echo "fullsize rc=$?"; tail -15 $O/fullsize_dist.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize_distributed.py > $O/pytest.log 2>&1; echo "suite rc=$?"; tail -5 $O/pytest.log
timeout -k 10 200 python tools/bench_dist_rank.py 512 8 60 box > $O/dr8_box.log 2>&1; tail -1 $O/dr8_box.log
timeout -k 10 200 python tools/bench_dist_rank.py 512 8 40 config4amd > $O/dr8_c4.log 2>&1; tail -1 $O/dr8_c4.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_box.json 2> $O/bench_box.err; cat $O/bench_box.json | cut -c1-300
