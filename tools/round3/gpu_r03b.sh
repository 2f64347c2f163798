#!/bin/bash
# round 3, call b: distributed correction-on-load stage: tests + per-rank cost (new vs old)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03b
mkdir -p $O
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_distributed.py tests/test_reference_fixtures.py -m gpu -x -q > $O/dist.log 2>&1; echo "dist rc=$?"; tail -15 $O/dist.log
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize_distributed.py -m gpu -x -q -k "config3" > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; tail -8 $O/fullsize.log
for mode in 1 0; do
  OCN_DIST_CORRECT_ON_LOAD=$mode timeout -k 10 200 python tools/bench_dist_rank.py 512 8 60 box > $O/dr8_box_col$mode.log 2>&1; tail -1 $O/dr8_box_col$mode.log
done
OCN_NARROW_TILE=0 timeout -k 10 200 python tools/bench_dist_rank.py 512 8 60 box > $O/dr8_box_wide.log 2>&1; tail -1 $O/dr8_box_wide.log
for R in 2 4; do timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 40 box > $O/dr${R}_box.log 2>&1; tail -1 $O/dr${R}_box.log; done
