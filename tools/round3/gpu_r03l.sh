#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03l
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "model_driver or c_distributed_driver" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_hydrostatic.py -m gpu -q -x > $O/hydro.log 2>&1; rc=$?; echo "hydro rc=$rc"; tail -5 $O/hydro.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_dist_rank.py 512 8 10 driver4 > $O/driver4.log 2>&1; tail -3 $O/driver4.log
timeout -k 10 400 python tools/bench_dist_rank.py 512 8 20 driver > $O/driver.log 2>&1; tail -3 $O/driver.log
