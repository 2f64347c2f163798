#!/bin/bash
# round 4, call zzd: the C RK3 driver on grids with walls: tests of the driver, the C-ABI library tests, the whole suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzd
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "driver" > $O/tests.txt 2>&1; rc=$?
tail -12 $O/tests.txt
exit $rc
