#!/bin/bash
# direct all-gather: distributed tests (local transport R = 2..8, RCCL world 1) + full-size distributed
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r03ag
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fullsize_distributed.py -m gpu -x -q > gpurun_out/r03ag/pytest.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03ag/pytest.log
timeout -k 10 200 python tools/bench_dist_rank.py 512 8 30 driver > gpurun_out/r03ag/rank8.txt 2>&1; tail -3 gpurun_out/r03ag/rank8.txt
