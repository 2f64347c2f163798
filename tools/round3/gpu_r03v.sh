#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03v
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_distributed.py -m gpu -q -x -k "poisson or Poisson or pipeline or slab or steps_match or transpose" > gpurun_out/r03v/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r03v/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
bash tools/ab_bench.sh "--steps 20 --warmup 5 --no-kernel-timing" ab/tw_base.so ab/tw_pad.so
for L in ab/tw_base.so ab/tw_pad.so; do cp $L oceananigans.jl_amd/lib/libocn_hip.so; python tools/bench_poisson.py 512 | tail -1 | cut -c1-90; python tools/bench_poisson.py 512 | tail -1 | cut -c1-90; done
