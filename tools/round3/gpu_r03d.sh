#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03d
mkdir -p $O
cd $ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_model.py -m gpu -x -q -k "driver or c_host" > $O/driver.log 2>&1; echo "driver rc=$?"; tail -15 $O/driver.log
timeout -k 10 300 python tools/bench_dist_rank.py 512 8 60 driver > $O/dr8_driver.log 2>&1; tail -2 $O/dr8_driver.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --driver c > $O/bench_c.json 2> $O/bench_c.err; cut -c1-400 $O/bench_c.json
OCN_DRIVER_DEFER_CORRECTION=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --driver c --no-strict > $O/bench_c_nodefer.json 2> $O/bench_c_nodefer.err; cut -c1-400 $O/bench_c_nodefer.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict > $O/bench_py.json 2> $O/bench_py.err; cut -c1-400 $O/bench_py.json
