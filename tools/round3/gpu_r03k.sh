#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03k
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for drv in python c; do timeout -k 10 400 python bench.py --workload config4 --driver $drv --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-kernel-timing > $O/bench_c4_$drv.json 2> $O/bench_c4_$drv.err || { tail -5 $O/bench_c4_$drv.err; exit 1; }; python3 -c "import json;d=json.load(open('$O/bench_c4_$drv.json'));print('config4 driver=$drv', round(d['ms_per_step'],2), d['driver'], d['value'])"; done
timeout -k 10 300 python tools/enqueue_time.py > $O/enqueue.log 2>&1; tail -6 $O/enqueue.log
