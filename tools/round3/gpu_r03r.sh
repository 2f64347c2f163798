#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03r
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_physics.py -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for ss in 0 1 0 1; do
  OCN_SHARE_STRESSES=$ss timeout -k 10 400 python bench.py --workload config4 --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-kernel-timing > $O/bench_c4_ss$ss.json 2> $O/bench_c4_ss$ss.err || { tail -5 $O/bench_c4_ss$ss.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_c4_ss$ss.json'));print('config4 share_stresses=$ss', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d.get('driver'))"
done
