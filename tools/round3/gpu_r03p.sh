#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03p
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_physics.py tests/test_gpu_distributed.py tests/test_gpu_fullsize_distributed.py -m gpu -q -x -k "not hydrostatic and not config5" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for ef in 0 1 0 1; do
  OCN_EXTRA_FIRST=$ef timeout -k 10 400 python bench.py --workload config4 --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-kernel-timing > $O/bench_c4_ef$ef.json 2> $O/bench_c4_ef$ef.err || { tail -5 $O/bench_c4_ef$ef.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_c4_ef$ef.json'));print('config4 extra_first=$ef', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d.get('driver'), d.get('state_checksum'))"
done
