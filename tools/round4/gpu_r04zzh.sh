#!/bin/bash
# round 4, call zzh: the last inverse cosine transform of a solve on real pairs stores straight into the pressure field (no copy pass):
# parity tests, then 256^3 step and solve times with and without (OCN_POISSON_DCT_TO_FIELD=0), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzh
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_model.py tests/test_gpu_physics.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -4 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  echo "== OCN_POISSON_DCT_TO_FIELD=$v" >> $O/bench.txt
  OCN_POISSON_DCT_TO_FIELD=$v timeout -k 10 300 python tools/bench_general.py 256 10 PPB,PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
done
cat $O/bench.txt
