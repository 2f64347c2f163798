#!/bin/bash
# round 4, call zzi: the source term evaluated inside the first transform of a solve on real pairs (ocn_solve_for_pressure): parity tests,
# then 256^3 step and solve times with and without (OCN_POISSON_SOURCE_ON_LOAD=0), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzi
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_model.py tests/test_gpu_physics.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -4 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  echo "== OCN_POISSON_SOURCE_ON_LOAD=$v" >> $O/bench.txt
  OCN_POISSON_SOURCE_ON_LOAD=$v timeout -k 10 300 python tools/bench_general.py 256 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
  OCN_POISSON_SOURCE_ON_LOAD=$v timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
done
cat $O/bench.txt
