#!/bin/bash
# round 4, call zs: the x lines of regular channels (Periodic x) in the row kernel too (colfft.hip rowdct_kernel MODE 8: transform, division, inverse in one
# in-place pass): parity tests of the general solver, then the 256^3 step and solve times with and without (OCN_POISSON_ROW_DCT=0), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zs
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -5 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  echo "== OCN_POISSON_ROW_DCT=$v" >> $O/bench_general.txt
  OCN_POISSON_ROW_DCT=$v timeout -k 10 300 python tools/bench_general.py 256 10 2>&1 | grep "ms/step" >> $O/bench_general.txt || exit 1
done
cat $O/bench_general.txt
