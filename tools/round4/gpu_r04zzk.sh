#!/bin/bash
# round 4, call zzk: (Periodic, Periodic, Bounded) with a regular z: cosine transform, division and inverse in one column pass instead of the
# Thomas sweep: solver / model / physics tests, then step times with and without (OCN_POISSON_DCT_Z=0), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzk
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_model.py tests/test_gpu_physics.py tests/test_gpu_kernels.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -6 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  echo "== OCN_POISSON_DCT_Z=$v" >> $O/bench.txt
  OCN_POISSON_DCT_Z=$v timeout -k 10 300 python tools/bench_general.py 256 10 PPP,PPB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
done
cat $O/bench.txt
