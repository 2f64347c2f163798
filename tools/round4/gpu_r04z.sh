#!/bin/bash
# round 4, call z: real y transforms of the slab pipeline with their loads batched (source values in two groups of 44, the spectrum rows of
# the inverse all in flight) against the previous kernels: per rank at R = 2, 4, 8, two repetitions; then the distributed tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04z
mkdir -p $O
cd $ROOT
for rep in 1 2; do
for v in new old; do
  L=""; [ $v = old ] && L="$ROOT/ab/lib_realy_old.so"
  for R in 2 4 8; do
    OCN_LIB_PATH=$L OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank$R.$v.$rep.txt 2>&1 || { tail -5 $O/rank$R.$v.$rep.txt; exit 1; }
    echo "$v rep$rep $(grep 'C driver' $O/rank$R.$v.$rep.txt)"
  done
done
done
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fullsize_distributed.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
