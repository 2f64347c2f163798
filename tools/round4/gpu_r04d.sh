#!/bin/bash
# round 4, call d: XCD-contiguous block order of the column FFT kernels: Poisson tests, then same-box A/B (OCN_FFT_XCD = 0 / 1) of the
# 512^3 step, the Poisson solve alone, and the per-rank cost at R = 2, 8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04d
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/ab_bench.sh "--steps 20 --warmup 5" xcd0::OCN_FFT_XCD=0 xcd1::OCN_FFT_XCD=1 > $O/ab.txt 2>&1; cat $O/ab.txt
for X in 0 1; do
  OCN_FFT_XCD=$X timeout -k 10 200 python tools/bench_poisson.py 512 > $O/poisson_xcd$X.txt 2>&1; tail -3 $O/poisson_xcd$X.txt
  for R in 2 8; do
    OCN_BDR_ONLY=c OCN_FFT_XCD=$X timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank${R}_xcd$X.txt 2>&1 || { tail -5 $O/rank${R}_xcd$X.txt; exit 1; }
    echo "XCD=$X $(grep 'C driver' $O/rank${R}_xcd$X.txt)"
  done
done
