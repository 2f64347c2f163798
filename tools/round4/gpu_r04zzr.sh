#!/bin/bash
# round 4, call zzr: the in-wave exchange also for the 256-point transforms with 16 columns per workgroup (column passes of 256^3 grids,
# real y transforms of the slab pipeline at 512^3): parity tests, then same-box A/B against ab/lib_fft_prev.so (= the previous commit's colfft.hip)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzr
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_distributed.py tests/test_gpu_general_topologies.py tests/test_gpu_fullsize_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for v in new prev; do
    L=""; [ $v = prev ] && L="OCN_LIB_PATH=$ROOT/ab/lib_fft_prev.so"
    env $L timeout -k 10 120 python tools/bench_poisson.py 256 2>&1 | grep poisson | sed "s/^/$v $rep: /" | cut -c1-90 | tee -a $O/poisson256_ab.txt
    for R in 2 8; do
      env $L OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver 2>&1 | grep 'C driver' | sed "s/^/$v $rep: /" | cut -c1-110 | tee -a $O/rank_ab.txt
    done
  done
done
