#!/bin/bash
# round 4, call t: LDS row padding of the 17 x 15 patches (slab tendencies), per rank at R = 2 and 8, two repetitions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04t
mkdir -p $O
cd $ROOT
OCN_LIB_PATH=$ROOT/ab/lib_pad1.so timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do
for P in 0 1 2 3; do
for R in 8 2; do
  OCN_LIB_PATH=$ROOT/ab/lib_pad$P.so OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank${R}_pad$P.$rep.txt 2>&1 || { tail -5 $O/rank${R}_pad$P.$rep.txt; exit 1; }
  echo "rep$rep PAD=$P $(grep 'C driver' $O/rank${R}_pad$P.$rep.txt)"
done
done
done
