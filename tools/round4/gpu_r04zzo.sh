#!/bin/bash
# round 4, call zzo: a third workgroup of the fused z pass per CU (the cross-wave exchange as real parts, then imaginary parts, through
# half the LDS; 80 VGPRs by launch bounds): parity tests of the variant (ab/lib_fft_split.so = HEAD's sources with
# -DOCN_FFT_SPLIT_EXCHANGE=1), then same-box A/B of the 512^3 solve and of the step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzo
mkdir -p $O
cd $ROOT
OCN_LIB_PATH=$ROOT/ab/lib_fft_split.so timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q > $O/pytest_split.log 2>&1; rc=$?; echo "pytest(split) rc=$rc"; tail -5 $O/pytest_split.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for v in default split; do
    L=""; [ $v = split ] && L="OCN_LIB_PATH=$ROOT/ab/lib_fft_split.so"
    env $L timeout -k 10 120 python tools/bench_poisson.py 512 2>&1 | grep poisson | sed "s/^/$v $rep: /" | cut -c1-100 | tee -a $O/poisson_ab.txt
  done
done
bash tools/ab_bench.sh "--steps 20 --warmup 5" default split:ab/lib_fft_split.so 2>&1 | tee $O/ab_box.txt
