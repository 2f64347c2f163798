#!/bin/bash
# round 4, call zzn: the exchange between the second and third radix-8 stage of the 512-point column FFTs as an in-wave transpose
# (v_permlane32_swap / v_permlane16_swap / DPP) instead of an LDS round trip: parity tests, then same-box A/B against the LDS exchange
# (ab/lib_fft_lds.so = the same sources with -DOCN_FFT_LANE_TRANSPOSE=0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzn
mkdir -p $O
cd $ROOT
timeout -k 10 800 python -m pytest tests/test_gpu_model.py tests/test_gpu_general_topologies.py tests/test_gpu_bench_line.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for v in new lds; do
    L=""; [ $v = lds ] && L="OCN_LIB_PATH=$ROOT/ab/lib_fft_lds.so"
    env $L timeout -k 10 120 python tools/bench_poisson.py 512 2>&1 | grep poisson | sed "s/^/$v $rep: /" | cut -c1-120 | tee -a $O/poisson_ab.txt
  done
done
bash tools/ab_bench.sh "--steps 20 --warmup 5" new lds:ab/lib_fft_lds.so 2>&1 | tee $O/ab_box.txt
