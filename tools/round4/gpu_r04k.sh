#!/bin/bash
# round 4, call k: tests of the epilogue-written strips, open boundary conditions; per-rank cost with / without the epilogue strips;
# column FFT workgroup width A/B with the XCD-contiguous block order; marker-delimited profiles of config 5 and config 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04k
mkdir -p $O
cd $ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_general_topologies.py tests/test_gpu_physics.py tests/test_gpu_fullsize_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for ES in 0 1; do
for R in 2 8; do
  OCN_DIST_EPILOGUE_STRIPS=$ES OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank${R}_es$ES.txt 2>&1 || { tail -5 $O/rank${R}_es$ES.txt; exit 1; }
  echo "EPILOGUE_STRIPS=$ES $(grep 'C driver' $O/rank${R}_es$ES.txt)"
done
done
for CB in 8 16; do
  OCN_COLFFT_CB=$CB OCN_COLFFT_CB2=$CB timeout -k 10 200 python tools/bench_poisson.py 512 > $O/poisson_cb$CB.txt 2>&1; echo "CB=$CB (all passes) $(tail -1 $O/poisson_cb$CB.txt)"
done
OCN_COLFFT_CB=16 OCN_COLFFT_CB2=8 timeout -k 10 200 python tools/bench_poisson.py 512 > $O/poisson_default.txt 2>&1; echo "default (16, fused z 8) $(tail -1 $O/poisson_default.txt)"
bash tools/profile_bench.sh r04b config5 512 4 > $O/profile5.log 2>&1; tail -2 $O/profile5.log
bash tools/profile_bench.sh r04b config4 512 3 > $O/profile4.log 2>&1; tail -2 $O/profile4.log
