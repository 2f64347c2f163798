#!/bin/bash
# round 4, call c: (1) distributed + model tests with the two-pass x solve, per-grid math mode, config 1 at its real size,
# (2) same-box A/B of the flagship variants, (3) per-rank cost at R = 2, 4, 8 (replica transport) with 16- and 32-column real y transforms
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04c
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/ab_bench.sh "--steps 20 --warmup 5" nosel:ab/lib_nosel.so sel:ab/lib_sel.so nosel_tall:ab/lib_nosel.so:OCN_TEND_TALL=1 sel_tall:ab/lib_sel.so:OCN_TEND_TALL=1 > $O/ab.txt 2>&1; cat $O/ab.txt
for CB in 16 32; do
for R in 2 4 8; do
  OCN_REALY_CB=$CB timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank${R}_cb$CB.txt 2>&1 || { tail -5 $O/rank${R}_cb$CB.txt; exit 1; }
  echo "CB=$CB $(grep 'C driver' $O/rank${R}_cb$CB.txt)"
done
done
