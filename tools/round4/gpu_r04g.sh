#!/bin/bash
# round 4, call g: config 4's term set per rank with / without the interior-buffer split, general-topology suite with the interior box,
# 256^3 grids with walls (box + frames against the per-cell kernel), flagship LDS-select variants A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04g
mkdir -p $O
cd $ROOT
for OV in 0 1; do
  OCN_DIST_GENERAL_OVERLAP=$OV timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_ov$OV.txt 2>&1 || { tail -5 $O/rank8_c4_ov$OV.txt; exit 1; }
  echo "OVERLAP=$OV"; grep 'driver4' $O/rank8_c4_ov$OV.txt
done
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -m gpu -x -q > $O/pytest_general.log 2>&1; echo "pytest general rc=$?"; tail -4 $O/pytest_general.log
for GT in 0 1; do
  OCN_GENERAL_TILED=$GT timeout -k 10 300 python tools/bench_general.py 256 5 > $O/general_gt$GT.txt 2>&1; grep "N=256" $O/general_gt$GT.txt
done
bash tools/ab_bench.sh "--steps 20 --warmup 5" sel0:ab/lib_sel0.so sel1:ab/lib_sel1.so sel2:ab/lib_sel2.so > $O/ab.txt 2>&1; cat $O/ab.txt
