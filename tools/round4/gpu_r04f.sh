#!/bin/bash
# round 4, call f: tests of the C model driver's interior / buffer split, the conservative switches of the box driver, the
# begin -> all_gather -> end sequence through RCCL, the drivers' flush hand-over; then per-rank A/B at R = 8: fused source term on / off,
# config 4's term set with and without the split
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04f
mkdir -p $O
cd $ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_model.py tests/test_gpu_physics.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for FS in 0 1; do
  OCN_BDR_ONLY=c OCN_DIST_FUSED_SOURCE=$FS timeout -k 10 200 python tools/bench_dist_rank.py 512 8 30 driver > $O/rank8_fs$FS.txt 2>&1 || { tail -5 $O/rank8_fs$FS.txt; exit 1; }
  echo "FUSED_SOURCE=$FS $(grep 'C driver' $O/rank8_fs$FS.txt)"
done
for OV in 0 1; do
  OCN_DIST_GENERAL_OVERLAP=$OV timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_ov$OV.txt 2>&1 || { tail -5 $O/rank8_c4_ov$OV.txt; exit 1; }
  echo "OVERLAP=$OV"; grep 'driver4' $O/rank8_c4_ov$OV.txt
done
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -m gpu -x -q > $O/pytest_general.log 2>&1; echo "pytest general rc=$?"; tail -4 $O/pytest_general.log
for GT in 0 1; do
  OCN_GENERAL_TILED=$GT timeout -k 10 300 python tools/bench_general.py 256 5 > $O/general_gt$GT.txt 2>&1; grep "N=256" $O/general_gt$GT.txt
done
