#!/bin/bash
# round 4, call h: general-topology suite again (lateral array / function conditions, box decomposition), replica all-to-all variants
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04h
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -m gpu -x -q > $O/pytest_general.log 2>&1; echo "pytest general rc=$?"; tail -4 $O/pytest_general.log
for B in zeros own; do
  OCN_REPLICA_A2A_BACK=$B OCN_BDR_ONLY=c timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_$B.txt 2>&1; echo "BACK=$B"; grep 'driver4' $O/rank8_c4_$B.txt
done
OCN_REPLICA_A2A_BACK=own OCN_DIST_GENERAL_OVERLAP=0 timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_own_ov0.txt 2>&1; echo "BACK=own OVERLAP=0"; grep 'driver4' $O/rank8_c4_own_ov0.txt
