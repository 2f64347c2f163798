#!/bin/bash
# round 4, call j: the whole GPU suite with the library as it stands, per-rank costs at R = 2, 4, 8, and the rocprofv3 passes (trace,
# FETCH / WRITE, SQ, stall, occupancy) of one rank of 2 -- what bounds the real y transforms of the slab pipeline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04j
mkdir -p $O
cd $ROOT
bash tools/gpu_suite.sh > $O/suite.log 2>&1; tail -4 $O/suite.log
for R in 2 4 8; do
  OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank$R.txt 2>&1 || { tail -5 $O/rank$R.txt; exit 1; }
  grep 'C driver' $O/rank$R.txt
done
OCN_BDR_ONLY=c bash tools/profile_cmd.sh r04j_rank2 tools/bench_dist_rank.py 512 2 8 driver > $O/profile.log 2>&1; tail -3 $O/profile.log
