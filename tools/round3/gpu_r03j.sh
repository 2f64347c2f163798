#!/bin/bash
# after the fast-math WENO5 rewrite: profiles of the three workloads (kernel trace + PMC passes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03j
mkdir -p $O
cd $ROOT
bash tools/profile_bench.sh r03j box 512 4 > $O/prof_box.log 2>&1; tail -2 $O/prof_box.log
bash tools/profile_bench.sh r03j config4 512 3 > $O/prof4.log 2>&1; tail -2 $O/prof4.log
bash tools/profile_bench.sh r03j config5 512 3 > $O/prof5.log 2>&1; tail -2 $O/prof5.log
