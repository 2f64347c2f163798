#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03s
mkdir -p $O
cd $ROOT
bash tools/profile_bench.sh r03h config4 512 3 > $O/prof4.log 2>&1; tail -2 $O/prof4.log
cd $ROOT && python3 tools/summarize_profile.py gpurun_out/prof_r03h_config4 r03h config4 $((512*512*256)) 4 > $O/summary4.log 2>&1
cp profiles/r03h_config4.* $O/
head -16 profiles/r03h_config4.md | cut -c1-200; grep -A8 "Where the wave" profiles/r03h_config4.md | cut -c1-160
