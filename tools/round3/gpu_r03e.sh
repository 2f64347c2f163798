#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03e
mkdir -p $O
cd $ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_model.py -m gpu -x -q -k "driver or c_host" > $O/driver.log 2>&1; echo "driver rc=$?"; tail -4 $O/driver.log
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $O/sq_counters.txt; wc -l $O/sq_counters.txt
bash tools/profile_bench.sh r03a config4 512 3 > $O/prof.log 2>&1; tail -5 $O/prof.log
cd $ROOT && python3 tools/summarize_profile.py gpurun_out/prof_r03a_config4 r03a config4 $((512*512*256)) 4 > $O/summary.log 2>&1; tail -30 profiles/r03a_config4.md
cp profiles/r03a_config4.* $O/
