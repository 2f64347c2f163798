#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03g
mkdir -p $O
cd $ROOT
for wl in config5 config4; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-strict > $O/bench_$wl.json 2> $O/bench_$wl.err; python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'])"
done
bash tools/profile_bench.sh r03b config4 512 3 > $O/prof4.log 2>&1; tail -2 $O/prof4.log
cd $ROOT && python3 tools/summarize_profile.py gpurun_out/prof_r03b_config4 r03b config4 $((512*512*256)) 4 > $O/summary4.log 2>&1
bash tools/profile_bench.sh r03b config5 512 3 > $O/prof5.log 2>&1; tail -2 $O/prof5.log
cd $ROOT && python3 tools/summarize_profile.py gpurun_out/prof_r03b_config5 r03b config5 $((1024*1024*128)) 4 > $O/summary5.log 2>&1
cp profiles/r03b_config4.* profiles/r03b_config5.* $O/
head -24 profiles/r03b_config4.md | cut -c1-200; grep -A12 "Where the wave" profiles/r03b_config4.md | cut -c1-160
