#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03o
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_general_topologies.py -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_box.json 2> $O/bench_box.err || { tail -5 $O/bench_box.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box.json'));print('box', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d['driver'], 'strict', d.get('strict_ms_per_step'))"
for wl in config4 config5; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d.get('driver'))"
done
bash tools/profile_bench.sh r03f box 512 4 > $O/prof_box.log 2>&1; tail -2 $O/prof_box.log
cd $ROOT && python3 tools/summarize_profile.py gpurun_out/prof_r03f_box r03f 512 $((512*512*512)) 5 > $O/summary_box.log 2>&1; tail -3 $O/summary_box.log
cp profiles/r03f_512.* $O/ 2>/dev/null
head -22 profiles/r03f_512.md | cut -c1-220
