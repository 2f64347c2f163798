#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03h
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_physics.py tests/test_gpu_hydrostatic.py -m gpu -x -q > $O/pytest.log 2>&1; echo "physics+hydro rc=$?"; tail -3 $O/pytest.log
for wl in config4 config5; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-strict > $O/bench_$wl.json 2> $O/bench_$wl.err; python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace4 -o c4 -- python3 $ROOT/bench.py --workload config4 --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-kernel-timing > $O/trace4.log 2>&1
cd $ROOT; python3 tools/summarize_rocpd.py $(find $O/trace4 -name "*.db" | head -1) $O/c4_summary.md "config4" > /dev/null 2>&1; head -16 $O/c4_summary.md | cut -c1-150
find $O -name "*.db" -size +30M -delete
