#!/bin/bash
# final state after the fast-math WENO5 rewrite: whole GPU suite, the three bench lines, one rank of 8 (box and config 4)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03l
mkdir -p $O
cd $ROOT
bash tools/gpu_suite.sh > $O/suite.log 2>&1; tail -4 $O/suite.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_box.json 2> $O/bench_box.err || { tail -5 $O/bench_box.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box.json'));print('box', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d['driver'], 'strict', d.get('strict_ms_per_step'), d['roofline']['pmc_source'], 'kernel', d['roofline']['kernel_ms'])"
for wl in config4 config5; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d.get('driver'))"
done
timeout -k 10 200 python tools/bench_dist_rank.py 512 8 30 driver > $O/rank8_box.txt 2>&1; tail -2 $O/rank8_box.txt
timeout -k 10 200 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_config4.txt 2>&1; tail -2 $O/rank8_config4.txt
