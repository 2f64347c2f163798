#!/bin/bash
# round 4, call zzb: the three bench lines and the per-rank numbers with the final library of the round (after the wall-grid work touched
# physics.hip / tendencies.hip / amd.hip / kernels.hip): nothing may have moved on the Periodic grids
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzb
mkdir -p $O
cd $ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_box.json 2> $O/bench_box.err || { tail -5 $O/bench_box.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box.json'));print('box', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d['driver'], 'strict', d.get('strict_ms_per_step'), 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
for wl in config4 config5; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'], d.get('driver'))"
done
for R in 2 4 8; do
  OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank$R.txt 2>&1 || { tail -5 $O/rank$R.txt; exit 1; }
  grep 'C driver' $O/rank$R.txt
done
