#!/bin/bash
# round 4, call zzp: the final library of the round (in-wave transpose in the 512-point column FFTs, bench.py's two legs): the whole GPU
# suite, smoke, the rocprofv3 passes of the 512^3 box (r04zzzz), the three bench lines and the per-rank numbers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzp
mkdir -p $O
cd $ROOT
bash tools/gpu_suite.sh > $O/suite.log 2>&1; tail -3 $O/suite.log
grep -q "pytest rc=0" $O/suite.log || exit 1
bash tools/profile_bench.sh r04zzzz box 512 4 > $O/profile.log 2>&1; tail -1 $O/profile.log
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
