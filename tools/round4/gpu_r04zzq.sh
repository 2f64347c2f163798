#!/bin/bash
# round 4, call zzq: long runs with the final library (300 steps of the 512^3 box through bench.py: the in-wave FFT exchange; tools/soak.py 300;
# the distributed soak) and the rocprofv3 passes of config 4 and config 5 (r04zzzz)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzq
mkdir -p $O
cd $ROOT
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-strict > $O/bench_box_300.json 2> $O/bench_box_300.err || { tail -5 $O/bench_box_300.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box_300.json'));print('box 300 steps', round(d['ms_per_step'],2),'ms finite', d['config']['finite'], d['config']['state_checksum']['sum_of_squares'])"
timeout -k 10 900 python tools/soak.py 300 > $O/soak.log 2>&1; echo "soak rc=$?"; grep -v Warn $O/soak.log | tail -6 | cut -c1-230
timeout -k 10 300 python tools/dist_soak.py > $O/dist_soak.log 2>&1; tail -1 $O/dist_soak.log | cut -c1-230
bash tools/profile_bench.sh r04zzzz config4 512 3 2>&1 | tail -1 && bash tools/profile_bench.sh r04zzzz config5 512 4 2>&1 | tail -1
