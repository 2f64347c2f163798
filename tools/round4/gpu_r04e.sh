#!/bin/bash
# round 4, call e: bench.py after its restructuring (roofline object, watchdog sync, markers, comm stats over a one-rank RCCL world),
# then the rocprofv3 passes of the 512^3 step with marker-delimited accounting
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04e
mkdir -p $O
cd $ROOT
timeout -k 10 500 python bench.py > $O/bench_box.json 2> $O/bench_box.err || { tail -5 $O/bench_box.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box.json'));print('box', round(d['ms_per_step'],2),'ms', d['roofline']['frac'], d['roofline']['valu']['frac'], d['step_roofline'])"
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strict > $O/bench_rccl1.json 2> $O/bench_rccl1.err || { tail -5 $O/bench_rccl1.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_rccl1.json'));print('rccl world 1', round(d['ms_per_step'],2),'ms', d['config']['rccl'], d['comm_stats'])"
bash tools/profile_bench.sh r04a box 512 4 > $O/profile.log 2>&1; tail -3 $O/profile.log
