#!/bin/bash
# round 4, call o: the replica transport against real ranks of a replicated flow; bench.py default line and a one-rank RCCL world line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04o
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "replica or c_distributed_driver" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 500 python bench.py > $O/bench_box.json 2> $O/bench_box.err || { tail -5 $O/bench_box.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box.json'));print('box', round(d['ms_per_step'],2),'ms', 'strict', d.get('strict_ms_per_step'), d['roofline']['frac'], d['roofline']['pmc_source'], d['step_roofline']['measured_over_algorithmic'])"
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rccl1.json 2> $O/bench_rccl1.err || { tail -5 $O/bench_rccl1.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_rccl1.json'));print('rccl world 1', round(d['ms_per_step'],2),'ms', d['strict_ms_per_step'], d['comm_stats'])"
