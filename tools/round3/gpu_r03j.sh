#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03j
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_model.py -m gpu -q > $O/general.log 2>&1; echo "general rc=$?"; tail -30 $O/general.log | cut -c1-220
for nt in; do OCN_NARROW_TILE=$nt timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict > $O/bench_nt$nt.json 2> $O/bench_nt$nt.err; python3 -c "import json;d=json.load(open('$O/bench_nt$nt.json'));print('narrow_tile=$nt', round(d['ms_per_step'],2), d['driver'], d['roofline']['kernel_ms'])"; done
