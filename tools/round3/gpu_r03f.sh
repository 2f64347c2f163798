#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03f
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize_distributed.py > $O/pytest.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest.log
for wl in config4 config5; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-strict > $O/bench_$wl.json 2> $O/bench_$wl.err; python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', round(d['ms_per_step'],2),'ms', '%.3e'%d['value'])"
done
for v in "0 2" "1 2" "2 2" "1 4" "2 4"; do set -- $v; OCN_COLFFT_PERSIST=$1 OCN_COLFFT_PERSIST_MINW=$2 timeout -k 10 120 python tools/bench_poisson.py 512 2>&1 | tail -1 | sed "s/^/persist=$1 minw=$2: /"; done
OCN_COLFFT_PERSIST=1 OCN_COLFFT_CB2=4 OCN_COLFFT_PERSIST_MINW=3 timeout -k 10 120 python tools/bench_poisson.py 512 2>&1 | tail -1 | sed "s/^/persist=1 cb2=4 minw=3: /"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --driver c > $O/bench_box_c.json 2> $O/bench_box_c.err; python3 -c "import json;d=json.load(open('$O/bench_box_c.json'));print('box c', round(d['ms_per_step'],2))"
