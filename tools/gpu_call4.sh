#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02d; mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_hydrostatic.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for cfg in "3 3" "4 3" "3 4" "4 4"; do set -- $cfg; echo "tracer waves $1, hydro waves $2"; OCN_TRACER_WAVES=$1 OCN_HYDRO_WAVES=$2 timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 8 config5 30 2>&1 | tail -1; done
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rccl world1 self-copy', d['ms_per_step'], d['config']['rccl'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ROOT/tools/bench_hydrostatic.py 1024 128 6 config5 30 > $O/trace.log 2>&1
