#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02f; mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rccl world1 self-copy', d['ms_per_step'], d['config']['rccl'])"
cd /tmp && export TMPDIR=/tmp
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict --no-kernel-timing > $O/trace.log 2>&1
find $O -name "*kernel_trace.csv" -size +30M -delete
