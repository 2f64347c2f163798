#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02j; mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_hydrostatic.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 python bench.py --workload config5 --steps 10 --warmup 3 --no-cpu-baseline --no-strict 2>&1 | tail -2 | cut -c1-600
