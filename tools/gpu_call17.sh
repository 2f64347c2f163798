#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02q; mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_physics.py tests/test_gpu_hydrostatic.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 8 config5 30 2>&1 | tail -1
timeout -k 10 300 python tools/bench_config4.py 512 256 5 2 2>&1 | tail -1
