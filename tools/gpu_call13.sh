#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02m; mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_physics.py tests/test_gpu_hydrostatic.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
bash tools/profile_bench.sh r02a config4 512 3
