#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02p; mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_physics.py -m gpu -x -q -k "pair or function" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
