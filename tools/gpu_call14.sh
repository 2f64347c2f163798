#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02n; mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.log
