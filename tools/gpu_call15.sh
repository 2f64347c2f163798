#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02o; mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
for pair in 0 1; do echo "OCN_TRACER_PAIR=$pair"; OCN_TRACER_PAIR=$pair timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 8 config5 30 2>&1 | tail -1; OCN_TRACER_PAIR=$pair timeout -k 10 300 python tools/bench_config4.py 512 256 5 2 2>&1 | tail -1; done
