#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03z
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_general_topologies.py -m gpu -q -x > gpurun_out/r03z/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r03z/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for f in 0 1; do OCN_POISSON_FUSE_SHUFFLES=$f timeout -k 10 600 python tools/bench_general.py 256 10 2>&1 | grep -E "PBB|BBB" | sed "s/^/fuse=$f /"; done
