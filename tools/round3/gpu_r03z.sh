#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03z
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -m gpu -q -x -k "cosine" > gpurun_out/r03z/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r03z/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc

