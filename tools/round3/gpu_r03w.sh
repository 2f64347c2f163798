#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03w
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -m gpu -q -x -k "one_dimensional or horizontal" > gpurun_out/r03w/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r03w/tests.log | cut -c1-220
