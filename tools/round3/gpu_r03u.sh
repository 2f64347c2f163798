#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03u
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "hydrostatic" > gpurun_out/r03u/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -30 gpurun_out/r03u/tests.log | cut -c1-250
