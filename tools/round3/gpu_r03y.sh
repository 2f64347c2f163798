#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03y
timeout -k 10 900 python examples/horizontal_convection.py > gpurun_out/r03y/horizontal_convection.log 2>&1; echo "rc=$?"; head -3 gpurun_out/r03y/horizontal_convection.log; tail -4 gpurun_out/r03y/horizontal_convection.log
