#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03soak
timeout -k 10 1000 python tools/soak.py 300 > gpurun_out/r03soak/soak.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r03soak/soak.log | cut -c1-220
