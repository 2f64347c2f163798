#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03i
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py -m gpu -q > $O/general.log 2>&1; echo "general rc=$?"; tail -40 $O/general.log | cut -c1-220
