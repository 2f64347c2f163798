#!/bin/bash
# round 4, call s: long runs with the final library (300 steps of the three workloads; distributed soak through the library transport)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04s
mkdir -p $O
cd $ROOT
timeout -k 10 1000 python tools/soak.py 300 > $O/soak.log 2>&1; echo "soak rc=$?"; grep -v Warn $O/soak.log | tail -6 | cut -c1-230
