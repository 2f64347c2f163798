#!/bin/bash
# round 4, call zk: long runs with the final library (300 steps of the three workloads; distributed soak through the library transport)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zk
mkdir -p $O
cd $ROOT
timeout -k 10 1000 python tools/soak.py 300 > $O/soak.log 2>&1; echo "soak rc=$?"; grep -v Warn $O/soak.log | tail -6 | cut -c1-230
timeout -k 10 300 python tools/dist_soak.py > $O/dist_soak.log 2>&1; tail -1 $O/dist_soak.log
timeout -k 10 600 python tools/dist_soak_walls.py 200 > $O/dist_soak_walls.log 2>&1; tail -2 $O/dist_soak_walls.log
