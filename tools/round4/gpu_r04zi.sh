#!/bin/bash
# round 4, call zi: soak of the wall-bounded distributed paths (channel and closed box with the LES term set on 4 ranks, 60 steps)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zi
mkdir -p $O
cd $ROOT
timeout -k 10 600 python tools/dist_soak_walls.py 60 > $O/soak.txt 2>&1; echo "rc=$?"; tail -6 $O/soak.txt
