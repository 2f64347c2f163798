#!/bin/bash
# round 4, call zzc: ranges (the interior / buffer split of a slab) keep the box + frames decomposition: parity tests, then one rank of a
# 256^3 channel / closed box on 4 ranks' slabs ... (per-rank timing of walls is not in tools/bench_dist_rank.py: the tests only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzc
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_distributed.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -6 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/dist_soak_walls.py 20 > $O/soak.txt 2>&1 || { tail -5 $O/soak.txt; exit 1; }
cat $O/soak.txt
