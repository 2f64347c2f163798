#!/bin/bash
# round 4, call zzj: long runs of the fused wall paths: 400 steps of config 4's term set on a 128 x 128 x 64 channel and closed box (finite?),
# 300 steps of the plain 128^3 channel / closed box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzj
mkdir -p $O
cd $ROOT
timeout -k 10 500 python tools/bench_general_terms.py 128 64 400 PBB,BBB > $O/soak_terms.txt 2>&1 || { tail -5 $O/soak_terms.txt; exit 1; }
grep "ms/step" $O/soak_terms.txt
timeout -k 10 500 python tools/bench_general.py 128 300 PBB,BBB > $O/soak_plain.txt 2>&1 || { tail -5 $O/soak_plain.txt; exit 1; }
grep "ms/step" $O/soak_plain.txt
