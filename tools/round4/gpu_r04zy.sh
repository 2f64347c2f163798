#!/bin/bash
# round 4, call zy: step times with the fused stage boundaries on grids with walls (OCN_FUSE_WALLS=0: the reference's launch sequence), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zy
mkdir -p $O
cd $ROOT
for v in 1 0 1 0; do
  echo "== OCN_FUSE_WALLS=$v" >> $O/bench.txt
  OCN_FUSE_WALLS=$v timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
  OCN_FUSE_WALLS=$v timeout -k 10 300 python tools/bench_general.py 256 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
done
cat $O/bench.txt
