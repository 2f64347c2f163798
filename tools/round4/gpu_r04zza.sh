#!/bin/bash
# round 4, call zza: the wall frames of every per-cell kernel in ONE launch (OCN_GENERAL_FRAMES=separate: one launch per frame): parity
# tests, then step times both ways, same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zza
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py tests/test_gpu_distributed.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -6 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for v in merged separate merged separate; do
  echo "== OCN_GENERAL_FRAMES=$v" >> $O/bench.txt
  OCN_GENERAL_FRAMES=$v timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
  OCN_GENERAL_FRAMES=$v timeout -k 10 300 python tools/bench_general.py 256 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
done
cat $O/bench.txt
