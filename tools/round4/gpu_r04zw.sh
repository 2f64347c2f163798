#!/bin/bash
# round 4, call zw: the AMD kernel with per-field layouts and immediate x offsets (GEN = 1; zero strides of Flat directions only in GEN = 2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zw
mkdir -p $O
cd $ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py -x -q -m gpu -k "amd or les or AMD" > $O/tests.txt 2>&1; rc=$?
tail -5 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 > $O/bench.txt 2>&1 || { tail -20 $O/bench.txt; exit 1; }
cat $O/bench.txt
