#!/bin/bash
# round 4, call zv: the finishing pass (Coriolis, buoyancy, pHY', stress divergence) and the tracer diffusion of grids with walls on the
# interior box through the tiled kernels with per-field layouts: parity tests, then the config-4 term set on a channel / closed box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zv
mkdir -p $O
cd $ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py tests/test_gpu_kernels.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -8 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 > $O/bench.txt 2>&1 || { tail -20 $O/bench.txt; exit 1; }
cat $O/bench.txt
OCN_GENERAL_TILED=0 timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 PBB,BBB > $O/bench_percell.txt 2>&1 || { tail -20 $O/bench_percell.txt; exit 1; }
echo "== OCN_GENERAL_TILED=0"; cat $O/bench_percell.txt
