#!/bin/bash
# round 4, call zx: fused stage boundaries on grids with walls (substep epilogues on the interior box, finishing kernels on the frames):
# parity tests (fused against unfused bit for bit, models against the oracle), then the channel / closed-box step times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zx
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_physics.py tests/test_gpu_general_topologies.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -25 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 > $O/bench_terms.txt 2>&1 || { tail -20 $O/bench_terms.txt; exit 1; }
cat $O/bench_terms.txt
timeout -k 10 300 python tools/bench_general.py 256 10 2>&1 | grep "ms/step" > $O/bench_general.txt || exit 1
cat $O/bench_general.txt
