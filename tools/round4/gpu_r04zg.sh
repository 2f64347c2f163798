#!/bin/bash
# round 4, call zg: the general solver on real pairs (x Periodic next to a Bounded y / z): Poisson / general-topology / distributed tests, then
# 256^3 steps with the packed path against OCN_POISSON_PACKED=0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zg
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_general_topologies.py tests/test_gpu_physics.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for v in 1 0; do
  OCN_POISSON_PACKED=$v timeout -k 10 300 python tools/bench_general.py > $O/general_packed$v.txt 2>&1; echo "packed=$v"; tail -8 $O/general_packed$v.txt
done
