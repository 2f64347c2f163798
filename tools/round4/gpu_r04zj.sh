#!/bin/bash
# round 4, call zj: the closed box with a stretched z on reals (templated Thomas sweep): solver / model / distributed tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zj
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_general_topologies.py tests/test_gpu_physics.py tests/test_gpu_distributed.py tests/test_gpu_hydrostatic.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
