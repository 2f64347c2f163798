#!/bin/bash
# round 4, call w: Distributed on (Periodic, Bounded, Bounded) grids -- the channel: new / extended tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04w
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py -x -q -k "amd" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
