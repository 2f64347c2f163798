#!/bin/bash
# round 4, call i: open boundary conditions with a value + lateral conditions (general-topology suite), config 4's term set per rank
# (one-rank RCCL world at the R = 8 local size) without / with the interior-buffer split (16-wide buffers) / with 3-wide buffers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04i
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
OCN_DIST_GENERAL_OVERLAP=0 timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_ov0.txt 2>&1; echo "OVERLAP=0"; grep 'driver4' $O/rank8_c4_ov0.txt
timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_ov1.txt 2>&1; echo "OVERLAP=1 (16-wide buffers in the C driver)"; grep 'driver4' $O/rank8_c4_ov1.txt
OCN_DIST_BUFFER_WIDTH=3 timeout -k 10 300 python tools/bench_dist_rank.py 512 8 20 driver4 > $O/rank8_c4_ov1w3.txt 2>&1; echo "OVERLAP=1, 3-wide buffers"; grep 'driver4' $O/rank8_c4_ov1w3.txt
