#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02i; mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -2
for cfg in "0 16" "1 16" "1 8" "1 4"; do set -- $cfg; echo "concurrent=$1 minkz=$2"; OCN_DIST_CONCURRENT_STRIPS=$1 OCN_STRIP_MIN_KZ=$2 timeout -k 10 300 python tools/bench_dist_rank.py 512 8 10 box 2>&1 | tail -1; done
OCN_DIST_CONCURRENT_STRIPS=1 timeout -k 10 300 python tools/bench_dist_rank.py 512 8 6 config4amd 2>&1 | tail -1
