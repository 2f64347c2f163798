#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict --no-kernel-timing > $O/trace.log 2>&1
echo rc=$?
find $O -name "*kernel_trace.csv" -size +30M -delete
