#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02b; mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_hydrostatic.py -m gpu -q > $O/pytest_hyd.log 2>&1; echo "pytest hyd rc=$?"
tail -15 $O/pytest_hyd.log
OCN_HYDRO_FUSED=0 timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 6 config5 30 2>&1 | tail -1
timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 6 config5 30 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ROOT/tools/bench_hydrostatic.py 1024 128 6 config5 30 > $O/trace.log 2>&1
echo "rocprof rc=$?"
find $O -name "*kernel_trace.csv" -size +30M -delete
