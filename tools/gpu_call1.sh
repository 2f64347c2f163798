#!/bin/bash
# round-2 GPU call 1: full GPU suite, then the config-5 step under rocprofv3 (kernel trace + stats)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r02a
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 6 config5 30 > gpurun_out/r02a/config5_plain.log 2>&1; cat gpurun_out/r02a/config5_plain.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02a/trace -- python3 $ROOT/tools/bench_hydrostatic.py 1024 128 6 config5 30 > $ROOT/gpurun_out/r02a/trace.log 2>&1
echo "rocprof rc=$?"
find $ROOT/gpurun_out/r02a -name "*kernel_trace.csv" -size +30M -delete
ls -R $ROOT/gpurun_out/r02a | head -30
