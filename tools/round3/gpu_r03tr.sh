#!/bin/bash
# tracer kernel without default-valued conditional loads: same-box A/B on config 4 / 5, then the whole suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/ab_bench.sh "--workload config4 --steps 6 --warmup 2" ab/libH_weno4.so ab/libJ_tracer.so
bash tools/ab_bench.sh "--workload config5 --steps 6 --warmup 2" ab/libH_weno4.so ab/libJ_tracer.so
cp ab/libJ_tracer.so oceananigans.jl_amd/lib/libocn_hip.so
mkdir -p gpurun_out/r03tr
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03tr/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03tr/pytest.log
