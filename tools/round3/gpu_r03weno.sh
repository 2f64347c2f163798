#!/bin/bash
# fast-math WENO5 combination with 3 fewer operations: same-box A/B on the three workloads, then the tolerance tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/ab_bench.sh "--workload box --size 512 --steps 10 --warmup 3" ab/libD_weno3.so ab/libH_weno4.so
bash tools/ab_bench.sh "--workload config4 --steps 6 --warmup 2" ab/libD_weno3.so ab/libH_weno4.so
bash tools/ab_bench.sh "--workload config5 --steps 6 --warmup 2" ab/libD_weno3.so ab/libH_weno4.so
cp ab/libH_weno4.so oceananigans.jl_amd/lib/libocn_hip.so
mkdir -p gpurun_out/r03weno
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03weno/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03weno/pytest.log
