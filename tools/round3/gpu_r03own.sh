#!/bin/bash
# own shared fluxes kept in registers (6 LDS reads fewer per plane): same-box A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/ab_bench.sh "--workload box --size 512 --steps 10 --warmup 3" ab/libJ.so ab/libN_ownflux.so
bash tools/ab_bench.sh "--workload config4 --steps 6 --warmup 2" ab/libJ.so ab/libN_ownflux.so
