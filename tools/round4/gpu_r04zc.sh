#!/bin/bash
# round 4, call zc: AMD kernel templated on the number of tracers with every load of a plane issued up front, against ab/lib_amd_old.so
# (the library of the previous commit), config 4, same box, two repetitions; physics / general-topology tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zc
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--workload config4 --steps 10 --warmup 3" new old:ab/lib_amd_old.so > $O/ab_config4.txt 2>&1; cat $O/ab_config4.txt
timeout -k 10 600 python -m pytest tests/test_gpu_physics.py tests/test_gpu_general_topologies.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
