#!/bin/bash
# round 4, call za: row source kernel of the box pipeline with its loads batched (two groups of four cell pairs) against the previous build
# (ab/lib_realy_old.so: the row and real y kernels as of r04y), same box, two repetitions; then the model / kernel tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04za
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--steps 20 --warmup 5" new old:ab/lib_realy_old.so > $O/ab_box.txt 2>&1; cat $O/ab_box.txt
bash tools/ab_bench.sh "--workload config4 --steps 10 --warmup 3" new old:ab/lib_realy_old.so > $O/ab_config4.txt 2>&1; cat $O/ab_config4.txt
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
