#!/bin/bash
# round-2 GPU call 3: whole GPU suite, the three bench lines, RCCL one-rank bench, profiles (box 512 + config5)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02c; mkdir -p $O
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $O/bench_box.json 2> $O/bench_box.err; echo "bench box rc=$?"; cat $O/bench_box.json
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict > $O/bench_box_rccl1.json 2> $O/bench_box_rccl1.err; echo "bench rccl1 rc=$?"; cat $O/bench_box_rccl1.json; tail -3 $O/bench_box_rccl1.err
timeout -k 10 400 python bench.py --workload config5 --steps 10 --warmup 3 > $O/bench_config5.json 2> $O/bench_config5.err; echo "bench config5 rc=$?"; cat $O/bench_config5.json
bash tools/profile_bench.sh r02a box 512 4 && bash tools/profile_bench.sh r02a config5 512 4
