#!/bin/bash
# Runs ON THE GPU BOX: the three bench lines + the rocprofv3 passes of each workload (tools/profile_bench.sh).  Usage: tools/profile_all.sh <tag>
set -o pipefail
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/bench_$TAG; mkdir -p $O
cd $ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_box.json 2> $O/bench_box.err; echo "bench box rc=$?"
timeout -k 10 400 python bench.py --workload config5 --steps 20 --warmup 5 > $O/bench_config5.json 2> $O/bench_config5.err; echo "bench config5 rc=$?"
timeout -k 10 400 python bench.py --workload config4 --steps 10 --warmup 3 > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench config4 rc=$?"
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict > $O/bench_box_rccl_world1.json 2> $O/bench_box_rccl_world1.err; echo "bench rccl1 rc=$?"
bash tools/profile_bench.sh $TAG box 512 4 && bash tools/profile_bench.sh $TAG config5 512 4 && bash tools/profile_bench.sh $TAG config4 512 3
