#!/bin/bash
# round 4, call zo: 17 x 15 patches for the tracer kernel (OCN_TRACER_TILE=17) on config 4 and config 5, same box, two repetitions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zo
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--workload config4 --steps 10 --warmup 3" t32 t17::OCN_TRACER_TILE=17 > $O/ab_config4.txt 2>&1; cat $O/ab_config4.txt
bash tools/ab_bench.sh "--workload config5 --steps 10 --warmup 3" t32 t17::OCN_TRACER_TILE=17 > $O/ab_config5.txt 2>&1; cat $O/ab_config5.txt
