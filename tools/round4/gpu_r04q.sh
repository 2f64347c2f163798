#!/bin/bash
# round 4, call q: tracer kernel with address-selected LDS stencils and the T + S pair kernel, same-box A/B on config 4 and config 5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04q
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--workload config4 --steps 8 --warmup 2" base:ab/lib_base.so trsel:ab/lib_trsel.so pair:ab/lib_base.so:OCN_TRACER_PAIR=1 > $O/ab_config4.txt 2>&1; cat $O/ab_config4.txt
bash tools/ab_bench.sh "--workload config5 --steps 10 --warmup 3" base:ab/lib_base.so trsel:ab/lib_trsel.so > $O/ab_config5.txt 2>&1; cat $O/ab_config5.txt
