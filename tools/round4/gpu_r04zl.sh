#!/bin/bash
# round 4, call zl: patch shapes of the correction-on-load tendency kernel at 512^3: 32 x 8 (default), 16 x 16, 17 x 15; same box, two repetitions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zl
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--steps 20 --warmup 5" t32x8 t17x15::OCN_PC_TILE=17 t32x8b t17x15b::OCN_PC_TILE=17 > $O/ab_tiles.txt 2>&1; cat $O/ab_tiles.txt
