#!/bin/bash
# round 4, call zp: 17 x 15 patches for the momentum kernel of config 4 (Bounded z, no correction on load: OCN_NARROW_TILE=1), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zp
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--workload config4 --steps 10 --warmup 3" t32 t17::OCN_NARROW_TILE=1 t32b t17b::OCN_NARROW_TILE=1 > $O/ab_config4.txt 2>&1; cat $O/ab_config4.txt
