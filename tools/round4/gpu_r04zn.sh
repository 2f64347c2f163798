#!/bin/bash
# round 4, call zn: z-chunk length of the tendency launches (OCN_TEND_MIN_BLOCKS: the launch is cut along z until it has that many workgroups) with
# the 17 x 15 patches, 512^3 box, same box, two repetitions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zn
mkdir -p $O
cd $ROOT
bash tools/ab_bench.sh "--steps 20 --warmup 5" b8192 b4096::OCN_TEND_MIN_BLOCKS=4096 b16384::OCN_TEND_MIN_BLOCKS=16384 b2048::OCN_TEND_MIN_BLOCKS=2048 b6144::OCN_TEND_MIN_BLOCKS=6144 > $O/ab_blocks.txt 2>&1; cat $O/ab_blocks.txt
