#!/bin/bash
# z-chunk length of the tendency launch on a slab (one rank of 8): OCN_TEND_MIN_BLOCKS sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03mb
for mb in 8192 6144 4096 3072 2048 1536; do
  OCN_TEND_MIN_BLOCKS=$mb timeout -k 10 200 python tools/bench_dist_rank.py 512 8 30 driver > gpurun_out/r03mb/mb$mb.txt 2>&1 || { tail -3 gpurun_out/r03mb/mb$mb.txt; exit 1; }
  echo "min_blocks $mb: $(grep 'C driver' gpurun_out/r03mb/mb$mb.txt)"
done
