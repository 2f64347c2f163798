#!/bin/bash
# fast-math AMD kernel: averages with one factor 1/4 (Q), plus the filter-width ratios folded into the derivative's reciprocal spacing (P)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/ab_bench.sh "--workload config4 --steps 6 --warmup 2" ab/libO.so ab/libQ_amd_i4.so ab/libP_amd.so
