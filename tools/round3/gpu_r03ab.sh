#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/ab_bench.sh "--steps 20 --warmup 5 --no-kernel-timing" ab/before.so ab/now.so
