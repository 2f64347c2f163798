#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# of the bench workload.  Usage: tools/profile_gpu.sh <tag> <n> [steps]
set -o pipefail
TAG=${1:-r01}; N=${2:-512}; STEPS=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${N}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --size $N --steps $STEPS --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || exit 1
# keep only what the summary needs (the per-dispatch traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -R $OUT | head -40
