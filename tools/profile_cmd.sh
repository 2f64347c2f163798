#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes of tools/profile_bench.sh for ANY python command of this repository.
# Usage: tools/profile_cmd.sh <tag> <script and args relative to the repository root ...>   [ENV=VALUE pairs come from the caller's environment]
#   -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq,pmc_stall,pmc_occ}; condense with tools/summarize_profile.py.
# The program itself follows `--` (python3 ...), never a wrapper; --pmc passes carry no trace domain besides the kernel dispatches.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
echo "$SCRIPT $*" > $OUT/command.txt
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $SCRIPT $ARGS > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
ARGS="$*"
run trace --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
run pmc_sq --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY || exit 1
run pmc_stall --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || echo "(stall pass failed, continuing)"
run pmc_occ --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS || echo "(occupancy pass failed, continuing)"
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
