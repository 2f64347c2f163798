#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters) of one
# bench.py workload.  Usage: tools/profile_bench.sh <tag> <workload> <size> [steps]
#   -> gpurun_out/prof_<tag>_<workload>/{trace,pmc_fetch,pmc_write,pmc_sq}; condense with tools/summarize_profile.py afterwards.
# The program itself follows `--` (python3 ...), never a wrapper; --pmc passes carry no trace domain besides the kernel dispatches.
set -o pipefail
TAG=${1:-r02}; WL=${2:-box}; N=${3:-512}; STEPS=${4:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --size $N --steps $STEPS --warmup 1 --no-cpu-baseline --no-strict --no-kernel-timing"
echo "$ARGS" > $OUT/command.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
# stall attribution: where the wave cycles go.  SQ_WAVE_CYCLES ~ SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY (issue stalls;
# SQ_WAIT_INST_LDS is its LDS part) + SQ_WAIT_ANY (parked at s_waitcnt / barrier) (MI355X_MICROARCH.md, rocprofv3 PMC slots: 8 SQ counters per pass)
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_stall -- python3 $ARGS > $OUT/pmc_stall.log 2>&1 || { tail -5 $OUT/pmc_stall.log; echo "(stall pass failed, continuing)"; }
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_occ -- python3 $ARGS > $OUT/pmc_occ.log 2>&1 || { tail -5 $OUT/pmc_occ.log; echo "(occupancy pass failed, continuing)"; }
# keep only what the summary needs (the per-dispatch traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -2 $OUT/trace.log
du -sh $OUT
