#!/bin/bash
# round 4, call v: latency / cache counters of the 512^3 step: average VMEM and LDS instruction latency (SQ_INST_LEVEL_* / SQ_INSTS_*),
# L2 hit rate, TA busy -- what the flagship kernel's parked wave cycles wait for
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r04v_box
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload box --size 512 --steps 4 --warmup 1 --no-cpu-baseline --no-strict --no-kernel-timing"
timeout -k 10 400 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_lat -- python3 $ARGS > $OUT/pmc_lat.log 2>&1 || { tail -5 $OUT/pmc_lat.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS > $OUT/pmc_l2.log 2>&1 || { tail -5 $OUT/pmc_l2.log; echo "(l2 pass failed)"; }
timeout -k 10 400 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- python3 $ARGS > $OUT/pmc_tcp.log 2>&1 || { tail -5 $OUT/pmc_tcp.log; echo "(tcp pass failed)"; }
du -sh $OUT
