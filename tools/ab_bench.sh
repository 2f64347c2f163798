#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants on the SAME box (box-to-box variance is ~7%, larger than most kernel changes).
# Usage: tools/ab_bench.sh "<bench args>" ab/libA.so ab/libB.so ...   (each is copied over lib/libocn_hip.so in turn;
# the first is restored at the end)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ARGS=$1; shift
mkdir -p gpurun_out/ab
cp oceananigans.jl_amd/lib/libocn_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for L in "$@"; do
  cp $L oceananigans.jl_amd/lib/libocn_hip.so
  n=$(basename $L .so)
  timeout -k 10 280 python bench.py $ARGS --no-cpu-baseline --no-strict > gpurun_out/ab/$n.$rep.json 2> gpurun_out/ab/$n.$rep.err || { echo "$n failed"; tail -3 gpurun_out/ab/$n.$rep.err; cp /tmp/lib_keep.so oceananigans.jl_amd/lib/libocn_hip.so; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab/$n.$rep.json"))
r=d.get("roofline") or {}
print("$n rep$rep ms/step %.3f kernel_ms %s plain %s" % (d["ms_per_step"], r.get("kernel_ms"), r.get("hbm",{}).get("plain",{}).get("kernel_ms")))
PY
done
done
cp /tmp/lib_keep.so oceananigans.jl_amd/lib/libocn_hip.so
