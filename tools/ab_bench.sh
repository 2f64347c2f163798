#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants / environment switches on the SAME box (box-to-box variance is ~7%, larger than most
# kernel changes).  The library under test is selected with OCN_LIB_PATH (oceananigans.jl_amd/_lib.py): lib/libocn_hip.so is never
# overwritten, so a failed run leaves nothing foreign behind.
# Usage: tools/ab_bench.sh "<bench args>" name[:lib.so[:ENV=V,ENV2=V2]] ...     (empty lib = the default library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ARGS=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
for spec in "$@"; do
  IFS=: read -r n L E <<< "$spec"
  envs=$(echo "$E" | tr ',' ' ')
  [ -n "$L" ] && envs="$envs OCN_LIB_PATH=$ROOT/$L"
  env $envs timeout -k 10 280 python bench.py $ARGS --no-cpu-baseline --no-strict > gpurun_out/ab/$n.$rep.json 2> gpurun_out/ab/$n.$rep.err || { echo "$n failed"; tail -3 gpurun_out/ab/$n.$rep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab/$n.$rep.json"))
r=d.get("roofline") or {}
print("$n rep$rep ms/step %.3f kernel_ms %s plain %s" % (d["ms_per_step"], r.get("kernel_ms"), r.get("hbm",{}).get("plain",{}).get("kernel_ms")))
PY
done
done
