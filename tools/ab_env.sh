#!/bin/bash
# Runs ON THE GPU BOX: the bench under several environment settings on the SAME box.
# Usage: tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b OTHER=c" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ARGS=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
idx=0
for E in "$@"; do
  idx=$((idx+1))
  env $E timeout -k 10 280 python bench.py $ARGS --no-cpu-baseline --no-strict > gpurun_out/ab/env$idx.$rep.json 2> gpurun_out/ab/env$idx.$rep.err || { echo "$E failed"; tail -3 gpurun_out/ab/env$idx.$rep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab/env$idx.$rep.json"))
print("[$E] rep$rep ms/step %.3f" % d["ms_per_step"])
PY
done
done
