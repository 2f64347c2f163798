#!/bin/bash
# round 4, call zh: rocprofv3 passes of the three workloads with the final library (r04zz: marker-delimited traffic of the timed steps)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_bench.sh r04zz box 512 4 2>&1 | tail -2 && bash tools/profile_bench.sh r04zz config4 512 3 2>&1 | tail -2 && bash tools/profile_bench.sh r04zz config5 512 4 2>&1 | tail -2
