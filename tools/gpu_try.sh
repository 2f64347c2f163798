#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for x in 0 1; do echo "OCN_XCD_REMAP=$x"; OCN_XCD_REMAP=$x timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 8 config5 30 2>&1 | tail -1; OCN_XCD_REMAP=$x timeout -k 10 300 python tools/bench_config4.py 512 256 5 2 2>&1 | tail -1; done
