#!/bin/bash
# Runs ON THE GPU BOX: the whole -m gpu suite with its slowest tests listed.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/suite
cd $ROOT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/suite/pytest.log 2>&1; echo "pytest rc=$?"; tail -24 gpurun_out/suite/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
