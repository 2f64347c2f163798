#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02l; mkdir -p $O
cd $ROOT
timeout -k 10 300 python tools/bench_tendency.py 512 fast 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_physics.py tests/test_gpu_hydrostatic.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('box', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['hbm']['plain']['kernel_ms'])"
timeout -k 10 300 python tools/bench_hydrostatic.py 1024 128 8 config5 30 2>&1 | tail -1
