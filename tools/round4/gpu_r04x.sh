#!/bin/bash
# round 4, call x: FourierTridiagonalPoissonSolver on grids with a Bounded / Flat x or y (stretched z under walls)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04x
mkdir -p $O
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_model.py -x -q -k "fourier_tridiagonal or stretched_z_under_walls or poisson" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
