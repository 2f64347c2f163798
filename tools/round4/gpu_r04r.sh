#!/bin/bash
# round 4, call r: real y transforms of the slab pipeline with 8 columns per workgroup (four workgroups per CU) against 16, per rank
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04r
mkdir -p $O
cd $ROOT
timeout -k 10 300 env OCN_REALY_CB=8 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "transpose_free or library_transport_ranks" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do
for CB in 16 8; do
for R in 2 8; do
  OCN_REALY_CB=$CB OCN_BDR_ONLY=c timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank${R}_cb$CB.$rep.txt 2>&1 || { tail -5 $O/rank${R}_cb$CB.$rep.txt; exit 1; }
  echo "rep$rep REALY_CB=$CB $(grep 'C driver' $O/rank${R}_cb$CB.$rep.txt)"
done
done
done
