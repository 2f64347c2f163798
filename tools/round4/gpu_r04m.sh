#!/bin/bash
# round 4, call m: kernel traces of one rank of 2, 4 and 8 (C driver over the replica transport, final library) and the rocprofv3 passes of
# the 512^3 step with the final library; the distributed driver tests once more
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04m
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "c_distributed_driver" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for R in 2 4 8; do
  OCN_BDR_ONLY=c timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace$R -o dr$R -- python3 $ROOT/tools/bench_dist_rank.py 512 $R 20 driver > $O/trace$R.log 2>&1 || { tail -5 $O/trace$R.log; exit 1; }
  DB=$(find $O/trace$R -name "*.db" | head -1)
  python3 $ROOT/tools/summarize_rocpd.py $DB $O/dist_rank${R}.md "one rank of $R, 512^3 box, C driver over the replica transport (OCN_BDR_ONLY=c tools/bench_dist_rank.py 512 $R 20 driver; 5 warm-up + 20 timed steps + set!)" > /dev/null 2>&1
  find $O/trace$R -name "*.db" -size +30M -delete
  grep 'C driver' $O/trace$R.log
done
cd $ROOT
bash tools/profile_bench.sh r04z box 512 4 > $O/profile.log 2>&1; tail -2 $O/profile.log
