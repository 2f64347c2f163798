#!/bin/bash
# round 4, call a: state of the tree at the start of the round (GPU suite) + kernel traces of one rank of 2 and of 4 (C driver, box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04a
mkdir -p $O
cd $ROOT
bash tools/gpu_suite.sh > $O/suite.log 2>&1; tail -4 $O/suite.log
for R in 2 4 8; do
  timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank$R.txt 2>&1 || { tail -5 $O/rank$R.txt; exit 1; }
  tail -2 $O/rank$R.txt
done
timeout -k 10 200 python tools/bench_dist_rank.py 512 1 30 box > $O/rank1.txt 2>&1; tail -1 $O/rank1.txt
cd /tmp && export TMPDIR=/tmp
for R in 2 4; do
  OCN_BDR_ONLY=c timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace$R -o dr$R -- python3 $ROOT/tools/bench_dist_rank.py 512 $R 20 driver > $O/trace$R.log 2>&1 || { tail -5 $O/trace$R.log; exit 1; }
  DB=$(find $O/trace$R -name "*.db" | head -1)
  python3 $ROOT/tools/summarize_rocpd.py $DB $O/dr${R}_summary.md "one rank of $R, 512^3 box, C driver (OCN_BDR_ONLY=c tools/bench_dist_rank.py 512 $R 20 driver; 5 warm-up + 20 timed steps)" > /dev/null 2>&1
  find $O/trace$R -name "*.db" -size +30M -delete
done
head -30 $O/dr2_summary.md
