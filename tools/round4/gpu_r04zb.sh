#!/bin/bash
# round 4, call zb: kernel traces of one rank of 2 and 8 with the batched real y / row kernels (C driver over the replica transport)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for R in 2 8; do
  OCN_BDR_ONLY=c timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace$R -o dr$R -- python3 $ROOT/tools/bench_dist_rank.py 512 $R 20 driver > $O/trace$R.log 2>&1 || { tail -5 $O/trace$R.log; exit 1; }
  DB=$(find $O/trace$R -name "*.db" | head -1)
  python3 $ROOT/tools/summarize_rocpd.py $DB $O/dist_rank${R}.md "one rank of $R, 512^3 box, C driver over the replica transport (OCN_BDR_ONLY=c tools/bench_dist_rank.py 512 $R 20 driver; 5 warm-up + 20 timed steps + set!), batched real y / row kernels" > /dev/null 2>&1
  find $O/trace$R -name "*.db" -size +30M -delete
  grep 'C driver' $O/trace$R.log
done
