#!/bin/bash
# round 3, call c: kernel trace of one rank of eight with the correction-on-load stage
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o dr8 -- python3 $ROOT/tools/bench_dist_rank.py 512 8 20 box > $O/trace.log 2>&1; echo "trace rc=$?"; tail -2 $O/trace.log
cd $ROOT
DB=$(find $O/trace -name "*.db" | head -1); echo $DB
python3 tools/summarize_rocpd.py $DB $O/dr8_summary.md "one rank of 8, 512^3 box, correction-on-load stage (tools/bench_dist_rank.py 512 8 20 box)" > /dev/null 2>&1
head -40 $O/dr8_summary.md
find $O/trace -name "*.db" -size +30M -delete
