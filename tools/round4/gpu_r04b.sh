#!/bin/bash
# round 4, call b: (1) kernel + model tests with the address-selected LDS stencils, (2) same-box A/B of the flagship variants
# (value selects / address selects) x (32 x 8 / 32 x 12 patches), (3) per-rank cost at R = 2, 4, 8 through the replica transport
# (the R-rank pipelines: transpose-free solve), (4) kernel traces of one rank of 2 and of 8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04b
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/ab_bench.sh "--steps 20 --warmup 5" nosel:ab/lib_nosel.so sel:ab/lib_sel.so nosel_tall:ab/lib_nosel.so:OCN_TEND_TALL=1 sel_tall:ab/lib_sel.so:OCN_TEND_TALL=1 > $O/ab.txt 2>&1; cat $O/ab.txt
for R in 2 4 8; do
  timeout -k 10 200 python tools/bench_dist_rank.py 512 $R 30 driver > $O/rank$R.txt 2>&1 || { tail -5 $O/rank$R.txt; exit 1; }
  grep "driver N" $O/rank$R.txt
done
cd /tmp && export TMPDIR=/tmp
for R in 2 8; do
  OCN_BDR_ONLY=c timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace$R -o dr$R -- python3 $ROOT/tools/bench_dist_rank.py 512 $R 20 driver > $O/trace$R.log 2>&1 || { tail -5 $O/trace$R.log; exit 1; }
  DB=$(find $O/trace$R -name "*.db" | head -1)
  python3 $ROOT/tools/summarize_rocpd.py $DB $O/dr${R}_summary.md "one rank of $R, 512^3 box, C driver over the replica transport (OCN_BDR_ONLY=c tools/bench_dist_rank.py 512 $R 20 driver; 5 warm-up + 20 timed steps)" > /dev/null 2>&1
  find $O/trace$R -name "*.db" -size +30M -delete
done
head -24 $O/dr2_summary.md
