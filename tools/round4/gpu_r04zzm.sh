#!/bin/bash
# round 4, call zzm: bench.py's two legs on a partitioned run (one-rank RCCL world): the new tests, the all-gather tests, and the 512^3 line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzm
mkdir -p $O
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_bench_line.py tests/test_gpu_distributed.py -m gpu -x -q -k "bench_line or all_gather or conservative or rccl" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 $O/pytest.log
[ $rc -eq 0 ] || exit 1
OCN_FORCE_DISTRIBUTED=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_box_rccl_world1.json 2> $O/bench_box_rccl_world1.err || { tail -20 $O/bench_box_rccl_world1.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_box_rccl_world1.json'));r=d['config']['rccl'];print('default', round(d['ms_per_step'],2), d['driver'], '| conservative', round(r['conservative']['ms_per_step'],2), r['fast_path'])"
grep -a "\[bench\]" $O/bench_box_rccl_world1.err
