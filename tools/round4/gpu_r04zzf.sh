#!/bin/bash
# round 4, call zzf: one velocity component per thread in the per-cell momentum kernels (the wall frames are latency-bound): parity tests,
# step times, kernel trace of the closed box (momentum_tendencies_general was 176.7 us per merged launch, momentum_finish_general 47.7: r04zze)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zzf
mkdir -p $O
cd $ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_general_topologies.py tests/test_gpu_physics.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?
tail -4 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_general_terms.py 256 128 10 PBB,BBB 2>&1 | grep "ms/step" > $O/bench.txt || exit 1
timeout -k 10 300 python tools/bench_general.py 256 10 PBB,BBB 2>&1 | grep "ms/step" >> $O/bench.txt || exit 1
cat $O/bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/BBB -- python3 $ROOT/tools/bench_general.py 256 30 BBB > $O/BBB.log 2>&1 || { tail -5 $O/BBB.log; exit 1; }
f=$(find $O/BBB -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $O/BBB_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
cat $O/BBB_stats.txt
find $O/BBB -name "*kernel_trace.csv" -delete
