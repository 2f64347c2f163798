#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ef in 0 1; do
  export OCN_EXTRA_FIRST=$ef
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ef$ef -- python3 $ROOT/bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline --no-strict --no-kernel-timing > $O/trace_ef$ef.log 2>&1 || exit 1
  f=$(find $O/trace_ef$ef -name "*kernel_stats.csv" | head -1)
  echo "== extra_first=$ef"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {r["Percentage"]}')
PY
  find $O/trace_ef$ef -name "*kernel_trace.csv" -delete
done
