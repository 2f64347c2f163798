#!/bin/bash
# round 4, call zze: kernel trace of the 256^3 channel and closed box with the fused stage boundaries and merged frames
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04zze
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in PBB BBB; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -- python3 $ROOT/tools/bench_general.py 256 30 $t > $O/$t.log 2>&1 || { tail -5 $O/$t.log; exit 1; }
  f=$(find $O/$t -name "*kernel_stats.csv" | head -1)
  python3 - "$f" > $O/${t}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:130]}')
PY
  cat $O/${t}_stats.txt
  find $O/$t -name "*kernel_trace.csv" -delete
done
