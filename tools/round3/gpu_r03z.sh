#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/step_bbb.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import oceananigans_jl_amd as ocn
N = 256
ocn.set_math_mode(ocn.MATH_FAST)
g = ocn.RectilinearGrid(ocn.GPU(), size=(N, N, N), x=(0, 1), y=(0, 1), z=(0, 1), topology=("Bounded",) * 3)
m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for f in m.velocities:
    v = f.interior_view(); v.copy_(2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1)
ocn.set(m)
for _ in range(6):
    ocn.time_step(m, 1e-4)
ocn.flush_tendencies(m); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step -- python3 /tmp/step_bbb.py > $O/trace_step.log 2>&1 || { tail -5 $O/trace_step.log; exit 1; }
f=$(find $O/trace_step -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:84]:84s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} avg_us {float(r["AverageNs"])/1e3:8.1f}')
PY
find $O/trace_step -name "*kernel_trace.csv" -delete
