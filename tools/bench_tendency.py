#!/usr/bin/env python3
"""Times the momentum-tendency launch alone (HIP events on the launching stream). Env: OCN_TILE, OCN_TENDENCY_KERNEL."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
ocn.set_math_mode(ocn.MATH_FAST if mode == "fast" else ocn.MATH_STRICT)
P = "Periodic"
g = ocn.RectilinearGrid(ocn.GPU(), size=(n, n, n), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for f in m.velocities:
    f.data.copy_(torch.rand(f.data.shape, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1)
ocn.compute_tendencies(m); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    ocn.compute_tendencies(m)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
chk = float(sum(f.interior_view().abs().sum() for f in m.timestepper.Gn))
print(f"tile={os.environ.get('OCN_TILE','0')} kernel={os.environ.get('OCN_TENDENCY_KERNEL','tiled')} n={n} {mode}: {ms:.3f} ms  ({n**3/ms/1e6:.2f} Gcell/s)  checksum={chk:.10e}")
