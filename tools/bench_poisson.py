#!/usr/bin/env python3
"""Times solve_for_pressure! alone (source term + transforms + solve) with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = "Periodic"
g = ocn.RectilinearGrid(ocn.GPU(), size=(n, n, n), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
U = [ocn.Field(l, g) for l in (1, 2, 4)]
for f in U:
    f.data.copy_(torch.rand(f.data.shape, device="cuda", dtype=torch.float64))
p = ocn.CenterField(g)
s = ocn.nonhydrostatic_pressure_solver(g)
ocn.solve_for_pressure(p, s, 1.0, U); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ocn.solve_for_pressure(p, s, 1.0, U)
e1.record(); torch.cuda.synchronize()
print(f"poisson n={n} cb={os.environ.get('OCN_COLFFT_CB','8')} fusedz={os.environ.get('OCN_POISSON_FUSED_Z','1')}: {e0.elapsed_time(e1)/10:.3f} ms per solve_for_pressure  info={s.info()}")
