#!/usr/bin/env python3
"""Config 4 of BASELINE.json (advection-only physics): 512x512x256 (Periodic, Periodic, Bounded), stretched z
(ocean_wind_mixing_and_convection spacing, examples/ocean_wind_mixing_and_convection.jl:38-62), WENO5, RK3,
FourierTridiagonalPoissonSolver.  Prints ms/step and cell-updates/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
Lz, refinement, stretching = 32.0, 1.2, 12.0
h = lambda k: (k - 1) / Nz
zeta0 = lambda k: 1 + (h(k) - 1) / refinement
Sigma = lambda k: (1 - np.exp(-stretching * h(k))) / (1 - np.exp(-stretching))
z_faces = np.array([Lz * (zeta0(k) * Sigma(k) - 1) for k in range(1, Nz + 2)])
ocn.set_math_mode(ocn.MATH_FAST)
g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, 64), y=(0, 64), z=z_faces, topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for f in m.velocities:
    iv = f.interior_view(); iv.copy_(torch.rand(iv.shape, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1)
ocn.set(m)
umax = float(torch.stack([f.interior_view().abs().max() for f in m.velocities]).max())
dt = 0.1 * min(g.dx, float(np.diff(z_faces).min())) / umax
for _ in range(2): ocn.time_step(m, dt)
ocn.flush_tendencies(m); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): ocn.time_step(m, dt)
ocn.flush_tendencies(m); torch.cuda.synchronize()
el = time.perf_counter() - t0
div = torch.zeros((Nz, Nx, Nx), dtype=torch.float64, device="cuda")
ocn._lib.call("ocn_divergence", g.cref, m.u.ptr, m.v.ptr, m.w.ptr, div.data_ptr(), 0)
print(f"config4 {Nx}x{Nx}x{Nz} PPB stretched: {el/steps*1e3:.2f} ms/step, {Nx*Nx*Nz*steps/el:.3e} cell-updates/s, max|div u| = {float(div.abs().max()):.2e}, finite={bool(torch.isfinite(m.u.data).all())}")
