#!/usr/bin/env python3
"""First HydrostaticFreeSurfaceModel slice at the size of BASELINE.json's config 5 (1024 x 1024 x 128, (Periodic, Periodic, Bounded)),
with what the slice has: ExplicitFreeSurface instead of the split-explicit one, flux-form WENO5 momentum and tracer advection,
T and S, linear SeawaterBuoyancy, FPlane, ScalarDiffusivity, QuasiAdamsBashforth2.  One QAB2 step = one tendency evaluation.

  tools/bench_hydrostatic.py [Nx] [Nz] [steps] [WENO | VectorInvariant | config5] [split_substeps]
      VectorInvariant: the model's defaults, Centered tracers;  config5: VectorInvariant momentum + WENO tracers (BASELINE config 5's
      "WENO tracer advection");  split_substeps = N: SplitExplicitFreeSurface(substeps = N) instead of the explicit free surface
      (NOT yet run on the GPU at this size: added after the round's GPU budget was spent; the baroclinic step is then 10x longer).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
scheme = sys.argv[4] if len(sys.argv) > 4 else "WENO"
split = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ocn.set_math_mode(ocn.MATH_FAST)
H, L = 1000.0, 1.0e6
g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, L), y=(0, L), z=(-H, 0.0), topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
m = ocn.HydrostaticFreeSurfaceModel(g, momentum_advection=ocn.WENO() if scheme == "WENO" else ocn.VectorInvariant(), tracers=("T", "S"),
                                    tracer_advection=ocn.WENO() if scheme == "config5" else None,
                                    free_surface=ocn.SplitExplicitFreeSurface(substeps=split) if split else ocn.ExplicitFreeSurface(),
                                    coriolis=ocn.FPlane(f=1e-4), closure=ocn.ScalarDiffusivity(ν=1e-2, κ=1e-3),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                    fused=None if os.environ.get("OCN_HYDRO_FUSED", "1") != "0" else False)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for f in (m.u, m.v):
    iv = f.interior_view()
    iv.copy_(1e-2 * (2 * torch.rand(iv.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
zc = torch.linspace(-H + H / (2 * Nz), -H / (2 * Nz), Nz, device="cuda", dtype=torch.float64)
m.field("T").interior_view().copy_((20 + 0.01 * zc)[:, None, None].expand(Nz, Nx, Nx))
m.field("S").interior_view().fill_(35.0)
m.update_state(compute_tendencies=False)
dt = (2.0 if split else 0.2) * g.dx / np.sqrt(ocn.hydrostatic.g_Earth * H)   # gravity-wave CFL 0.2 (explicit) / 2 (split-explicit)
for _ in range(3):
    m.time_step(dt)
ocn.sync_device()
t0 = time.perf_counter()
for _ in range(steps):
    m.time_step(dt)
ocn.sync_device()
ms = (time.perf_counter() - t0) / steps * 1e3
finite = bool(torch.isfinite(m.eta).all()) and all(bool(torch.isfinite(f.interior_view()).all()) for f in m.velocities)
print(f"hydrostatic slice {Nx}x{Nx}x{Nz} PPB, fused={m.fused}, {'split-explicit(' + str(split) + ')' if split else 'explicit'} free surface, {scheme}, T+S, QAB2: {ms:.2f} ms/step, "
      f"{Nx * Nx * Nz / ms * 1e3:.3e} cell-updates/s, max|eta| = {float(m.eta.abs().max()):.2e}, finite={finite}")
