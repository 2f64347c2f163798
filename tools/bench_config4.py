#!/usr/bin/env python3
"""Config 4 of BASELINE.json: 512x512x256 (Periodic, Periodic, Bounded), stretched z (ocean_wind_mixing_and_convection
spacing, examples/ocean_wind_mixing_and_convection.jl:38-62), WENO5, RK3, FourierTridiagonalPoissonSolver.

  tools/bench_config4.py [Nx] [Nz] [steps] [physics]
physics = 0: advection only (SURVEY §8d first form);  1: the example's physics (:79-152) with the LES closure replaced by a
constant ScalarDiffusivity: SeawaterBuoyancy(linear EOS), T and S tracers, FPlane(f=1e-4), wind stress, surface heat flux,
bottom temperature gradient, evaporation;  2: the example as written, closure = AnisotropicMinimumDissipation().
Prints ms/step and cell-updates/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
physics = int(sys.argv[4]) if len(sys.argv) > 4 else 0
Lz, refinement, stretching = 32.0, 1.2, 12.0
h = lambda k: (k - 1) / Nz
zeta0 = lambda k: 1 + (h(k) - 1) / refinement
Sigma = lambda k: (1 - np.exp(-stretching * h(k))) / (1 - np.exp(-stretching))
z_faces = np.array([Lz * (zeta0(k) * Sigma(k) - 1) for k in range(1, Nz + 2)])
ocn.set_math_mode(ocn.MATH_FAST)
g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, 64), y=(0, 64), z=z_faces, topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
if physics:
    Q, rho, cp, dTdz = 200.0, 1026.0, 3991.0, 0.01
    taux = -1.225 / rho * 2.5e-3 * 10 * 10
    bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(taux)),
           "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(Q / (rho * cp)), bottom=ocn.GradientBoundaryCondition(dTdz)),
           "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-1e-3 / 3600))}
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                closure=ocn.AnisotropicMinimumDissipation() if physics == 2 else ocn.ScalarDiffusivity(ν=1e-4, κ=1e-4),
                                buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                boundary_conditions=bcs)
    zc = 0.5 * (z_faces[1:] + z_faces[:-1])
    T = m.field("T").interior_view()
    T.copy_(torch.from_numpy(20 + dTdz * zc)[:, None, None].to("cuda") + 1e-6 * torch.rand(T.shape, generator=gen, device="cuda", dtype=torch.float64))
    m.field("S").interior_view().fill_(35.0)
    amp = 1e-2
else:
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    amp = 1.0
for f in m.velocities:
    iv = f.interior_view(); iv.copy_(amp * (torch.rand(iv.shape, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1))
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
print(f"config4 {Nx}x{Nx}x{Nz} PPB stretched physics={physics}: {el/steps*1e3:.2f} ms/step, {Nx*Nx*Nz*steps/el:.3e} cell-updates/s, "
      f"max|div u| = {float(div.abs().max()):.2e}, finite={bool(all(torch.isfinite(f.data).all() for f in m.prognostic_fields()))}")
