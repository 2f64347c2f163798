#!/usr/bin/env python3
"""The reference's examples/ocean_wind_mixing_and_convection.jl (:1-170) on the MI355X backend: same grid, physics, boundary
conditions, initial condition, time-step wizard and progress message; plotting / JLD2 output left out.

    python examples/ocean_wind_mixing_and_convection.py [--advection UpwindBiased] [--size 32 32 24] [--stop-minutes 40]

The example uses UpwindBiased(order=5) (the default here); BASELINE.json's config 4 swaps in WENO().
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oceananigans_jl_amd as ocn

minute, hour = 60.0, 3600.0
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=3, default=(32, 32, 24))
ap.add_argument("--advection", choices=("UpwindBiased", "WENO", "Centered"), default="UpwindBiased")
ap.add_argument("--stop-minutes", type=float, default=40.0)
ap.add_argument("--math", choices=("fast", "strict"), default="fast")
a = ap.parse_args()
ocn.set_math_mode(ocn.MATH_FAST if a.math == "fast" else ocn.MATH_STRICT)

Nx, Ny, Nz = a.size
Lx = Ly = 64.0     # (m) domain horizontal extents
Lz = 32.0          # (m) domain depth
refinement = 1.2   # controls spacing near surface (higher means finer spaced)
stretching = 12    # controls rate of stretching at bottom
h = lambda k: (k - 1) / Nz
zeta0 = lambda k: 1 + (h(k) - 1) / refinement
Sigma = lambda k: (1 - np.exp(-stretching * h(k))) / (1 - np.exp(-stretching))
z_faces = np.array([Lz * (zeta0(k) * Sigma(k) - 1) for k in range(1, Nz + 2)])
grid = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Ny, Nz), x=(0, Lx), y=(0, Ly), z=z_faces,
                           topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))

buoyancy = ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4))
Q, rho_o, cP = 200.0, 1026.0, 3991.0            # W m⁻² surface heat flux, kg m⁻³, J K⁻¹ kg⁻¹
JT = Q / (rho_o * cP)                           # K m s⁻¹ surface temperature flux
dTdz = 0.01                                     # K m⁻¹
T_bcs = ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(JT), bottom=ocn.GradientBoundaryCondition(dTdz))
u10, cD, rho_a = 10.0, 2.5e-3, 1.225
tau_x = -rho_a / rho_o * cD * u10 * abs(u10)    # m² s⁻²
u_bcs = ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(tau_x))
evaporation_rate = 1e-3 / hour                  # m s⁻¹;  Jˢ(x, y, t, S, rate) = -rate * S
S_bcs = ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-evaporation_rate))

model = ocn.NonhydrostaticModel(grid, buoyancy=buoyancy, advection={"UpwindBiased": ocn.UpwindBiased(order=5), "WENO": ocn.WENO(), "Centered": ocn.Centered()}[a.advection],
                                tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4), closure=ocn.AnisotropicMinimumDissipation(),
                                boundary_conditions={"u": u_bcs, "T": T_bcs, "S": S_bcs})

rng = np.random.default_rng(0)
Xi = lambda z: rng.standard_normal(np.broadcast_shapes(np.shape(z), (Nx, Ny, 1))) * z / grid.Lz * (1 + z / grid.Lz)  # noise
Ti = lambda x, y, z: 20 + dTdz * z + dTdz * grid.Lz * 1e-6 * Xi(z)
ui = lambda x, y, z: np.sqrt(abs(tau_x)) * 1e-3 * Xi(z)
ocn.set(model, u=ui, w=lambda x, y, z: np.sqrt(abs(tau_x)) * 1e-3 * rng.standard_normal((Nx, Ny, np.shape(z)[-1])) * z / grid.Lz * (1 + z / grid.Lz),
        T=Ti, S=35.0)

dt, stop_time = 10.0, a.stop_minutes * minute
wizard = ocn.TimeStepWizard(cfl=1.0, max_change=1.1, max_dt=1 * minute)
nan_checker = ocn.NaNChecker({"u": model.u}, erroring=True)
t0 = time.perf_counter()
while model.clock.time < stop_time:
    if model.clock.iteration % 10 == 0:
        dt = wizard(model, dt)
    if model.clock.iteration % 20 == 0:
        wmax = float(model.w.data.abs().max())
        print(f"Iteration: {model.clock.iteration:04d}, time: {model.clock.time / minute:7.3f} min, Δt: {dt:6.2f} s, "
              f"max(|w|) = {wmax:.1e} m s⁻¹, wall time: {time.perf_counter() - t0:.1f} s", flush=True)
    if model.clock.iteration % 100 == 0:
        nan_checker(model)
    ocn.time_step(model, min(dt, stop_time - model.clock.time))
ocn.sync_device()
nu = model.diffusivity_fields["nu_e"].interior()
T = model.field("T").interior()
print(f"done: {model.clock.iteration} iterations, {time.perf_counter() - t0:.1f} s; max(|w|) = {float(model.w.data.abs().max()):.2e}, "
      f"νₑ in [{nu.min():.1e}, {nu.max():.1e}] m² s⁻¹, surface T = {T[:, :, -1].mean():.4f} °C, T range [{T.min():.3f}, {T.max():.3f}]")
