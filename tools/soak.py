#!/usr/bin/env python3
"""Long runs on one GPU (not part of the test suite): the three bench workloads for hundreds of steps -- finite fields, discrete
incompressibility after every projection, kinetic energy of the unforced box not increasing (WENO5 is dissipative), volume
conservation of the free surface.   tools/soak.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ocn.set_math_mode(ocn.MATH_FAST)
P = "Periodic"
# ---- box 256^3
N = 256
g = ocn.RectilinearGrid(ocn.GPU(), size=(N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
for f in m.velocities:
    v = f.interior_view(); v.copy_(2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1)
ocn.set(m)
ke = lambda: float(sum((f.interior_view() ** 2).sum() for f in m.velocities))
def div_max():
    u, v, w = (f.interior_view() for f in m.velocities)
    d = (torch.roll(u, -1, 2) - u) / g.dx + (torch.roll(v, -1, 1) - v) / g.dy + (torch.roll(w, -1, 0) - w) / g.dz
    return float(d.abs().max())
e0, worst_div, last = ke(), 0.0, None
for n in range(steps):
    umax = max(float(f.interior_view().abs().max()) for f in m.velocities)
    ocn.time_step(m, 0.3 * g.dx / umax)
    if n % 20 == 19:
        ocn.flush_tendencies(m)
        e, d = ke(), div_max()
        assert np.isfinite(e) and (last is None or e <= last * (1 + 1e-12)), (n, e, last)
        worst_div, last = max(worst_div, d), e
print(f"box 256^3: {steps} steps at CFL 0.3, KE {e0:.6e} -> {last:.6e} (monotone), max|div u| {worst_div:.2e} (|u| ~ 1, 1/dx = {1 / g.dx:.0f})")
assert worst_div < 1e-9
del m, g
torch.cuda.empty_cache()
# ---- config 5 at 512 x 512 x 64
Nx, Nz, H, L = 512, 64, 1000.0, 5.0e5
g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, L), y=(0, L), z=(-H, 0.0), topology=(P, P, "Bounded"), halo=(3, 3, 3))
m = ocn.HydrostaticFreeSurfaceModel(g, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                    free_surface=ocn.SplitExplicitFreeSurface(substeps=30), coriolis=ocn.FPlane(f=1e-4),
                                    closure=ocn.ScalarDiffusivity(ν=1e-2, κ=1e-3),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)))
for f in (m.u, m.v):
    v = f.interior_view(); v.copy_(0.1 * (2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
zc = torch.linspace(-H + H / (2 * Nz), -H / (2 * Nz), Nz, device="cuda", dtype=torch.float64)
m.field("T").interior_view().copy_((20 + 0.01 * zc)[:, None, None].expand(Nz, Nx, Nx))
m.field("S").interior_view().fill_(35.0)
m.update_state(compute_tendencies=False)
dt = 2.0 * g.dx / np.sqrt(9.80665 * H)
eta0 = float(m.eta_interior().mean())
for n in range(steps):
    m.time_step(dt)
m.flush_tendencies()
ok = all(bool(torch.isfinite(f.interior_view()).all()) for f in (m.u, m.v, m.w) + tuple(m.tracers))
T = m.field("T").interior_view()
print(f"config 5 at 512x512x64: {steps} steps, finite={ok}, max|u| {float(m.u.interior_view().abs().max()):.3f}, max|eta| {float(m.eta_interior().abs().max()):.3e}, "
      f"mean(eta) drift {abs(float(m.eta_interior().mean()) - eta0):.2e}, T in [{float(T.min()):.3f}, {float(T.max()):.3f}]")
assert ok and abs(float(m.eta_interior().mean()) - eta0) < 1e-12 and 9.9 < float(T.min()) and float(T.max()) < 20.1
