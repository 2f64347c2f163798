#!/usr/bin/env python3
"""Step time of the config-4 term set (T, S, SeawaterBuoyancy, FPlane, AMD, a wind stress and surface fluxes, WENO, stretched z) on grids with
walls in x / y next to the (Periodic, Periodic, Bounded) grid of the same size:  tools/bench_general_terms.py [Nx] [Nz] [steps] [topologies]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oceananigans_jl_amd as ocn
from helpers import stretched_faces
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
topos = sys.argv[4].split(",") if len(sys.argv) > 4 else ("PPB", "PBB", "BBB")
ocn.set_math_mode(ocn.MATH_FAST)
T = {"P": "Periodic", "B": "Bounded"}
for topo in topos:
    g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, 128.0), y=(0, 128.0), z=stretched_faces(Nz, 64.0), topology=tuple(T[t] for t in topo),
                            halo=(3, 3, 3))
    bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-2e-4)),
           "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(4e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
           "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4), closure=ocn.AnisotropicMinimumDissipation(),
                                buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), boundary_conditions=bcs)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for f in m.velocities:
        v = f.interior_view()
        v.copy_(1e-2 * (2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
    m.tracers[0].interior_view().add_(20.0)
    m.tracers[1].interior_view().add_(35.0)
    ocn.set(m)
    dt = 1.0
    for _ in range(3):
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m); ocn.sync_device()
    t0 = time.perf_counter()
    for _ in range(steps):
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m); ocn.sync_device()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ok = all(bool(torch.isfinite(f.interior_view()).all()) for f in m.prognostic_fields())
    print(f"{topo} {Nx}x{Nx}x{Nz}, config-4 terms: {ms:.2f} ms/step ({Nx*Nx*Nz/ms/1e6:.2f} Gcell-updates/s), fused={m.fuse_stage_boundaries}, finite={ok}", flush=True)
    del m, g
    torch.cuda.empty_cache()
