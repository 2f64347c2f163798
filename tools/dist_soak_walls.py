#!/usr/bin/env python3
"""Soak of the wall-bounded distributed paths: 60 RK3 steps of a stratified 64 x 64 x 32 channel (Periodic, Bounded, Bounded) and closed box
(Bounded, Bounded, Bounded) with the config-4 term set (T, S, SeawaterBuoyancy, FPlane, AMD, a wind stress and surface fluxes, stretched z)
on 4 ranks (threads on one GPU, the library's in-process transport) against the single-rank model: drift between the distributed
Fourier-tridiagonal solver (cosine transforms after / around the transposes) and the single-process one, finiteness, incompressibility.
Usage: tools/dist_soak_walls.py [steps]"""
import os
import sys
import threading

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oceananigans_jl_amd as ocn
from helpers import stretched_faces
from test_gpu_distributed import _run_ranks_local

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N, R = (64, 64, 32), 4
ocn.set_math_mode(ocn.MATH_FAST)
for topo in ("PBB", "BBB"):
    names = {"P": "Periodic", "B": "Bounded"}
    ext = dict(x=(0, 128.0), y=(0, 128.0), z=stretched_faces(N[2], 64.0), topology=tuple(names[t] for t in topo), halo=(3, 3, 3))
    rng = np.random.default_rng(11)
    shape = {"u": (N[0] + (topo[0] == "B"), N[1], N[2]), "v": (N[0], N[1] + 1, N[2]), "w": (N[0], N[1], N[2] + 1)}
    init = {k: 1e-2 * rng.uniform(-1, 1, s) for k, s in shape.items()}
    zc = 0.5 * (ext["z"][1:] + ext["z"][:-1])
    init["T"] = 20 + 0.01 * zc[None, None, :] + 1e-3 * rng.uniform(-1, 1, N)
    init["S"] = 35 + 1e-3 * rng.uniform(-1, 1, N)
    dt = 2.0

    def build(arch, r=None):
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-2e-4)),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(4e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(arch, size=N, **ext), advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                    closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), boundary_conditions=bcs)
        nx = m.grid.Nx
        sel = {}
        for k, v in init.items():
            if r is None:
                sel[k] = v
            else:
                extra = 1 if (topo[0] == "B" and k == "u" and r == R - 1) else 0
                sel[k] = v[r * nx:(r + 1) * nx + extra]
        ocn.set(m, **sel)
        return m

    single = build(ocn.GPU())
    for _ in range(steps):
        ocn.time_step(single, dt)
    ocn.flush_tendencies(single)
    ocn.sync_device()
    ref = [f.interior() for f in single.prognostic_fields()]

    def rank_main(r, fabric):
        m = build(ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric), r)
        for _ in range(steps):
            ocn.time_step(m, dt)
        ocn.flush_tendencies(m)
        g = m.grid
        dd = torch.zeros((g.Nz, g.Ny, g.Nx), dtype=torch.float64, device="cuda")
        ocn.fill_halo_regions(m.velocities)
        ocn._lib.call("ocn_divergence", g.cref, m.u.ptr, m.v.ptr, m.w.ptr, dd.data_ptr(), 0)
        ocn.sync_device()
        fabric.barrier()
        return [f.interior() for f in m.prognostic_fields()], float(dd.abs().max())

    outs = _run_ranks_local(ocn, R, rank_main)
    nx = N[0] // R
    vmax = max(np.abs(a).max() for a in ref[:3])
    du = dT = 0.0
    finite = True
    for r, (fields, _) in enumerate(outs):
        for a, b, name in zip(fields, ref, ("u", "v", "w", "T", "S")):
            extra = 1 if (topo[0] == "B" and name == "u" and r == R - 1) else 0
            e = np.abs(a - b[r * nx:(r + 1) * nx + extra]).max()
            finite = finite and bool(np.isfinite(a).all())
            if name in "uvw":
                du = max(du, e)
            else:
                dT = max(dT, e)
    print(f"{topo}: {steps} steps on {R} ranks vs single rank: max|du| = {du:.3e} (max|u| = {vmax:.3e}), max|dT, dS| = {dT:.3e}, "
          f"max|div u| per rank = {max(o[1] for o in outs):.3e}, finite = {finite}")
