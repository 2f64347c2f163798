#!/usr/bin/env python3
"""fused against unfused stage boundaries on one grid, array by array after every step: tools/debug_fused_walls.py TOPO Nx Ny Nz closure [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oceananigans_jl_amd as ocn
from helpers import stretched_faces
topo, N, closure = sys.argv[1], tuple(int(a) for a in sys.argv[2:5]), sys.argv[5]
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
rng = np.random.default_rng(41)
T = {"P": "Periodic", "B": "Bounded", "F": "Flat"}
z = stretched_faces(N[2], 32.0)
shape = {"u": (N[0] + (topo[0] == "B"), N[1], N[2]), "v": (N[0], N[1] + (topo[1] == "B"), N[2])}
init = {n: 1e-2 * rng.uniform(-1, 1, shape[n]) for n in "uv"}
init["T"] = 20 + 1e-2 * rng.uniform(-1, 1, N)
init["S"] = 35 + 1e-2 * rng.uniform(-1, 1, N)
def build():
    kw = dict(x=(0, 64), z=z, topology=tuple(T[t] for t in topo))
    if topo[1] != "F":
        kw["y"] = (0, 64)
    g = ocn.RectilinearGrid(ocn.GPU(), size=tuple(n for n, t in zip(N, topo) if t != "F"), **kw)
    if closure == "none":
        return ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"))
    parts = closure.split("+")
    bcs = {}
    if "bcs" in parts:
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-3e-4)),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
    kw = {}
    if "coriolis" in parts: kw["coriolis"] = ocn.FPlane(f=1e-4)
    if "scalar" in parts: kw["closure"] = ocn.ScalarDiffusivity(ν=1e-3, κ={"T": 2e-3, "S": 5e-4})
    if "amd" in parts: kw["closure"] = ocn.AnisotropicMinimumDissipation()
    if "buoyancy" in parts: kw["buoyancy"] = ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4))
    return ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), boundary_conditions=bcs, **kw)
poison = os.environ.get("POISON")
if poison:  # garbage in the torch pools (and, after empty_cache, in what the driver hands to the library's own allocations)
    import torch
    for rep in range(2):
        for n in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
            ts = [torch.full((n,), float(poison), dtype=torch.float64, device="cuda") for _ in range(4)]
            del ts
        if rep == 0 and os.environ.get("POISON_DRIVER") == "1":
            torch.cuda.empty_cache()
ocn.set_math_mode(ocn.MATH_STRICT)
def build_on(tp, n, fused):
    zz = stretched_faces(n[2], 32.0)
    kw = dict(x=(0, 64), z=zz, topology=tuple(T[t] for t in tp))
    if tp[1] != "F":
        kw["y"] = (0, 64)
    gg = ocn.RectilinearGrid(ocn.GPU(), size=tuple(q for q, t in zip(n, tp) if t != "F"), **kw)
    mm = ocn.NonhydrostaticModel(gg, advection=ocn.WENO(), tracers=("T", "S"))
    if not fused:
        mm.fuse_stage_boundaries = mm.defer_final_tendencies = False
    return mm
if os.environ.get("PRE"):  # other grids first, fused then unfused as the parametrised test runs them: PRE="PPB,32,16,12;BPB,30,12,10"
    for spec in os.environ["PRE"].split(";"):
        ptopo, *pn = spec.split(",")
        pn = tuple(int(a) for a in pn)
        r2 = np.random.default_rng(41)
        sh = {"u": (pn[0] + (ptopo[0] == "B"), pn[1], pn[2]), "v": (pn[0], pn[1] + (ptopo[1] == "B"), pn[2])}
        ini = {n: 1e-2 * r2.uniform(-1, 1, sh[n]) for n in "uv"}
        ini["T"] = 20 + 1e-2 * r2.uniform(-1, 1, pn)
        ini["S"] = 35 + 1e-2 * r2.uniform(-1, 1, pn)
        res = []
        for fused in (True, False):
            mp = build_on(ptopo, pn, fused)
            ocn.set(mp, **ini)
            for _ in range(3):
                ocn.time_step(mp, 1.5)
            G = [f.parent() for f in mp.timestepper.Gn]
            ocn.sync_device()
            res.append([f.parent() for f in mp.prognostic_fields()] + G)
        print("pre", spec, "max diff", max(np.abs(a - b).max() for a, b in zip(*res)))
ms = []
for fused in (True, False):
    if os.environ.get("SEQ") == "1":
        break
    m = build()
    if not fused:
        m.fuse_stage_boundaries = m.defer_final_tendencies = False
    ocn.set(m, **init)
    ms.append(m)
names = ["u", "v", "w", "T", "S", "Gu", "Gv", "Gw", "GT", "GS", "p"]
if os.environ.get("SOLVE_ONLY") == "1":  # two solver instances on the same velocities
    ps = []
    for q in range(2):
        m = build()
        ocn.set(m, **init)
        ocn.solve_for_pressure(m.pNHS, m.pressure_solver, 1.5, m.velocities)
        ocn.sync_device()
        ps.append((m, m.pNHS.interior().copy(), [f.parent() for f in m.velocities]))
    print("velocities equal:", all(np.array_equal(a, b) for a, b in zip(ps[0][2], ps[1][2])))
    d = np.abs(ps[0][1] - ps[1][1])
    print(f"p of two solver instances: max diff {d.max():.3e} ({int((d != 0).sum())} of {d.size}), info {ps[0][0].pressure_solver.info()}")
    m = ps[0][0]
    ocn.solve_for_pressure(m.pNHS, m.pressure_solver, 1.5, m.velocities)
    ocn.sync_device()
    d = np.abs(m.pNHS.interior() - ps[0][1])
    print(f"the first instance again: max diff {d.max():.3e}")
    sys.exit(0)
if os.environ.get("SEQ") == "1":  # one model after the other, like the test
    ms = []
    res = []
    for fused in (True, False):
        m = build()
        if not fused:
            m.fuse_stage_boundaries = m.defer_final_tendencies = False
        ocn.set(m, **init)
        per = []
        for s_ in range(steps):
            ocn.time_step(m, 1.5)
            G = [f.parent() for f in m.timestepper.Gn]
            ocn.sync_device()
            per.append([f.parent() for f in m.prognostic_fields()] + G + [m.pNHS.interior()])
        res.append(per)
        ms.append(m)
    for s_ in range(steps):
        print(f"step {s_ + 1}:", ", ".join(f"{n} {np.abs(a - b).max():.2e} ({int((a != b).sum())})" for n, a, b in zip(names, res[0][s_], res[1][s_])))
    sys.exit(0)
for s in range(steps):
    outs = []
    for m in ms:
        ocn.time_step(m, 1.5)
        G = [f.parent() for f in m.timestepper.Gn]
        ocn.sync_device()
        outs.append([f.parent() for f in m.prognostic_fields()] + G + [m.pNHS.interior()])
    print(f"step {s + 1}:", ", ".join(f"{n} {np.abs(a - b).max():.2e} ({int((a != b).sum())})" for n, a, b in zip(names, *outs)))
