#!/usr/bin/env python3
"""Debug helper: run the ocean-mixing parity case with the torch caching allocator pre-filled with garbage, to expose
kernels whose results depend on out-of-bounds / uninitialised reads.  tools/debug_poison.py [poison_value]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oceananigans_jl_amd as ocn
from oracle import oracle as O
from helpers import make_pair, stretched_faces, from_dev

poison = float(sys.argv[1]) if len(sys.argv) > 1 else None
if poison is not None:
    for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):  # small and large pools
        ts = [torch.full((n,), poison, dtype=torch.float64, device="cuda") for _ in range(4)]
        del ts
if os.environ.get("PRE") == "tracer":  # the sequence of tests/test_gpu_model.py::test_tracer_conservation
    rng = np.random.default_rng(2)
    og, pg = make_pair(O, ocn, (16, 16, 16), "PPP")
    om = O.NonhydrostaticModel(og, tracers=("c",))
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("c",))
    init = {n: rng.uniform(-1, 1, (16, 16, 16)) for n in "uvw"}
    init["c"] = rng.uniform(0, 1, (16, 16, 16))
    om.set(**init)
    ocn.set(pm, **init)
    for _ in range(3):
        om.time_step(0.01)
        ocn.time_step(pm, 0.01)
    ocn.sync_device()
    print("pre done", np.abs(pm.tracers[0].interior() - og.interior(om.tracers[0])).max())
    if os.environ.get("KEEP") != "1":
        del om, pm, og, pg
    if os.environ.get("GC") == "1":
        import gc
        gc.collect()
        torch.cuda.synchronize()
size = (32, 8, 16)
ts = "QuasiAdamsBashforth2"
z = stretched_faces(size[2], 32.0)
og, pg = make_pair(O, ocn, size, "PPB", x=(0, 64), y=(0, 64), z=z)
rng = np.random.default_rng(26)
variant = os.environ.get("VARIANT", "full")
kw_o, kw_p = {}, {}
if variant == "full":
    obcs = {"u": {"top": O.FluxBoundaryCondition(-3e-4)}, "T": {"top": O.FluxBoundaryCondition(5e-5), "bottom": O.GradientBoundaryCondition(0.01)},
            "S": {"top": O.BC("flux", 0.0, -2.8e-7)}}
    pbcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-3e-4)),
            "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
            "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
    kw_o = dict(tracers=("T", "S"), coriolis_f=1e-4, closure=(1e-3, {"T": 2e-3, "S": 5e-4}), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4), boundary_conditions=obcs)
    kw_p = dict(tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4), closure=ocn.ScalarDiffusivity(ν=1e-3, κ={"T": 2e-3, "S": 5e-4}),
                buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), boundary_conditions=pbcs)
om = O.NonhydrostaticModel(og, timestepper=ts, advection="WENO5", **kw_o)
ocn.set_math_mode(ocn.MATH_STRICT)
pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), timestepper=ts, **kw_p)
zc = 0.5 * (z[1:] + z[:-1])
init = {n: 1e-2 * rng.uniform(-1, 1, og.interior(f).shape) for n, f in zip("uvw", (om.u, om.v, om.w))}
if variant == "full":
    init["T"] = 20 + 0.01 * zc[None, None, :] + 1e-3 * rng.uniform(-1, 1, size)
    init["S"] = 35 + 1e-3 * rng.uniform(-1, 1, size)
om.set(**init); ocn.set(pm, **init)
def report(tag):
    ocn.sync_device()
    errs = []
    names = ("u", "v", "w") + tuple(om.tracer_names)
    for name, a, d in zip(names, om.fields, pm.prognostic_fields()):
        errs.append(f"{name}:{np.abs(og.interior(from_dev(d)) - og.interior(a)).max():.2e}")
    for name, a, d in zip(names, om.Gn, pm.timestepper._Gn):
        errs.append(f"G{name}:{np.abs(og.interior(from_dev(d)) - og.interior(a)).max():.2e}")
    errs.append(f"p:{np.abs(og.interior_N(from_dev(pm.pNHS)) - og.interior_N(om.p)).max():.2e}")
    errs.append("| sums oracle " + " ".join(f"{float(np.sum(og.interior(a))):.12e}" for a in om.fields))
    errs.append("| product " + " ".join(f"{float(np.sum(og.interior(from_dev(d)))):.12e}" for d in pm.prognostic_fields()))
    print(tag, " ".join(errs), flush=True)
report("after set")
for n in range(3):
    om.time_step(2.0); ocn.time_step(pm, 2.0)
    report(f"step {n + 1}")
