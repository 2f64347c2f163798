#!/usr/bin/env python3
"""Debug: two Poisson solver handles alive at once.  Checks ||lap(p) - R|| / ||R|| of each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oceananigans_jl_amd as ocn
from oracle import oracle as O
from helpers import make_pair, stretched_faces, from_dev, to_dev

def check(tag, size, topo, z, solver=None):
    og, pg = make_pair(O, ocn, size, topo, x=(0, 64), y=(0, 64), z=z)
    rng = np.random.default_rng(1)
    U = []
    for loc in (1, 2, 4):
        a = og.zeros(loc); og.interior(a)[...] = rng.uniform(-1, 1, og.interior(a).shape); O.fill_halo_regions(og, a, loc); U.append(a)
    R = O.divergence(og, *U)
    dU = [to_dev(ocn, pg, l, a) for l, a in zip((1, 2, 4), U)]
    S = solver or ocn.nonhydrostatic_pressure_solver(pg)
    p = ocn.CenterField(pg)
    ocn.solve_for_pressure(p, S, 1.0, dU)
    ocn.fill_halo_regions(p); ocn.sync_device()
    lap = O.laplacian(og, from_dev(p))
    print(tag, S.info(), "residual", np.linalg.norm(lap - R) / np.linalg.norm(R), flush=True)
    return S, (og, pg)

order = os.environ.get("ORDER", "AB")
keep = []
zB = stretched_faces(16, 32.0)
for ch in order:
    if ch == "A":
        keep.append(check("A 16^3 PPP", (16, 16, 16), "PPP", (0, 64)))
    elif ch == "B":
        keep.append(check("B 32x8x16 PPB", (32, 8, 16), "PPB", zB))
    elif ch == "C":
        keep.append(check("C 16x12x10 PPB", (16, 12, 10), "PPB", stretched_faces(10, 32.0)))
    elif ch == "b":  # re-solve with the first B solver
        S, (og, pg) = [k for k in keep if k[1][1].Nx == 32][0]
        check("B again", (32, 8, 16), "PPB", zB, solver=S)
