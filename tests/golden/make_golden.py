#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/).

These vectors are produced by THIS repository's restatement of the reference algorithm, not by a run of the
reference (Julia is unavailable here): they freeze the oracle's behaviour so that any later edit
to the oracle or to the HIP kernels that changes results is caught, and they travel to the GPU box as data.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import stretched_faces  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    "ppp_10x9x8": dict(size=(10, 9, 8), topo="PPP", z=(0.0, 1.5)),
    "ppb_stretched_8x8x10": dict(size=(8, 8, 10), topo="PPB", z="stretched"),
    "ppf_12x10": dict(size=(12, 10, 1), topo="PPF", z=None),
}


def build(name, size, topo, z):
    if isinstance(z, str):
        z = stretched_faces(size[2], 1.2)
    g = O.Grid(size, x=(0, 2.0), y=(0, 1.0), z=z, topology=topo, halo=(3, 3, 3))
    rng = np.random.default_rng(20250418)
    # (1) tendencies of arbitrary (non-solenoidal) parent data, halos included
    u, v, w = (g.zeros(l) for l in (1, 2, 4))
    for a in (u, v, w):
        a[...] = rng.uniform(-1, 1, a.shape)
    Gu, Gv, Gw = (g.zeros(l) for l in (1, 2, 4))
    O.momentum_tendencies(g, u, v, w, Gu, Gv, Gw)
    c = g.zeros(0)
    c[...] = rng.uniform(0, 1, c.shape)
    Gc = g.zeros(0)
    O.tracer_tendency(g, u, v, w, c, Gc)
    # (2) two RK3 steps from a projected random state
    m = O.NonhydrostaticModel(g)
    init = {n: rng.uniform(-1, 1, g.interior(f).shape) for n, f in zip("uvw", (m.u, m.v, m.w))}
    if topo[2] == "F":
        init["w"][...] = 0
    m.set(**init)
    dt = 0.02
    for _ in range(2):
        m.time_step(dt)
    out = dict(u=u, v=v, w=w, c=c, Gu=Gu, Gv=Gv, Gw=Gw, Gc=Gc, init_u=init["u"], init_v=init["v"], init_w=init["w"],
               dt=np.float64(dt), u2=m.u, v2=m.v, w2=m.w, p2=m.p)
    if z is not None and not np.isscalar(z) and len(z) != 2:
        out["z_faces"] = np.asarray(z)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    for name, kw in CASES.items():
        build(name, **kw)
        print("wrote", name)
