"""Committed golden vectors (tests/golden/, made by tests/golden/make_golden.py from the CPU oracle):
 - CPU: the oracle still reproduces them bit for bit;
 - GPU: the HIP library reproduces them (bitwise for the strict stencils, 1e-11 after time steps)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CASES  # noqa: E402

from helpers import from_dev, make_pair, stretched_faces, to_dev  # noqa: E402


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def _z(kw):
    return stretched_faces(kw["size"][2], 1.2) if isinstance(kw["z"], str) else kw["z"]


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_golden(oracle, name):
    O, kw, d = oracle, CASES[name], _load(name)
    g = O.Grid(kw["size"], x=(0, 2.0), y=(0, 1.0), z=_z(kw), topology=kw["topo"], halo=(3, 3, 3))
    G = [g.zeros(l) for l in (1, 2, 4)]
    O.momentum_tendencies(g, *(np.asfortranarray(d[k]) for k in "uvw"), *G)
    for a, k in zip(G, ("Gu", "Gv", "Gw")):
        np.testing.assert_array_equal(a, d[k])
    m = O.NonhydrostaticModel(g)
    m.set(u=d["init_u"], v=d["init_v"], w=d["init_w"])
    for _ in range(2):
        m.time_step(float(d["dt"]))
    # the FFT (pocketfft) may differ in the last bits between scipy builds: stepped fields to 1e-13
    for a, k in zip((m.u, m.v, m.w, m.p), ("u2", "v2", "w2", "p2")):
        np.testing.assert_allclose(a, d[k], rtol=0, atol=1e-13 * max(1.0, np.abs(d[k]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_reproduces_golden(oracle, ocn, name):
    kw, d = CASES[name], _load(name)
    og, pg = make_pair(oracle, ocn, kw["size"], kw["topo"], x=(0, 2.0), y=(0, 1.0), z=_z(kw))
    ocn.set_math_mode(ocn.MATH_STRICT)
    du, dv, dw = (to_dev(ocn, pg, l, np.asfortranarray(d[k])) for l, k in zip((1, 2, 4), "uvw"))
    dG = [ocn.Field(l, pg) for l in (1, 2, 4)]
    ocn._lib.call("ocn_compute_momentum_tendencies", pg.cref, du.ptr, dv.ptr, dw.ptr, dG[0].ptr, dG[1].ptr, dG[2].ptr, None, 0)
    dc, dGc = to_dev(ocn, pg, 0, np.asfortranarray(d["c"])), ocn.Field(0, pg)
    ocn._lib.call("ocn_compute_tracer_tendency", pg.cref, du.ptr, dv.ptr, dw.ptr, dc.ptr, dGc.ptr, None, 0)
    ocn.sync_device()
    for f, k in zip(dG + [dGc], ("Gu", "Gv", "Gw", "Gc")):
        np.testing.assert_array_equal(from_dev(f), d[k])
    m = ocn.NonhydrostaticModel(pg, advection=ocn.WENO())
    ocn.set(m, u=d["init_u"], v=d["init_v"], w=d["init_w"])
    for _ in range(2):
        ocn.time_step(m, float(d["dt"]))
    ocn.sync_device()
    for f, k in zip(m.velocities + (m.pNHS,), ("u2", "v2", "w2", "p2")):
        a, b = og.interior(from_dev(f)), og.interior(np.asfortranarray(d[k]))
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())
