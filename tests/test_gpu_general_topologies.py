"""Fields, halos, tendencies and time steps on grids with a Bounded or Flat x / y (VERDICT r2 Missing 1): the direction-generic
kernels of csrc/general.hip and kernels.hip behind the same C entry points, against the CPU oracle -- bit for bit in strict math
for everything upstream of the Poisson solver.  Topologies: (Periodic, Bounded, Bounded) channels, (Bounded, Bounded, Bounded) boxes,
(Periodic, Flat, Bounded) x-z slices, (Bounded, Periodic, Periodic), (Flat, Bounded, Bounded) -- what most reference examples and
test/test_dynamics.jl:35-50 (PBB / BBB budgets) run on.
Reference: topologically_conditional_interpolation.jl:37-128, flat_advective_fluxes.jl:8-44, fill_halo_regions.jl:50-296,
fill_halo_regions_open.jl:65-70, fill_halo_regions_value_gradient.jl:5-103, kernel_launching.jl:113-161 (exclude_periphery)."""
import ctypes as C

import numpy as np
import pytest

from helpers import from_dev, make_pair, random_parent, to_dev

pytestmark = pytest.mark.gpu

LOCS = (1, 2, 4)
CASES = [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB"), ((16, 1, 10), "PFB"), ((9, 12, 8), "BPP"), ((1, 10, 9), "FBB"), ((10, 9, 8), "BPB"),
         ((7, 8, 9), "PBP")]


# large enough for the interior box of csrc/general.hip (cells a full stencil away from the x / y walls -> the LDS-tiled kernels with
# per-field layouts; the frames next to the walls -> the per-cell kernel): every launch combination, ragged tile edges included
BOX_CASES = [((40, 20, 9), "PBB"), ((30, 18, 8), "BBB"), ((26, 12, 10), "BPP"), ((45, 16, 12), "BPB"), ((33, 19, 9), "PBP"), ((70, 41, 6), "BBB")]


def _pair(O, ocn, size, topo):
    return make_pair(O, ocn, size, topo, x=(0, 1.3), y=(0, 0.9), z=(-0.7, 0))


def _filled(O, og, rng):
    """u, v, w, c with random parents whose halos then follow the reference's fills (walls, mirrors, periodic images)"""
    out = []
    for loc in LOCS + (0,):
        a = random_parent(og, loc, rng)
        O.fill_halo_regions(og, a, loc)
        out.append(a)
    return out


@pytest.mark.parametrize("size,topo", CASES)
def test_halo_fills_match_oracle_on_every_parent_cell(oracle, ocn, size, topo):
    """fill_halo_regions! of u, v, w and a tracer as ONE tupled launch: walls first, non-periodic sides over the interior cross-section,
    periodic sides over the whole parent -- every parent cell (corners and never-filled cells included) equals the oracle's"""
    O = oracle
    rng = np.random.default_rng(3)
    og, pg = _pair(O, ocn, size, topo)
    arrs = [random_parent(og, loc, rng) for loc in LOCS + (0,)]
    devs = [to_dev(ocn, pg, loc, a) for loc, a in zip(LOCS + (0,), arrs)]
    for loc, a in zip(LOCS + (0,), arrs):
        O.fill_halo_regions(og, a, loc)
    ocn.fill_halo_regions(tuple(devs))
    ocn.sync_device()
    for loc, a, d in zip(LOCS + (0,), arrs, devs):
        np.testing.assert_array_equal(from_dev(d), a, err_msg=f"{topo} loc {loc}")


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB")])
def test_value_and_gradient_conditions_on_x_y_walls(oracle, ocn, size, topo):
    """Value / Gradient conditions on south / north (and west / east on the closed box): the first halo cell is the linear extrapolation
    of fill_halo_regions_value_gradient.jl:5-103, bit for bit"""
    O = oracle
    rng = np.random.default_rng(4)
    og, pg = _pair(O, ocn, size, topo)
    a = random_parent(og, 0, rng)
    d = to_dev(ocn, pg, 0, a)
    obcs = {"south": O.ValueBoundaryCondition(0.3), "north": O.GradientBoundaryCondition(-1.7), "top": O.ValueBoundaryCondition(2.0)}
    pbcs = dict(south=ocn.ValueBoundaryCondition(0.3), north=ocn.GradientBoundaryCondition(-1.7), top=ocn.ValueBoundaryCondition(2.0))
    if topo == "BBB":
        obcs.update(west=O.GradientBoundaryCondition(0.5), east=O.ValueBoundaryCondition(-0.25))
        pbcs.update(west=ocn.GradientBoundaryCondition(0.5), east=ocn.ValueBoundaryCondition(-0.25))
    O.fill_halo_regions(og, a, 0, bcs=obcs)
    d.boundary_conditions = ocn.FieldBoundaryConditions(**pbcs)
    ocn.fill_halo_regions(d)
    ocn.sync_device()
    np.testing.assert_array_equal(from_dev(d), a)


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB"), ((16, 12, 10), "PPB")])
def test_open_boundary_conditions_with_a_value(oracle, ocn, size, topo):
    """OpenBoundaryCondition(value) (boundary_condition.jl; fill_halo_regions_open.jl:9-70): the wall-normal velocity ON the two boundary
    faces of a Bounded direction is set to getbc -- a number or an array over the tangential directions -- by every fill that fills the
    boundary-normal velocities; the default Impenetrable condition writes 0.  Every parent cell against the oracle, on the general path
    (walls in x / y) and on the Periodic-x-y path (w at bottom / top); a tracer refuses the kind."""
    O = oracle
    rng = np.random.default_rng(8)
    og, pg = _pair(O, ocn, size, topo)
    Nx, Ny, Nz = size
    cases = [(4, dict(bottom=0.25, top=rng.uniform(-1, 1, (Nx, Ny))))]
    if topo[1] == "B":
        cases.append((2, dict(south=rng.uniform(-1, 1, (Nx, Nz)), north=-0.5)))
    if topo[0] == "B":
        cases.append((1, dict(west=1.5, east=rng.uniform(-1, 1, (Ny, Nz)))))
    for loc, sides in cases:
        a = random_parent(og, loc, rng)
        d = to_dev(ocn, pg, loc, a)
        O.fill_halo_regions(og, a, loc, bcs={k: O.OpenBoundaryCondition(v) for k, v in sides.items()})
        d.boundary_conditions = ocn.FieldBoundaryConditions(**{k: ocn.OpenBoundaryCondition(v) for k, v in sides.items()})
        ocn.fill_halo_regions(d)
        ocn.sync_device()
        np.testing.assert_array_equal(from_dev(d), a, err_msg=f"{topo} loc {loc}")
        lo, hi = (a[0 + 3], a[-1 - 3]) if loc == 1 else (a[:, 3], a[:, -4]) if loc == 2 else (a[:, :, 3], a[:, :, -4])
        assert np.abs(lo).max() > 0 and np.abs(hi).max() > 0  # the faces carry the prescribed values
        # fill_boundary_normal_velocities = false (update_state!) leaves the faces alone
        b = random_parent(og, loc, rng)
        e = to_dev(ocn, pg, loc, b)
        e.boundary_conditions = d.boundary_conditions
        O.fill_halo_regions(og, b, loc, fill_boundary_normal_velocities=False, bcs={k: O.OpenBoundaryCondition(v) for k, v in sides.items()})
        ocn.fill_halo_regions(e, fill_boundary_normal_velocities=False)
        ocn.sync_device()
        np.testing.assert_array_equal(from_dev(e), b)
    c = ocn.Field(0, pg)
    c.boundary_conditions = ocn.FieldBoundaryConditions(top=ocn.OpenBoundaryCondition(1.0))
    with pytest.raises(ocn.OcnError):
        ocn.fill_halo_regions(c)


def test_uniform_through_flow_with_open_boundaries_is_steady(ocn):
    """NonhydrostaticModel on a (Bounded, Periodic, Bounded) grid with u = U0 prescribed on the west and east faces
    (OpenBoundaryCondition(U0)) and u = U0 inside: the uniform through-flow is an exact steady state of the discrete equations (zero
    advective tendency, zero divergence, zero pressure), so three RK3 steps leave u = U0 on every face, boundary faces included, and
    v = w = 0 -- a closed-form check (the reference has no offline value for an open-boundary run: parity unpinned)."""
    U0 = 0.3
    g = ocn.RectilinearGrid(ocn.GPU(), size=(24, 12, 10), x=(0, 2.0), y=(0, 1.0), z=(-1.0, 0), topology=("Bounded", "Periodic", "Bounded"))
    bcs = {"u": ocn.FieldBoundaryConditions(west=ocn.OpenBoundaryCondition(U0), east=ocn.OpenBoundaryCondition(U0))}
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), boundary_conditions=bcs, math_mode=ocn.MATH_STRICT)
    ocn.set(m, u=np.full((25, 12, 10), U0))
    for _ in range(3):
        ocn.time_step(m, 0.01)
    ocn.flush_tendencies(m)
    ocn.sync_device()
    u = m.u.interior()
    assert u.shape[0] == 25 and np.abs(u - U0).max() <= 1e-13, np.abs(u - U0).max()
    assert np.abs(m.v.interior()).max() <= 1e-13 and np.abs(m.w.interior()).max() <= 1e-13
    with pytest.raises(ValueError, match="normal"):
        ocn.NonhydrostaticModel(g, advection=ocn.WENO(), boundary_conditions={"v": ocn.FieldBoundaryConditions(west=ocn.OpenBoundaryCondition(1.0))})


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB")])
def test_array_and_function_conditions_on_x_y_walls(oracle, ocn, size, topo):
    """Array- and function-valued conditions on the lateral walls (boundary_condition.jl getbc for AbstractArray; continuous_boundary_
    function.jl:17-115 without field dependencies): arrays over the two tangential directions -- (Nx, Nz) on south / north, (Ny, Nz) on
    west / east -- and functions f(x, z, t) / f(y, z, t) of the tangential coordinates at the field's nodes.  Value / Gradient halos and
    the flux contributions to G against the oracle (which reads arrays through the same index rule), bit for bit; the function-valued
    conditions against the arrays of their values."""
    O = oracle
    rng = np.random.default_rng(14)
    og, pg = _pair(O, ocn, size, topo)
    Nx, Ny, Nz = size
    xb = topo[0] == "B"
    # ---- halos: Value / Gradient arrays
    a = random_parent(og, 0, rng)
    d = to_dev(ocn, pg, 0, a)
    sv, ng = rng.uniform(-1, 1, (Nx, Nz)), rng.uniform(-1, 1, (Nx, Nz))
    obcs = {"south": O.ValueBoundaryCondition(sv), "north": O.GradientBoundaryCondition(ng)}
    pbcs = dict(south=ocn.ValueBoundaryCondition(sv), north=ocn.GradientBoundaryCondition(ng))
    if xb:
        wg, ev = rng.uniform(-1, 1, (Ny, Nz)), rng.uniform(-1, 1, (Ny, Nz))
        obcs.update(west=O.GradientBoundaryCondition(wg), east=O.ValueBoundaryCondition(ev))
        pbcs.update(west=ocn.GradientBoundaryCondition(wg), east=ocn.ValueBoundaryCondition(ev))
    O.fill_halo_regions(og, a, 0, bcs=obcs)
    d.boundary_conditions = ocn.FieldBoundaryConditions(**pbcs)
    ocn.fill_halo_regions((d,))
    ocn.sync_device()
    np.testing.assert_array_equal(from_dev(d), a)
    # ---- fluxes: arrays on every lateral wall of a tracer
    c, G = random_parent(og, 0, rng), random_parent(og, 0, rng)
    dc, dG = to_dev(ocn, pg, 0, c), to_dev(ocn, pg, 0, G)
    fl = {"south": rng.uniform(-1, 1, (Nx, Nz)), "north": rng.uniform(-1, 1, (Nx, Nz))}
    if xb:
        fl.update(west=rng.uniform(-1, 1, (Ny, Nz)), east=rng.uniform(-1, 1, (Ny, Nz)))
    O.apply_flux_bcs(og, 0, c, G, {k: O.FluxBoundaryCondition(v) for k, v in fl.items()})
    pb = ocn.FieldBoundaryConditions(**{k: ocn.FluxBoundaryCondition(v) for k, v in fl.items()})
    arr = (C.POINTER(ocn._lib.CFieldBcs) * 1)(C.pointer(pb.c_struct(pg)))
    ocn._lib.call("ocn_apply_flux_bcs", pg.cref, ocn._lib.ptr_array([dG.ptr]), ocn._lib.ptr_array([dc.ptr]), ocn._lib.i32_array([0]), arr, 1, 0)
    ocn.sync_device()
    np.testing.assert_array_equal(from_dev(dG), G)
    # ---- functions of the tangential coordinates and time = the arrays of their values
    t0 = 0.75
    fs = lambda x, z, t: np.sin(3 * x) * z + t
    fw = lambda y, z, t, p: p * np.cos(2 * y) - z * t
    xc, yc, zc = (np.asarray(pg.nodes_1d(dd, 0)).reshape(-1)[:n] for dd, n in zip(range(3), size))
    e = ocn.Field(0, pg)
    e.data.copy_(d.data)
    fb = dict(south=ocn.ValueBoundaryCondition(fs))
    ab = dict(south=ocn.ValueBoundaryCondition(fs(xc[:, None], zc[None, :], t0)))
    if xb:
        fb.update(west=ocn.GradientBoundaryCondition(fw, parameters=0.4))
        ab.update(west=ocn.GradientBoundaryCondition(fw(yc[:, None], zc[None, :], t0, 0.4)))
    d.boundary_conditions = ocn.FieldBoundaryConditions(**fb)
    d.boundary_conditions.refresh(pg, 0, t0)
    e.boundary_conditions = ocn.FieldBoundaryConditions(**ab)
    ocn.fill_halo_regions((d,))
    ocn.fill_halo_regions((e,))
    ocn.sync_device()
    np.testing.assert_array_equal(from_dev(d), from_dev(e))
    with pytest.raises(ValueError):
        ocn.FieldBoundaryConditions(south=ocn.ValueBoundaryCondition(np.zeros((Nx, Ny)))).c_struct(pg)  # wrong shape for a south wall


@pytest.mark.parametrize("size,topo", CASES + BOX_CASES)
@pytest.mark.parametrize("scheme", ["WENO5", "Centered2", "UpwindBiased5"])
def test_advective_tendencies_strict_bitwise(oracle, ocn, size, topo, scheme):
    """compute_Gu! / Gv! / Gw! / Gc! with the order reduction near x / y walls, Flat shortcuts, per-location parent shapes and the
    excluded periphery: bit-identical to the oracle (strict math)"""
    O = oracle
    if scheme != "Centered2" and any(n < 3 for n, t in zip(size, topo) if t != "F"):
        pytest.skip("WENO needs 3 cells")
    rng = np.random.default_rng(11)
    og, pg = _pair(O, ocn, size, topo)
    u, v, w, c = _filled(O, og, rng)
    sch = {"WENO5": O.ADV_WENO5, "Centered2": O.ADV_CENTERED2, "UpwindBiased5": O.ADV_UPWIND5}[scheme]
    G = [og.zeros(l) for l in LOCS]
    Gc = og.zeros(0)
    O.momentum_tendencies(og, u, v, w, *G, scheme=sch)
    O.tracer_tendency(og, u, v, w, c, Gc, scheme=sch)
    ocn.set_math_mode(ocn.MATH_STRICT)
    du, dv, dw, dc = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS + (0,), (u, v, w, c)))
    dG = [ocn.Field(l, pg) for l in LOCS]
    dGc = ocn.Field(0, pg)
    t = ocn._lib.CModelTerms()
    t.advection = {"WENO5": ocn._lib.ADVECTION_WENO5, "Centered2": ocn._lib.ADVECTION_CENTERED2, "UpwindBiased5": ocn._lib.ADVECTION_UPWIND5}[scheme]
    ocn._lib.call("ocn_compute_momentum_tendencies_terms", pg.cref, C.byref(t), du.ptr, dv.ptr, dw.ptr, dG[0].ptr, dG[1].ptr, dG[2].ptr, None, 0)
    ocn._lib.call("ocn_compute_tracer_tendency_terms", pg.cref, C.byref(t), 0.0, None, du.ptr, dv.ptr, dw.ptr, dc.ptr, dGc.ptr, None, 0)
    ocn.sync_device()
    for a, b, name in zip(G + [Gc], dG + [dGc], ("Gu", "Gv", "Gw", "Gc")):
        np.testing.assert_array_equal(from_dev(b), a, err_msg=f"{topo} {scheme} {name}")
    assert sum(np.abs(g).max() > 0 for g in G) >= 2
    if scheme == "WENO5":  # the plain entry points take the same path
        dG2 = [ocn.Field(l, pg) for l in LOCS]
        ocn._lib.call("ocn_compute_momentum_tendencies", pg.cref, du.ptr, dv.ptr, dw.ptr, dG2[0].ptr, dG2[1].ptr, dG2[2].ptr, None, 0)
        ocn.sync_device()
        for a, b in zip(G, dG2):
            np.testing.assert_array_equal(from_dev(b), a)


@pytest.mark.parametrize("size,topo", BOX_CASES)
def test_interior_box_decomposition_equals_the_per_cell_kernel(ocn, size, topo):
    """box (tiled, per-field layouts) + wall frames (per-cell) against the per-cell kernel over the whole grid (OCN_GENERAL_TILED=0, read
    once per process: a child process): the two launch plans evaluate the same expressions on the same operands -- bit for bit in strict
    math, within the fast-math tolerance otherwise"""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
import oceananigans_jl_amd as ocn
size, topo, mode = eval(sys.argv[1]), sys.argv[2], int(sys.argv[3])
T = {"P": "Periodic", "B": "Bounded"}
g = ocn.RectilinearGrid(ocn.GPU(), size=size, x=(0, 1.3), y=(0, 0.9), z=(-0.7, 0), topology=tuple(T[t] for t in topo), halo=(3, 3, 3)).with_math_mode(mode)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
f = [ocn.Field(l, g) for l in (1, 2, 4, 0)]
for q in f:
    q.data.copy_(torch.rand(q.data.shape, generator=gen, device="cuda", dtype=torch.float64) - 0.5)
ocn.fill_halo_regions(f)
G = [ocn.Field(l, g) for l in (1, 2, 4, 0)]
ocn._lib.call("ocn_compute_momentum_tendencies", g.cref, f[0].ptr, f[1].ptr, f[2].ptr, G[0].ptr, G[1].ptr, G[2].ptr, None, 0)
ocn._lib.call("ocn_compute_tracer_tendency", g.cref, f[0].ptr, f[1].ptr, f[2].ptr, f[3].ptr, G[3].ptr, None, 0)
# ... and over a RANGE with the full k extent (the interior range of a slab: the box is intersected with it, the periphery is not excluded),
# with the other terms (Coriolis, stress divergence, tracer diffusion) behind the advective ones
import ctypes as C
R = [ocn.Field(l, g) for l in (1, 2, 4, 0)]
rng = ocn._lib.i32_array([2, g.Nx - 1, 1, g.Ny, 1, g.Nz])
t = ocn._lib.CModelTerms()
t.advection, t.coriolis, t.f, t.closure, t.nu = ocn._lib.ADVECTION_WENO5, 1, 0.7, 1, 0.013
ocn._lib.call("ocn_compute_momentum_tendencies_terms", g.cref, C.byref(t), f[0].ptr, f[1].ptr, f[2].ptr, R[0].ptr, R[1].ptr, R[2].ptr, rng, 0)
ocn._lib.call("ocn_compute_tracer_tendency_terms", g.cref, C.byref(t), 0.021, None, f[0].ptr, f[1].ptr, f[2].ptr, f[3].ptr, R[3].ptr, rng, 0)
ocn.sync_device()
np.savez(sys.argv[4], *[q.data.cpu().numpy() for q in G + R])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    for mode in (ocn.MATH_STRICT, ocn.MATH_FAST):
        outs = []
        with tempfile.TemporaryDirectory() as td:
            for tiled in ("1", "0"):
                path = os.path.join(td, f"g{tiled}.npz")
                p = subprocess.run([sys.executable, "-c", code, repr(size), topo, str(mode), path], env=dict(os.environ, OCN_GENERAL_TILED=tiled),
                                   capture_output=True, text=True, timeout=300, cwd=root)
                assert p.returncode == 0, p.stderr[-2000:]
                outs.append(np.load(path))
            for k in outs[0].files:
                if mode == ocn.MATH_STRICT:
                    np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=f"{topo} strict {k}")
                else:  # fast math: two compilations of the same expressions contract into different FMAs
                    assert np.abs(outs[0][k] - outs[1][k]).max() <= 1e-12 * np.abs(outs[1][k]).max(), f"{topo} fast {k}"
                assert np.abs(outs[0][k]).max() > 0


@pytest.mark.parametrize("size,topo", CASES + BOX_CASES)
def test_advective_tendencies_fast_tolerance(oracle, ocn, size, topo):
    O = oracle
    if any(n < 3 for n, t in zip(size, topo) if t != "F"):
        pytest.skip("WENO needs 3 cells")
    rng = np.random.default_rng(12)
    og, pg = _pair(O, ocn, size, topo)
    u, v, w, c = _filled(O, og, rng)
    G = [og.zeros(l) for l in LOCS]
    O.momentum_tendencies(og, u, v, w, *G)
    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        du, dv, dw = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS, (u, v, w)))
        dG = [ocn.Field(l, pg) for l in LOCS]
        ocn._lib.call("ocn_compute_momentum_tendencies", pg.cref, du.ptr, dv.ptr, dw.ptr, dG[0].ptr, dG[1].ptr, dG[2].ptr, None, 0)
        ocn.sync_device()
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)
    for a, b in zip(G, dG):
        assert np.abs(from_dev(b) - a).max() <= 1e-12 * max(np.abs(a).max(), 1e-300)


@pytest.mark.parametrize("size,topo", CASES + BOX_CASES)
@pytest.mark.parametrize("variant", ["numbers-centered", "numbers-weno", "fields-weno"])
def test_extra_terms_and_diffusion_strict_bitwise(oracle, ocn, size, topo, variant):
    """FPlane Coriolis with the active-node weighting near walls, ScalarDiffusivity stress divergence and tracer diffusion, BuoyancyTracer
    acting on w (no separate pHY'): added to G in the reference's order, bit-identical to the oracle.  On the BOX_CASES the cells a full
    stencil away from the walls take the tiled finishing pass with per-field layouts (stresses shared through LDS) and the tiled tracer
    kernel with the diffusive flux divergence folded in, the frames the per-cell kernels.  "fields": the eddy viscosity / diffusivity arrays
    of an LES closure in place of the numbers, and a separate hydrostatic pressure anomaly with SeawaterBuoyancy."""
    O = oracle
    weno = variant.endswith("weno")
    fields = variant.startswith("fields")
    if weno and any(n < 3 for n, t in zip(size, topo) if t != "F"):
        pytest.skip("WENO needs 3 cells")
    if not weno and (size, topo) in BOX_CASES[2:]:
        pytest.skip("covered by the WENO variants")
    rng = np.random.default_rng(13)
    og, pg = _pair(O, ocn, size, topo)
    u, v, w, c = _filled(O, og, rng)
    sch = O.ADV_WENO5 if weno else O.ADV_CENTERED2
    G = [og.zeros(l) for l in LOCS]
    Gc = og.zeros(0)
    O.momentum_tendencies(og, u, v, w, *G, scheme=sch)
    O.tracer_tendency(og, u, v, w, c, Gc, scheme=sch)
    nu_e = kappa_e = pHY = S = None
    if fields:
        nu_e, kappa_e, pHY, S = (np.abs(random_parent(og, 0, rng)) * sc for sc in (0.02, 0.03, 1.0, 1.0))
        for a in (nu_e, kappa_e, pHY, S):
            O.fill_halo_regions(og, a, 0)
        ph = O.Physics(f=0.7, nu=0.0, kappa=0.0, buoyancy=("SeawaterBuoyancy", 9.81, 2e-4, 8e-4))
        O.momentum_extra_tendencies(og, ph, u, v, w, c, S, pHY, *G, nu_e=nu_e)
        O.tracer_diffusion(og, 0.0, c, Gc, kappa_e=kappa_e)
    else:
        ph = O.Physics(f=0.7, nu=0.013, kappa=0.021, buoyancy="BuoyancyTracer")
        O.momentum_extra_tendencies(og, ph, u, v, w, c, None, None, *G)
        O.tracer_diffusion(og, 0.021, c, Gc)
    ocn.set_math_mode(ocn.MATH_STRICT)
    du, dv, dw, dc = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS + (0,), (u, v, w, c)))
    dG = [ocn.Field(l, pg) for l in LOCS]
    dGc = ocn.Field(0, pg)
    t = ocn._lib.CModelTerms()
    t.advection = ocn._lib.ADVECTION_WENO5 if weno else ocn._lib.ADVECTION_CENTERED2
    t.coriolis, t.f = 1, 0.7
    keep = []
    if fields:
        dn, dk, dp, dS = (to_dev(ocn, pg, 0, a) for a in (nu_e, kappa_e, pHY, S))
        keep = [dn, dk, dp, dS]
        t.closure, t.nu_e = 2, dn.ptr
        t.buoyancy, t.g, t.alpha, t.beta, t.T, t.S, t.pHY = ocn._lib.BUOYANCY_SEAWATER_TS, 9.81, 2e-4, 8e-4, dc.ptr, dS.ptr, dp.ptr
    else:
        t.closure, t.nu = 1, 0.013
        t.buoyancy, t.T = ocn._lib.BUOYANCY_TRACER, dc.ptr
    ocn._lib.call("ocn_compute_momentum_tendencies_terms", pg.cref, C.byref(t), du.ptr, dv.ptr, dw.ptr, dG[0].ptr, dG[1].ptr, dG[2].ptr, None, 0)
    ocn._lib.call("ocn_compute_tracer_tendency_terms", pg.cref, C.byref(t), 0.0 if fields else 0.021, dk.ptr if fields else None, du.ptr, dv.ptr,
                  dw.ptr, dc.ptr, dGc.ptr, None, 0)
    ocn.sync_device()
    for a, b, name in zip(G + [Gc], dG + [dGc], ("Gu", "Gv", "Gw", "Gc")):
        np.testing.assert_array_equal(from_dev(b), a, err_msg=f"{topo} {variant} {name}")
    del keep


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB"), ((16, 1, 10), "PFB")])
def test_substeps_divergence_and_correction_bitwise(oracle, ocn, size, topo):
    """rk3_substep! / ab2_step! skip the wall faces of u, v, w (exclude_periphery), divᶜᶜᶜ and pressure_correct_velocities! use each
    field's own parent shape"""
    O = oracle
    rng = np.random.default_rng(14)
    og, pg = _pair(O, ocn, size, topo)
    U = [random_parent(og, l, rng) for l in LOCS]
    Gn = [random_parent(og, l, rng) for l in LOCS]
    Gm = [random_parent(og, l, rng) for l in LOCS]
    p = random_parent(og, 0, rng)
    dU = [to_dev(ocn, pg, l, a) for l, a in zip(LOCS, U)]
    dGn = [to_dev(ocn, pg, l, a) for l, a in zip(LOCS, Gn)]
    dGm = [to_dev(ocn, pg, l, a) for l, a in zip(LOCS, Gm)]
    dp = to_dev(ocn, pg, 0, p)
    for l, a, gn, gm in zip(LOCS, U, Gn, Gm):
        O.rk3_substep(og, l, a, gn, gm, 0.01, 5 / 12, -17 / 60)
        O.ab2_step(og, l, a, gn, gm, 0.02, 0.1)
    O.pressure_correct(og, *U, p, 0.03)
    div = O.divergence(og, *U)
    pa, ia = ocn._lib.ptr_array, ocn._lib.i32_array
    args = (pg.cref, 3, pa([f.ptr for f in dU]), pa([f.ptr for f in dGn]), pa([f.ptr for f in dGm]), ia(list(LOCS)))
    ocn._lib.call("ocn_rk3_substep", *args, 0.01, 5 / 12, -17 / 60, 1, 0)
    ocn._lib.call("ocn_ab2_step", *args, 0.02, 0.1, 0)
    ocn._lib.call("ocn_pressure_correct_velocities", pg.cref, dU[0].ptr, dU[1].ptr, dU[2].ptr, dp.ptr, 0.03, 0)
    import torch
    dd = torch.zeros((pg.Nz, pg.Ny, pg.Nx), dtype=torch.float64, device="cuda")
    ocn._lib.call("ocn_divergence", pg.cref, dU[0].ptr, dU[1].ptr, dU[2].ptr, dd.data_ptr(), 0)
    ocn.sync_device()
    for a, b in zip(U, dU):
        np.testing.assert_array_equal(from_dev(b), a)
    np.testing.assert_array_equal(dd.cpu().numpy().T, np.asarray(div).reshape(pg.Nx, pg.Ny, pg.Nz))


@pytest.mark.parametrize("size,topo", [((16, 12, 10), "PBB"), ((12, 12, 10), "BBB"), ((32, 1, 16), "PFB")])
@pytest.mark.parametrize("stepper", ["RungeKutta3", "QuasiAdamsBashforth2"])
def test_time_steps_match_oracle_on_closed_and_sliced_grids(oracle, ocn, size, topo, stepper):
    """Three time steps of NonhydrostaticModel(advection = WENO()) with a tracer on a channel, a closed box and an x-z slice against the
    oracle's model (strict math; the cosine-transform Poisson solver differs from pocketfft's DCT in rounding: 1e-11), incompressible
    to 5e-8 like test_time_stepping.jl:125-158, walls impenetrable"""
    O = oracle
    og, pg = _pair(O, ocn, size, topo)
    rng = np.random.default_rng(15)
    ocn.set_math_mode(ocn.MATH_STRICT)
    om = O.NonhydrostaticModel(og, tracers=("c",), timestepper=stepper)
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("c",), timestepper=stepper)
    assert pm.fuse_stage_boundaries == (stepper == "RungeKutta3")  # (round 4: the stage boundaries of grids with walls fuse too)
    init = {}
    for name, loc in zip("uvw", LOCS):
        if topo[{"u": 0, "v": 1, "w": 2}[name]] == "F":
            continue
        init[name] = rng.uniform(-1, 1, og.interior(og.zeros(loc)).shape)
    init["c"] = rng.uniform(0, 1, og.interior(og.zeros(0)).shape)
    om.set(**init)
    ocn.set(pm, **init)
    dt = 2e-3
    for _ in range(3):
        om.time_step(dt)
        ocn.time_step(pm, dt)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    scale = max(np.abs(a).max() for a in (om.u, om.v, om.w))
    for a, f, name in zip((om.u, om.v, om.w, om.tracers[0]), pm.velocities + pm.tracers, ("u", "v", "w", "c")):
        err = np.abs(og.interior(from_dev(f)) - og.interior(a)).max()
        assert err <= 1e-11 * max(scale, 1.0), f"{topo} {stepper} {name}: {err}"
    assert np.abs(og.interior(from_dev(pm.pNHS)) - og.interior(om.p)).max() <= 1e-10 * max(1.0, np.abs(om.p).max())
    import torch
    dd = torch.zeros((pg.Nz, pg.Ny, pg.Nx), dtype=torch.float64, device="cuda")
    ocn._lib.call("ocn_divergence", pg.cref, pm.u.ptr, pm.v.ptr, pm.w.ptr, dd.data_ptr(), 0)
    assert float(dd.abs().max()) < 5e-8
    for f, d, t in zip(pm.velocities, range(3), topo):
        if t == "B":  # impenetrable walls
            a = from_dev(f)
            idx = [slice(None)] * 3
            H = (pg.Hx, pg.Hy, pg.Hz)[d]
            N = (pg.Nx, pg.Ny, pg.Nz)[d]
            for face in (H, H + N):
                idx[d] = face
                assert np.all(a[tuple(idx)] == 0.0)


@pytest.mark.parametrize("size,topo", [((16, 12, 10), "PBB"), ((12, 12, 10), "BBB"), ((32, 1, 16), "PFB"), ((64, 128, 12), "PBB"), ((10, 9, 8), "BPB"),
                                       ((128, 64, 10), "BBB")])
@pytest.mark.parametrize("zkind", ["stretched", "regular"])
def test_fourier_tridiagonal_solver_with_walls_in_x_or_y(oracle, ocn, size, topo, zkind):
    """FourierTridiagonalPoissonSolver on XYRegularRG grids whose x / y are Bounded or Flat (fourier_tridiagonal_poisson_solver.jl:82-147:
    cosine / Fourier transforms along x and y, batched Thomas sweep along z): the only solver of a channel with a stretched z.  Source =
    the divergence of a random velocity (the library's K9 with its Δzᶜ factor); the solution equals the oracle's solver to 1e-10 and
    satisfies the discrete Poisson equation; (64, 128, ·) runs the y lines through the column kernel, (128, 64, ·) "BBB" also the x lines
    through the row kernel (forward and inverse around the Thomas sweep)."""
    from helpers import stretched_faces
    O = oracle
    z = stretched_faces(size[2], 0.7) if zkind == "stretched" else (-0.7, 0)
    og, pg = make_pair(O, ocn, size, topo, x=(0, 1.3), y=(0, 0.9), z=z)
    rng = np.random.default_rng(19)
    U = []
    for loc in LOCS:
        a = og.zeros(loc)
        og.interior(a)[...] = rng.uniform(-1, 1, og.interior(a).shape)
        if topo[{1: 0, 2: 1, 4: 2}[loc]] == "F":
            a[...] = 0
        O.fill_halo_regions(og, a, loc)
        U.append(a)
    dU = [to_dev(ocn, pg, l, a) for l, a in zip(LOCS, U)]
    solver = ocn.nonhydrostatic_pressure_solver(pg) if zkind == "stretched" else ocn.FourierTridiagonalPoissonSolver(pg)
    assert isinstance(solver, ocn.FourierTridiagonalPoissonSolver) and solver.info()["kind"] == 3
    p = ocn.CenterField(pg)
    ocn.solve_for_pressure(p, solver, 0.7, dU)
    ocn.fill_halo_regions(p)
    ocn.sync_device()
    S = O.FourierTridiagonalPoissonSolver(og)
    p0 = og.zeros(0)
    S.source_term(*U, 0.7)
    S.solve(p0)
    got = np.asfortranarray(from_dev(p))
    assert np.abs(og.interior(got) - og.interior(p0)).max() <= 1e-10 * max(1.0, np.abs(p0).max())
    R = O.divergence(og, *U) / 0.7
    lap = O.laplacian(og, got)
    assert np.linalg.norm(lap - R) <= 1e-9 * np.linalg.norm(R)
    # set_source_term! (fourier_tridiagonal_poisson_solver.jl:155-177: multiplies by Δzᶜ itself) gives the same solution
    solver.set_source_term(R)
    q = ocn.CenterField(pg)
    solver.solve(q)
    ocn.sync_device()
    assert np.abs(q.interior() - p.interior()).max() <= 1e-10 * max(1.0, np.abs(p0).max())


@pytest.mark.parametrize("size,topo", [((16, 12, 10), "PBB"), ((12, 12, 10), "BBB"), ((32, 1, 16), "PFB")])
def test_time_steps_match_oracle_with_a_stretched_z_under_walls(oracle, ocn, size, topo):
    """Three RK3 steps (WENO, a tracer) on a channel, a closed box and an x-z slice with a STRETCHED z against the oracle's model: the model
    picks the Fourier-tridiagonal solver (NonhydrostaticModels.jl:25-62: GridWithFourierTridiagonalSolver), strict math, 1e-11."""
    from helpers import stretched_faces
    O = oracle
    og, pg = make_pair(O, ocn, size, topo, x=(0, 1.3), y=(0, 0.9), z=stretched_faces(size[2], 0.7))
    rng = np.random.default_rng(16)
    ocn.set_math_mode(ocn.MATH_STRICT)
    om = O.NonhydrostaticModel(og, tracers=("c",))
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("c",))
    assert isinstance(pm.pressure_solver, ocn.FourierTridiagonalPoissonSolver) and pm.pressure_solver.info()["kind"] == 3
    init = {}
    for name, loc in zip("uvw", LOCS):
        if topo[{"u": 0, "v": 1, "w": 2}[name]] == "F":
            continue
        init[name] = rng.uniform(-1, 1, og.interior(og.zeros(loc)).shape)
    init["c"] = rng.uniform(0, 1, og.interior(og.zeros(0)).shape)
    om.set(**init)
    ocn.set(pm, **init)
    dt = 2e-3
    for _ in range(3):
        om.time_step(dt)
        ocn.time_step(pm, dt)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    scale = max(np.abs(a).max() for a in (om.u, om.v, om.w))
    for a, f, name in zip((om.u, om.v, om.w, om.tracers[0]), pm.velocities + pm.tracers, ("u", "v", "w", "c")):
        err = np.abs(og.interior(from_dev(f)) - og.interior(a)).max()
        assert err <= 1e-11 * max(scale, 1.0), f"{topo} {name}: {err}"
    assert np.abs(og.interior(from_dev(pm.pNHS)) - og.interior(om.p)).max() <= 1e-10 * max(1.0, np.abs(om.p).max())


@pytest.mark.parametrize("topo,fieldname", [("PBB", "c"), ("PBB", "u"), ("BBB", "c"), ("BPB", "c"), ("BPB", "v"), ("PPB", "u"), ("PPB", "v")])
def test_scalar_diffusivity_budget(ocn, topo, fieldname):
    """test/test_dynamics.jl:35-59, 436-455 (test_ScalarDiffusivity_budget on (Periodic, Bounded, Bounded) and (Bounded, Bounded, Bounded);
    velocity components only along Periodic directions, as the reference's loop -- a wall-normal component is projected to zero by set!):
    with no flow and ScalarDiffusivity(ν = κ = 1), the mean of a random u / v / c is conserved over 10 steps of
    Δt = 1e-4 Δz² / κ (isapprox: rtol = sqrt(eps))"""
    P, B = "Periodic", "Bounded"
    t = tuple(B if ch == "B" else P for ch in topo)
    g = ocn.RectilinearGrid(ocn.GPU(), size=(4, 4, 4), extent=(1, 1, 1), topology=t)
    ocn.set_math_mode(ocn.MATH_STRICT)
    m = ocn.NonhydrostaticModel(g, closure=ocn.ScalarDiffusivity(ν=1, κ=1), tracers=("c",), buoyancy=None)
    assert m.grid.Hx >= 1
    rng = np.random.default_rng(16)
    f = m.field(fieldname)
    shape = tuple(reversed(f.interior_view().shape))
    ocn.set(m, u=0, v=0, w=0, c=0)
    ocn.set(m, **{fieldname: rng.random(shape)})
    init_mean = f.interior().mean()
    dz = m.grid.dz
    dt = 1e-4 * dz ** 2 / 1.0
    for _ in range(10):
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m)
    ocn.sync_device()
    final_mean = f.interior().mean()
    assert np.isfinite(final_mean) and abs(final_mean - init_mean) <= np.sqrt(np.finfo(float).eps) * max(abs(init_mean), abs(final_mean))


@pytest.mark.parametrize("size,topo", [((7, 11, 16), "BBB"), ((16, 7, 11), "PBB"), ((32, 20, 12), "BPB"), ((9, 1, 14), "BFB"), ((64, 48, 40), "BBB"),
                                       # y / z lengths the column FFT kernels take (64 ... 512): spectra in stage order, permuted eigenvalues / twiddles
                                       ((16, 64, 128), "BBB"), ((24, 128, 64), "PBB"), ((8, 64, 64), "BPB"), ((12, 64, 256), "PPB"), ((128, 64, 64), "BBB"),
                                       ((4, 512, 64), "PBB"), ((6, 64, 512), "BBB"), ((256, 6, 4), "BBB"), ((512, 4, 1), "BBF"),
                                       # x lines in the row kernel: cosine transforms of a closed box, the FFT of a channel's Periodic x
                                       ((64, 12, 10), "PBB"), ((256, 64, 6), "PBB"), ((128, 6, 1), "PBF"), ((512, 4, 64), "PBB")])
@pytest.mark.parametrize("fused", ["1", "0"])
def test_fft_based_cosine_transforms_equal_direct_sums(ocn, size, topo, fused, monkeypatch):
    """K11 (index_permutations.jl:38-90, discrete_transforms.jl:141-176): the cosine transforms of the general FFTBasedPoissonSolver built
    from complex FFTs of the same length (even / odd permutation + twiddle factors, Makhoul) against the direct O(N) sums of their
    definitions (OCN_POISSON_NAIVE_DCT=1), on odd, prime and mixed sizes: the two solutions of the same Poisson problem agree to 1e-12
    of max|ϕ|, and ∇²ϕ reproduces the zero-mean source to sqrt(eps).  fused = "1" (default): a y / z cosine transform of length 64 ... 512 is
    ONE pass of the column kernel (permutation and twiddle inside, natural wavenumber order); "0": gather, FFT, twiddle passes"""
    import torch
    if fused == "0" and max(size[1:]) < 64 and not (topo in ("BBB", "BBF", "PBB", "PBF") and size[0] in (64, 128, 256, 512)):
        pytest.skip("no column- or row-kernel lengths: the switches change nothing")
    monkeypatch.setenv("OCN_POISSON_FUSED_DCT", fused)
    monkeypatch.setenv("OCN_POISSON_ROW_DCT", fused)  # the x lines of a closed box or a channel: one pass (transform, division, inverse) or 7 / 3
    T = {"P": "Periodic", "B": "Bounded", "F": "Flat"}
    nonflat = [d for d in range(3) if topo[d] != "F"]
    kw = dict(size=tuple(size[d] for d in nonflat), topology=tuple(T[t] for t in topo), halo=tuple(3 for _ in nonflat))
    ext = {"x": (0, 1.3), "y": (0, 0.9), "z": (-0.7, 0)}
    for d, name in enumerate("xyz"):
        if topo[d] != "F":
            kw[name] = ext[name]
    g = ocn.RectilinearGrid(ocn.GPU(), **kw)
    rng = np.random.default_rng(17)
    R = rng.normal(size=size)
    R -= R.mean()
    Rd = torch.from_numpy(np.ascontiguousarray(R.T)).cuda()
    sols = []
    for naive in ("0", "1"):
        monkeypatch.setenv("OCN_POISSON_NAIVE_DCT", naive)
        solver = ocn.FFTBasedPoissonSolver(g, general=True)  # (general: the cosine / Fourier line transforms also where x and y are Periodic)
        p = ocn.CenterField(g)
        ocn._lib.call("ocn_poisson_set_source_term", solver._h, Rd.data_ptr(), 0)
        solver.solve(p)
        ocn.fill_halo_regions(p)
        ocn.sync_device()
        sols.append(p.interior())
        if naive == "0":
            a = p.parent()
            H = [g.Hx, g.Hy, g.Hz]
            N = [g.Nx, g.Ny, g.Nz]
            D = [g.dx, g.dy, g.dz]
            c = a[H[0]:H[0] + N[0], H[1]:H[1] + N[1], H[2]:H[2] + N[2]]
            lap = np.zeros_like(c)
            for d in range(3):
                if topo[d] == "F":
                    continue
                sl = lambda o: tuple(slice(H[q] + (o if q == d else 0), H[q] + N[q] + (o if q == d else 0)) for q in range(3))
                lap += (a[sl(1)] - 2 * c + a[sl(-1)]) / D[d] ** 2
            assert np.abs(lap - R).max() <= np.sqrt(np.finfo(float).eps) * np.abs(R).max()
    assert np.abs(sols[0] - sols[1]).max() <= 1e-12 * np.abs(sols[1]).max()


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB"), ((1, 9, 8), "FBB")])
def test_flux_conditions_on_x_y_walls_bitwise(oracle, ocn, size, topo):
    """apply_x_bcs! / apply_y_bcs! / apply_z_bcs! (apply_flux_bcs.jl:38-160): fluxes through west / east / south / north / bottom / top of a
    tracer, through south / north / top of u and through west / east of w (a Face-in-z field: the face spacing enters area and volume) are
    added to G in the reference's order x, y, z -- corner cells take up to three contributions -- bit for bit, on a stretched z"""
    O = oracle
    rng = np.random.default_rng(21)
    zf = -0.7 * np.linspace(1, 0, size[2] + 1) ** 1.3
    og, pg = make_pair(O, ocn, size, topo, x=(0, 1.3), y=(0, 0.9), z=zf)
    assert og.zf is not None
    xb = topo[0] == "B"
    sides = {
        0: dict(south=(0.3, 0.0), north=(-1.1, 0.2), bottom=(0.7, 0.0), top=(2.0, -0.4), **({"west": (0.9, 0.0), "east": (-0.6, 0.5)} if xb else {})),
        1: dict(south=(1.3, 0.0), north=(0.0, 0.7), top=(-0.2, 0.0)),
        4: ({"west": (0.4, 0.3), "east": (1.9, 0.0)} if xb else {}) | dict(south=(-0.8, 0.0)),
    }
    locs = [0, 1, 4]
    c = [random_parent(og, l, rng) for l in locs]
    G = [random_parent(og, l, rng) for l in locs]
    dc = [to_dev(ocn, pg, l, a) for l, a in zip(locs, c)]
    dG = [to_dev(ocn, pg, l, a) for l, a in zip(locs, G)]
    pb = []
    for l, a, g in zip(locs, c, G):
        O.apply_flux_bcs(og, l, a, g, {k: O.BC("flux", v, coeff) for k, (v, coeff) in sides[l].items()})
        pb.append(ocn.FieldBoundaryConditions(**{k: ocn.FluxBoundaryCondition(v, coeff) for k, (v, coeff) in sides[l].items()}))
    arr = (C.POINTER(ocn._lib.CFieldBcs) * 3)(*[C.pointer(b.c_struct(pg)) for b in pb])
    ocn._lib.call("ocn_apply_flux_bcs", pg.cref, ocn._lib.ptr_array([f.ptr for f in dG]), ocn._lib.ptr_array([f.ptr for f in dc]),
                  ocn._lib.i32_array(locs), arr, 3, 0)
    ocn.sync_device()
    for l, a, b in zip(locs, G, dG):
        np.testing.assert_array_equal(from_dev(b), a, err_msg=f"{topo} loc {l}")


def test_channel_model_with_wall_fluxes_matches_oracle(oracle, ocn):
    """(Periodic, Bounded, Bounded) channel with ScalarDiffusivity, a tracer heated through the south wall and cooled through the north
    one, u dragged at the north wall (condition + coeff * u): 3 RK3 steps against the oracle's model, 1e-11 (cosine-transform rounding)"""
    O = oracle
    og, pg = _pair(O, ocn, (16, 12, 10), "PBB")
    rng = np.random.default_rng(22)
    ocn.set_math_mode(ocn.MATH_STRICT)
    ob = {"c": {"south": O.BC("flux", 0.8), "north": O.BC("flux", 0.8), "top": O.BC("flux", -0.3)}, "u": {"north": O.BC("flux", 0.0, 0.6)}}
    pbc = {"c": ocn.FieldBoundaryConditions(south=ocn.FluxBoundaryCondition(0.8), north=ocn.FluxBoundaryCondition(0.8), top=ocn.FluxBoundaryCondition(-0.3)),
           "u": ocn.FieldBoundaryConditions(north=ocn.FluxBoundaryCondition(0.0, 0.6))}
    om = O.NonhydrostaticModel(og, tracers=("c",), closure=(0.02, {"c": 0.03}), boundary_conditions=ob)
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("c",), closure=ocn.ScalarDiffusivity(ν=0.02, κ=0.03), boundary_conditions=pbc)
    init = {n: rng.uniform(-1, 1, og.interior(og.zeros(l)).shape) for n, l in zip("uvw", LOCS)}
    init["c"] = rng.uniform(0, 1, og.interior(og.zeros(0)).shape)
    om.set(**init)
    ocn.set(pm, **init)
    c0 = og.interior(om.tracers[0]).mean()
    for _ in range(3):
        om.time_step(2e-3)
        ocn.time_step(pm, 2e-3)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    scale = max(np.abs(a).max() for a in (om.u, om.v, om.w))
    for a, f, name in zip((om.u, om.v, om.w, om.tracers[0]), pm.velocities + pm.tracers, ("u", "v", "w", "c")):
        assert np.abs(og.interior(from_dev(f)) - og.interior(a)).max() <= 1e-11 * max(scale, 1.0), name
    # the south flux enters, the north and top fluxes leave: d<c>/dt = 0.8/Ly - 0.8/Ly + 0.3/Lz
    assert abs((og.interior(from_dev(pm.tracers[0])).mean() - c0) / (3 * 2e-3) - 0.3 / 0.7) < 1e-9


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB"), ((16, 1, 10), "PFB"), ((1, 10, 9), "FBB")])
def test_hydrostatic_pressure_anomaly_on_closed_and_sliced_grids(oracle, ocn, size, topo):
    """_update_hydrostatic_pressure! over p_kernel_parameters = 0:N+1 (1:N along a Flat direction, update_hydrostatic_pressure.jl:48-56)
    from mirrored tracer halos, and the tendencies with -∂x pHY', -∂y pHY' on the faces that are not walls: bit-identical to the oracle"""
    O = oracle
    rng = np.random.default_rng(31)
    zf = -0.7 * np.linspace(1, 0, size[2] + 1) ** 1.3
    og, pg = make_pair(O, ocn, size, topo, x=(0, 1.3), y=(0, 0.9), z=zf)
    u, v, w, c = _filled(O, og, rng)
    ph = O.Physics(f=0.7, nu=0.013, kappa=0.021, buoyancy="BuoyancyTracer")
    pHY = og.zeros(0)
    O.update_hydrostatic_pressure(og, ph, c, None, pHY)
    G = [og.zeros(l) for l in LOCS]
    O.momentum_tendencies(og, u, v, w, *G)
    O.momentum_extra_tendencies(og, ph, u, v, w, c, None, pHY, *G)
    ocn.set_math_mode(ocn.MATH_STRICT)
    du, dv, dw, dc = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS + (0,), (u, v, w, c)))
    dp = ocn.Field(0, pg)
    dG = [ocn.Field(l, pg) for l in LOCS]
    t = ocn._lib.CModelTerms()
    t.advection, t.coriolis, t.f, t.closure, t.nu = ocn._lib.ADVECTION_WENO5, 1, 0.7, 1, 0.013
    t.buoyancy, t.T, t.pHY = ocn._lib.BUOYANCY_TRACER, dc.ptr, dp.ptr
    ocn._lib.call("ocn_update_hydrostatic_pressure", pg.cref, C.byref(t), dp.ptr, 0)
    ocn._lib.call("ocn_compute_momentum_tendencies_terms", pg.cref, C.byref(t), du.ptr, dv.ptr, dw.ptr, dG[0].ptr, dG[1].ptr, dG[2].ptr, None, 0)
    ocn.sync_device()
    np.testing.assert_array_equal(from_dev(dp), pHY, err_msg=f"{topo} pHY")
    assert np.abs(pHY).max() > 0
    for a, b, name in zip(G, dG, ("Gu", "Gv", "Gw")):
        np.testing.assert_array_equal(from_dev(b), a, err_msg=f"{topo} {name}")


@pytest.mark.parametrize("topo", ["PBB", "BBB"])
def test_stratified_channel_with_separate_hydrostatic_pressure_matches_oracle(oracle, ocn, topo):
    """BuoyancyTracer with the model's default separate pHY' (nonhydrostatic_model.jl:143-158) on a channel and a closed box, FPlane and
    ScalarDiffusivity: 3 RK3 steps against the oracle's model (1e-11: cosine-transform rounding), walls impenetrable"""
    O = oracle
    og, pg = _pair(O, ocn, (16, 12, 10), topo)
    rng = np.random.default_rng(33)
    ocn.set_math_mode(ocn.MATH_STRICT)
    om = O.NonhydrostaticModel(og, tracers=("b",), coriolis_f=0.3, closure=(0.02, {"b": 0.03}), buoyancy="BuoyancyTracer")
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("b",), coriolis=ocn.FPlane(f=0.3), closure=ocn.ScalarDiffusivity(ν=0.02, κ=0.03),
                                 buoyancy=ocn.BuoyancyTracer())
    assert pm.pHY is not None and pm.fuse_stage_boundaries
    init = {n: rng.uniform(-1, 1, og.interior(og.zeros(l)).shape) for n, l in zip("uvw", LOCS)}
    init["b"] = rng.uniform(0, 1, og.interior(og.zeros(0)).shape)
    om.set(**init)
    ocn.set(pm, **init)
    for _ in range(3):
        om.time_step(2e-3)
        ocn.time_step(pm, 2e-3)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    scale = max(np.abs(a).max() for a in (om.u, om.v, om.w))
    for a, f, name in zip((om.u, om.v, om.w, om.tracers[0]), pm.velocities + pm.tracers, ("u", "v", "w", "b")):
        assert np.abs(og.interior(from_dev(f)) - og.interior(a)).max() <= 1e-11 * max(scale, 1.0), name
    np.testing.assert_allclose(og.interior(from_dev(pm.pHY)), og.interior(om.pHY), rtol=0, atol=1e-12)


@pytest.mark.parametrize("size,topo", [((12, 10, 9), "PBB"), ((11, 9, 8), "BBB"), ((9, 12, 8), "BPP"), ((34, 17, 20), "BBB"), ((16, 1, 10), "PFB"),
                                       ((1, 10, 9), "FBB"), ((12, 1, 9), "BFB")])
def test_amd_diffusivities_on_closed_grids_bitwise(oracle, ocn, size, topo):
    """_compute_AMD_viscosity! / _compute_AMD_diffusivity! (anisotropic_minimum_dissipation.jl:125-169) on grids with a Bounded x / y: every
    staggered field has its own parent shape (the z-marching register-carrying kernel with per-field strides) -- νₑ, κₑ and the variable-ν /
    variable-κ flux divergences bit for bit against the oracle.  x-z and y-z slices (a Flat y / x): the derivatives along the Flat direction
    vanish and its interpolations are the identity (zero strides in the kernel, the one cell there is in the oracle; Δ = 1 there)."""
    O = oracle
    rng = np.random.default_rng(34)
    og, pg = _pair(O, ocn, size, topo)
    u, v, w, c = _filled(O, og, rng)
    nu, ka = og.zeros(0), og.zeros(0)
    O.amd_viscosity(og, 1 / 12, u, v, w, nu)
    O.amd_diffusivity(og, 1 / 7, u, v, w, c, ka)
    ocn.set_math_mode(ocn.MATH_STRICT)
    du, dv, dw, dc = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS + (0,), (u, v, w, c)))
    dnu, dka = ocn.Field(0, pg), ocn.Field(0, pg)
    Ck = (C.c_double * 1)(1 / 7)
    ocn._lib.call("ocn_compute_amd_diffusivities", pg.cref, 1 / 12, du.ptr, dv.ptr, dw.ptr, dnu.ptr, 1, Ck, ocn._lib.ptr_array([dc.ptr]),
                  ocn._lib.ptr_array([dka.ptr]), 0)
    ocn.sync_device()
    np.testing.assert_array_equal(from_dev(dnu), nu, err_msg=f"{topo} nu_e")
    np.testing.assert_array_equal(from_dev(dka), ka, err_msg=f"{topo} kappa_e")
    assert og.interior_N(nu).max() > 0 and og.interior_N(ka).max() > 0
    for a, l in ((nu, 0), (ka, 0)):
        O.fill_halo_regions(og, a, l)
    G = [og.zeros(l) for l in LOCS] + [og.zeros(0)]
    O.momentum_tendencies(og, u, v, w, *G[:3])
    O.momentum_extra_tendencies(og, O.Physics(nu=0.0), u, v, w, None, None, None, *G[:3], nu_e=nu)
    O.tracer_tendency(og, u, v, w, c, G[3])
    O.tracer_diffusion(og, 0.0, c, G[3], kappa_e=ka)
    ocn.fill_halo_regions((dnu, dka))
    t = ocn._lib.CModelTerms()
    t.advection, t.closure, t.nu_e = ocn._lib.ADVECTION_WENO5, 2, dnu.ptr
    dG = [ocn.Field(l, pg) for l in LOCS + (0,)]
    ocn._lib.call("ocn_compute_momentum_tendencies_terms", pg.cref, C.byref(t), du.ptr, dv.ptr, dw.ptr, dG[0].ptr, dG[1].ptr, dG[2].ptr, None, 0)
    ocn._lib.call("ocn_compute_tracer_tendency_terms", pg.cref, C.byref(t), 0.0, dka.ptr, du.ptr, dv.ptr, dw.ptr, dc.ptr, dG[3].ptr, None, 0)
    ocn.sync_device()
    for a, b, name in zip(G, dG, ("Gu", "Gv", "Gw", "Gc")):
        np.testing.assert_array_equal(from_dev(b), a, err_msg=f"{topo} {name}")


@pytest.mark.parametrize("size,topo", [((16, 12, 10), "PBB"), ((32, 1, 16), "PFB"), ((24, 1, 12), "BFB")])
def test_les_channel_with_amd_matches_oracle(oracle, ocn, size, topo):
    """A (Periodic, Bounded, Bounded) channel -- and x-z slices with a Flat y -- with AnisotropicMinimumDissipation, a buoyancy tracer with
    its pHY' and an f-plane: 3 RK3 steps against the oracle's model (the eddy diffusivities are ratios of small numbers: 1e-10 on the fields)"""
    O = oracle
    og, pg = _pair(O, ocn, size, topo)
    rng = np.random.default_rng(35)
    ocn.set_math_mode(ocn.MATH_STRICT)
    om = O.NonhydrostaticModel(og, tracers=("b",), coriolis_f=0.3, closure=("AMD",), buoyancy="BuoyancyTracer")
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("b",), coriolis=ocn.FPlane(f=0.3), closure=ocn.AnisotropicMinimumDissipation(),
                                 buoyancy=ocn.BuoyancyTracer())
    init = {n: rng.uniform(-1, 1, og.interior(og.zeros(l)).shape) for n, l in zip("uvw", LOCS)}
    init["b"] = rng.uniform(0, 1, og.interior(og.zeros(0)).shape)
    om.set(**init)
    ocn.set(pm, **init)
    for _ in range(3):
        om.time_step(2e-3)
        ocn.time_step(pm, 2e-3)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    scale = max(np.abs(a).max() for a in (om.u, om.v, om.w))
    for a, f, name in zip((om.u, om.v, om.w, om.tracers[0]), pm.velocities + pm.tracers, ("u", "v", "w", "b")):
        assert np.abs(og.interior(from_dev(f)) - og.interior(a)).max() <= 1e-10 * max(scale, 1.0), name
    assert np.abs(om.nu_e).max() > 0
    assert np.abs(from_dev(pm.diffusivity_fields["nu_e"]) - om.nu_e).max() <= 1e-6 * np.abs(om.nu_e).max()


def test_horizontal_convection_example_setup_matches_oracle(oracle, ocn):
    """examples/horizontal_convection.jl:23-88 at a reduced size: (Bounded, Flat, Bounded), WENO, RK3, BuoyancyTracer with its pHY',
    ScalarDiffusivity and the surface Value condition b = -cos(2π x / Lx) given as a FUNCTION f(x, t, p) (no y argument on a Flat y):
    5 steps against the oracle's model with the same values as an array, 1e-11; u, w impenetrable, v identically 0"""
    O = oracle
    Nx, Nz, Lx = 32, 16, 2.0
    og = O.Grid((Nx, 1, Nz), x=(-1, 1), y=(0, 1), z=(-1, 0), topology="BFB", halo=(3, 3, 3))
    pg = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nz), x=(-1, 1), z=(-1, 0), topology=("Bounded", "Flat", "Bounded"))
    nu = np.sqrt(1.0 * Lx ** 3 / 1e5)
    xc = og.nodes(0, False)
    vals = (-np.cos(2 * np.pi * xc / Lx)).reshape(Nx, 1)
    ocn.set_math_mode(ocn.MATH_STRICT)
    om = O.NonhydrostaticModel(og, tracers=("b",), closure=(nu, {"b": nu}), buoyancy="BuoyancyTracer",
                               boundary_conditions={"b": {"top": O.ValueBoundaryCondition(vals)}})
    surface = lambda x, t, p: -p["bstar"] * np.cos(2 * np.pi * x / p["Lx"])
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("b",), buoyancy=ocn.BuoyancyTracer(), closure=ocn.ScalarDiffusivity(ν=nu, κ=nu),
                                 boundary_conditions={"b": ocn.FieldBoundaryConditions(top=ocn.ValueBoundaryCondition(surface, parameters=dict(bstar=1.0, Lx=Lx)))})
    rng = np.random.default_rng(36)
    init = {"u": 1e-2 * rng.uniform(-1, 1, og.interior(og.zeros(1)).shape), "w": 1e-2 * rng.uniform(-1, 1, og.interior(og.zeros(4)).shape),
            "b": 1e-2 * rng.uniform(-1, 1, (Nx, 1, Nz))}
    om.set(**init)
    ocn.set(pm, **init)
    for _ in range(5):
        om.time_step(1e-2)
        ocn.time_step(pm, 1e-2)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    scale = max(np.abs(a).max() for a in (om.u, om.w))
    for a, f, name in zip((om.u, om.v, om.w, om.tracers[0]), pm.velocities + pm.tracers, ("u", "v", "w", "b")):
        assert np.abs(og.interior(from_dev(f)) - og.interior(a)).max() <= 1e-11 * max(scale, 1.0), name
    assert np.all(from_dev(pm.v) == 0.0)
    # the surface value enters through the first halo cell: b[0.5 (Nz, Nz + 1)] = the condition
    b = from_dev(pm.tracers[0])
    top = 0.5 * (b[pg.Hx:pg.Hx + Nx, 0, pg.Hz + Nz - 1] + b[pg.Hx:pg.Hx + Nx, 0, pg.Hz + Nz])
    np.testing.assert_allclose(top, vals[:, 0], rtol=0, atol=1e-14)
    assert 0 < ocn.AdvectiveCFL(1e-2)(pm) < 1


def test_horizontal_convection_example_runs(ocn):
    """the example script itself, 60 iterations at 64 x 32: finite, |b| within its surface values, the wizard raises Δt from rest"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "horizontal_convection.py"), "--nx", "64", "--nz", "32", "--max-iterations", "60",
                        "--Ra", "1e6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Iter:     50" in r.stdout and "Nu =" in r.stdout


def test_one_dimensional_diffusion_example(oracle, ocn):
    """examples/one_dimensional_diffusion.jl:21-60: RectilinearGrid(size = 128, z = (-0.5, 0.5), topology = (Flat, Flat, Bounded)),
    ScalarDiffusivity(κ = 1), tracer T = exp(-z² / 2 width²), Δt = 0.1 Δz² / κ, the reference's default Centered advection and RK3:
    1000 iterations.  Bit-identical to the oracle's model on the same grid; the Gaussian spreads as σ² = width² + 2 κ t (second-order
    accurate: 1e-3 of the peak, with the image sources of the two walls); the column integral is conserved by the no-flux walls."""
    O = oracle
    N, width = 128, 0.1
    og = O.Grid((1, 1, N), x=(0, 1), y=(0, 1), z=(-0.5, 0.5), topology="FFB", halo=(3, 3, 3))
    pg = ocn.RectilinearGrid(ocn.GPU(), size=N, z=(-0.5, 0.5), topology=("Flat", "Flat", "Bounded"))
    assert (pg.Nx, pg.Ny, pg.Nz) == (1, 1, N)
    ocn.set_math_mode(ocn.MATH_STRICT)
    om = O.NonhydrostaticModel(og, tracers=("T",), advection="Centered2", closure=(0.0, {"T": 1.0}))
    pm = ocn.NonhydrostaticModel(pg, closure=ocn.ScalarDiffusivity(κ=1), tracers="T")
    zc = og.nodes(2, False)
    T0 = np.exp(-zc ** 2 / (2 * width ** 2)).reshape(1, 1, N)
    om.set(T=T0)
    ocn.set(pm, T=lambda x, y, z: np.exp(-z ** 2 / (2 * width ** 2)))
    dt = 0.1 * pg.dz ** 2 / 1.0
    for _ in range(1000):
        om.time_step(dt)
        ocn.time_step(pm, dt)
    ocn.flush_tendencies(pm)
    ocn.sync_device()
    T = pm.tracers[0].interior()[0, 0, :]
    np.testing.assert_array_equal(T, og.interior(om.tracers[0])[0, 0, :])
    t = 1000 * dt
    s2 = width ** 2 + 2 * t
    gauss = lambda z: np.sqrt(width ** 2 / s2) * np.exp(-z ** 2 / (2 * s2))
    exact = gauss(zc) + gauss(-1 - zc) + gauss(1 - zc)   # the no-flux walls at z = -0.5, 0.5: first image sources
    assert np.abs(T - exact).max() < 1e-3
    assert abs(T.sum() - T0.sum()) < 1e-12 * T0.sum()
    for f in pm.velocities:
        assert np.all(f.interior() == 0.0)
