"""Every value the reference still holds OFFLINE for the hot path's inputs -- jldoctest outputs and @test known answers -- checked on
the oracle (test infrastructure) AND on the product's host side (oceananigans.jl_amd/grids.py, fields.py, physics.py, models.py).
Data: tests/golden/reference_fixtures.json (values + file:line provenance; no reference code).  Julia is absent, so these are the
only reference-produced numbers that exist in this project; DESIGN.md section 6 lists what each pins and what stays unpinned.

What they pin on the hot path: Δx, Δy, Δz of regular grids; Δzᵃᵃᶜ / Δzᵃᵃᶠ of stretched grids BIT FOR BIT (docs/src/fields.md prints
them with 17 digits) -- the metric vectors every kernel reads; the halo layout and the periodic / no-flux fill (fields.md prints a
filled parent array); node coordinates (inputs of set! and of boundary-condition functions) including Julia's TwicePrecision range
arithmetic; the halo the model gives its grid; the constants of FPlane / SeawaterBuoyancy."""
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")))
P, B, F = "Periodic", "Bounded", "Flat"


def num(v):
    if isinstance(v, str):
        import re
        return eval(re.sub(r"(\d)\s*pi", r"\1*pi", v).replace("pi", "math.pi"), {"math": math})  # "2pi" -> 2 * π like Julia's 2π
    return v


def sig6(x):
    """Julia prints these through prettysummary: 6 significant digits"""
    return float("%.6g" % x)


def same_print(x, text):
    want = float(text)
    return sig6(x) == want and (math.copysign(1, x) == math.copysign(1, want) or x != 0)


def product_grid(spec, halo=None):
    import oceananigans_jl_amd as ocn
    kw = dict(size=tuple(spec["size"]), topology=tuple(spec["topology"]))
    if "extent" in spec:
        kw["extent"] = tuple(num(v) for v in spec["extent"])
    for c in "xy":
        if c in spec:
            kw[c] = tuple(num(v) for v in spec[c])
    if "z" in spec:
        kw["z"] = tuple(num(v) for v in spec["z"])
    if "z_faces" in spec:
        kw["z"] = np.array(spec["z_faces"], dtype=float)
    if "halo_arg" in spec:
        kw["halo"] = tuple(spec["halo_arg"])
    if halo is not None:
        kw["halo"] = halo
    return ocn.RectilinearGrid(None, **kw)  # architecture None: host metadata only, no device


def oracle_grid(spec):
    from oracle import oracle as O
    topo = "".join(t[0] for t in spec["topology"])
    nonflat = [d for d in range(3) if topo[d] != "F"]
    size = list(spec["size"])
    if "extent" in spec:
        Ls = [num(v) for v in spec["extent"]]
        it = iter(Ls)
        full = [None if topo[d] == "F" else next(it) for d in range(3)]
        x, y, z = (None if full[0] is None else (0.0, full[0]), None if full[1] is None else (0.0, full[1]),
                   None if full[2] is None else (-full[2], 0.0))
    else:
        x = tuple(num(v) for v in spec["x"]) if "x" in spec else None
        y = tuple(num(v) for v in spec["y"]) if "y" in spec else None
        z = tuple(num(v) for v in spec["z"]) if "z" in spec else (np.array(spec["z_faces"], dtype=float) if "z_faces" in spec else None)
    halo = None
    if "halo_arg" in spec:
        it = iter(spec["halo_arg"])
        halo = [0 if topo[d] == "F" else next(it) for d in range(3)]
    assert len(size) == len(nonflat)
    return O.Grid(size, x=x, y=y, z=z, topology=topo, halo=halo)


@pytest.mark.parametrize("spec", FX["grid_show"], ids=[s["source"] for s in FX["grid_show"]])
def test_grid_show_doctests(spec):
    """the `show(grid)` outputs of the reference's jldoctests: halo, domain end points ξ[1], ξ[N+1] and spacings"""
    pg, og = product_grid(spec), oracle_grid(spec)
    assert [pg.Hx, pg.Hy, pg.Hz] == spec["halo"] == [og.Hx, og.Hy, og.Hz]
    for d in range(3):
        if spec["domain"][d] is None:
            assert pg.topology[d] == F
            continue
        lo, hi = pg.domain(d)
        olo, ohi = og.coordinate(d, True)[[(og.Hx, og.Hy, og.Hz)[d], (og.Hx, og.Hy, og.Hz)[d] + (og.Nx, og.Ny, og.Nz)[d]]]
        assert (lo, hi) == (olo, ohi)                                    # product == oracle, bit for bit
        assert same_print(lo, spec["domain"][d][0]) and same_print(hi, spec["domain"][d][1]), (d, lo, hi, spec["domain"][d])
        if spec["spacing"][d] is not None:
            assert same_print((pg.dx, pg.dy, pg.dz)[d], spec["spacing"][d]) and (pg.dx, pg.dy, pg.dz)[d] == (og.dx, og.dy, og.dz)[d]
    if "dz_min_max" in spec:
        lo, hi = pg.spacing_extrema(2)
        assert same_print(lo, spec["dz_min_max"][0]) and same_print(hi, spec["dz_min_max"][1])
        assert (lo, hi) == (og.dzc[og.Hz:og.Hz + og.Nz].min(), og.dzc[og.Hz:og.Hz + og.Nz].max())


def _faces(case):
    N, L = case["N"], case["L"]
    if case["formula"] == "hyperbolic":
        s = case["sigma"]
        return [-L * (1 - math.tanh(s * (k - 1) / N) / math.tanh(s)) for k in range(1, N + 2)]
    if case["formula"] == "chebychev_centered":
        return [-L / 2 * math.cos(math.pi * (j - 1) / N) for j in range(1, N + 2)]
    if case["formula"] == "chebychev_y":
        return [L * (1 - math.cos(math.pi * (j - 1) / N)) / 2 for j in range(1, N + 2)]
    if case["formula"] == "chebychev_z":
        return [-L * (1 + math.cos(math.pi * (k - 1) / N)) / 2 for k in range(1, N + 2)]
    raise KeyError(case["formula"])


@pytest.mark.parametrize("case", FX["stretched_faces"], ids=[c["source"] for c in FX["stretched_faces"]])
def test_variably_spaced_doctests(case):
    """`variably spaced with min(Δ)=…, max(Δ)=…` of the reference's jldoctests, through generate_coordinate's explicit-face branch
    (grid_generation.jl:34-95).  The y-direction examples run through the same branch as z (the product stretches z only)."""
    import oceananigans_jl_amd as ocn
    from oracle import oracle as O
    faces = np.array(_faces(case))
    N = case["N"]
    pg = ocn.RectilinearGrid(None, size=(4, 4, N), x=(0, 1), y=(0, 1), z=faces, topology=(P, P, B))
    og = O.Grid((4, 4, N), x=(0, 1), y=(0, 1), z=faces, topology="PPB")
    lo, hi = pg.spacing_extrema(2)
    assert same_print(lo, case["min"]) and same_print(hi, case["max"]), (lo, hi)
    assert np.array_equal(pg._dzc_host, og.dzc) and np.array_equal(pg._dzf_host, og.dzf)
    zlo, zhi = pg.domain(2)
    assert same_print(zlo, case["domain"][0]) and same_print(zhi, case["domain"][1])
    # the same faces given as a function of the face index, as the doctests give them
    pf = ocn.RectilinearGrid(None, size=(4, 4, N), x=(0, 1), y=(0, 1), z=lambda k: _faces(case)[k - 1], topology=(P, P, B))
    assert np.array_equal(pf._dzc_host, pg._dzc_host)


def test_stretched_spacings_and_nodes_bit_for_bit():
    """docs/src/fields.md prints Δzᵃᵃᶜ, Δzᵃᵃᶠ and the z nodes of z = [0, 0.1, 0.3, 0.6, 1] with all their digits: the metric vectors the
    kernels read are reproduced BIT FOR BIT by the oracle and by grids.py"""
    import oceananigans_jl_amd as ocn
    from oracle import oracle as O
    fx = FX["fields_md_stretched"]
    z = np.array(fx["z_faces"], dtype=float)
    H = fx["halo"]
    pg = ocn.RectilinearGrid(None, size=(4, 5, 4), x=(0, 1), y=(0, 1), z=z, topology=(P, P, B), halo=(H, H, H))
    og = O.Grid((4, 5, 4), x=(0, 1), y=(0, 1), z=z, topology="PPB", halo=(H, H, H))
    for dzc, dzf in ((pg._dzc_host, pg._dzf_host), (og.dzc, og.dzf)):        # element 0 <-> k = 1 - H
        assert list(dzc[H:H + 4]) == fx["zspacings_center_1_4"]
        assert list(dzf[H:H + 5]) == fx["zspacings_face_1_5"]
    assert list(pg.nodes_1d(2, False, with_halos=True)) == fx["znodes_center_with_halos_0_5"] == list(og.nodes(2, False, True))
    assert list(pg.nodes_1d(2, False)) == fx["znodes_center"] == list(og.nodes(2, False))
    assert list(pg.nodes_1d(2, True)) == fx["znodes_face"] == list(og.nodes(2, True))


def test_minimum_spacings_doctest():
    import oceananigans_jl_amd as ocn
    fx = FX["minimum_spacings"]
    g = ocn.RectilinearGrid(None, size=tuple(fx["size"]), extent=tuple(fx["extent"]))
    assert (g.dx, g.dy, g.dz) == (fx["x"], fx["y"], fx["z"])
    assert oracle_grid(dict(size=fx["size"], extent=fx["extent"], topology=[P, P, B])).d == [fx["x"], fx["y"], fx["z"]]


def test_test_grids_known_answers():
    """test/test_grids.jl: halo faces, first cell centres, end faces (exact ==), array lengths, the (0, π) grid"""
    import oceananigans_jl_amd as ocn
    t = FX["test_grids"]
    c = t["halo_faces"]
    N, H, L = c["N"], c["H"], c["L"]
    D = L / N
    for mk in ("product", "oracle"):
        if mk == "product":
            g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, L), y=(0, L), z=(0, L), halo=(H, H, H), topology=tuple(c["topology"]))
            xf, yf, zf = (g.nodes_1d(d, True, with_halos=True) for d in range(3))
        else:
            from oracle import oracle as O
            g = O.Grid((N, N, N), x=(0, L), y=(0, L), z=(0, L), halo=(H, H, H), topology="PBB")
            xf, yf, zf = (g.nodes(d, True, True) for d in range(3))
        at = lambda a, i: a[i - 1 + H]                       # OffsetArray index -> position
        assert at(xf, 0) == -H * D and at(yf, 0) == -H * D and at(zf, 0) == -H * D
        assert at(xf, N + 1) == L and at(yf, N + 2) == L + H * D and at(zf, N + 2) == L + H * D
    c = t["first_cells"]
    N, H, L = c["N"], c["H"], c["L"]
    g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, L), y=(0, L), z=(0, L), halo=(H, H, H))
    assert all(g.nodes_1d(d, False)[0] == (L / N) / 2 for d in range(3))
    c = t["lengths"]
    g = ocn.RectilinearGrid(None, size=tuple(c["size"]), extent=(1, 1, 1), halo=tuple(c["halo"]), topology=tuple(c["topology"]))
    for d, (n, h) in enumerate(zip(c["size"], c["halo"])):
        assert len(g.nodes_1d(d, False, with_halos=True)) == n + 2 * h and len(g.nodes_1d(d, True, with_halos=True)) == n + 1 + 2 * h
    N = t["pi_grid"]["N"]
    g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, math.pi), y=(0, math.pi), z=(0, math.pi))
    for d in range(3):
        assert np.isclose(g.nodes_1d(d, False)[1], math.pi / 2, rtol=1e-15) and np.isclose(g.nodes_1d(d, True)[1], math.pi / 3, rtol=1e-15)
        assert np.isclose((g.dx, g.dy, g.dz)[d], math.pi / 3, rtol=1e-15)


def test_field_nodes_set_and_layout_doctests():
    """docs/src/fields.md: node vectors, set!(c, f(x, y, z)) values, the one-dimensional grid -- host side of the product and the oracle"""
    import oceananigans_jl_amd as ocn
    from oracle import oracle as O
    fx = FX["fields_md_nodes_and_set"]
    z = np.array(FX["fields_md_stretched"]["z_faces"], dtype=float)
    pg = ocn.RectilinearGrid(None, size=(4, 5, 4), x=(0, 1), y=(0, 1), z=z, topology=(P, P, B), halo=(1, 1, 1))
    og = O.Grid((4, 5, 4), x=(0, 1), y=(0, 1), z=z, topology="PPB", halo=(1, 1, 1))
    assert list(pg.nodes_1d(0, False)) == fx["xnodes_center"] == list(og.nodes(0, False))
    assert list(pg.nodes_1d(0, True)) == fx["xnodes_face"] == list(og.nodes(0, True))
    x, y, zz = pg.nodes(0)                                      # Center, Center, Center
    c = 2 * x + 0 * y + 0 * zz                                  # fun_stuff(x, y, z) = 2x
    assert list(c[:, 0, 0]) == fx["set_2x_center_column"]
    assert (c.max(), c.min(), c.mean()) == (fx["set_2x_center_stats"]["max"], fx["set_2x_center_stats"]["min"], fx["set_2x_center_stats"]["mean"])
    xu = pg.nodes(1)[0]
    assert list((2 * xu)[:, 0, 0]) == fx["set_2x_xface_column"]
    od = fx["one_d"]
    g1 = ocn.RectilinearGrid(None, size=od["size"], x=tuple(od["x"]), topology=(P, F, F))
    v = 3 * g1.nodes_1d(0, False)
    assert (v.max(), v.min(), v.mean()) == (od["max"], od["min"], od["mean"])
    assert g1.parent_shape(0) == (od["parent_x_indices"][1] - od["parent_x_indices"][0] + 1, 1, 1)
    # the halo fill the doctest prints (periodic x, y; the k = 1 plane), on the oracle's restatement of fill_halo_regions!
    a = og.zeros(0)
    og.interior(a)[...] = c
    assert np.array_equal(a[:, :, 1], np.array(fx["before_fill_k1"]))
    O.fill_halo_regions(og, a, 0)
    assert np.array_equal(a[:, :, 1], np.array(fx["after_fill_k1"]))
    assert list(a[0:2, 1, 1]) == fx["parent_1_2__2_2"] and list(og.interior(a)[0:2, 0, 0]) == fx["data_1_2__1_1"]


@pytest.mark.gpu
def test_field_halo_fill_doctest_on_the_gpu(ocn):
    """the same printed parent array, produced by ocn_fill_halo_regions through the C ABI"""
    fx = FX["fields_md_nodes_and_set"]
    z = np.array(FX["fields_md_stretched"]["z_faces"], dtype=float)
    g = ocn.RectilinearGrid(ocn.GPU(), size=(4, 5, 4), x=(0, 1), y=(0, 1), z=z, topology=(P, P, B), halo=(1, 1, 1))
    c = ocn.CenterField(g)
    c.set(lambda x, y, z: 2 * x)
    a = c.parent()
    assert np.array_equal(a[:, :, 1], np.array(fx["before_fill_k1"]))
    ocn.fill_halo_regions(c)
    ocn.sync_device()
    a = c.parent()
    assert np.array_equal(a[:, :, 1], np.array(fx["after_fill_k1"]))
    assert list(a[0:2, 1, 1]) == fx["parent_1_2__2_2"]


def test_halo_inflation_known_answers(monkeypatch):
    """test/test_nonhydrostatic_models.jl:34-66: the model rebuilds its grid with the halo the advection scheme needs
    (inflate_grid_halo_size, nonhydrostatic_model.jl:243-257).  Checked on the constructor logic alone (no device)."""
    import oceananigans_jl_amd as ocn
    from oceananigans_jl_amd import models
    fx = FX["halo_inflation"]
    seen = {}

    class Stop(Exception):
        pass

    def spy(grid, *a, **k):          # first thing the constructor does with the (possibly rebuilt) grid
        seen["halo"] = [grid.Hx, grid.Hy, grid.Hz]
        raise Stop

    monkeypatch.setattr(models, "XFaceField", spy)
    schemes = {"Centered2": ocn.Centered(), "WENO": ocn.WENO(), "UpwindBiased5": ocn.UpwindBiased(order=5)}
    for case in fx["cases"]:
        g = ocn.RectilinearGrid(None, size=tuple(fx["size"]), extent=tuple(fx["extent"]), halo=tuple(case["halo"]))
        with pytest.raises(Stop):
            ocn.NonhydrostaticModel(g, advection=schemes[case["advection"]])
        assert seen["halo"] == case["expect"], case


def test_coriolis_and_buoyancy_constants():
    import oceananigans_jl_amd as ocn
    from oceananigans_jl_amd import physics
    from oracle import julia_base
    c, b = FX["coriolis"], FX["buoyancy"]
    assert ocn.FPlane(f=math.pi).f == math.pi
    assert ocn.FPlane(rotation_rate=2, latitude=30).f == c["fplane_rate2_lat30"]          # exact, as Julia's sind(30) == 0.5
    assert ocn.FPlane.OMEGA_EARTH == c["omega_earth"] and ocn.FPlane(latitude=90).f == 2 * c["omega_earth"]
    for deg in (0, 17, 30, 45, 70, 90, 135, 180, 210, 270, 330, -30, -45, 360, 390.5):
        assert physics.sind(deg) == julia_base.sind(deg)
        assert abs(physics.sind(deg) - math.sin(math.radians(deg))) <= 1e-15
    eos = ocn.LinearEquationOfState()
    assert (eos.thermal_expansion, eos.haline_contraction) == (b["default_thermal_expansion"], b["default_haline_contraction"])
    assert ocn.SeawaterBuoyancy().gravitational_acceleration == b["g_Earth"]
    eos = ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)      # test_buoyancy.jl:10-13
    assert (eos.thermal_expansion, eos.haline_contraction) == (2e-4, 8e-4)


def test_julia_range_restatements_agree():
    """the product's _JuliaRange and the oracle's julia_range are independent restatements of Base._linspace: they must agree on
    nice and on irrational end points, and reduce to exact arithmetic where the end points are exact ratios"""
    from oceananigans_jl_amd.grids import _JuliaRange
    from oracle import julia_base
    rng = np.random.default_rng(7)
    cases = [(0.0, 1.0, 5), (-0.75, 8.5, 38), (-3 * (2 * math.pi / 32), 34 * (2 * math.pi / 32), 38), (1e-3, 7e5, 1000), (-1.0, -0.25, 4)]
    cases += [(float(a), float(a + abs(b) + 1e-9), int(n)) for a, b, n in zip(rng.normal(size=20), rng.normal(size=20), rng.integers(2, 600, 20))]
    for a, b, n in cases:
        r = _JuliaRange(a, b, n)
        got = [r[i] for i in range(1, n + 1)]
        assert got == julia_base.julia_range(a, b, n)
        assert got[0] == a and got[-1] == b
        assert np.allclose(got, np.linspace(a, b, n), rtol=1e-14, atol=1e-15 * max(abs(a), abs(b)))


def test_beta_plane_constructor_known_answers():
    """test/test_coriolis.jl:40-53, 112-120: BetaPlane(f₀ = π, β = 2π) keeps its arguments; BetaPlane(latitude = 70, radius = 2π, rotation_rate = 3π)
    gives f₀ = 6π sind(70), β = 6π cosd(70) / 2π; neither / both argument sets is an ArgumentError; and cosd has Julia's exact values"""
    import math
    import oceananigans_jl_amd as ocn
    from oceananigans_jl_amd import physics
    c = ocn.BetaPlane(f0=math.pi, beta=2 * math.pi)
    assert c.f0 == math.pi and c.beta == 2 * math.pi
    c = ocn.BetaPlane(latitude=70, radius=2 * math.pi, rotation_rate=3 * math.pi)
    assert c.f0 == 6 * math.pi * physics.sind(70)
    assert c.beta == 6 * math.pi * physics.cosd(70) / (2 * math.pi)
    assert abs(physics.cosd(70) - math.cos(math.radians(70))) <= 1e-15
    assert physics.cosd(60) == 0.5 and physics.cosd(90) == 0.0 and physics.cosd(0) == 1.0 and physics.cosd(180) == -1.0 and physics.cosd(120) == -0.5
    for kw in ({}, dict(f0=1.0), dict(f0=1.0, beta=2.0, latitude=10)):
        with pytest.raises(ValueError):
            ocn.BetaPlane(**kw)
    earth = ocn.BetaPlane(latitude=45)
    assert earth.f0 == 2 * 7.292115e-5 * physics.sind(45) and earth.beta == 2 * 7.292115e-5 * physics.cosd(45) / 6371.0e3



def test_diffusive_cfl_doctest():
    """src/Diagnostics/cfl.jl:66-77: DiffusiveCFL(0.1)(model) == 0.256 for ScalarDiffusivity(ν = 1e-2) on a 16^3 grid of extent 1
    (cell_diffusion_timescale = min(Δ² / ν, Δ² / max κ) with no tracers: turbulence_closure_diagnostics.jl:20-47) -- host logic only"""
    from types import SimpleNamespace
    import oceananigans_jl_amd as ocn
    g = ocn.RectilinearGrid(None, size=(16, 16, 16), extent=(1, 1, 1))
    model = SimpleNamespace(grid=g, closure=ocn.ScalarDiffusivity(ν=1e-2), tracer_names=(), diffusivity_fields=None)
    assert ocn.DiffusiveCFL(0.1)(model) == 0.256
    assert ocn.cell_diffusion_timescale(SimpleNamespace(grid=g, closure=None, tracer_names=(), diffusivity_fields=None)) == float("inf")
    model = SimpleNamespace(grid=g, closure=ocn.ScalarDiffusivity(ν=1e-2, κ={"T": 4e-2, "S": 1e-3}), tracer_names=("T", "S"), diffusivity_fields=None)
    assert ocn.cell_diffusion_timescale(model) == (1 / 16) ** 2 / 4e-2
