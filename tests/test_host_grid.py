"""Host logic of the product (grid generation, argument checks) against the oracle's independent restatement."""
import numpy as np
import pytest

from helpers import stretched_faces


@pytest.fixture(scope="module")
def pkg():
    import oceananigans_jl_amd as ocn
    return ocn


@pytest.mark.parametrize("N,ext", [(16, (0, 2 * np.pi)), (13, (-0.3, 1.7)), (7, (1, 2)), (512, (0, 2 * np.pi))])
def test_regular_spacing_matches_oracle(oracle, pkg, N, ext):
    og = oracle.Grid((N, N, N), x=ext, y=ext, z=ext, topology="PPP", halo=(3, 3, 3))
    pg = pkg.RectilinearGrid(None, size=(N, N, N), x=ext, y=ext, z=ext, topology=("Periodic",) * 3, halo=(3, 3, 3))
    assert (pg.dx, pg.dy, pg.dz) == (og.dx, og.dy, og.dz)
    assert (pg.Lx, pg.Ly, pg.Lz) == (og.Lx, og.Ly, og.Lz)
    assert pg.parent_shape(1) == og.shape(1) == (N + 6,) * 3


def test_stretched_z_matches_oracle(oracle, pkg):
    zf = stretched_faces(12, 3.0)
    og = oracle.Grid((8, 8, 12), x=(0, 1), y=(0, 1), z=zf, topology="PPB", halo=(3, 3, 3))
    pg = pkg.RectilinearGrid(None, size=(8, 8, 12), x=(0, 1), y=(0, 1), z=zf, topology=("Periodic", "Periodic", "Bounded"))
    np.testing.assert_array_equal(pg._dzc_host, og.dzc)
    np.testing.assert_array_equal(pg._dzf_host, og.dzf)
    assert pg.Lz == og.Lz
    assert pg.parent_shape(4) == og.shape(4) == (14, 14, 19)  # Face field in a Bounded direction: N+1+2H
    assert pg.parent_shape(0) == (14, 14, 18)


def test_flat_dimension_and_defaults(pkg):
    g = pkg.RectilinearGrid(None, size=(128, 128), x=(0, 2 * np.pi), y=(0, 2 * np.pi), topology=("Periodic", "Periodic", "Flat"))
    assert g.size == (128, 128, 1) and (g.Hx, g.Hy, g.Hz) == (3, 3, 0)
    assert g.dz == 1.0 and g.Lz == 1.0  # Flat metrics (grid_generation.jl:138-155)
    assert g.parent_shape(4) == (134, 134, 1)


def test_grid_argument_errors(pkg):
    P = "Periodic"
    with pytest.raises(ValueError, match="must have 3 elements"):
        pkg.RectilinearGrid(None, size=(8, 8), x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, P))
    with pytest.raises(ValueError, match="increasing interval"):
        pkg.RectilinearGrid(None, size=(8, 8, 8), x=(1, 0), y=(0, 1), z=(0, 1), topology=(P, P, P))
    with pytest.raises(ValueError, match="halo"):
        pkg.RectilinearGrid(None, size=(2, 8, 8), x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, P), halo=(3, 3, 3))
    with pytest.raises(ValueError, match="increasing"):
        pkg.RectilinearGrid(None, size=(8, 8, 2), x=(0, 1), y=(0, 1), z=[0, 2, 1], topology=(P, P, "Bounded"))
    with pytest.raises(NotImplementedError):
        pkg.RectilinearGrid(None, size=(8, 8, 2), x=[0, 1, 2, 3, 4, 5, 6, 7, 9], y=(0, 1), z=(0, 1), topology=(P, P, P))


def test_weno_scheme_arguments(pkg):
    assert pkg.WENO().buffer == 3
    with pytest.raises(ValueError, match="odd orders"):
        pkg.WENO(order=4)
    with pytest.raises(NotImplementedError):
        pkg.WENO(order=7)


def test_node_coordinates(pkg):
    """grid_generation.jl:34-135: faces c1 + (i-1) Δ, centres half a spacing further; N+1 faces in a Bounded dimension;
    stretched z centres are face mid-points."""
    g = pkg.RectilinearGrid(None, size=(4, 5, 3), x=(0, 1), y=(-1, 1.5), z=[-4.0, -2.0, -0.5, 0.0],
                            topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
    assert np.array_equal(g.nodes_1d(0, 1), [0.0, 0.25, 0.5, 0.75])
    assert np.array_equal(g.nodes_1d(0, 0), [0.125, 0.375, 0.625, 0.875])
    assert np.allclose(g.nodes_1d(1, 0), -1 + 0.5 * (np.arange(5) + 0.5), rtol=0, atol=1e-16)
    assert np.array_equal(g.nodes_1d(2, 4), [-4.0, -2.0, -0.5, 0.0])
    assert np.array_equal(g.nodes_1d(2, 0), [-3.0, -1.25, -0.25])
    x, y, z = g.nodes(1)
    assert x.shape == (4, 1, 1) and y.shape == (1, 5, 1) and z.shape == (1, 1, 3)


# ---- SplitExplicitFreeSurface constructor (split_explicit_free_surface.jl:120-263) --------------------------------------------------
def test_split_explicit_free_surface_settings(pkg):
    """substeps = N, the MINIMUM_SUBSTEPS default of the disambiguation method (:178-179), cfl-based FixedTimeStepSize (:217-235)
    with substeps recomputed from the baroclinic Δt (step_split_explicit_free_surface.jl:54-58), cfl + fixed_Δt (:171-176); the
    weights equal the oracle's independent restatement, are normalised and centred (Σ aₘ m / M ≈ 1, the docstring's condition)."""
    from oracle import hydrostatic as Hy
    S = pkg.SplitExplicitFreeSurface
    for n in (5, 12, 30, 51):
        fs = S(substeps=n)
        frac, w = fs.settings(1.0)
        ofrac, ow = Hy.weights_from_substeps(n)
        assert frac == ofrac == 2.0 / n and np.array_equal(w, ow)
        assert abs(w.sum() - 1) < 1e-15 and len(w) <= n
        if n >= 12:
            assert abs((w * np.arange(1, len(w) + 1)).sum() * frac - 1) < 0.05
    assert len(S().settings(1.0)[1]) == len(S(substeps=5).settings(1.0)[1])           # MINIMUM_SUBSTEPS
    g = pkg.RectilinearGrid(None, size=(8, 8, 4), x=(0, 8e3), y=(0, 4e3), z=(-100.0, 0.0), topology=("Periodic", "Periodic", "Bounded"))
    with pytest.raises(ValueError):
        S(cfl=0.7)                                                                     # the grid is required with cfl
    fs = S(g, cfl=0.7)
    ds = np.sqrt(1 / (1 / g.dx ** 2 + 1 / g.dy ** 2))
    dtb = 0.7 * ds / np.sqrt(fs.gravitational_acceleration * 100.0)
    assert abs(fs.Δt_barotropic - dtb) < 1e-15 * dtb
    for dt in (0.5 * dtb, 10 * dtb, 33.3 * dtb):
        frac, w = fs.settings(dt)
        n = max(5, int(np.ceil(2 * dt / dtb)))
        assert frac == 2.0 / n and np.array_equal(w, Hy.weights_from_substeps(n)[1])
    fs = S(g, cfl=0.7, fixed_Δt=10 * dtb)
    assert fs.settings(123.0)[0] == 2.0 / int(np.ceil(2 * 10 * dtb / dtb))


def test_implicit_free_surface_arguments(pkg):
    """ImplicitFreeSurface(; solver_method, gravitational_acceleration) (implicit_free_surface.jl:79-80): only the FFT solver exists here"""
    fs = pkg.ImplicitFreeSurface(gravitational_acceleration=3.0)
    assert fs.gravitational_acceleration == 3.0
    pkg.ImplicitFreeSurface(solver_method=":FastFourierTransform")
    with pytest.raises(NotImplementedError):
        pkg.ImplicitFreeSurface(solver_method=":PreconditionedConjugateGradient")


def test_hydrostatic_timestepper_argument(pkg):
    """HydrostaticFreeSurfaceModel(; timestepper = :QuasiAdamsBashforth2 | :SplitRungeKutta3) (hydrostatic_free_surface_model.jl): the
    name is validated before anything touches a device"""
    import inspect
    sig = inspect.signature(pkg.HydrostaticFreeSurfaceModel.__init__)
    assert sig.parameters["timestepper"].default == "QuasiAdamsBashforth2"


def test_julia_range_restatement_properties():
    """grids._JuliaRange restates Base's TwicePrecision `range(start, stop, length = n)`; Julia is not installed, so the general case has
    no reference value to compare with (the pins are the doctest prints of tests/golden/reference_fixtures.json).  What bounds a wrong
    branch (the rational one, the truncated-step one with its bits = min(27, ...) rule, either reference index): for random (start,
    stop, n) -- negative, irrational, tiny and huge end points -- the first and last elements ARE the end points, the elements are strictly
    increasing (wherever the step is resolved by doubles of that magnitude), and each is within 1 ulp of the exactly interpolated rational start + (i - 1) (stop - start) / (n - 1)."""
    import math
    from fractions import Fraction
    from oceananigans_jl_amd.grids import _JuliaRange
    rng = np.random.default_rng(2024)
    cases = [(0.0, 2 * math.pi, 129), (-math.pi, math.pi, 65), (0.0, 1.0, 4), (-1.0, 0.0, 1025), (0.1, 0.3, 3), (1e-9, 3e-9, 17),
             (-5e6, 7.25e6, 513), (-0.7, 0.0, 9), (1 / 3, 2 / 3, 11), (-2.0 ** 0.5, 3.0 ** 0.5, 258)]
    for _ in range(300):
        a = float(rng.choice([rng.uniform(-10, 10), rng.uniform(-1e-6, 1e-6), rng.uniform(-1e7, 1e7), float(rng.integers(-50, 50)), rng.normal()]))
        L = float(rng.choice([rng.uniform(1e-3, 10), rng.uniform(1e-9, 1e-6), rng.uniform(1e3, 1e7), float(rng.integers(1, 100)), math.pi * rng.uniform(0.1, 4)]))
        cases.append((a, a + L, int(rng.integers(2, 2000))))
    for start, stop, n in cases:
        if not stop > start:
            continue
        r = _JuliaRange(start, stop, n)
        vals = [r[i] for i in range(1, n + 1)]
        assert vals[0] == start and vals[-1] == stop, (start, stop, n)
        resolved = (stop - start) / (n - 1) > 4 * math.ulp(max(abs(start), abs(stop)))  # (a step below the spacing of doubles cannot be strict)
        assert all((b > a) if resolved else (b >= a) for a, b in zip(vals, vals[1:])), (start, stop, n)
        fs, fe = Fraction(start), Fraction(stop)
        for i in (1, 2, n // 3 + 1, n // 2 + 1, n - 1, n):
            if not 1 <= i <= n:
                continue
            exact = fs + (fe - fs) * Fraction(i - 1, n - 1)
            scale = max(abs(start), abs(stop))  # ulp of the range's magnitude: elements near zero of a wide range are not relatively exact
            assert abs(Fraction(vals[i - 1]) - exact) <= Fraction(math.ulp(scale)), (start, stop, n, i)
