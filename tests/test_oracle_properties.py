"""The reference's property / known-answer tests for the hot path, re-expressed against the CPU oracle
(SURVEY.md 8c).  These are what pins the oracle, since the reference itself (Julia) cannot run here."""
import ctypes as C

import numpy as np
import pytest

from helpers import stretched_faces


def _rand_velocity(O, g, rng, fill=True):
    out = []
    for loc in (1, 2, 4):
        a = g.zeros(loc)
        g.interior(a)[...] = rng.random(g.interior(a).shape)
        if fill:
            O.fill_halo_regions(g, a, loc)
        out.append(a)
    return out


# ---- test/test_poisson_solvers.jl:58-106: ∇²ϕ ≈ R for random divergent U, all 8 topologies, N in {7, 16} + mixed ----
TOPOS = ["PPP", "PPB", "PBP", "PBB", "BPP", "BPB", "BBP", "BBB"]


@pytest.mark.parametrize("topo", TOPOS)
@pytest.mark.parametrize("size", [(7, 7, 7), (16, 16, 16), (11, 16, 7)])
def test_fft_poisson_laplacian_equals_source(oracle, topo, size):
    O = oracle
    rng = np.random.default_rng(1234)
    g = O.Grid(size, x=(0, 1), y=(0, 2.5), z=(0, 4), topology=topo, halo=(3, 3, 3))
    u, v, w = _rand_velocity(O, g, rng)
    R = O.divergence(g, u, v, w)
    S = O.FFTBasedPoissonSolver(g)
    p = g.zeros(0)
    S.source_term(u, v, w, 1.0)
    S.solve(p)
    O.fill_halo_regions(g, p, 0)
    lap = O.laplacian(g, p)
    assert np.linalg.norm(lap - R) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(R)  # isapprox default rtol


@pytest.mark.parametrize("size,topo", [((16, 16), "PPF"), ((7, 16), "PBF"), ((11, 9), "BBF")])
def test_fft_poisson_2d(oracle, size, topo):
    O = oracle
    rng = np.random.default_rng(2)
    g = O.Grid(size, x=(0, 1), y=(0, 1), topology=topo, halo=(3, 3))
    u, v, w = _rand_velocity(O, g, rng)
    w[...] = 0
    R = O.divergence(g, u, v, w)
    S = O.FFTBasedPoissonSolver(g)
    p = g.zeros(0)
    S.source_term(u, v, w, 1.0)
    S.solve(p)
    O.fill_halo_regions(g, p, 0)
    assert np.linalg.norm(O.laplacian(g, p) - R) <= 1.5e-8 * np.linalg.norm(R)


# ---- test/test_poisson_solvers.jl:109-141: 2nd-order convergence to analytic cos modes ----
@pytest.mark.parametrize("topo", ["PPP", "PPB", "BBB"])
def test_poisson_second_order_convergence(oracle, topo):
    O = oracle
    errs = []
    for N in (32, 64):
        g = O.Grid((N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=topo, halo=(3, 3, 3))
        xc = (np.arange(N) + 0.5) * g.dx
        psi = [np.cos(xc / 2) if t == "B" else np.cos(xc) for t in topo]
        k2 = sum(0.25 if t == "B" else 1.0 for t in topo)
        Psi = psi[0][:, None, None] * psi[1][None, :, None] * psi[2][None, None, :]
        S = O.FFTBasedPoissonSolver(g)
        S.storage[...] = -k2 * Psi
        p = g.zeros(0)
        S.solve(p)
        errs.append(np.abs(g.interior_N(p) - Psi).max())
    rate = np.log2(errs[0] / errs[1])
    assert abs(rate - 2) < 5e-2 * 2 + 0.05


# ---- test/test_poisson_solvers_stretched_grids.jl:12-50: faces [1,2,4,7,11,16,22,29,37] ----
@pytest.mark.parametrize("topo", ["PPB", "PBB", "BPB", "BBB"])
def test_fourier_tridiagonal_poisson_stretched(oracle, topo):
    O = oracle
    rng = np.random.default_rng(3)
    zf = np.array([1, 2, 4, 7, 11, 16, 22, 29, 37], dtype=float)
    g = O.Grid((8, 8, 8), x=(0, 1), y=(0, 1), z=zf, topology=topo, halo=(3, 3, 3))
    u, v, w = _rand_velocity(O, g, rng)
    R = O.divergence(g, u, v, w)
    S = O.FourierTridiagonalPoissonSolver(g)
    p = g.zeros(0)
    S.source_term(u, v, w, 1.0)
    S.solve(p)
    O.fill_halo_regions(g, p, 0)
    assert np.linalg.norm(O.laplacian(g, p) - R) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(R)
    assert abs(g.interior_N(p).mean()) < 1e-12 * np.abs(p).max()  # zero-mean gauge (:142)


def test_fourier_tridiagonal_equals_fft_on_regular_grid(oracle):
    """Both direct solvers solve the same discrete system; on a regular (P,P,B) grid they must agree."""
    O = oracle
    rng = np.random.default_rng(4)
    g = O.Grid((12, 10, 9), x=(0, 1), y=(0, 1), z=(-1, 0), topology="PPB", halo=(3, 3, 3))
    u, v, w = _rand_velocity(O, g, rng)
    p1, p2 = g.zeros(0), g.zeros(0)
    S1, S2 = O.FFTBasedPoissonSolver(g), O.FourierTridiagonalPoissonSolver(g)
    S1.source_term(u, v, w, 0.5); S1.solve(p1)
    S2.source_term(u, v, w, 0.5); S2.solve(p2)
    np.testing.assert_allclose(g.interior_N(p2), g.interior_N(p1), rtol=0, atol=1e-11 * np.abs(p1).max())


# ---- test/test_batched_tridiagonal_solver.jl:7-114 ----
def test_batched_tridiagonal_vs_dense(oracle):
    rng = np.random.default_rng(5)
    Nx, Ny, Nz = 4, 3, 16
    a, c = rng.random(Nz - 1), rng.random(Nz - 1)
    b = 3 + rng.random((Nx, Ny, Nz))
    f = rng.standard_normal((Nx, Ny, Nz)) + 1j * rng.standard_normal((Nx, Ny, Nz))
    phi = oracle.batched_tridiagonal_solve_z(a, b, c, f)
    for i in range(Nx):
        for j in range(Ny):
            M = np.diag(b[i, j]) + np.diag(a, -1) + np.diag(c, 1)
            np.testing.assert_allclose(phi[i, j], np.linalg.solve(M, f[i, j]), rtol=1e-12, atol=1e-13)


# ---- test/test_halo_regions.jl:1-65 ----
@pytest.mark.parametrize("topo", ["PPP", "PPB", "BBB"])
def test_halo_regions(oracle, topo):
    O = oracle
    rng = np.random.default_rng(6)
    g = O.Grid((9, 8, 7), x=(0, 1), y=(0, 1), z=(0, 1), topology=topo, halo=(3, 3, 3))
    a = g.zeros(0)
    g.interior(a)[...] = rng.random((9, 8, 7))
    mask = np.ones(a.shape, bool)
    mask[3:-3, 3:-3, 3:-3] = False
    assert np.all(a[mask] == 0.0)  # halos are zero after set! (new_data zero-initialises)
    O.fill_halo_regions(g, a, 0)
    H = 3
    for d, (t, N) in enumerate(zip(topo, (9, 8, 7))):
        inter = [slice(H, H + n) for n in (9, 8, 7)]
        lo, hi, lo_src, hi_src = list(inter), list(inter), list(inter), list(inter)
        if t == "P":  # periodic: halo == opposite interior
            lo[d], lo_src[d] = slice(0, H), slice(N, N + H)
            hi[d], hi_src[d] = slice(N + H, N + 2 * H), slice(H, 2 * H)
        else:         # no-flux: one mirrored cell
            lo[d], lo_src[d] = slice(H - 1, H), slice(H, H + 1)
            hi[d], hi_src[d] = slice(N + H, N + H + 1), slice(N + H - 1, N + H)
        np.testing.assert_array_equal(a[tuple(lo)], a[tuple(lo_src)])
        np.testing.assert_array_equal(a[tuple(hi)], a[tuple(hi_src)])


def test_impenetrable_wall_velocity(oracle):
    O = oracle
    g = O.Grid((4, 4, 5), x=(0, 1), y=(0, 1), z=(0, 1), topology="PPB", halo=(3, 3, 3))
    w = g.zeros(4)
    assert w.shape == (10, 10, 12)  # Nz+1 faces in a Bounded direction (grid_utils.jl:68)
    w[...] = 1.0
    O.fill_halo_regions(g, w, 4)
    assert np.all(g.interior(w)[:, :, 0] == 0) and np.all(g.interior(w)[:, :, -1] == 0)
    assert np.all(g.interior(w)[:, :, 1:-1] == 1)


# ---- test/test_operators.jl: exact identities of δ/∂/divergence/Laplacian on polynomial data ----
def test_operator_identities(oracle):
    O = oracle
    zf = np.array([0.0, 0.5, 1.5, 3.0, 5.0, 7.5, 10.0])
    g = O.Grid((6, 5, 6), x=(0, 3), y=(0, 2.5), z=zf, topology="PPB", halo=(3, 3, 3))
    # metrics: Δzᶜ equals the face differences, Δzᶠ the centre differences (grid_generation.jl:60-75)
    np.testing.assert_array_equal(g.dzc[3:9], np.diff(zf))
    zc = (zf[1:] + zf[:-1]) / 2
    np.testing.assert_array_equal(g.dzf[4:9], np.diff(zc))
    # div of (a x, b y, c z) linear velocity = a + b + c exactly representable case
    u, v, w = g.zeros(1), g.zeros(2), g.zeros(4)
    xf = np.arange(-3, 9) * g.dx
    u[...] = 2.0 * xf[:, None, None]
    zfh = g.zf  # face positions with halos
    w[...] = 3.0 * zfh[None, None, :w.shape[2]]
    div = O.divergence(g, u, v, w)
    np.testing.assert_allclose(div, 5.0, rtol=0, atol=1e-13)
    # Laplacian of a z-quadratic on the stretched grid matches the finite-volume formula
    p = g.zeros(0)
    zch = np.concatenate([[zc[0] - g.dzf[3] * (3 - i) for i in range(3)], zc, [zc[-1] + g.dzf[9] * (i + 1) for i in range(3)]])
    p[...] = (zch ** 2)[None, None, :]
    lap = O.laplacian(g, p)
    k = 2
    expect = ((zc[k + 1] ** 2 - zc[k] ** 2) / (zc[k + 1] - zc[k]) - (zc[k] ** 2 - zc[k - 1] ** 2) / (zc[k] - zc[k - 1])) / (zf[k + 1] - zf[k])
    np.testing.assert_allclose(lap[2, 2, k], expect, rtol=1e-13)


# ---- test/test_time_stepping.jl:125-158: incompressibility after Nt steps, RK3 and AB2; :160+ tracer conservation ----
@pytest.mark.parametrize("ts", ["RungeKutta3", "QAB2"])
@pytest.mark.parametrize("topo,z", [("PPP", (0, 2 * np.pi)), ("PPB", (0, 2 * np.pi)), ("PPB", "stretched"), ("PPF", None)])
def test_incompressibility_and_conservation(oracle, ts, topo, z):
    O = oracle
    rng = np.random.default_rng(7)
    size = (12, 12, 1) if topo == "PPF" else (12, 12, 12)
    if isinstance(z, str):
        z = stretched_faces(12, 2 * np.pi)
    g = O.Grid(size, x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=topo, halo=(3, 3, 3))
    m = O.NonhydrostaticModel(g, tracers=("c",), timestepper=ts)
    init = {n: rng.uniform(-1, 1, g.interior(f).shape) for n, f in zip("uvw", (m.u, m.v, m.w))}
    if topo == "PPF":
        init["w"][...] = 0
    init["c"] = rng.uniform(0, 1, g.interior(m.tracers[0]).shape)
    m.set(**init)
    vol = (g.dzc[3:3 + g.Nz] if g.dzc is not None else np.full(g.Nz, g.dz))[None, None, :]
    c0 = (g.interior_N(m.tracers[0]) * vol).sum()
    for _ in range(10):
        m.time_step(0.01)
    assert np.abs(O.divergence(g, m.u, m.v, m.w)).max() < 5e-8
    assert np.isfinite(m.u).all()
    c1 = (g.interior_N(m.tracers[0]) * vol).sum()
    assert abs(c1 - c0) <= 1e-12 * abs(c0)
    if topo == "PPF":
        assert np.all(m.w == 0) and np.all(m.Gn[2] == 0)  # Gw ≡ 0 on a Flat z grid (flat_advective_fluxes.jl)


def test_first_ab2_step_is_euler(oracle):
    """test_time_stepping.jl:90-118: the first QAB2 step is forward Euler (χ = -0.5) even if G⁻ holds garbage."""
    O = oracle
    g = O.Grid((4, 4, 4), x=(0, 1), y=(0, 1), z=(0, 1), topology="PPP", halo=(3, 3, 3))
    U, Gn, Gm = g.zeros(0), g.zeros(0), g.zeros(0)
    U[...] = 1.0; Gn[...] = 2.0; Gm[...] = np.nan
    O.ab2_step(g, 0, U, Gn, Gm, 0.5, -0.5)
    assert np.all(g.interior(U) == 2.0)
    O.ab2_step(g, 0, U, Gn, Gn, 0.5, 0.1)  # regular AB2: u += Δt((1.5+χ)G - (0.5+χ)G) = Δt G
    np.testing.assert_allclose(g.interior(U), 3.0, rtol=1e-15)


# ---- validation/convergence_tests/one_dimensional_advection_schemes.jl:47-63: WENO(order=5) converges at 5th order ----
def test_weno5_fifth_order_and_upwinding(oracle):
    L = oracle.lib()
    errs = []
    for N in (16, 32, 64):
        h = 2 * np.pi / N
        avg = lambda a, b: (np.cos(a) - np.cos(b)) / h  # cell average of sin over [a, b]
        e = 0.0
        for i in range(N):
            xf = i * h  # face
            S = np.array([avg(xf + (m) * h, xf + (m + 1) * h) for m in range(-3, 3)])
            for left in (1, 0):
                val = L.ocn_oracle_weno5(S.ctypes.data_as(C.c_void_p), left)
                e = max(e, abs(val - np.sin(xf)))
        errs.append(e)
    r1, r2 = np.log2(errs[0] / errs[1]), np.log2(errs[1] / errs[2])
    assert r1 > 4.5 and r2 > 4.7
    # a discontinuity: the reconstruction stays within the data range (essentially non-oscillatory), both biases
    S = np.array([0.0, 0.0, 0.0, 1.0, 1.0, 1.0])
    for left in (1, 0):
        v = L.ocn_oracle_weno5(S.ctypes.data_as(C.c_void_p), left)
        assert -1e-6 <= v <= 1 + 1e-6
    # constants are reproduced exactly up to round-off (weights sum to one)
    S = np.full(6, 3.25)
    assert abs(L.ocn_oracle_weno5(S.ctypes.data_as(C.c_void_p), 1) - 3.25) < 1e-14
    S4 = np.array([1.0, 2.0, 3.0, 4.0])
    assert abs(L.ocn_oracle_centered4(S4.ctypes.data_as(C.c_void_p)) - 2.5) < 1e-14  # exact for linear data


def test_upwind_biased5_fifth_order(oracle):
    """validation/convergence_tests/one_dimensional_advection_schemes.jl:47-63 lists UpwindBiased(order=5) next to WENO: the
    fixed 5-point stencils converge at 5th order from cell averages, both biases; the 3-point fallback at 3rd order."""
    L = oracle.lib()
    e5, e3 = [], []
    for N in (16, 32, 64):
        h = 2 * np.pi / N
        avg = lambda a, b: (np.cos(a) - np.cos(b)) / h
        m5 = m3 = 0.0
        for i in range(N):
            xf = i * h
            S = np.array([avg(xf + m * h, xf + (m + 1) * h) for m in range(-3, 3)])
            S4 = np.ascontiguousarray(S[1:5])
            for left in (1, 0):
                m5 = max(m5, abs(L.ocn_oracle_upwind5(S.ctypes.data_as(C.c_void_p), left) - np.sin(xf)))
                m3 = max(m3, abs(L.ocn_oracle_upwind3(S4.ctypes.data_as(C.c_void_p), left) - np.sin(xf)))
        e5.append(m5)
        e3.append(m3)
    assert np.log2(e5[0] / e5[1]) > 4.7 and np.log2(e5[1] / e5[2]) > 4.9
    assert np.log2(e3[0] / e3[1]) > 2.7 and np.log2(e3[1] / e3[2]) > 2.9
    S = np.full(6, 3.25)
    assert abs(L.ocn_oracle_upwind5(S.ctypes.data_as(C.c_void_p), 1) - 3.25) < 1e-14
