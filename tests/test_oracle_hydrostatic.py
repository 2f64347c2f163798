"""CPU oracle of the first HydrostaticFreeSurfaceModel slice (oracle/hydrostatic.py): identities of the
restated definitions and a linear free-surface wave.  PARITY UNPINNED against the reference (its tests of this model are
time-stepping smoke tests, test_hydrostatic_free_surface_models.jl:10-30, re-expressed here as `time_step works`)."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import hydrostatic as Hy


def _grid(N=(16, 12, 6), L=(2.0e3, 1.5e3), H=50.0, stretched=False):
    z = (-H, 0.0) if not stretched else -H * (1 - np.linspace(0, 1, N[2] + 1) ** 1.5)[::-1] * 1.0
    if stretched:
        z = np.sort(-H * (np.linspace(1, 0, N[2] + 1) ** 1.5))
    return O.Grid(N, x=(0, L[0]), y=(0, L[1]), z=z, topology="PPB", halo=(3, 3, 3))


@pytest.mark.parametrize("stretched", [False, True])
def test_w_from_continuity_makes_the_flow_nondivergent(stretched):
    """compute_w_from_continuity.jl:31-40: w integrates -div_xy(u, v) from w[1] = 0, so divᶜᶜᶜ(u, v, w) vanishes cell by cell."""
    g = _grid(stretched=stretched)
    rng = np.random.default_rng(3)
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="WENO5")
    m.set(u=rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)), v=rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)))
    assert np.all(m.w[:, :, g.Hz] == 0)
    d = O.divergence(g, m.u, m.v, m.w)
    scale = np.abs(m.u).max() / g.dx
    assert np.abs(d).max() <= 1e-13 * scale
    # the surface value is minus the divergence of the depth-integrated transport (what the free surface feels)
    dzc = np.full(g.Nz, g.dz) if g.dzc is None else np.asarray(g.dzc[g.Hz:g.Hz + g.Nz])
    U = (g.interior_N(m.u) * dzc).sum(axis=2)
    V = (g.interior_N(m.v) * dzc).sum(axis=2)
    divUV = (np.roll(U, -1, 0) - U) / g.dx + (np.roll(V, -1, 1) - V) / g.dy
    wtop = m.w[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, g.Hz + g.Nz]
    assert np.abs(wtop + divUV).max() <= 1e-12 * np.abs(divUV).max()


@pytest.mark.parametrize("advection", ["Centered2", "WENO5"])
def test_time_step_works_and_conserves_volume(advection):
    """time_step_hydrostatic_model_works (test_hydrostatic_free_surface_models.jl:10-30) + the explicit free surface conserves
    the mean elevation on a periodic domain: sum(Gη) = sum(w_top) = -sum(div_xy(transport)) = 0."""
    g = _grid()
    rng = np.random.default_rng(4)
    m = Hy.HydrostaticFreeSurfaceModel(g, tracers=("T", "S"), momentum_advection=advection, coriolis_f=1e-4, closure=(1e-2, 1e-3),
                                       buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4))
    m.set(u=1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)), v=1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)),
          eta=1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny)), T=20 + 1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)), S=35.0)
    eta0 = m.eta[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny].sum()
    for _ in range(5):
        m.time_step(1.0)
    assert all(np.isfinite(f).all() for f in m.fields) and np.isfinite(m.eta).all()
    eta1 = m.eta[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny].sum()
    assert abs(eta1 - eta0) <= 1e-12 * g.Nx * g.Ny * 1e-2
    assert m.iteration == 5 and abs(m.time - 5.0) < 1e-14


def test_linear_surface_gravity_wave_period():
    """A small standing wave η = a cos(kx) over a flat bottom, no rotation, no stratification: the barotropic mode of the
    hydrostatic equations oscillates at ω² = g H k_d², k_d = (2/Δx) sin(kΔx/2) the discrete wavenumber of the C-grid
    gradient / divergence pair (continuous limit: the shallow-water speed sqrt(gH))."""
    Nx, H, L = 32, 20.0, 4.0e3
    g = O.Grid((Nx, 4, 4), x=(0, L), y=(0, 500.0), z=(-H, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="Centered2")
    a, k = 1e-4, 2 * np.pi / L
    x = (np.arange(Nx) + 0.5) * g.dx
    m.set(eta=a * np.cos(k * x)[:, None] * np.ones((1, 4)))
    kd = 2 / g.dx * np.sin(k * g.dx / 2)
    omega = np.sqrt(Hy.g_Earth * H) * kd
    period = 2 * np.pi / omega
    nsteps = 800
    dt = period / nsteps
    amp = []
    for n in range(nsteps):
        m.time_step(dt)
        e = m.eta[g.Hx:g.Hx + Nx, g.Hy]
        amp.append(2 * np.mean(e * np.cos(k * x)))   # projection on the initial mode
    amp = np.array(amp) / a
    t = dt * np.arange(1, nsteps + 1)
    # follows cos(ω t): back at +1 after one period, -1 at half a period, 0 at the quarter
    assert abs(amp[-1] - 1.0) < 5e-3
    assert abs(amp[nsteps // 2 - 1] + 1.0) < 5e-3
    assert abs(amp[nsteps // 4 - 1]) < 1e-2
    assert np.abs(amp - np.cos(omega * t)).max() < 1e-2
    # the velocity is depth-independent (barotropic) and 90 degrees out of phase
    u = g.interior_N(m.u)
    assert np.abs(u - u[:, :, :1]).max() <= 1e-12 * max(np.abs(u).max(), 1e-30)


def test_vector_invariant_advection_is_second_order():
    """VectorInvariant() (enstrophy-conserving vorticity flux + energy-conserving vertical advection and KE gradient,
    vector_invariant_advection.jl:269-361) against the analytic U·∇u of a horizontally non-divergent Taylor-Green flow: zero for a
    uniform flow, error ratio ~4 between resolutions 16 and 32."""
    errs = []
    for N in (16, 32):
        L = 2.0e3
        g = O.Grid((N, N, 4), x=(0, L), y=(0, L), z=(-40.0, 0.0), topology="PPB", halo=(3, 3, 3))
        m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="VectorInvariant")
        k = 2 * np.pi / L
        xc, xf = (np.arange(N) + 0.5) * g.dx, np.arange(N) * g.dx
        u0 = np.sin(k * xf)[:, None, None] * np.cos(k * xc)[None, :, None] * np.ones((1, 1, 4))
        v0 = -np.cos(k * xc)[:, None, None] * np.sin(k * xf)[None, :, None] * np.ones((1, 1, 4))
        m.set(u=u0, v=v0)
        m.update_state(True)
        assert np.abs(g.interior(m.w)).max() <= 1e-12          # discretely non-divergent in the horizontal
        # u ∂x u + v ∂y u = k sin(kx) cos(kx) (cos² + sin²)(ky) = (k/2) sin(2 k x) at the u points
        exact = 0.5 * k * np.sin(2 * k * xf)[:, None, None] * np.ones((1, N, 4))
        errs.append(np.abs(-g.interior_N(m.Gn[0]) - exact).max())
    assert errs[0] / errs[1] > 3.5
    g = O.Grid((8, 8, 4), x=(0, 1.0), y=(0, 1.0), z=(-1.0, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="VectorInvariant")
    m.set(u=np.full((8, 8, 4), 0.3), v=np.full((8, 8, 4), -0.2))
    m.update_state(True)
    assert np.abs(g.interior_N(m.Gn[0])).max() == 0 and np.abs(g.interior_N(m.Gn[1])).max() == 0


def test_default_model_time_steps():
    """the reference's default configuration of this slice: VectorInvariant momentum, Centered tracers
    (time_step_hydrostatic_model_works, test_hydrostatic_free_surface_models.jl:10-30)"""
    g = _grid(stretched=True)
    rng = np.random.default_rng(8)
    m = Hy.HydrostaticFreeSurfaceModel(g, tracers=("T", "S"), momentum_advection="VectorInvariant", coriolis_f=1e-4, closure=(1e-2, 1e-3),
                                       buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4))
    m.set(u=1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)), v=1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)),
          T=20 + 1e-2 * rng.uniform(-1, 1, (g.Nx, g.Ny, g.Nz)), S=35.0)
    for _ in range(4):
        m.time_step(1.0)
    assert all(np.isfinite(f).all() for f in m.fields) and np.isfinite(m.eta).all() and np.abs(m.eta).max() > 0


# ---- SplitExplicitFreeSurface: the reference's own solver tests re-expressed (test_split_explicit_free_surface_solver.jl) ----------
def _sefs_setup():
    Nx, Ny = 128, 64
    L = 2 * np.pi
    grav = Hy.g_Earth
    H = 1 / grav                     # Lz = 1 / g_Earth: g H = 1 (:21)
    dx, dy = L / Nx, L / Ny
    xc, xf = (np.arange(Nx) + 0.5) * dx, np.arange(Nx) * dx
    z = lambda: np.zeros((Nx, Ny))
    return Nx, Ny, dx, dy, xc, xf, grav, H, z


def test_split_explicit_one_timestep():
    """:36-56: η = sin x, one substep with Δτ = 1: U = -cos(x_face) within 1e-3"""
    Nx, Ny, dx, dy, xc, xf, grav, H, z = _sefs_setup()
    _, weights = Hy.weights_from_substeps(200, Hy.constant_averaging_kernel)
    eta = np.sin(xc)[:, None] * np.ones((1, Ny))
    U, V = z(), z()
    Hy.iterate_split_explicit(eta, U, V, z(), z(), z(), z(), z(), 1.0, weights[:1], grav, H, dx, dy)
    assert np.abs(U - (-np.cos(xf))[:, None]).max() < 1e-3


def test_split_explicit_wave_returns_after_one_period():
    """:58-100: g H = 1, k = 1: after T = 2π the forward-backward substepping brings η back to sin x (1e-6) and U to 0 (1e-3)"""
    Nx, Ny, dx, dy, xc, xf, grav, H, z = _sefs_setup()
    T = 2 * np.pi
    dtau = 2 * np.pi / max(Nx, Ny) * 5e-2
    Nt = int(np.floor(T / dtau))
    dtau_end = T - Nt * dtau
    _, weights = Hy.weights_from_substeps(Nt, Hy.constant_averaging_kernel)
    eta0 = np.sin(xc)[:, None] * np.ones((1, Ny))
    eta, U, V = eta0.copy(), z(), z()
    etab, Ub, Vb, GU, GV = z(), z(), z(), z(), z()
    for _ in range(Nt):
        Hy.iterate_split_explicit(eta, U, V, etab, Ub, Vb, GU, GV, dtau, weights[:1], grav, H, dx, dy)
    Hy.iterate_split_explicit(eta, U, V, etab, Ub, Vb, GU, GV, dtau_end, weights[:1], grav, H, dx, dy)
    assert np.abs(U).max() < 1e-3
    assert np.abs(eta - eta0).max() < 1e-6


def test_split_explicit_averaging_does_nothing_to_a_uniform_state():
    """:107-155: uniform η, U, V are fixed points and their constant-kernel averages reproduce them (100 eps)"""
    Nx, Ny, dx, dy, xc, xf, grav, H, z = _sefs_setup()
    frac, weights = Hy.weights_from_substeps(200, Hy.constant_averaging_kernel)
    assert len(weights) == 200 and abs(frac - 0.01) < 1e-15
    eta, U, V = z() + 1.0, z() + 2.0, z() + 3.0
    etab, Ub, Vb = z(), z(), z()
    dtau = 2 * np.pi / max(Nx, Ny) * 1e-2
    for _ in range(len(weights)):
        Hy.iterate_split_explicit(eta, U, V, etab, Ub, Vb, z(), z(), dtau, weights[:1], grav, H, dx, dy)
    tol = 100 * np.finfo(float).eps
    for a, v in ((eta, 1.0), (U, 2.0), (V, 3.0), (etab, 1.0), (Ub, 2.0), (Vb, 3.0)):
        assert np.abs(a - v).max() < tol


def test_split_explicit_forced_two_dimensional_wave():
    """:157-238 ("Complex Multi-Timestep"): ∂ₜη + ∇·U = 0, ∂ₜU + ∇η = G with η₀ = sin(2x) sin(3y) + 1, U₀ = V₀ = 0, constant
    forcing G = (1, 2): mean(η) conserved to 10 eps; η, U, V and their time averages match the analytic solution within 1e-2."""
    Nx, Ny, dx, dy, xc, xf, grav, H, z = _sefs_setup()
    yc, yf = (np.arange(Ny) + 0.5) * dy, np.arange(Ny) * dy
    kx, ky = 2, 3
    om = np.sqrt(kx ** 2 + ky ** 2)
    T = 2 * np.pi / om / 3 * 2
    dtau = 2 * np.pi / max(Nx, Ny) * 1e-2
    Nt = int(np.floor(T / dtau))
    dtau_end = T - Nt * dtau
    _, weights = Hy.weights_from_substeps(Nt + 1, Hy.constant_averaging_kernel)
    eta0 = np.sin(kx * xc)[:, None] * np.sin(ky * yc)[None, :] + 1
    eta, U, V = eta0.copy(), z(), z()
    etab, Ub, Vb = z(), z(), z()
    GU, GV = z() + 1.0, z() + 2.0
    mean_before = eta.mean()
    for _ in range(Nt):
        Hy.iterate_split_explicit(eta, U, V, etab, Ub, Vb, GU, GV, dtau, weights[:1], grav, H, dx, dy)
    Hy.iterate_split_explicit(eta, U, V, etab, Ub, Vb, GU, GV, dtau_end, weights[:1], grav, H, dx, dy)
    assert abs(eta.mean() - mean_before) < 10 * np.finfo(float).eps
    e0 = eta0[:, 0]
    U0 = kx * np.cos(kx * xf) * np.sin(ky * yc[0])
    V0 = ky * np.sin(kx * xc) * np.cos(ky * yf[0])
    eta_exact = np.cos(om * T) * (e0 - 1) + 1
    U_exact = -(np.sin(om * T) / om) * U0 + 1.0 * T
    V_exact = -(np.sin(om * T) / om) * V0 + 2.0 * T
    etab_exact = (np.sin(om * T) / om) / T * (e0 - 1) + 1
    Ub_exact = (np.cos(om * T) / om ** 2 - 1 / om ** 2) / T * U0 + 1.0 * T / 2
    Vb_exact = (np.cos(om * T) / om ** 2 - 1 / om ** 2) / T * V0 + 2.0 * T / 2
    tol = 1e-2
    assert np.abs(U[:, 0] - U_exact).max() / np.abs(U_exact).max() < tol
    assert np.abs(V[:, 0] - V_exact).max() / np.abs(V_exact).max() < tol
    assert np.abs(eta[:, 0] - eta_exact).max() / np.abs(eta_exact).max() < tol
    assert np.abs(Ub[:, 0] - Ub_exact).max() < tol
    assert np.abs(Vb[:, 0] - Vb_exact).max() < tol
    assert np.abs(etab[:, 0] - etab_exact).max() < tol


def test_split_explicit_weights():
    """weights_from_substeps (split_explicit_free_surface.jl:228-241) with the default Shchepetkin-McWilliams kernel: truncated where
    the kernel turns negative, normalised, centred on the baroclinic step (Σ aₘ m/M ≈ 1, the kernel's design condition)"""
    frac, w = Hy.weights_from_substeps(30)
    assert len(w) == 21 and abs(frac - 2 / 30) < 1e-15
    assert abs(w.sum() - 1) < 1e-14 and w[-1] > 0 and w[:3].max() < 0   # the kernel starts slightly negative (its -r τ/τ₀ term)
    tau = np.linspace(0, 2, 31)[1:]
    assert Hy.averaging_shape_function(tau[len(w)]) < 0                   # the first dropped weight would be negative
    assert abs((w * np.arange(1, len(w) + 1)).sum() * frac - 1) < 0.02


def test_split_explicit_model_wave_and_barotropic_consistency():
    """The whole QAB2 step with the split-explicit free surface (ab2_step! -> substepping -> barotropic corrector): a long
    surface wave at a baroclinic gravity-wave CFL of 0.8 stays stable and close to cos(ω t) (the averaging damps it slightly),
    mean(η) is conserved and Σ Δz u equals the barotropic transport after every step."""
    Nx, H, L = 32, 20.0, 4.0e3
    g = O.Grid((Nx, 4, 4), x=(0, L), y=(0, 500.0), z=(-H, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="Centered2", split_explicit_substeps=30)
    a, k = 1e-4, 2 * np.pi / L
    x = (np.arange(Nx) + 0.5) * g.dx
    m.set(eta=a * np.cos(k * x)[:, None] * np.ones((1, 4)))
    kd = 2 / g.dx * np.sin(k * g.dx / 2)
    omega = np.sqrt(Hy.g_Earth * H) * kd
    dt = 2 * np.pi / omega / 40
    assert np.sqrt(Hy.g_Earth * H) * dt / g.dx > 0.75
    amp = []
    for n in range(80):
        m.time_step(dt)
        e = m.eta[g.Hx:g.Hx + Nx, g.Hy:g.Hy + 4]
        amp.append(2 * np.mean(e[:, 0] * np.cos(k * x)) / a)
        assert abs(e.sum()) < 1e-16 * Nx * 4
        assert np.abs(m._barotropic_mode(m.u) - m.U).max() <= 8 * np.finfo(float).eps * max(np.abs(m.U).max(), 1e-30)
    amp = np.array(amp)
    t = dt * np.arange(1, 81)
    assert np.abs(amp - np.cos(omega * t)).max() < 0.12
    assert 0.9 < amp[79] < 1.0


# ---- test_split_explicit_vertical_integrals.jl re-expressed --------------------------------------------------------------------
def _barotropic_kernels_model():
    Nx, Ny, Nz = 128, 64, 32
    L = 2 * np.pi
    g = O.Grid((Nx, Ny, Nz), x=(0, L), y=(0, L), z=(-L, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="Centered2", split_explicit_substeps=200)
    xc, xf = (np.arange(Nx) + 0.5) * g.dx, np.arange(Nx) * g.dx
    yc, yf = (np.arange(Ny) + 0.5) * g.dy, np.arange(Ny) * g.dy
    zc = -L + (np.arange(Nz) + 0.5) * g.dz
    return g, m, L, xc, xf, yc, yf, zc


def test_barotropic_mode_vertical_integrals():
    """:52-111: compute_barotropic_mode! integrates cos(πz/2Lz) to 2Lz/π within 1e-3, a z-independent field exactly to Lz x field,
    and sin(x) z cos(y) to -sin(x) Lz²/2 cos(y)"""
    g, m, L, xc, xf, yc, yf, zc = _barotropic_kernels_model()
    ones = np.ones((g.Nx, g.Ny, 1))
    g.interior_N(m.u)[...] = ones * np.cos(np.pi / 2 * zc / L)[None, None, :]
    assert np.abs(m._barotropic_mode(m.u) - 2 * L / np.pi).max() < 1e-3
    g.interior_N(m.v)[...] = np.sin(xc[:, None, None] * yf[None, :, None]) * np.cos(np.pi / 2 * zc / L)[None, None, :]
    assert np.abs(m._barotropic_mode(m.v) - np.sin(xc[:, None] * yf[None, :]) * 2 * L / np.pi).max() < 1e-3
    g.interior_N(m.u)[...] = 0.0
    assert np.all(m._barotropic_mode(m.u) == 0.0)
    g.interior_N(m.u)[...] = 1.0
    np.testing.assert_allclose(m._barotropic_mode(m.u), L, rtol=1e-14)
    g.interior_N(m.u)[...] = np.sin(xf)[:, None, None] * np.ones((1, g.Ny, g.Nz))
    np.testing.assert_allclose(m._barotropic_mode(m.u), np.sin(xf)[:, None] * L * np.ones((1, g.Ny)), rtol=1e-13, atol=1e-14)
    g.interior_N(m.v)[...] = np.sin(xc)[:, None, None] * zc[None, None, :] * np.cos(yf)[None, :, None]
    np.testing.assert_allclose(m._barotropic_mode(m.v), -np.sin(xc)[:, None] * L ** 2 / 2 * np.cos(yf)[None, :], rtol=1e-12, atol=1e-13)


def test_barotropic_correction():
    """:113-146: u = z + Lz/2 + sin x with the barotropic transport set to cos(x) Lz is corrected to z + Lz/2 + cos x (1e-14)"""
    g, m, L, xc, xf, yc, yf, zc = _barotropic_kernels_model()
    zz = (zc + L / 2)[None, None, :]
    g.interior_N(m.u)[...] = zz + np.sin(xf)[:, None, None] + 0 * yc[None, :, None]
    m.U[...] = np.cos(xf)[:, None] * L * np.ones((1, g.Ny))
    g.interior_N(m.v)[...] = zz * np.sin(yf)[None, :, None] + np.sin(xc)[:, None, None]
    m.V[...] = (np.cos(xc) + xc)[:, None] * L * np.ones((1, g.Ny))
    m._barotropic_corrector()
    assert np.abs(g.interior_N(m.u) - (zz + np.cos(xf)[:, None, None] + 0 * yc[None, :, None])).max() < 1e-14
    assert np.abs(g.interior_N(m.v) - (zz * np.sin(yf)[None, :, None] + (np.cos(xc) + xc)[:, None, None])).max() < 1e-13


# ---- BASELINE.json configs[4]: VectorInvariant momentum + WENO tracer advection + split-explicit free surface ----------------------
def test_config5_combination_budgets():
    """The combination BASELINE.json names (VectorInvariant() momentum, tracer_advection = WENO(), SplitExplicitFreeSurface, T / S
    with linear SeawaterBuoyancy, FPlane, ScalarDiffusivity) steps on the oracle; mean(η) is conserved by the substepping
    (test_split_explicit_free_surface_solver.jl: mean(η) conserved to 10 eps), Σ Δz u equals the barotropic transport after the
    corrector, w[k = 1] = 0, and the tracer totals change only by the advective flux through the moving surface of the static
    grid, Σ Az w[Nz+1] c_top dt (a linear free surface on z-coordinates does not conserve tracers exactly)."""
    g = _grid(N=(20, 14, 8), stretched=False)
    rng = np.random.default_rng(17)
    shp = (g.Nx, g.Ny, g.Nz)
    m = Hy.HydrostaticFreeSurfaceModel(g, tracers=("T", "S"), momentum_advection="VectorInvariant", tracer_advection="WENO5",
                                       coriolis_f=1e-4, closure=(1e-2, 1e-3), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4),
                                       split_explicit_substeps=20)
    assert m.tracer_scheme == O.ADV_WENO5 and m.vector_invariant
    m.set(u=1e-2 * rng.uniform(-1, 1, shp), v=1e-2 * rng.uniform(-1, 1, shp), eta=1e-3 * rng.uniform(-1, 1, shp[:2]),
          T=20 + 1e-2 * rng.uniform(-1, 1, shp), S=35 + 1e-2 * rng.uniform(-1, 1, shp))
    ii, jj = slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny)
    eta0 = m.eta[ii, jj].mean()
    T0, S0 = g.interior(m.tracers[0]).sum(), g.interior(m.tracers[1]).sum()
    dt = 1.5 * g.dx / np.sqrt(Hy.g_Earth * g.Lz)
    for _ in range(5):
        m.time_step(dt)
        assert abs(m.eta[ii, jj].mean() - eta0) < 10 * np.finfo(float).eps * 1e-3 * 20
        assert np.abs(m._barotropic_mode(m.u) - m.U).max() <= 16 * np.finfo(float).eps * np.abs(m.U).max()
        assert np.abs(m.w[ii, jj, g.Hz]).max() == 0
    assert np.isfinite(m.eta).all() and all(np.isfinite(f).all() for f in m.fields)
    wtop = np.abs(m.w[ii, jj, g.Hz + g.Nz]).max()
    for c, c0 in zip(m.tracers, (T0, S0)):
        drift = abs(g.interior(c).sum() - c0)
        assert drift <= 5 * dt * wtop / g.dz * np.abs(g.interior(c)).max() * g.Nx * g.Ny * 2   # surface flux bound
        assert drift < 1e-7 * abs(c0)
    assert np.abs(m.U).max() > 0 and np.abs(g.interior(m.w)).max() > 0


def test_split_explicit_ab3_scheme_wave():
    """timestepper = AdamsBashforth3Scheme() (split_explicit_timesteppers.jl:19-159; the reference has no test of it): the same long surface
    wave as the ForwardBackward test stays stable at a baroclinic gravity-wave CFL of 0.8, follows cos(ω t), conserves mean(η), and with
    coefficients (α, θ, β) = (1, 0, 0), (δ, μ, γ, ϵ) = (1, 0, 0, 0) the scheme reduces to the ForwardBackward one bit for bit."""
    Nx, H, L = 32, 20.0, 4.0e3
    a, k = 1e-4, 2 * np.pi / L

    def run(ts, coeffs=None, nsteps=80):
        g = O.Grid((Nx, 4, 4), x=(0, L), y=(0, 500.0), z=(-H, 0.0), topology="PPB", halo=(3, 3, 3))
        m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="Centered2", split_explicit_substeps=30, split_explicit_timestepper=ts)
        if coeffs is not None:
            m.ab3 = coeffs
        x = (np.arange(Nx) + 0.5) * g.dx
        m.set(eta=a * np.cos(k * x)[:, None] * np.ones((1, 4)))
        kd = 2 / g.dx * np.sin(k * g.dx / 2)
        omega = np.sqrt(Hy.g_Earth * H) * kd
        dt = 2 * np.pi / omega / 40
        amp = []
        for _ in range(nsteps):
            m.time_step(dt)
            e = m.eta[g.Hx:g.Hx + Nx, g.Hy:g.Hy + 4]
            amp.append(2 * np.mean(e[:, 0] * np.cos(k * x)) / a)
            assert abs(e.sum()) < 1e-16 * Nx * 4
        return m, np.array(amp), omega * dt * np.arange(1, nsteps + 1)

    m, amp, phase = run("AdamsBashforth3")
    assert m.ab3 is not None and abs(m.ab3["alpha"] + m.ab3["theta"] + m.ab3["beta"] - 1) < 1e-15
    assert abs(m.ab3["delta"] + m.ab3["mu"] + m.ab3["gamma"] + m.ab3["epsilon"] - 1) < 1e-15       # both extrapolations are consistent
    assert np.abs(amp - np.cos(phase)).max() < 0.12 and 0.9 < amp[79] < 1.01
    fb, amp_fb, _ = run("ForwardBackward", nsteps=10)
    deg, amp_deg, _ = run("AdamsBashforth3", coeffs=dict(alpha=1.0, theta=0.0, beta=0.0, delta=1.0, mu=0.0, gamma=0.0, epsilon=0.0), nsteps=10)
    np.testing.assert_array_equal(deg.eta, fb.eta)
    np.testing.assert_array_equal(deg.U, fb.U)


# ---- ImplicitFreeSurface, FFT solver (test_implicit_free_surface_solver.jl:20-97 re-expressed) ---------------------------------------------
def test_implicit_free_surface_solver_satisfies_its_equation():
    """run_implicit_free_surface_solver_tests: u = v = η = 0 except one surface cell carrying a transport of 1e5 m³/s; after
    step_free_surface!(Δt = 900) the linear operator of the implicit η equation applied to the solution equals the right-hand side
    (the reference: max |L η - rhs| < 1e-9 and std < 1e-9 in its own scaling); here in the FFT solver's scaled form
    (∇² - 1 / (g Lz Δt²)) η = (∇ʰ·Q★ - ηⁿ / Δt) / (g Lz Δt)."""
    Nx, Ny, Nz, H = 16, 12, 5, 50.0
    g = O.Grid((Nx, Ny, Nz), x=(0, 2e3), y=(0, 1.5e3), z=(-H, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="Centered2", implicit_free_surface=True)
    u = np.zeros((Nx, Ny, Nz))
    u[Nx // 2, Ny // 2, Nz - 1] = 1e5 / (g.dy * g.dz)
    m.set(u=u)
    ustar = g.interior_N(m.u).copy()
    dt = 900.0
    m._implicit_step(dt)
    e = m.eta
    ii, jj = slice(3, 3 + Nx), slice(3, 3 + Ny)
    lap = (e[4:4 + Nx, jj] - 2 * e[ii, jj] + e[2:2 + Nx, jj]) / g.dx ** 2 + (e[ii, 4:4 + Ny] - 2 * e[ii, jj] + e[ii, 2:2 + Ny]) / g.dy ** 2
    lhs = lap - e[ii, jj] / (Hy.g_Earth * H * dt ** 2)
    Qu = (g.dy * g.dz) * ustar.sum(2)
    rhs = (np.roll(Qu, -1, 0) - Qu) / (Hy.g_Earth * H * dt * g.dx * g.dy)
    assert np.abs(rhs).max() > 1e-6
    assert np.abs(lhs - rhs).max() < 1e-9 * np.abs(rhs).max() and (lhs - rhs).std() < 1e-9 * np.abs(rhs).max()
    # the pressure correction made the barotropic flow consistent with the new surface: Δη / Δt = -∇ʰ·Q / Az
    Qn = (g.dy * g.dz) * g.interior_N(m.u).sum(2)
    Vn = (g.dx * g.dz) * g.interior_N(m.v).sum(2)
    div = ((np.roll(Qn, -1, 0) - Qn) + (np.roll(Vn, -1, 1) - Vn)) / (g.dx * g.dy)
    assert np.abs(e[ii, jj] / dt + div).max() < 1e-9 * np.abs(div).max()   # (a small difference of the much larger predictor divergence)


def test_implicit_free_surface_model_is_stable_at_large_gravity_wave_cfl():
    """The point of the implicit free surface: the whole model steps stably at a gravity-wave CFL of 20, conserves mean(η), and damps
    rather than amplifies the surface wave (backward Euler in η)."""
    Nx, H, L = 32, 20.0, 4.0e3
    g = O.Grid((Nx, 4, 4), x=(0, L), y=(0, 500.0), z=(-H, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, momentum_advection="Centered2", implicit_free_surface=True)
    a, k = 1e-3, 2 * np.pi / L
    x = (np.arange(Nx) + 0.5) * g.dx
    m.set(eta=a * np.cos(k * x)[:, None] * np.ones((1, 4)))
    dt = 20 * g.dx / np.sqrt(Hy.g_Earth * H)
    amp = []
    for _ in range(20):
        m.time_step(dt)
        e = m.eta[g.Hx:g.Hx + Nx, g.Hy:g.Hy + 4]
        amp.append(np.abs(e).max() / a)
        assert abs(e.sum()) < 1e-14 * Nx * 4 * a
    assert max(amp) <= 1.0 + 1e-12 and np.isfinite(m.u).all()


# ---- SplitRungeKutta3 (split_hydrostatic_runge_kutta_3.jl; the reference ships no test of it and calls it experimental) -----------------------
def _rk3_case(ts, n, T=160.0, N=(16, 12, 6)):
    from helpers import stretched_faces
    g = O.Grid(N, x=(0, 4.0e3), y=(0, 3.0e3), z=stretched_faces(N[2], 40.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, tracers=("T", "S"), momentum_advection="VectorInvariant", tracer_advection="WENO5", coriolis_f=1e-4,
                                       closure=(1e-2, 2e-3), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4), split_explicit_substeps=12,
                                       timestepper=ts)
    x, y = (np.arange(N[0]) + 0.5) / N[0], (np.arange(N[1]) + 0.5) / N[1]
    one = np.ones(N)
    m.set(u=0.05 * np.sin(2 * np.pi * x)[:, None, None] * np.cos(2 * np.pi * y)[None, :, None] * one,
          v=-0.05 * np.cos(2 * np.pi * x)[:, None, None] * np.sin(2 * np.pi * y)[None, :, None] * one,
          eta=0.02 * np.cos(2 * np.pi * x)[:, None] * np.ones(N[:2]), T=20 + 0.1 * np.sin(2 * np.pi * x)[:, None, None] * one, S=35.0 * one)
    for _ in range(n):
        m.time_step(T / n)
    return g.interior_N(m.u).copy(), m.eta[3:-3, 3:-3].copy(), m


def test_split_runge_kutta_3_is_consistent_with_qab2_and_converges():
    """Both time steppers integrate the same equations with the same split-explicit free surface: at the same Δt their solutions agree
    far better than either agrees with a finer step, and halving Δt halves the distance to a 4x finer run (the barotropic averaging over
    a Δt-wide window makes both first order in Δt for the fast surface wave).  Volume is conserved and the barotropic corrector leaves
    Σ Δz u = U at the end of every stage."""
    u_ref, e_ref, _ = _rk3_case("SplitRungeKutta3", 64)
    u8, e8, _ = _rk3_case("SplitRungeKutta3", 8)
    u16, e16, m = _rk3_case("SplitRungeKutta3", 16)
    uq, eq, _ = _rk3_case("QuasiAdamsBashforth2", 16)
    su, se = np.abs(u_ref).max(), np.abs(e_ref).max()
    err8, err16 = np.abs(u8 - u_ref).max() / su, np.abs(u16 - u_ref).max() / su
    assert err16 < 0.01 and 1.6 < err8 / err16 < 2.6, (err8, err16)
    assert np.abs(e16 - e_ref).max() / se < 0.12
    assert np.abs(u16 - uq).max() / su < 0.1 * err16 and np.abs(e16 - eq).max() / se < 0.01   # same Δt: the two schemes nearly coincide
    g = m.grid
    assert abs(m.eta[3:-3, 3:-3].mean() - 0.0) < 1e-15 + 1e-13 * se                          # mean(η) of the cosine initial condition is 0
    assert np.abs(m._barotropic_mode(m.u) - m.U).max() <= 1e-12 * np.abs(m.U).max()
    assert m.iteration == 16 and abs(m.time - 160.0) < 1e-9
