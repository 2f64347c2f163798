"""HIP side of the first HydrostaticFreeSurfaceModel slice against the CPU oracle (oracle/hydrostatic.py),
through the C ABI: ocn_compute_w_from_continuity, ocn_add_barotropic_pressure_gradient, ocn_add_momentum_terms,
ocn_explicit_free_surface_ab2_step and the model built from them.  Strict math: bit for bit."""
import numpy as np
import pytest

from helpers import from_dev, make_pair, stretched_faces, to_dev

pytestmark = pytest.mark.gpu


def _pair(O, ocn, size, stretched):
    z = stretched_faces(size[2], 40.0) if stretched else (-40.0, 0.0)
    return make_pair(O, ocn, size, "PPB", x=(0, 2.0e3), y=(0, 1.5e3), z=z)


@pytest.mark.parametrize("stretched", [False, True])
def test_w_from_continuity_bitwise(oracle, ocn, stretched):
    from oracle import hydrostatic as Hy
    O = oracle
    og, pg = _pair(O, ocn, (16, 12, 7), stretched)
    rng = np.random.default_rng(11)
    u, v, w = og.zeros(1), og.zeros(2), og.zeros(4)
    u[...] = rng.uniform(-1, 1, u.shape)
    v[...] = rng.uniform(-1, 1, v.shape)
    Hy.compute_w_from_continuity(og, u, v, w)
    du, dv, dw = to_dev(ocn, pg, 1, u), to_dev(ocn, pg, 2, v), ocn.Field(4, pg)
    ocn._lib.call("ocn_compute_w_from_continuity", pg.cref, du.ptr, dv.ptr, dw.ptr, 0)
    ocn.sync_device()
    got = from_dev(dw)
    # every column with east / north neighbours in the parent
    np.testing.assert_array_equal(got[:-1, :-1, og.Hz:og.Hz + og.Nz + 1], w[:-1, :-1, og.Hz:og.Hz + og.Nz + 1])


@pytest.mark.parametrize("advection", ["Centered2", "WENO5", "VectorInvariant", "VectorInvariant-unfused"])
@pytest.mark.parametrize("physics", [False, True])
def test_hydrostatic_model_steps_match_oracle(oracle, ocn, advection, physics):
    """3 QAB2 steps (the first one Euler) of the explicit-free-surface model: u, v, w, η and the tracers equal the oracle's bit
    for bit in strict math."""
    from oracle import hydrostatic as Hy
    O = oracle
    size = (16, 12, 7)
    og, pg = _pair(O, ocn, size, stretched=True)
    rng = np.random.default_rng(12)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]))
    kw_o, kw_p, tracers = {}, {}, ()
    if physics:
        tracers = ("T", "S")
        init["T"] = 20 + 1e-2 * rng.uniform(-1, 1, size)
        init["S"] = 35 + 1e-2 * rng.uniform(-1, 1, size)
        kw_o = dict(coriolis_f=1e-4, closure=(1e-2, 2e-3), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4),
                    boundary_conditions={"u": {"top": O.FluxBoundaryCondition(-1e-4)}, "T": {"top": O.FluxBoundaryCondition(5e-5)}})
        kw_p = dict(coriolis=ocn.FPlane(f=1e-4), closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                    boundary_conditions={"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1e-4)),
                                         "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5))})
    fused = None
    if advection == "VectorInvariant-unfused":
        advection, fused = "VectorInvariant", False
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=tracers, momentum_advection=advection, **kw_o)
    om.set(**init)
    ocn.set_math_mode(ocn.MATH_STRICT)
    scheme = {"Centered2": ocn.Centered, "WENO5": ocn.WENO, "VectorInvariant": ocn.VectorInvariant}[advection]()
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=scheme, tracers=tracers, free_surface=ocn.ExplicitFreeSurface(), fused=fused, **kw_p)
    assert pm.fused == (advection == "VectorInvariant" and fused is None)
    pm.set(**init)
    for dt in (2.0, 2.0, 2.0):
        om.time_step(dt)
        pm.time_step(dt)
    ocn.sync_device()
    for name, a, d in zip(("u", "v", "w"), (om.u, om.v, om.w), pm.velocities):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a), err_msg=name)
    eta = pm.eta_interior().cpu().numpy().T
    np.testing.assert_array_equal(eta, om.eta[og.Hx:og.Hx + og.Nx, og.Hy:og.Hy + og.Ny])
    for a, d in zip(om.tracers, pm.tracers):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a))
    assert np.abs(eta).max() > 0 and np.isfinite(eta).all()


@pytest.mark.parametrize("fused", [True, False])
def test_split_explicit_free_surface_model_steps_match_oracle(oracle, ocn, fused):
    """SplitExplicitFreeSurface(substeps = 12), ForwardBackwardScheme: slow forcing, substepping with the averaging weights, barotropic
    corrector -- 3 QAB2 steps of the default configuration (VectorInvariant momentum, Centered tracers) with Coriolis,
    diffusivity, buoyancy and flux conditions equal the oracle's bit for bit, barotropic transports included."""
    from oracle import hydrostatic as Hy
    O = oracle
    size = (16, 12, 7)
    og, pg = _pair(O, ocn, size, stretched=True)
    rng = np.random.default_rng(21)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]),
                T=20 + 1e-2 * rng.uniform(-1, 1, size), S=35 + 1e-2 * rng.uniform(-1, 1, size))
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=("T", "S"), momentum_advection="VectorInvariant", coriolis_f=1e-4, closure=(1e-2, 2e-3),
                                        buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4), split_explicit_substeps=12,
                                        boundary_conditions={"u": {"top": O.FluxBoundaryCondition(-1e-4)}, "T": {"top": O.FluxBoundaryCondition(5e-5)}})
    om.set(**init)
    ocn.set_math_mode(ocn.MATH_STRICT)
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=ocn.VectorInvariant(), tracers=("T", "S"),
                                         free_surface=ocn.SplitExplicitFreeSurface(substeps=12), coriolis=ocn.FPlane(f=1e-4),
                                         closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                         buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                         boundary_conditions={"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1e-4)),
                                                              "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5))},
                                         fused=fused)
    pm.set(**init)
    for dt in (20.0, 20.0, 20.0):
        om.time_step(dt)
        pm.time_step(dt)
    ocn.sync_device()
    for name, a, d in zip(("u", "v", "w"), (om.u, om.v, om.w), pm.velocities):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a), err_msg=name)
    ii, jj = slice(og.Hy, og.Hy + og.Ny), slice(og.Hx, og.Hx + og.Nx)
    np.testing.assert_array_equal(pm.eta[ii, jj].cpu().numpy().T, om.eta[og.Hx:og.Hx + og.Nx, og.Hy:og.Hy + og.Ny])
    np.testing.assert_array_equal(pm.U[ii, jj].cpu().numpy().T, om.U)
    np.testing.assert_array_equal(pm.V[ii, jj].cpu().numpy().T, om.V)
    for a, d in zip(om.tracers, pm.tracers):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a))
    assert np.abs(om.U).max() > 0


# ---- BASELINE.json configs[4]: VectorInvariant momentum + WENO tracer advection + SplitExplicitFreeSurface -----------------------------
def _config5_pair(oracle, ocn, size, substeps, stretched, seed=31, fused=None, timestepper="QuasiAdamsBashforth2", beta=None):
    from oracle import hydrostatic as Hy
    O = oracle
    og, pg = _pair(O, ocn, size, stretched=stretched)
    rng = np.random.default_rng(seed)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]),
                T=20 + 1e-2 * rng.uniform(-1, 1, size), S=35 + 1e-2 * rng.uniform(-1, 1, size))
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=("T", "S"), momentum_advection="VectorInvariant", tracer_advection="WENO5",
                                        coriolis_f=1e-4, coriolis_beta=beta, closure=(1e-2, 2e-3), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4),
                                        split_explicit_substeps=substeps, timestepper=timestepper,
                                        boundary_conditions={"u": {"top": O.FluxBoundaryCondition(-1e-4)}, "T": {"top": O.FluxBoundaryCondition(5e-5)}})
    om.set(**init)
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                         free_surface=ocn.SplitExplicitFreeSurface(substeps=substeps),
                                         coriolis=ocn.FPlane(f=1e-4) if beta is None else ocn.BetaPlane(f0=1e-4, beta=beta),
                                         closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                         buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                         boundary_conditions={"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1e-4)),
                                                              "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5))},
                                         fused=fused, timestepper=timestepper)
    pm.set(**init)
    return og, om, pm


def _compare_hydrostatic(og, om, pm, tol):
    """tol = 0: bit for bit; otherwise max-abs difference relative to the field's max-abs."""
    def cmp(name, got, want):
        if tol == 0:
            np.testing.assert_array_equal(got, want, err_msg=name)
        else:
            scale = max(np.abs(want).max(), 1e-300)
            assert np.abs(got - want).max() <= tol * scale, (name, np.abs(got - want).max() / scale)
    for name, a, d in zip(("u", "v", "w"), (om.u, om.v, om.w), pm.velocities):
        cmp(name, og.interior(from_dev(d)), og.interior(a))
    ii, jj = slice(og.Hy, og.Hy + og.Ny), slice(og.Hx, og.Hx + og.Nx)
    cmp("eta", pm.eta[ii, jj].cpu().numpy().T, om.eta[og.Hx:og.Hx + og.Nx, og.Hy:og.Hy + og.Ny])
    cmp("U", pm.U[ii, jj].cpu().numpy().T, om.U)
    cmp("V", pm.V[ii, jj].cpu().numpy().T, om.V)
    for n, (a, d) in enumerate(zip(om.tracers, pm.tracers)):
        cmp(f"tracer{n}", og.interior(from_dev(d)), og.interior(a))


@pytest.mark.parametrize("size,substeps,stretched,dt", [((16, 12, 7), 12, True, 20.0), ((130, 70, 5), 30, False, 4.0), ((67, 9, 12), 8, True, 3.0)])
@pytest.mark.parametrize("math,fused", [("strict", True), ("strict", False), ("fast", True)])
def test_config5_combination_matches_oracle(oracle, ocn, size, substeps, stretched, dt, math, fused):
    """BASELINE.json configs[4] as written: VectorInvariant() momentum, tracer_advection = WENO(), SplitExplicitFreeSurface(substeps),
    T / S + linear SeawaterBuoyancy + FPlane + ScalarDiffusivity + flux conditions; 3 QAB2 steps (the first Euler).
    (130, 70, 5): more than one 64-wide block in x and Ny not a multiple of the block height, so the periodic wrap across blocks
    of the barotropic kernels is exercised.  fused = True: the re-cut launch sequence (one momentum pass with the barotropic sums, one
    launch per tracer, temporally blocked substeps, corrector + w in one pass); False: the reference's launch sequence.  dt keeps the barotropic gravity-wave CFL sqrt(g H) Δτ / Δx below 0.7 on each grid.
    Strict math: bit for bit; fast math: 1e-10 of each field's max."""
    og, om, pm = None, None, None
    ocn.set_math_mode(ocn.MATH_STRICT if math == "strict" else ocn.MATH_FAST)
    try:
        og, om, pm = _config5_pair(oracle, ocn, size, substeps, stretched, fused=fused)
        assert pm.fused == fused
        for _ in range(3):
            om.time_step(dt)
            pm.time_step(dt)
        ocn.sync_device()
        _compare_hydrostatic(og, om, pm, 0 if math == "strict" else 1e-10)
        assert np.abs(om.U).max() > 0 and np.abs(og.interior(om.w)).max() > 0
        assert np.abs(og.interior(om.u)).max() < 1.0 and np.abs(om.eta).max() < 1.0      # a stable run, not a common blow-up
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)


@pytest.mark.parametrize("fused", [True, False])
def test_config5_combination_on_a_beta_plane_matches_oracle(oracle, ocn, fused):
    """The same combination with coriolis = BetaPlane(f₀, β) (beta_plane.jl:43-57; the hydrostatic model's usual Coriolis term): 3 QAB2
    steps bit for bit against the oracle, fused and reference launch sequences."""
    ocn.set_math_mode(ocn.MATH_STRICT)
    og, om, pm = _config5_pair(oracle, ocn, (16, 12, 7), 12, True, fused=fused, beta=2e-8)
    assert pm.fused == fused
    for _ in range(3):
        om.time_step(20.0)
        pm.time_step(20.0)
    ocn.sync_device()
    _compare_hydrostatic(og, om, pm, 0)
    og2, om2, _ = _config5_pair(oracle, ocn, (16, 12, 7), 12, True, fused=fused)
    for _ in range(3):
        om2.time_step(20.0)
    assert np.abs(og.interior(om.u) - og2.interior(om2.u)).max() > 0  # β matters


@pytest.mark.parametrize("size,substeps,stretched,dt", [((16, 12, 7), 12, True, 20.0), ((67, 9, 12), 8, True, 3.0)])
@pytest.mark.parametrize("math", ["strict", "fast"])
def test_split_runge_kutta_3_matches_oracle(oracle, ocn, size, substeps, stretched, dt, math):
    """HydrostaticFreeSurfaceModel(timestepper = :SplitRungeKutta3) with the split-explicit free surface
    (split_hydrostatic_runge_kutta_3.jl:76-133, hydrostatic_free_surface_rk3_step.jl:7-60, compute_slow_tendencies.jl:85-108,
    initialize_split_explicit_substepping.jl:44-63): three stages per step, each with the complete barotropic substepping, the
    stage-2 average and the stage-3 restart from step n.  The reference holds no test of this (experimental) time stepper, so the
    oracle is its only pin: 2 steps, bit for bit in strict math, 1e-10 in fast math."""
    ocn.set_math_mode(ocn.MATH_STRICT if math == "strict" else ocn.MATH_FAST)
    try:
        og, om, pm = _config5_pair(oracle, ocn, size, substeps, stretched, timestepper="SplitRungeKutta3")
        assert pm.split_rk3 and not pm.fused
        for _ in range(2):
            om.time_step(dt)
            pm.time_step(dt)
        ocn.sync_device()
        _compare_hydrostatic(og, om, pm, 0 if math == "strict" else 1e-10)
        assert pm.clock.iteration == 2 and abs(pm.clock.time - 2 * dt) < 1e-12
        assert np.abs(og.interior(om.u)).max() < 1.0 and np.abs(om.eta).max() < 1.0
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)


def test_config5_full_size_properties(ocn):
    """configs[4] at its full size, 1024 x 1024 x 128 (the oracle is far too slow there): size-independent properties.
      * mean(η) is conserved by the split-explicit substepping (test_split_explicit_free_surface_solver.jl: the forced wave conserves
        mean(η) to 10 eps): the η update is a flux divergence on a periodic plane;
      * w = 0 at the bottom face and the flow is discretely nondivergent after compute_w_from_continuity!;
      * the barotropic corrector leaves Σ Δz u == U (barotropic_split_explicit_corrector.jl:44-71);
      * tracer budgets: Σ T, Σ S change only by the advective flux through the moving surface of the static grid
        (Σ Az w[Nz+1] c_top dt: a linear free surface on z-coordinates does not conserve tracers exactly);
      * everything finite."""
    import torch
    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        Nx, Nz, H, L = 1024, 128, 1000.0, 1.0e6
        g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, L), y=(0, L), z=(-H, 0.0), topology=("Periodic", "Periodic", "Bounded"),
                                halo=(3, 3, 3))
        m = ocn.HydrostaticFreeSurfaceModel(g, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                            free_surface=ocn.SplitExplicitFreeSurface(substeps=30), coriolis=ocn.FPlane(f=1e-4),
                                            closure=ocn.ScalarDiffusivity(ν=1e-2, κ=1e-3),
                                            buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)))
        gen = torch.Generator(device="cuda")
        gen.manual_seed(5)
        for f in (m.u, m.v):
            iv = f.interior_view()
            iv.copy_(1e-1 * (2 * torch.rand(iv.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
        zc = torch.linspace(-H + H / (2 * Nz), -H / (2 * Nz), Nz, device="cuda", dtype=torch.float64)
        T = m.field("T").interior_view()
        T.copy_((20 + 0.01 * zc)[:, None, None] + 1e-3 * torch.rand(T.shape, generator=gen, device="cuda", dtype=torch.float64))
        m.field("S").interior_view().fill_(35.0)
        eta0 = 1e-2 * (2 * torch.rand((Nx, Nx), generator=gen, device="cuda", dtype=torch.float64) - 1)
        m.eta_interior().copy_(eta0)
        m.update_state(compute_tendencies=False)
        mean_eta0 = float(m.eta_interior().mean())
        T0, S0 = float(m.field("T").interior_view().sum()), float(m.field("S").interior_view().sum())
        dt = 2.0 * g.dx / np.sqrt(9.80665 * H)
        for _ in range(3):
            m.time_step(dt)
        ocn.sync_device()
        eta = m.eta_interior()
        assert bool(torch.isfinite(eta).all()) and all(bool(torch.isfinite(f.interior_view()).all()) for f in m.velocities)
        assert float(eta.abs().max()) > 0
        assert abs(float(eta.mean()) - mean_eta0) <= 10 * np.finfo(float).eps * max(1.0, float(eta0.abs().max())) * 3 * 30
        Hx, Hy, Hz = g.Hx, g.Hy, g.Hz
        w = m.w.data
        assert float(w[Hz, Hy:Hy + Nx, Hx:Hx + Nx].abs().max()) == 0.0                   # w[i, j, 1] = 0
        u, v = m.u.data, m.v.data
        k = slice(Hz, Hz + Nz)
        div = ((u[k, Hy:Hy + Nx, Hx + 1:Hx + Nx + 1] - u[k, Hy:Hy + Nx, Hx:Hx + Nx]) / g.dx
               + (v[k, Hy + 1:Hy + Nx + 1, Hx:Hx + Nx] - v[k, Hy:Hy + Nx, Hx:Hx + Nx]) / g.dy
               + (w[Hz + 1:Hz + Nz + 1, Hy:Hy + Nx, Hx:Hx + Nx] - w[k, Hy:Hy + Nx, Hx:Hx + Nx]) / g.dz)
        umax = float(u.abs().max())
        assert float(div.abs().max()) < 1e-12 * umax / g.dz * Nz
        del div
        # barotropic consistency: Σ Δz u == U after the corrector (to round-off of a 128-term sum)
        Usum = (u[k, Hy:Hy + Nx, Hx:Hx + Nx] * g.dz).sum(0)
        Uint = m.U[Hy:Hy + Nx, Hx:Hx + Nx]
        assert float((Usum - Uint).abs().max()) < 1e-11 * max(float(Uint.abs().max()), umax * H)
        # tracer budgets (no flux conditions): the totals change only by the flux through the top face
        wtop = float(w[Hz + Nz, Hy:Hy + Nx, Hx:Hx + Nx].abs().max())
        for name, c0 in (("T", T0), ("S", S0)):
            c = m.field(name).interior_view()
            drift = abs(float(c.sum()) - c0)
            assert drift <= 3 * dt * wtop / g.dz * float(c.abs().max()) * Nx * Nx * 2 + 1e-12 * abs(c0), (name, drift)
            assert drift < 1e-6 * abs(c0)
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)


def test_split_explicit_ab3_scheme_matches_oracle(oracle, ocn):
    """SplitExplicitFreeSurface(substeps = 12, timestepper = AdamsBashforth3Scheme()): the whole model (config-5 combination), 3 steps, bit
    for bit against the oracle in strict math -- the AB3 extrapolations U★, η★ and the history updates of
    split_explicit_timesteppers.jl:128-159."""
    from oracle import hydrostatic as Hy
    O = oracle
    size = (20, 12, 7)
    og, pg = _pair(O, ocn, size, stretched=True)
    rng = np.random.default_rng(41)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]),
                T=20 + 1e-2 * rng.uniform(-1, 1, size), S=35 + 1e-2 * rng.uniform(-1, 1, size))
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=("T", "S"), momentum_advection="VectorInvariant", tracer_advection="WENO5", coriolis_f=1e-4,
                                        closure=(1e-2, 2e-3), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4), split_explicit_substeps=12,
                                        split_explicit_timestepper="AdamsBashforth3")
    om.set(**init)
    ocn.set_math_mode(ocn.MATH_STRICT)
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                         free_surface=ocn.SplitExplicitFreeSurface(substeps=12, timestepper=ocn.AdamsBashforth3Scheme()),
                                         coriolis=ocn.FPlane(f=1e-4), closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                         buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)))
    pm.set(**init)
    for _ in range(3):
        om.time_step(15.0)
        pm.time_step(15.0)
    ocn.sync_device()
    _compare_hydrostatic(og, om, pm, 0)
    assert np.abs(om.U).max() > 0 and np.abs(om.eta).max() < 1.0


@pytest.mark.parametrize("advection", ["VectorInvariant", "WENO5"])
def test_implicit_free_surface_model_matches_oracle(oracle, ocn, advection):
    """ImplicitFreeSurface() with the FFT solver -- the reference's DEFAULT free surface on this grid (hydrostatic_free_surface_model.jl:
    51-52; HydrostaticFreeSurfaceModel(grid) with no free_surface argument selects it here too): 3 QAB2 steps at a gravity-wave CFL of 5
    with all physics against the oracle.  Everything but η's FFT solve is the same arithmetic (strict math): η to 1e-12 of its
    maximum (rocFFT vs pocketfft round-off), the fields it feeds back into to 1e-11."""
    from oracle import hydrostatic as Hy
    O = oracle
    size = (32, 16, 7)
    og, pg = _pair(O, ocn, size, stretched=True)
    rng = np.random.default_rng(51)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]),
                T=20 + 1e-2 * rng.uniform(-1, 1, size), S=35 + 1e-2 * rng.uniform(-1, 1, size))
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=("T", "S"), momentum_advection=advection, coriolis_f=1e-4, closure=(1e-2, 2e-3),
                                        buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4), implicit_free_surface=True)
    om.set(**init)
    ocn.set_math_mode(ocn.MATH_STRICT)
    scheme = ocn.VectorInvariant() if advection == "VectorInvariant" else ocn.WENO()
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=scheme, tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                         closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                         buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)))
    assert isinstance(pm.free_surface, ocn.ImplicitFreeSurface) and not pm.fused
    pm.set(**init)
    dt = 5 * og.dx / np.sqrt(9.80665 * 40.0)
    for _ in range(3):
        om.time_step(dt)
        pm.time_step(dt)
    ocn.sync_device()
    ii, jj = slice(og.Hy, og.Hy + og.Ny), slice(og.Hx, og.Hx + og.Nx)
    eo = om.eta[og.Hx:og.Hx + og.Nx, og.Hy:og.Hy + og.Ny]
    assert np.abs(pm.eta[ii, jj].cpu().numpy().T - eo).max() <= 1e-12 * np.abs(eo).max() and np.abs(eo).max() > 1e-4
    for name, a, d in zip(("u", "v", "w"), (om.u, om.v, om.w), pm.velocities):
        ref = og.interior(a)
        assert np.abs(og.interior(from_dev(d)) - ref).max() <= 1e-11 * np.abs(ref).max(), name
    for a, d in zip(om.tracers, pm.tracers):
        assert np.abs(og.interior(from_dev(d)) - og.interior(a)).max() <= 1e-13 * np.abs(og.interior(a)).max()
