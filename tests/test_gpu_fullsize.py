"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle is too slow there):
incompressibility after set!/time steps, momentum conservation of the flux-form advection + projection, linearity of the
Poisson solve, periodic-halo identities, strict == fast to tolerance, fused == unfused bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
P, B = "Periodic", "Bounded"


def _model(ocn, size, topo=(P, P, P), z=(0, 2 * np.pi), seed=1234):
    g = ocn.RectilinearGrid(ocn.GPU(), size=size, x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=topo, halo=(3, 3, 3))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    for f in m.velocities:
        iv = f.interior_view()
        iv.copy_(torch.rand(iv.shape, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1)
    ocn.set(m)
    return g, m


def _div_max(ocn, g, m):
    d = torch.zeros((g.Nz, g.Ny, g.Nx), dtype=torch.float64, device="cuda")
    ocn._lib.call("ocn_divergence", g.cref, m.u.ptr, m.v.ptr, m.w.ptr, d.data_ptr(), 0)
    return float(d.abs().max())


def _interior_N(f, g):
    return f.data[g.Hz:g.Hz + g.Nz, g.Hy:g.Hy + g.Ny, g.Hx:g.Hx + g.Nx]


@pytest.mark.parametrize("n", [256, 512])
def test_full_size_periodic_box(ocn, n):
    """configs[1] / configs[2] of BASELINE.json: n^3 triply periodic, WENO5, RK3."""
    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        g, m = _model(ocn, (n, n, n))
        assert m.fuse_stage_boundaries and m.correct_on_load and m.pressure_solver.info()["fused_z"]
        umax = float(torch.stack([f.interior_view().abs().max() for f in m.velocities]).max())
        assert _div_max(ocn, g, m) < 1e-10 * umax / g.dx          # set! projected the random field
        mom0 = [float(_interior_N(f, g).sum()) for f in m.velocities]
        dt = 0.1 * g.dx / umax
        for _ in range(3):
            ocn.time_step(m, dt)
        ocn.flush_tendencies(m)
        ocn.sync_device()
        assert bool(torch.isfinite(m.u.data).all())
        assert _div_max(ocn, g, m) < 5e-8                            # test_time_stepping.jl:125-158
        scale = n ** 3 * umax
        for a, f in zip(mom0, m.velocities):                          # flux form + periodic pressure gradient conserve momentum
            assert abs(float(_interior_N(f, g).sum()) - a) < 1e-11 * scale
        # periodic halo identity on the filled fields (K18), a few planes
        H = 3
        u = m.u.data
        assert torch.equal(u[:, :, :H], u[:, :, n:n + H]) and torch.equal(u[:, :, n + H:], u[:, :, H:2 * H])
        assert torch.equal(u[:, :H, :], u[:, n:n + H, :]) and torch.equal(u[:H], u[n:n + H])
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)


def test_full_size_strict_vs_fast_and_fused_vs_unfused(ocn):
    """256^3: strict and fast math agree to 1e-10 after 2 steps; the fused stage boundaries are bit-identical to the
    unfused call sequence (strict)."""
    n = 256
    res = {}
    for mode, fused in (("strict", True), ("strict", False), ("fast", True)):
        ocn.set_math_mode(ocn.MATH_STRICT if mode == "strict" else ocn.MATH_FAST)
        g, m = _model(ocn, (n, n, n))
        if not fused:
            m.fuse_stage_boundaries = m.defer_final_tendencies = m.correct_on_load = False
        for _ in range(2):
            ocn.time_step(m, 0.002)
        ocn.flush_tendencies(m)
        ocn.sync_device()
        res[(mode, fused)] = [f.data.clone() for f in m.velocities + (m.pNHS,)]
        del m
    ocn.set_math_mode(ocn.MATH_STRICT)
    for a, b in zip(res[("strict", True)], res[("strict", False)]):
        assert torch.equal(a, b)
    scale = float(res[("strict", True)][0].abs().max())
    for a, b in zip(res[("strict", True)][:3], res[("fast", True)][:3]):
        assert float((a - b).abs().max()) <= 1e-10 * scale


def test_full_size_poisson_linearity(ocn):
    """solve!(ϕ, solver, a R1 + b R2) == a solve!(R1) + b solve!(R2) at 512^3 (the solver is linear), and the solution of
    a zero-mean source has zero mean."""
    n = 512
    g = ocn.RectilinearGrid(ocn.GPU(), size=(n, n, n), x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, P), halo=(3, 3, 3))
    s = ocn.FFTBasedPoissonSolver(g)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    R = [torch.randn((n, n, n), generator=gen, device="cuda", dtype=torch.float64) for _ in range(2)]
    R = [r - r.mean() for r in R]
    out = []
    for r in (R[0], R[1], 0.3 * R[0] - 1.7 * R[1]):
        phi = ocn.CenterField(g)
        s.set_source_term(r)
        s.solve(phi)
        out.append(_interior_N(phi, g).clone())
    err = float((out[2] - (0.3 * out[0] - 1.7 * out[1])).abs().max())
    assert err <= 1e-12 * float(out[2].abs().max())
    assert abs(float(out[0].mean())) <= 1e-14 * float(out[0].abs().max())


def test_full_size_config4_stretched(ocn):
    """configs[3] (advection-only physics): 512 x 512 x 256 (Periodic, Periodic, Bounded), stretched z of
    examples/ocean_wind_mixing_and_convection.jl:38-62, Fourier-tridiagonal solver."""
    Nz, Lz, refinement, stretching = 256, 32.0, 1.2, 12.0
    k = np.arange(1, Nz + 2)
    h = (k - 1) / Nz
    zf = Lz * ((1 + (h - 1) / refinement) * (1 - np.exp(-stretching * h)) / (1 - np.exp(-stretching)) - 1)
    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        g, m = _model(ocn, (512, 512, Nz), topo=(P, P, B), z=zf)
        assert m.pressure_solver.info()["kind"] == 1 and not m.correct_on_load
        umax = float(torch.stack([f.interior_view().abs().max() for f in m.velocities]).max())
        dzmin = float(np.diff(zf).min())
        assert _div_max(ocn, g, m) < 1e-10 * umax / dzmin
        # thickness-weighted horizontal momentum is an invariant of flux-form advection + projection with no-flux walls
        dzc = torch.from_numpy(np.diff(zf)).to("cuda").reshape(-1, 1, 1)
        mom0 = [float((_interior_N(f, g) * dzc).sum()) for f in (m.u, m.v)]
        dt = 0.1 * min(g.dx, dzmin) / umax
        for _ in range(2):
            ocn.time_step(m, dt)
        ocn.flush_tendencies(m)
        ocn.sync_device()
        assert bool(torch.isfinite(m.u.data).all())
        assert _div_max(ocn, g, m) < 5e-8
        w = m.w.data
        assert float(w[g.Hz].abs().max()) == 0.0 and float(w[g.Hz + Nz].abs().max()) == 0.0  # impenetrable walls
        scale = 512 * 512 * Lz * umax
        for a, f in zip(mom0, (m.u, m.v)):
            assert abs(float((_interior_N(f, g) * dzc).sum()) - a) < 1e-11 * scale
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)


def test_full_size_config4_ocean_wind_mixing(ocn):
    """configs[3] of BASELINE.json at full size (512 x 512 x 256 (P,P,B), stretched z, the example's physics incl. the AMD
    closure), through properties that do not need the oracle: incompressibility, finiteness, νₑ/κₑ >= 0, and exact tracer
    budgets -- the volume integral of S changes only by the evaporation flux (-rate·S at the surface, `value + coeff·c` BC)
    and that of T only by the surface heat flux plus the diffusive flux through the bottom face (bottom Gradient BC)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import C4, config4_faces
    N, Nz = 512, 256
    zf = config4_faces(Nz)
    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        g = ocn.RectilinearGrid(ocn.GPU(), size=(N, N, Nz), x=(0, 64), y=(0, 64), z=zf, topology=(P, P, B), halo=(3, 3, 3))
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(C4["taux"])),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(C4["JT"]), bottom=ocn.GradientBoundaryCondition(C4["dTdz"])),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-C4["evap"]))}
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=C4["f"]),
                                    closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(C4["alpha"], C4["beta"])),
                                    boundary_conditions=bcs)
        assert m.fuse_stage_boundaries and m.pressure_solver.info()["kind"] == 1
        gen = torch.Generator(device="cuda")
        gen.manual_seed(7)
        zc = torch.from_numpy(0.5 * (zf[1:] + zf[:-1])).to("cuda")
        T = m.field("T").interior_view()
        T.copy_(20 + C4["dTdz"] * zc[:, None, None] + 1e-4 * torch.rand(T.shape, generator=gen, device="cuda", dtype=torch.float64))
        S = m.field("S").interior_view()
        S.copy_(35 + 1e-3 * torch.rand(S.shape, generator=gen, device="cuda", dtype=torch.float64))
        for f in m.velocities:
            iv = f.interior_view()
            iv.copy_(1e-2 * (torch.rand(iv.shape, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1))
        ocn.set(m)
        dzc = torch.from_numpy(np.diff(zf)).to("cuda")[:, None, None]

        def integral(name):  # per unit horizontal area
            return float((m.field(name).interior_view() * dzc).sum()) / (N * N)

        # one short RK3 step: flux-form advection and the interior diffusive fluxes cancel in the column integral, so the
        # integrals move only by the boundary fluxes, which vary by O(1e-3) relative across the three stages
        S0, T0 = integral("S"), integral("T")
        dt = 1e-3
        ocn.time_step(m, dt)
        ocn.flush_tendencies(m)
        ocn.sync_device()
        assert all(bool(torch.isfinite(f.data).all()) for f in m.prognostic_fields())
        assert _div_max(ocn, g, m) < 5e-8
        nu = m.diffusivity_fields["nu_e"].interior_view()
        assert float(nu.min()) >= 0 and float(nu.max()) > 0
        assert all(float(k.interior_view().min()) >= 0 for k in m.diffusivity_fields["kappa_e"])
        S_top = float(m.field("S").interior_view()[-1].mean())
        # dS/dt integrated = -(top flux) = rate * <S_top>;  S_top varies by O(1e-3) relative across the step
        dS = integral("S") - S0
        assert abs(dS - C4["evap"] * S_top * dt) < 1e-3 * C4["evap"] * 35 * dt + 1e-13 * 35 * float(zf[-1] - zf[0])
        # T: top flux JT leaves; the bottom Gradient condition lets -κₑ dTdz through the bottom face, κₑ >= 0 and tiny (AMD)
        dT = integral("T") - T0
        kb = float(m.diffusivity_fields["kappa_e"][0].interior_view()[0].mean())
        assert abs(dT - (-C4["JT"] * dt - kb * C4["dTdz"] * dt)) < 0.05 * C4["JT"] * dt + 1e-13 * 20 * float(zf[-1] - zf[0])
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)
