"""GPU parity of the Poisson solvers and of whole time steps against the CPU oracle.

Tolerances (fp64): anything downstream of an FFT differs from the oracle by FFT round-off
(rocFFT real-to-complex vs pocketfft complex): pressure within 1e-10 * max|p| (the north-star bound);
velocities after n steps within 1e-11 * max|u| (strict math)."""
import numpy as np
import pytest

from helpers import from_dev, make_pair, random_parent, stretched_faces, to_dev

pytestmark = pytest.mark.gpu
LOCS = (1, 2, 4)

POISSON_CASES = [((16, 16, 16), "PPP", (0, 2 * np.pi)),
                 ((7, 11, 16), "PPP", (0, 1.0)),      # odd sizes (test_poisson_solvers.jl uses 7, 11)
                 ((32, 16, 8), "PPP", (0, 1.0)),
                 ((16, 12, 64), "PPP", (0, 1.0)),     # Nz in {64,128,256,512}: fused FFT_z + solve + IFFT_z column kernel
                 ((20, 16, 128), "PPP", (0, 2.0)),
                 ((9, 8, 256), "PPP", (0, 1.0)),
                 ((8, 8, 512), "PPP", (0, 4.0)),
                 ((128, 64, 64), "PPP", (0, 1.0)),    # Nx in {128..1024} and Ny, Nz in {64..512}: fully custom transform pipeline
                 ((256, 128, 64), "PPP", (0, 2.0)),
                 ((16, 12, 10), "PPB", (-1.0, 0.0)),
                 ((16, 12, 9), "PPB", "stretched"),
                 ((11, 7, 9), "PPB", "stretched"),
                 ((128, 64, 12), "PPB", "stretched"),  # row / column kernels for (x, y) + Thomas sweep, no rocFFT
                 ((256, 128, 9), "PPB", (-2.0, 0.0)),
                 ((24, 16, 1), "PPF", None)]


def _grid(O, ocn, size, topo, z):
    if isinstance(z, str):
        z = stretched_faces(size[2], 2.0)
    return make_pair(O, ocn, size, topo, x=(0, 2 * np.pi), y=(0, 3.0), z=z)


@pytest.mark.parametrize("size,topo,z", POISSON_CASES)
def test_poisson_laplacian_equals_source(oracle, ocn, size, topo, z):
    """test_poisson_solvers.jl:58-106 / test_poisson_solvers_stretched_grids.jl:12-50 re-expressed:
    for a random divergent velocity, ∇²ϕ ≈ R = ∇·U (isapprox, rtol = sqrt(eps))."""
    O = oracle
    rng = np.random.default_rng(1234)
    og, pg = _grid(O, ocn, size, topo, z)
    hosts = []
    for l in LOCS:
        a = og.zeros(l)
        og.interior(a)[...] = rng.random(og.interior(a).shape)
        O.fill_halo_regions(og, a, l)
        hosts.append(a)
    R = O.divergence(og, *hosts)
    du, dv, dw = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS, hosts))
    solver = ocn.nonhydrostatic_pressure_solver(pg)
    phi = ocn.CenterField(pg)
    if topo == "PPP":
        assert bool(solver.info()["direct_out"] & 2) == (size[2] in (64, 128, 256, 512))  # fused z kernel in use
        assert bool(solver.info()["direct_out"] & 4) == (size[0] in (128, 256) and size[1] in (64, 128))  # custom x, y passes
    if topo == "PPB":
        assert bool(solver.info()["direct_out"] & 4) == (size[0] in (128, 256) and size[1] in (64, 128))
    ocn.solve_for_pressure(phi, solver, 1.0, (du, dv, dw))
    ocn.fill_halo_regions(phi)
    ocn.sync_device()
    lap = O.laplacian(og, np.asfortranarray(from_dev(phi)))
    assert np.linalg.norm(lap - R) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(R)
    assert np.abs(lap - R).max() <= 1e-10 * max(1.0, np.abs(R).max())
    # and against the oracle's own solve (pressure to the north-star tolerance)
    S = O.FourierTridiagonalPoissonSolver(og) if topo[2] == "B" else O.FFTBasedPoissonSolver(og)
    p0 = og.zeros(0)
    S.source_term(*hosts, 1.0)
    S.solve(p0)
    err = np.abs(og.interior_N(np.asfortranarray(from_dev(phi))) - og.interior_N(p0)).max()
    assert err <= 1e-10 * max(1.0, np.abs(p0).max())


ALL_TOPOS = ["PPP", "PPB", "PBP", "PBB", "BPP", "BPB", "BBP", "BBB"]


@pytest.mark.parametrize("topo", ALL_TOPOS + ["PBF", "BBF", "BPF"])
@pytest.mark.parametrize("size", [(7, 7, 7), (16, 16, 16), (11, 16, 7)])
def test_fft_poisson_all_topologies(oracle, ocn, topo, size):
    """test_poisson_solvers.jl:58-106 re-expressed on the GPU for every regular topology: the FFT-based solver with cosine transforms along
    Bounded dimensions (the reference's K11 path; csrc/poisson.hip kind 2).  R = div(U) of a random velocity computed by the oracle (fields
    on grids with a Bounded x or y exist only there), set as the source term; ∇²ϕ ≈ R to sqrt(eps) and ϕ equals the oracle's solution to
    1e-10.  (Periodic, Periodic, ·) topologies run the same kernels through `general=True`."""
    O = oracle
    if topo[2] == "F":
        size = size[:2]
    rng = np.random.default_rng(1234)
    kw = dict(x=(0, 1), y=(0, 2.5)) if topo[2] == "F" else dict(x=(0, 1), y=(0, 2.5), z=(0, 4))
    og = O.Grid(size, topology=topo, halo=(3, 3, 3)[:len(size)], **kw)
    names = {"P": "Periodic", "B": "Bounded", "F": "Flat"}
    pg = ocn.RectilinearGrid(ocn.GPU(), size=size, topology=tuple(names[t] for t in topo), halo=(3, 3, 3)[:len(size)], **kw)
    U = []
    for loc in (1, 2, 4):
        a = og.zeros(loc)
        og.interior(a)[...] = rng.random(og.interior(a).shape)
        O.fill_halo_regions(og, a, loc)
        U.append(a)
    if topo[2] == "F":
        U[2][...] = 0
    R = O.divergence(og, *U)
    solver = ocn.FFTBasedPoissonSolver(pg, general=True)
    assert solver.info()["kind"] == 2
    solver.set_source_term(R)
    phi = ocn.CenterField(pg)
    solver.solve(phi)
    ocn.sync_device()
    p = np.asfortranarray(from_dev(phi))
    O.fill_halo_regions(og, p, 0)
    lap = O.laplacian(og, p)
    assert np.linalg.norm(lap - R) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(R)
    S = O.FFTBasedPoissonSolver(og)
    p0 = og.zeros(0)
    S.source_term(*U, 1.0)
    S.solve(p0)
    assert np.abs(og.interior_N(p) - og.interior_N(p0)).max() <= 1e-10 * max(1.0, np.abs(p0).max())


def test_fft_poisson_bounded_converges_to_analytic_modes(oracle, ocn):
    """test_poisson_solvers.jl:109-141: second-order convergence of the (Bounded, Bounded, Bounded) solve to cos(x/2) cos(y/2) cos(z/2)."""
    errs = []
    for N in (16, 32):
        g = ocn.RectilinearGrid(ocn.GPU(), size=(N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=("Bounded",) * 3)
        xc = (np.arange(N) + 0.5) * g.dx
        psi = np.cos(xc / 2)
        Psi = psi[:, None, None] * psi[None, :, None] * psi[None, None, :]
        solver = ocn.nonhydrostatic_pressure_solver(g)
        assert solver.info()["kind"] == 2
        solver.set_source_term(-0.75 * Psi)
        phi = ocn.CenterField(g)
        solver.solve(phi)
        ocn.sync_device()
        errs.append(np.abs(phi.interior() - (Psi - Psi.mean())).max())
    assert 3.5 < errs[0] / errs[1] < 4.5


def test_two_solver_handles_coexist(oracle, ocn):
    """Regression for the rocFFT (ROCm 7.2) real-plan collision (tools/rocfft_repro.hip, profiles/r02_rocfft_repro.md): with a
    16^3 (P,P,P) solver alive, the real-to-complex plans of a (32, 8, k) solver come out wrong -- power-of-two pairs whose 2-D kernel
    lengths (Nx/2, Ny) are the transpose of a live plan's; the EARLIER plan stays correct.  Every new plan pair is verified over its
    full spectrum at creation (plane waves + a host DFT) and replaced by complex plans if it fails.  Both handles must solve
    correctly in either creation order, and the first one must still be correct after the second has been created."""
    O = oracle

    def make(size, topo, z):
        og, pg = make_pair(O, ocn, size, topo, x=(0, 64), y=(0, 64), z=z)
        rng = np.random.default_rng(1)
        U = []
        for loc in (1, 2, 4):
            a = og.zeros(loc)
            og.interior(a)[...] = rng.uniform(-1, 1, og.interior(a).shape)
            O.fill_halo_regions(og, a, loc)
            U.append(a)
        R = O.divergence(og, *U)
        dU = [to_dev(ocn, pg, l, a) for l, a in zip((1, 2, 4), U)]
        S = ocn.nonhydrostatic_pressure_solver(pg)

        def residual():
            p = ocn.CenterField(pg)
            ocn.solve_for_pressure(p, S, 1.0, dU)
            ocn.fill_halo_regions(p)
            ocn.sync_device()
            return np.linalg.norm(O.laplacian(og, from_dev(p)) - R) / np.linalg.norm(R)

        return S, residual

    A = ((16, 16, 16), "PPP", (0, 64))
    B = ((32, 8, 16), "PPB", stretched_faces(16, 32.0))
    C = ((32, 8, 8), "PPP", (0, 64))          # the exact pair of the standalone reproduction: 16^3 3-D, then 32 x 8 batched over 8
    D = ((16, 16, 4), "PPB", (-4.0, 0.0))
    for order in ((A, B), (B, A), (A, C), (C, A), (D, C), (A, B, C, D)):
        handles = [make(*case) for case in order]
        for _, residual in handles:               # every handle, after ALL of them exist
            assert residual() < 1e-12
        for _, residual in reversed(handles):
            assert residual() < 1e-12
        del handles


def test_poisson_set_source_term(oracle, ocn):
    """solve!(ϕ, solver, b): ∇²ϕ = b for a zero-mean b."""
    O = oracle
    rng = np.random.default_rng(3)
    og, pg = make_pair(O, ocn, (16, 12, 20), "PPP", z=(0, 1.0))
    R = rng.standard_normal((16, 12, 20))
    R -= R.mean()
    solver = ocn.FFTBasedPoissonSolver(pg)
    solver.set_source_term(R)
    phi = ocn.CenterField(pg)
    solver.solve(phi)
    ocn.fill_halo_regions(phi)
    ocn.sync_device()
    lap = O.laplacian(og, np.asfortranarray(from_dev(phi)))
    assert np.abs(lap - R).max() <= 1e-10 * np.abs(R).max()


@pytest.mark.parametrize("size", [(64, 64, 64), (128, 64, 256), (64, 128, 512)])
def test_cosine_transform_z_pass_equals_the_thomas_sweep(oracle, ocn, size, monkeypatch):
    """(Periodic, Periodic, Bounded) with a REGULAR z of a column-kernel length: the Fourier-tridiagonal handle replaces its Thomas sweep by
    the exact spectral twin -- REDFT10_z, division by λx + λy + λz, REDFT01_z in one column pass (what the reference's FFTBasedPoissonSolver
    does on such a grid, plan_transforms.jl:129-140).  The same source through both (OCN_POISSON_DCT_Z=0: the sweep): solutions agree to
    1e-11 of max|ϕ|, ∇²ϕ reproduces the source, and the source given explicitly (set_source_term!) takes the same path."""
    import torch
    O = oracle
    og, pg = _grid(O, ocn, size, "PPB", (-1.0, 0.0))
    rng = np.random.default_rng(77)
    hosts = []
    for l in LOCS:
        a = og.zeros(l)
        og.interior(a)[...] = rng.random(og.interior(a).shape)
        O.fill_halo_regions(og, a, l)
        hosts.append(a)
    R = O.divergence(og, *hosts)
    du, dv, dw = (to_dev(ocn, pg, l, a) for l, a in zip(LOCS, hosts))
    sols = []
    for flag in ("1", "0"):
        monkeypatch.setenv("OCN_POISSON_DCT_Z", flag)
        solver = ocn.nonhydrostatic_pressure_solver(pg)
        phi = ocn.CenterField(pg)
        ocn.solve_for_pressure(phi, solver, 1.0, (du, dv, dw))
        ocn.fill_halo_regions(phi)
        ocn.sync_device()
        sols.append(np.asfortranarray(from_dev(phi)))
        if flag == "1":
            lap = O.laplacian(og, sols[0])
            assert np.abs(lap - R).max() <= 1e-10 * np.abs(R).max()
            phi2 = ocn.CenterField(pg)
            solver.set_source_term(R)
            solver.solve(phi2)
            ocn.sync_device()
            assert np.abs(og.interior(np.asfortranarray(from_dev(phi2))) - og.interior(sols[0])).max() <= 1e-11 * np.abs(sols[0]).max()
    assert np.abs(og.interior(sols[0]) - og.interior(sols[1])).max() <= 1e-11 * np.abs(sols[1]).max()


def test_batched_tridiagonal_solver_matches_dense(ocn):
    """test_batched_tridiagonal_solver.jl:7-114 re-expressed: equals Tridiagonal(a,b,c) \\ f, random diag-dominant."""
    rng = np.random.default_rng(5)
    Nx, Ny, Nz = 5, 4, 17
    a, c = rng.random(Nz - 1), rng.random(Nz - 1)
    b = 3 + rng.random((Nx, Ny, Nz))
    f = rng.standard_normal((Nx, Ny, Nz)) + 1j * rng.standard_normal((Nx, Ny, Nz))
    solver = ocn.BatchedTridiagonalSolver(ocn.GPU(), a, b, c)
    phi = solver.solve(f)
    for i in range(Nx):
        for j in range(Ny):
            M = np.diag(b[i, j]) + np.diag(a, -1) + np.diag(c, 1)
            np.testing.assert_allclose(phi[i, j], np.linalg.solve(M, f[i, j]), rtol=1e-12, atol=1e-13)


def test_batched_tridiagonal_solver_bitwise_vs_oracle(oracle, ocn):
    rng = np.random.default_rng(6)
    Nx, Ny, Nz = 70, 3, 12
    a, c = rng.random(Nz - 1), rng.random(Nz - 1)
    b = 3 + rng.random((Nx, Ny, Nz))
    f = rng.standard_normal((Nx, Ny, Nz)) + 1j * rng.standard_normal((Nx, Ny, Nz))
    ref = oracle.batched_tridiagonal_solve_z(a, b, c, f)
    phi = ocn.BatchedTridiagonalSolver(ocn.GPU(), a, b, c).solve(f)
    np.testing.assert_array_equal(phi, ref)


def test_batched_tridiagonal_solver_keeps_storage_at_a_singular_pivot(ocn):
    """batched_tridiagonal_solver.jl:224-228: where a pivot is not definitely diagonally dominant (|β| <= 10 eps) ϕ[k] keeps what the storage
    held -- the caller's ϕ for the stand-alone solver (the Poisson solvers take 0 there: the free constant of the singular column).
    b = (1, 1), a = c = (1): β₂ = 1 - 1 * 1 = 0, so ϕ₂ = ϕ⁰₂ and ϕ₁ = f₁ / b₁ - ϕ⁰₂."""
    a, c = np.array([1.0]), np.array([1.0])
    b = np.ones((3, 2, 2))
    f = np.arange(12, dtype=np.float64).reshape(3, 2, 2) + 1j * np.arange(12, dtype=np.float64).reshape(3, 2, 2)[::-1]
    phi0 = np.full((3, 2, 2), 7.0 - 2.0j)
    phi = ocn.BatchedTridiagonalSolver(ocn.GPU(), a, b, c).solve(f, phi0)
    np.testing.assert_array_equal(phi[..., 1], phi0[..., 1])
    np.testing.assert_array_equal(phi[..., 0], f[..., 0] / 1.0 - 1.0 * phi0[..., 1])


MODEL_CASES = [((16, 16, 16), "PPP", (0, 2 * np.pi), "RungeKutta3"),
               ((16, 16, 16), "PPP", (0, 2 * np.pi), "QuasiAdamsBashforth2"),
               ((13, 17, 19), "PPP", (0, 1.0), "RungeKutta3"),
               ((16, 16, 64), "PPP", (0, 8 * np.pi), "RungeKutta3"),
               ((128, 64, 64), "PPP", (0, np.pi), "RungeKutta3"),
               ((16, 12, 10), "PPB", (-1.0, 0.0), "RungeKutta3"),
               ((16, 12, 10), "PPB", "stretched", "RungeKutta3"),
               ((24, 16, 1), "PPF", None, "RungeKutta3")]


@pytest.mark.parametrize("size,topo,z,ts", MODEL_CASES)
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_time_steps_match_oracle(oracle, ocn, size, topo, z, ts, mode):
    """set! (with the Δt = 1 projection) then 3 time steps; fields vs the oracle; max|∇·u| < 5e-8
    (test_time_stepping.jl:125-158)."""
    O = oracle
    rng = np.random.default_rng(1234)
    if isinstance(z, str):
        z = stretched_faces(size[2], 1.0)
    og, pg = make_pair(O, ocn, size, topo, z=z)
    om = O.NonhydrostaticModel(og, timestepper="RungeKutta3" if ts == "RungeKutta3" else "QAB2")
    ocn.set_math_mode(ocn.MATH_STRICT if mode == "strict" else ocn.MATH_FAST)
    try:
        pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), timestepper=ts)
        pm.copy_cached_tendencies = (mode == "strict" and topo == "PPB")  # also exercise the literal K7 copy kernel
        init = {n: rng.uniform(-1, 1, og.interior(f).shape) for n, f in zip("uvw", (om.u, om.v, om.w))}
        if topo[2] == "F":
            init["w"] = np.zeros_like(init["w"])
        om.set(**init)
        ocn.set(pm, **init)
        umax = max(np.abs(om.u).max(), np.abs(om.v).max(), np.abs(om.w).max())
        dt = 0.1 * min(og.dx, og.dy) / umax
        for _ in range(3):
            om.time_step(dt)
            ocn.time_step(pm, dt)
        ocn.sync_device()
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)
    tol = 1e-11 if mode == "strict" else 1e-10
    scale = max(np.abs(om.u).max(), np.abs(om.v).max())
    for name, a, d in zip("uvw", (om.u, om.v, om.w), pm.velocities):
        err = np.abs(og.interior(from_dev(d)) - og.interior(a)).max()
        assert err <= tol * scale, f"{name}: {err} > {tol * scale}"
    pscale = max(np.abs(om.p).max(), 1e-30)
    assert np.abs(og.interior_N(from_dev(pm.pNHS)) - og.interior_N(om.p)).max() <= 1e-10 * max(1.0, pscale)
    import torch
    ddiv = torch.zeros((og.Nz, og.Ny, og.Nx), dtype=torch.float64, device=pm.u.data.device)
    ocn._lib.call("ocn_divergence", pg.cref, pm.u.ptr, pm.v.ptr, pm.w.ptr, ddiv.data_ptr(), 0)
    assert float(ddiv.abs().max()) < 5e-8
    assert pm.clock.iteration == 3


def test_math_mode_is_a_property_of_the_model(oracle, ocn):
    """Two models of ONE process with different arithmetic variants (ocn_grid.math): the strict one stays bit-identical to a model run
    under the strict process default while the process default is FAST and a fast model steps in between; the fast one equals a model
    run under the FAST default."""
    O = oracle
    rng = np.random.default_rng(7)
    size = (32, 16, 12)
    og, pg = make_pair(O, ocn, size, "PPP", z=(0, 2.0))
    init = {n: rng.uniform(-1, 1, size) for n in "uvw"}
    dt = 0.02 * og.dx

    def run(default, pinned, other=None):
        ocn.set_math_mode(default)
        try:
            m = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), math_mode=pinned)
            ocn.set(m, **init)
            o = None
            if other is not None:
                o = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), math_mode=other)
                ocn.set(o, **init)
            for _ in range(2):
                ocn.time_step(m, dt)
                if o is not None:
                    ocn.time_step(o, dt)
            ocn.flush_tendencies(m)
            ocn.sync_device()
            return [from_dev(f).copy() for f in m.velocities], ([from_dev(f).copy() for f in o.velocities] if o is not None else None)
        finally:
            ocn.set_math_mode(ocn.MATH_STRICT)

    ref_strict, _ = run(ocn.MATH_STRICT, None)
    ref_fast, _ = run(ocn.MATH_FAST, None)
    assert any(np.abs(a - b).max() > 0 for a, b in zip(ref_strict, ref_fast))  # the variants do differ in the last bits
    pinned_strict, interleaved_fast = run(ocn.MATH_FAST, ocn.MATH_STRICT, other=ocn.MATH_FAST)
    for a, b in zip(ref_strict, pinned_strict):
        assert np.array_equal(a, b)
    for a, b in zip(ref_fast, interleaved_fast):
        assert np.array_equal(a, b)
    assert pg.math_mode is None and ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), math_mode=ocn.MATH_FAST).grid.math_mode == ocn.MATH_FAST


@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_config1_two_dimensional_turbulence_at_its_real_size(oracle, ocn, mode):
    """BASELINE.json configs[0] at its real size: the README example (README.md:111-121) -- 128 x 128 (Periodic, Periodic, Flat), extent
    2 pi, NonhydrostaticModel with advection = WENO(), RungeKutta3, u, v ~ rand, Dt = 0.01 -- 10 time steps against the oracle, both
    arithmetic variants.  (The README example also sets a ScalarDiffusivity closure for its long run; the hot path of the benchmark
    configuration is advection + projection, which is what is compared here.)"""
    O = oracle
    rng = np.random.default_rng(1234)
    og, pg = make_pair(O, ocn, (128, 128, 1), "PPF", x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=None, halo=(3, 3, 0))
    om = O.NonhydrostaticModel(og, timestepper="RungeKutta3")
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), math_mode=ocn.MATH_STRICT if mode == "strict" else ocn.MATH_FAST)
    init = {"u": rng.uniform(0, 1, og.interior(om.u).shape), "v": rng.uniform(0, 1, og.interior(om.v).shape), "w": np.zeros(og.interior(om.w).shape)}
    om.set(**init)
    ocn.set(pm, **init)
    for _ in range(10):
        om.time_step(0.01)
        ocn.time_step(pm, 0.01)
    ocn.sync_device()
    tol = 1e-11 if mode == "strict" else 1e-10
    scale = max(np.abs(om.u).max(), np.abs(om.v).max())
    for name, a, d in zip("uv", (om.u, om.v), pm.velocities):
        err = np.abs(og.interior(from_dev(d)) - og.interior(a)).max()
        assert err <= tol * scale, f"{name}: {err} > {tol * scale}"
    assert float(pm.w.data.abs().max()) == 0.0  # Gw = 0 and w stays 0 on a Flat z (flat_advective_fluxes.jl:8-44)
    assert np.abs(og.interior_N(from_dev(pm.pNHS)) - og.interior_N(om.p)).max() <= 1e-10 * max(1.0, np.abs(om.p).max())
    assert pm.clock.iteration == 10


def test_first_ab2_step_is_euler(ocn):
    """test_time_stepping.jl:90-118 idea: with Δt != last_Δt the QAB2 step is forward Euler (χ = -0.5), so
    u¹ = u⁰ + Δt G⁰ exactly before the projection; here checked through the kernel entry."""
    import torch
    pg = ocn.RectilinearGrid(ocn.GPU(), size=(8, 8, 8), x=(0, 1), y=(0, 1), z=(0, 1), topology=("Periodic",) * 3)
    U, Gn, Gm = ocn.Field(0, pg), ocn.Field(0, pg), ocn.Field(0, pg)
    U.data.fill_(1.0); Gn.data.fill_(2.0); Gm.data.fill_(float("nan"))
    pa, ia = ocn._lib.ptr_array, ocn._lib.i32_array
    ocn._lib.call("ocn_ab2_step", pg.cref, 1, pa([U.ptr]), pa([Gn.ptr]), pa([Gm.ptr]), ia([0]), 0.5, -0.5, 0)
    ocn.sync_device()
    iv = U.interior_view()
    # NaN * 0 = NaN in IEEE: the reference multiplies G⁻ by `not_euler` = false, which in Julia is a *strong zero*
    # (false * NaN == 0.0); the kernels reproduce that.
    assert torch.all(iv == 2.0)


def test_tracer_conservation(oracle, ocn):
    """tracer advection in flux form conserves the tracer sum (test_time_stepping.jl:160+ idea), and matches the oracle."""
    O = oracle
    rng = np.random.default_rng(2)
    og, pg = make_pair(O, ocn, (16, 16, 16), "PPP")
    om = O.NonhydrostaticModel(og, tracers=("c",))
    pm = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), tracers=("c",))
    init = {n: rng.uniform(-1, 1, (16, 16, 16)) for n in "uvw"}
    init["c"] = rng.uniform(0, 1, (16, 16, 16))
    om.set(**init)
    ocn.set(pm, **init)
    c0 = pm.tracers[0].interior().sum()
    for _ in range(3):
        om.time_step(0.01)
        ocn.time_step(pm, 0.01)
    ocn.sync_device()
    c1 = pm.tracers[0].interior()
    assert abs(c1.sum() - c0) <= 1e-10 * abs(c0)
    assert np.abs(c1 - og.interior(om.tracers[0])).max() <= 1e-11


@pytest.mark.parametrize("topo,z", [("PPP", (0, 2 * np.pi)), ("PPB", "stretched")])
def test_fused_stage_boundaries_equal_unfused(ocn, topo, z):
    """The fused launch (tendencies + next rk3 substep) is bit-identical to the two separate launches (strict math),
    and the model's velocity storage is back in place after every full step."""
    rng = np.random.default_rng(77)
    size = (20, 12, 10)
    if isinstance(z, str):
        z = stretched_faces(size[2], 1.0)
    P, B = "Periodic", "Bounded"
    kw = dict(size=size, x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=(P, P, P if topo[2] == "P" else B), halo=(3, 3, 3))
    ocn.set_math_mode(ocn.MATH_STRICT)
    models = []
    init = None
    for fused in (True, False):  # fused = stage-boundary fusion (+ pressure correction on load when all-periodic)
        g = ocn.RectilinearGrid(ocn.GPU(), **kw)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
        assert m.fuse_stage_boundaries
        if fused:
            assert m.correct_on_load == (topo == "PPP") and m.defer_final_tendencies
        else:
            m.fuse_stage_boundaries = m.defer_final_tendencies = m.correct_on_load = False
        if init is None:
            init = {n: rng.uniform(-1, 1, tuple(reversed(f.interior_view().shape))) for n, f in zip("uvw", m.velocities)}
        ocn.set(m, **init)
        for _ in range(3):
            ocn.time_step(m, 0.01)
        assert m._pending_tendencies == fused  # deferred final compute_tendencies! ...
        _ = m.timestepper.Gn                    # ... is completed by reading Gⁿ
        assert not m._pending_tendencies
        ocn.sync_device()
        models.append(m)
    a, b = models
    for fa, fb in zip(a.velocities + (a.pNHS,) + tuple(a.timestepper.Gn), b.velocities + (b.pNHS,) + tuple(b.timestepper.Gn)):
        np.testing.assert_array_equal(fa.parent(), fb.parent())


def test_hasnan_and_nan_checker(ocn):
    """src/Models/nan_checker.jl:33-52: hasnan(field) scans the whole parent array (halos included); NaNChecker names the field."""
    import torch
    g = ocn.RectilinearGrid(ocn.GPU(), size=(9, 7, 5), x=(0, 1), y=(0, 1), z=(0, 1), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers="c")
    assert not ocn.hasnan(m) and not ocn.hasnan(m.tracers[0])
    chk = ocn.NaNChecker({"u": m.u, "c": m.tracers[0]})
    assert chk(m) is None
    m.tracers[0].data.view(-1)[-1] = float("nan")  # the very last (odd-count) element, in a halo
    assert ocn.hasnan(m.tracers[0]) and not ocn.hasnan(m)
    assert chk(m) == "c"
    with pytest.raises(RuntimeError, match="NaN found in field c"):
        ocn.NaNChecker({"c": m.tracers[0]}, erroring=True)(m)
    m.u.data[2, 3, 4] = float("nan")
    assert ocn.hasnan(m)


@pytest.mark.parametrize("ts", ["RungeKutta3", "QuasiAdamsBashforth2"])
def test_checkpoint_pickup_is_exact(ocn, ts, tmp_path):
    """OutputWriters/checkpointer.jl:177-288 (test_checkpointer.jl idea): run 2 steps, checkpoint, run 2 more; a fresh model
    restored from the checkpoint and run 2 steps ends bit-identical."""
    rng = np.random.default_rng(3)
    N = (16, 12, 10)

    def build():
        g = ocn.RectilinearGrid(ocn.GPU(), size=N, x=(0, 1), y=(0, 1), z=stretched_faces(N[2], 1.0), topology=("Periodic", "Periodic", "Bounded"))
        return ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("b",), timestepper=ts, buoyancy=ocn.BuoyancyTracer(),
                                       closure=ocn.ScalarDiffusivity(ν=1e-3, κ=1e-3))

    ocn.set_math_mode(ocn.MATH_STRICT)
    m = build()
    ocn.set(m, u=rng.uniform(-1, 1, N), v=rng.uniform(-1, 1, N), b=rng.uniform(0, 1, N))
    dt = 2e-3
    for _ in range(2):
        ocn.time_step(m, dt)
    path = ocn.write_checkpoint(m, str(tmp_path / "checkpoint_iteration2"))
    for _ in range(2):
        ocn.time_step(m, dt)
    r = build()
    ocn.set_from_checkpoint(r, path)
    assert r.clock.iteration == 2 and r.clock.time == 2 * dt if ts != "RungeKutta3" else r.clock.iteration == 2
    for _ in range(2):
        ocn.time_step(r, dt)
    ocn.sync_device()
    for a, b in zip(m.prognostic_fields(), r.prognostic_fields()):
        np.testing.assert_array_equal(a.parent(), b.parent())
    assert r.clock.time == m.clock.time


def test_checkpoint_refuses_a_different_grid(ocn, tmp_path):
    """set!(model, filepath) compares the grids (checkpointer.jl:241-246): a different extent, stretching or topology with the same
    sizes must be refused (round 1 compared sizes and halos only)."""
    P = "Periodic"
    N = (16, 12, 10)
    m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, P)), advection=ocn.WENO())
    path = ocn.write_checkpoint(m, str(tmp_path / "ckpt"))
    for kw in (dict(x=(0, 2), y=(0, 1), z=(0, 1), topology=(P, P, P)), dict(x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, "Bounded")),
               dict(x=(0, 1), y=(0, 1), z=stretched_faces(N[2], 1.0), topology=(P, P, "Bounded"))):
        other = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, **kw), advection=ocn.WENO())
        with pytest.raises(ValueError, match="not the same"):
            ocn.set_from_checkpoint(other, path)
    same = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, P)), advection=ocn.WENO())
    ocn.set_from_checkpoint(same, path)


@pytest.mark.parametrize("fused", [True, False])
def test_hydrostatic_checkpoint_pickup_is_exact(ocn, fused, tmp_path):
    """QAB2 pickup of BASELINE.json configs[4]'s model: u, v, T, S, η, the barotropic velocities and Gⁿ / G⁻ (η included) are saved;
    a fresh model restored after 2 steps and run 2 more ends bit-identical to the uninterrupted run (the pickup step is a regular
    AB2 step, not an Euler one, because last_Δt is restored)."""
    rng = np.random.default_rng(9)
    N = (32, 12, 7)

    def build():
        g = ocn.RectilinearGrid(ocn.GPU(), size=N, x=(0, 4e3), y=(0, 1.5e3), z=stretched_faces(N[2], 40.0), topology=("Periodic", "Periodic", "Bounded"),
                                halo=(3, 3, 3))
        return ocn.HydrostaticFreeSurfaceModel(g, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                               free_surface=ocn.SplitExplicitFreeSurface(substeps=12), coriolis=ocn.FPlane(f=1e-4),
                                               closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                               buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), fused=fused)

    ocn.set_math_mode(ocn.MATH_STRICT)
    m = build()
    m.set(u=1e-2 * rng.uniform(-1, 1, N), v=1e-2 * rng.uniform(-1, 1, N), eta=1e-2 * rng.uniform(-1, 1, N[:2]),
          T=20 + 1e-2 * rng.uniform(-1, 1, N), S=35 + 1e-2 * rng.uniform(-1, 1, N))
    for _ in range(2):
        m.time_step(20.0)
    path = ocn.write_checkpoint(m, str(tmp_path / "hydrostatic_iteration2"))
    for _ in range(2):
        m.time_step(20.0)
    r = build()
    ocn.set_from_checkpoint(r, path)
    assert r.clock.iteration == 2 and r.clock.time == 40.0
    for _ in range(2):
        r.time_step(20.0)
    ocn.sync_device()
    for a, b in zip([m.u, m.v, m.w] + list(m.tracers), [r.u, r.v, r.w] + list(r.tracers)):
        np.testing.assert_array_equal(a.interior(), b.interior())
    g = m.grid
    ii, jj = slice(g.Hy, g.Hy + g.Ny), slice(g.Hx, g.Hx + g.Nx)
    for a, b in ((m.eta, r.eta), (m.U, r.U), (m.V, r.V)):
        np.testing.assert_array_equal(a[ii, jj].cpu().numpy(), b[ii, jj].cpu().numpy())


def test_set_with_functions(ocn):
    """set!(model, u=f(x, y, z), ...) (set_nonhydrostatic_model.jl:33-60): functions are evaluated at each field's own nodes."""
    g = ocn.RectilinearGrid(ocn.GPU(), size=(8, 6, 5), x=(0, 2), y=(-1, 1), z=[-3.0, -2.0, -1.2, -0.5, -0.1, 0.0],
                            topology=("Periodic", "Periodic", "Bounded"))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers="c")
    ocn.set(m, enforce_incompressibility=False, u=lambda x, y, z: x + 10 * y + 100 * z, w=lambda x, y, z: z, c=lambda x, y, z: 0 * x + np.cos(z))
    xF, xC = np.arange(8) * 0.25, (np.arange(8) + 0.5) * 0.25
    yC = -1 + (np.arange(6) + 0.5) / 3
    zF = np.array([-3.0, -2.0, -1.2, -0.5, -0.1, 0.0])
    zC = 0.5 * (zF[1:] + zF[:-1])
    np.testing.assert_allclose(m.u.interior(), xF[:, None, None] + 10 * yC[None, :, None] + 100 * zC[None, None, :], rtol=0, atol=1e-13)
    wcol = m.w.interior()[3, 2, :]                            # Face in Bounded z: Nz + 1 nodes
    assert np.array_equal(wcol[1:-1], zF[1:-1]) and wcol[0] == 0 and wcol[-1] == 0  # walls: impenetrable fill of set!
    assert np.array_equal(m.tracers[0].interior()[0, 0, :], np.cos(zC))
    ocn.set(m, enforce_incompressibility=False, v=lambda x, y, z: float(1.5))                  # scalar-valued function: element-wise fallback / broadcast
    assert np.all(m.v.interior() == 1.5)


@pytest.mark.parametrize("topo,z", [("PPP", (0, 1.0)), ("PPB", "stretched")])
def test_cell_advection_timescale_and_wizard(ocn, topo, z):
    """src/Advection/cell_advection_timescale.jl:13-35 and src/Simulations/time_step_wizard.jl:101-115: the device
    reduction equals the numpy restatement of the formula exactly; the wizard applies cfl, max_change, min_change, max_Δt."""
    rng = np.random.default_rng(8)
    N = (20, 9, 7)
    zz = stretched_faces(N[2], 1.0) if isinstance(z, str) else z
    g = ocn.RectilinearGrid(ocn.GPU(), size=N, x=(0, 2), y=(0, 3), z=zz, topology=tuple({"P": "Periodic", "B": "Bounded"}[t] for t in topo))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    ocn.set(m, u=rng.uniform(-1, 1, N), v=rng.uniform(-1, 1, N))
    u, v, w = (f.interior()[:, :, :N[2]] for f in m.velocities)
    dzf = g._dzf_host[g.Hz:g.Hz + N[2]] if g._dzf_host is not None else np.full(N[2], g.dz)
    ref = np.min(1 / ((np.abs(u) / g.dx + np.abs(v) / g.dy) + np.abs(w) / dzf[None, None, :]))
    tau = ocn.cell_advection_timescale(m)
    assert tau == ref
    assert ocn.AdvectiveCFL(0.01)(m) == 0.01 / ref
    wiz = ocn.TimeStepWizard(cfl=1.0, max_change=1.1, max_dt=60.0)
    assert wiz(m, 10 * tau) == max(0.5 * 10 * tau, tau)   # shrink limited by min_change
    assert wiz(m, 0.5 * tau) == 1.1 * 0.5 * tau           # growth limited by max_change
    assert ocn.TimeStepWizard(cfl=1.0, max_dt=tau / 3, max_change=100)(m, tau) == tau / 3
    rest = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    assert ocn.cell_advection_timescale(rest) == float("inf")


def test_advective_cfl_reference_doctest_value(ocn):
    """A known answer the reference itself holds (jldoctest in src/Diagnostics/cfl.jl:36-49): RectilinearGrid(size = (16, 16, 16),
    extent = (8, 8, 8)) (default topology (Periodic, Periodic, Bounded)), u .= pi, AdvectiveCFL(1.0)(model) == 6.283185307179586."""
    g = ocn.RectilinearGrid(ocn.GPU(), size=(16, 16, 16), x=(0, 8), y=(0, 8), z=(-8, 0), topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    m.u.data.fill_(np.pi)
    assert ocn.AdvectiveCFL(1.0)(m) == 6.283185307179586


@pytest.mark.parametrize("size,topo,z,own", [((32, 16, 12), "PPP", (0, 2.0), False), ((16, 12, 9), "PPB", "stretched", False),
                                             ((12, 9, 5), "PPP", (0, 1.0), False), ((24, 16, 1), "PPF", None, False),
                                             ((128, 64, 64), "PPP", (0, 2.0), True),  # own solver handle: hand-written FFT pipeline
                                             # grids with walls (round 4): tiled epilogue on the interior box, finishing kernel on the frames
                                             ((40, 20, 9), "PBB", (-0.7, 0), False), ((30, 18, 8), "BBB", "stretched", True),
                                             ((12, 10, 9), "BBB", (-0.7, 0), False), ((26, 12, 10), "BPP", (0, 1.0), False)])
@pytest.mark.parametrize("defer", [None, False])
def test_c_driver_equals_host_orchestration(oracle, ocn, size, topo, z, own, defer):
    """ocn_rk3_driver_time_step (the whole RK3 step behind one C entry point, csrc/driver.hip) against the Python host's
    time_step: same launches in the same order, so velocities, pressure and G^n agree bit for bit after 3 steps (and after an
    intermediate flush, which brings the velocities back into the caller's arrays and breaks the deferral chain).  defer = None: the
    library's default, which on all-periodic grids also defers the THIRD stage's pressure correction into the next step's first
    launch (ocn_rk3_driver_configure); flush restores the reference's state bit for bit either way."""
    import ctypes as C
    O = oracle
    rng = np.random.default_rng(5)
    og, pg = _grid(O, ocn, size, topo, z)
    ocn.set_math_mode(ocn.MATH_STRICT)
    init = {}
    for name, l in zip("uvw", LOCS):
        a = og.zeros(l)
        init[name] = rng.uniform(-1, 1, og.interior(a).shape)
    if topo == "PPF":
        init.pop("w")

    def build():
        m = ocn.NonhydrostaticModel(pg, advection=ocn.WENO())
        ocn.set(m, **init)
        return m

    dt = 0.01
    ref = build()
    for _ in range(3):
        ocn.time_step(ref, dt)
    ocn.flush_tendencies(ref)
    m = build()
    drv = ocn.RK3Driver(m, own_solver=own, defer_correction=defer)
    drv.time_step(dt)
    drv.flush()            # exercise the copy-home path between steps
    drv.time_step(dt)
    drv.time_step(dt)
    drv.flush()
    ocn.sync_device()
    for a, b in zip(ref.velocities + (ref.pNHS,), m.velocities + (m.pNHS,)):
        np.testing.assert_array_equal(from_dev(a), from_dev(b))
    Gd = drv.tendency_pointers()
    for G, ptr in zip(ref.timestepper.Gn, Gd):
        import torch
        got = torch.as_tensor(ocn.distributed._DevBuf(ptr, G.data.numel()), device="cuda").reshape(G.data.shape).cpu().numpy().T
        np.testing.assert_array_equal(og.interior_N(from_dev(G)), og.interior_N(got))
    del drv


@pytest.mark.parametrize("size,topo,z", [((32, 16, 12), "PPP", (0, 2.0)), ((16, 12, 9), "PPB", "stretched")])
def test_python_host_continues_after_a_c_driver(oracle, ocn, size, topo, z):
    """Two steps through the C driver, flush, then ocn.time_step(model, dt) on the same model: flush hands the driver's G^n to the model's
    time stepper and the clock has advanced, so the mixed run equals three steps of the Python host bit for bit (strict math) --
    velocities, pressure, G^n, clock."""
    O = oracle
    rng = np.random.default_rng(11)
    og, pg = _grid(O, ocn, size, topo, z)
    init = {name: rng.uniform(-1, 1, og.interior(og.zeros(l)).shape) for name, l in zip("uvw", LOCS)}
    dt = 0.01

    def build():
        m = ocn.NonhydrostaticModel(pg, advection=ocn.WENO(), math_mode=ocn.MATH_STRICT)
        ocn.set(m, **init)
        return m

    ref = build()
    for _ in range(3):
        ocn.time_step(ref, dt)
    ocn.flush_tendencies(ref)
    m = build()
    drv = ocn.RK3Driver(m)
    drv.time_step(dt)
    drv.time_step(dt)
    drv.flush()
    assert m.clock.iteration == 2 and abs(m.clock.time - 2 * dt) < 1e-15
    ocn.time_step(m, dt)
    ocn.flush_tendencies(m)
    ocn.sync_device()
    for a, b in zip(ref.velocities + (ref.pNHS,) + tuple(ref.timestepper.Gn), m.velocities + (m.pNHS,) + tuple(m.timestepper.Gn)):
        np.testing.assert_array_equal(og.interior_N(from_dev(a)), og.interior_N(from_dev(b)))
    assert m.clock.iteration == ref.clock.iteration == 3 and m.clock.time == ref.clock.time and m.clock.last_stage_dt == ref.clock.last_stage_dt
    del drv


def test_plain_c_host_of_the_c_abi_matches_the_python_host(ocn, tmp_path):
    """examples/c_abi_rk3.c (built by __graft_entry__.build()): a C99 program with no Python and no torch allocates through
    ocn_malloc, runs set! and three RK3 time_step!s through ocn_rk3_driver_* and writes u, v, w.  The Python host driving the same
    library from the same initial condition must agree bit for bit (strict math): the C ABI is self-sufficient."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "c_abi_rk3")
    if not os.path.exists(exe):  # normally built by __graft_entry__.build(); gcc is in the image
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["gcc", "-std=c99", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_rk3.c"), "-o", exe,
                        "-L" + os.path.join(root, "oceananigans.jl_amd", "lib"), "-locn_hip", "-lm",
                        "-Wl,-rpath,$ORIGIN/../../oceananigans.jl_amd/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    N, steps, dt = 24, 3, 2e-3
    out = tmp_path / "uvw.bin"
    r = subprocess.run([exe, str(N), str(steps), repr(dt), "strict", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.float64).reshape(3, N, N, N)           # [field][k][j][i]
    # the same 64-bit LCG as the C program, i fastest, for u, v, w
    state, vals = 0x9E3779B97F4A7C15, np.empty(3 * N ** 3)
    for q in range(vals.size):
        state = (state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        vals[q] = (state >> 11) / 9007199254740992.0 * 2.0 - 1.0
    init = vals.reshape(3, N, N, N)
    ocn.set_math_mode(ocn.MATH_STRICT)
    g = ocn.RectilinearGrid(ocn.GPU(), size=(N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi),
                            topology=("Periodic", "Periodic", "Periodic"), halo=(3, 3, 3))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    ocn.set(m, u=init[0].T, v=init[1].T, w=init[2].T)                      # set takes [i, j, k]
    for _ in range(steps):
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m)
    ocn.sync_device()
    for f, a, name in zip(m.velocities, got, "uvw"):
        np.testing.assert_array_equal(f.interior().T, a, err_msg=name)      # interior() is [i, j, k]
    assert "max|div u|" in r.stdout
