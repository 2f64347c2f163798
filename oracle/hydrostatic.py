"""TEST INFRASTRUCTURE (CPU restatement, never imported by the product): first slice of SURVEY §8(f) rank 4, the
HydrostaticFreeSurfaceModel of the reference with an ExplicitFreeSurface, VectorInvariant() (its default) or flux-form momentum
advection (Centered(order=2) / WENO / UpwindBiased passed as `momentum_advection`), QuasiAdamsBashforth2, on a static (Periodic, Periodic, Bounded)
RectilinearGrid.  Built on the operators of oracle.py (same C kernels as the nonhydrostatic oracle), so only what is new is
restated here:

  * compute_w_from_continuity!            src/Models/HydrostaticFreeSurfaceModels/compute_w_from_continuity.jl:22-47
  * free-surface tendency and AB2 step     explicit_free_surface.jl:36-40, 84-96, 112-160
  * u, v tendencies                        hydrostatic_free_surface_tendency_kernel_functions.jl:29-56, 70-97
  * tracer AB2 step (σ = 1 on a static grid)   hydrostatic_free_surface_ab2_step.jl:95-111
  * ab2_step! / update_state! / time_step! order   hydrostatic_free_surface_ab2_step.jl:9-26,
        update_hydrostatic_free_surface_model_state.jl:35-53, 74-96, TimeSteppers/quasi_adams_bashforth_2.jl:74-115

PARITY UNPINNED: the reference's own tests of this model are time-stepping smoke tests (test_hydrostatic_free_surface_models.jl);
tests/test_oracle_hydrostatic.py checks identities, a linear free-surface wave and the order of accuracy of the vector-invariant
advection; the split-explicit free surface (ForwardBackwardScheme, `split_explicit_substeps`; SplitExplicitFreeSurfaces/*.jl) is
pinned by the reference's own solver tests re-expressed.  Not covered: AdamsBashforth3Scheme substepping, SplitRungeKutta3,
implicit free surfaces, upwinding vector-invariant variants, z-star coordinates, vertically implicit diffusion, immersed
boundaries, forcing.  HIP counterpart: oceananigans.jl_amd/hydrostatic.py
(tests/test_gpu_hydrostatic.py compares the two bit for bit).
"""
import numpy as np

from . import oracle as O

g_Earth = 9.80665  # Oceananigans.BuoyancyFormulations.g_Earth


def compute_w_from_continuity(g, u, v, w):
    """w[i, j, 1] = 0;  w[i, j, k] = w[i, j, k-1] - flux_div_xyᶜᶜᶜ(i, j, k-1, u, v) / Azᶜᶜᶜ   (:31-40; ∂t_σ = 0 on a static grid).
    Columns: every (i, j) of the parent whose east / north neighbours exist (a superset of w_kernel_parameters, :43-51)."""
    Hz, Nz = g.Hz, g.Nz
    dzc = np.full(Nz, g.dz) if g.dzc is None else np.asarray(g.dzc[Hz:Hz + Nz])
    Az = g.dx * g.dy
    sx, sy = u.shape[0], u.shape[1]
    w[:, :, Hz] = 0.0
    for k in range(1, Nz + 1):  # 0-based source cell k-1 -> face k
        kc = Hz + k - 1
        Ax = g.dy * dzc[k - 1]
        Ay = g.dx * dzc[k - 1]
        dxu = np.zeros((sx, sy))
        dyv = np.zeros((sx, sy))
        dxu[:-1, :] = Ax * u[1:, :, kc] - Ax * u[:-1, :, kc]          # δxᶜᵃᵃ(Ax_qᶠᶜᶜ, u)
        dyv[:, :-1] = Ay * v[:, 1:, kc] - Ay * v[:, :-1, kc]          # δyᵃᶜᵃ(Ay_qᶜᶠᶜ, v)
        dh = (dxu + dyv) / Az
        w[:-1, :-1, Hz + k] = w[:-1, :-1, Hz + k - 1] - (dh[:-1, :-1] + 0.0)  # (the last parent row / column has no neighbour)


def vector_invariant_momentum_tendencies(g, u, v, w, Gu, Gv):
    """Gu = -U_dot_∇u, Gv = -U_dot_∇v with the reference's default VectorInvariant() scheme: EnstrophyConserving vorticity flux,
    EnergyConserving vertical advection and kinetic-energy gradient (Advection/vector_invariant_advection.jl:269-275, 304-319,
    360-361; Operators/vorticity_operators.jl:4-11; interpolation ℑ = 0.5 (a + b), differences a - b, derivatives δ / Δ)."""
    Hx, Hy, Hz, Nx, Ny, Nz = g.Hx, g.Hy, g.Hz, g.Nx, g.Ny, g.Nz
    dx, dy = g.dx, g.dy
    Az = dx * dy
    dzf = np.full(Nz + 2, g.dz) if g.dzf is None else np.asarray(g.dzf[Hz - 0:Hz + Nz + 2])  # Δzᶠ at faces k = 1 .. Nz+2 (1 .. Nz+1 used)

    def sh(a, di=0, dj=0, dk=0):
        return a[Hx + di:Hx + Nx + di, Hy + dj:Hy + Ny + dj, Hz + dk:Hz + Nz + dk]

    def zeta(di, dj):  # ζ₃ᶠᶠᶜ at (i + di, j + dj)
        gam = (dy * sh(v, di, dj) - dy * sh(v, di - 1, dj)) - (dx * sh(u, di, dj) - dx * sh(u, di, dj - 1))
        return gam / Az

    # ---- U
    m = lambda di: 0.5 * (dx * sh(v, di, 0) + dx * sh(v, di, 1))                      # ℑyᵃᶜᵃ(Δx_qᶜᶠᶜ v) at (i + di, j)
    hadv_u = -(0.5 * (zeta(0, 0) + zeta(0, 1))) * (0.5 * (m(-1) + m(0))) / dx
    dzfk = lambda dk: dzf[dk:dk + Nz][None, None, :]                                    # Δzᶠ at face k + dk
    Zu = lambda dk: (0.5 * (Az * sh(w, -1, 0, dk) + Az * sh(w, 0, 0, dk))) * ((sh(u, 0, 0, dk) - sh(u, 0, 0, dk - 1)) / dzfk(dk))
    vadv_u = (0.5 * (Zu(0) + Zu(1))) / Az
    Kh = lambda di, dj: (0.5 * (sh(u, di, dj) * sh(u, di, dj) + sh(u, di + 1, dj) * sh(u, di + 1, dj))
                         + 0.5 * (sh(v, di, dj) * sh(v, di, dj) + sh(v, di, dj + 1) * sh(v, di, dj + 1))) / 2
    bern_u = (Kh(0, 0) - Kh(-1, 0)) / dx
    g.interior_N(Gu)[...] = -((hadv_u + vadv_u) + bern_u)
    # ---- V
    n = lambda dj: 0.5 * (dy * sh(u, 0, dj) + dy * sh(u, 1, dj))                      # ℑxᶜᵃᵃ(Δy_qᶠᶜᶜ u) at (i, j + dj)
    hadv_v = (0.5 * (zeta(0, 0) + zeta(1, 0))) * (0.5 * (n(-1) + n(0))) / dy
    Zv = lambda dk: (0.5 * (Az * sh(w, 0, -1, dk) + Az * sh(w, 0, 0, dk))) * ((sh(v, 0, 0, dk) - sh(v, 0, 0, dk - 1)) / dzfk(dk))
    vadv_v = (0.5 * (Zv(0) + Zv(1))) / Az
    bern_v = (Kh(0, 0) - Kh(0, -1)) / dy
    g.interior_N(Gv)[...] = -((hadv_v + vadv_v) + bern_v)


def averaging_shape_function(tau, p=2, q=4, r=0.18927):
    """Shchepetkin & McWilliams (2005) averaging kernel (split_explicit_free_surface.jl:191-194)"""
    tau0 = (p + 2) * (p + q + 2) / (p + 1) / (p + q + 1)
    return (tau / tau0) ** p * (1 - (tau / tau0) ** q) - r * (tau / tau0)


def weights_from_substeps(substeps, averaging_kernel=averaging_shape_function):
    """weights_from_substeps (split_explicit_free_surface.jl:228-241): fractional step size Δτ and the normalised averaging
    weights, truncated at Julia's searchsortedlast(weights, 0, rev=true)."""
    from fractions import Fraction
    # Julia's range(0.0, 2.0, length = N+1) is a TwicePrecision StepRangeLen: its elements are the correctly rounded 2k/N
    tau = np.array([float(Fraction(2 * k, substeps)) for k in range(substeps + 1)])
    dtau = tau[1] - tau[0]
    w = np.array([averaging_kernel(t) for t in tau[1:]])
    lo, hi = 0, len(w) + 1                      # Base.searchsortedlast with the Reverse ordering, on 1-based indices
    while lo < hi - 1:
        mid = (lo + hi) >> 1
        if w[mid - 1] < 0:                      # lt(Reverse, 0, v[m]) = isless(v[m], 0)
            hi = mid
        else:
            lo = mid
    w = w[:lo]
    w = w / w.sum()
    return float(dtau), w


def constant_averaging_kernel(tau):
    return 1.0


def iterate_split_explicit(eta, U, V, etab, Ub, Vb, GU, GV, dtau, weights, grav, H, dx, dy):
    """iterate_split_explicit! with the ForwardBackwardScheme on a (Periodic, Periodic) static grid of column depth H
    (step_split_explicit_free_surface.jl:3-46, 62-98): for every weight, η -= Δτ (δx(Δy U) + δy(Δx V)) / Az, then
    U += Δτ (-g H ∂x η + Gᵁ), V likewise, and the filtered state accumulates weight x (η, U, V).  In place, interior arrays
    [i, j]; the periodic wrap is the topology-aware δxTᶜᵃᵃ / ∂xTᶠᶜᶠ."""
    Az = dx * dy
    for wgt in weights:
        eta[...] = eta - dtau * ((dy * np.roll(U, -1, 0) - dy * U) + (dx * np.roll(V, -1, 1) - dx * V)) / Az
        Un = U + dtau * (-grav * H * ((eta - np.roll(eta, 1, 0)) / dx) + GU)
        Vn = V + dtau * (-grav * H * ((eta - np.roll(eta, 1, 1)) / dy) + GV)
        etab += wgt * eta
        Ub += wgt * Un
        Vb += wgt * Vn
        U[...] = Un
        V[...] = Vn


AB3_DEFAULTS = dict(beta=0.281105, gamma=0.088, delta=0.614, epsilon=0.013)  # AdamsBashforth3Scheme() (split_explicit_timesteppers.jl:69-70)


def ab3_coefficients(beta=0.281105, alpha=None, theta=None, gamma=0.088, delta=0.614, epsilon=0.013, mu=None):
    """AdamsBashforth3Scheme(; β, α = 1.5 + β, θ = -0.5 - 2β, γ, δ, ϵ, μ = 1 - δ - γ - ϵ) in Julia's evaluation order"""
    alpha = 1.5 + beta if alpha is None else alpha
    theta = -0.5 - 2 * beta if theta is None else theta
    mu = 1 - delta - gamma - epsilon if mu is None else mu
    return dict(beta=beta, alpha=alpha, theta=theta, gamma=gamma, delta=delta, epsilon=epsilon, mu=mu)


def iterate_split_explicit_ab3(eta, U, V, etab, Ub, Vb, GU, GV, dtau, weights, grav, H, dx, dy, c):
    """iterate_split_explicit! with the AdamsBashforth3Scheme (split_explicit_timesteppers.jl:19-159; step_split_explicit_free_surface.jl:
    3-46): η -= Δτ (δx(Δy U★) + δy(Δx V★)) / Az with U★ = α Uᵐ + θ Uᵐ⁻¹ + β Uᵐ⁻²; U += Δτ (-g H ∂x η★ + Gᵁ) with
    η★ = δ ηᵐ⁺¹ + μ ηᵐ + γ ηᵐ⁻¹ + ϵ ηᵐ⁻².  The history is re-initialised from the current state at the start of every baroclinic step
    (initialize_free_surface_timestepper!, :116-127)."""
    Az = dx * dy
    Um1, Um2, Vm1, Vm2 = U.copy(), U.copy(), V.copy(), V.copy()
    em, em1, em2 = eta.copy(), eta.copy(), eta.copy()
    a, th, be, de, mu, ga, ep = c["alpha"], c["theta"], c["beta"], c["delta"], c["mu"], c["gamma"], c["epsilon"]
    for wgt in weights:
        # _split_explicit_free_surface!: cache_previous_free_surface!, then the update with U★, V★
        em2, em1, em = em1, em, eta.copy()
        Us = a * U + th * Um1 + be * Um2
        Vs = a * V + th * Vm1 + be * Vm2
        eta[...] = eta - dtau * ((dy * np.roll(Us, -1, 0) - dy * Us) + (dx * np.roll(Vs, -1, 1) - dx * Vs)) / Az
        # _split_explicit_barotropic_velocity!: cache_previous_velocities!, then the update with η★
        Um2, Um1 = Um1, U.copy()
        Vm2, Vm1 = Vm1, V.copy()
        es = de * eta + mu * em + ga * em1 + ep * em2
        Un = U + dtau * (-grav * H * ((es - np.roll(es, 1, 0)) / dx) + GU)
        Vn = V + dtau * (-grav * H * ((es - np.roll(es, 1, 1)) / dy) + GV)
        etab += wgt * eta
        Ub += wgt * Un
        Vb += wgt * Vn
        U[...] = Un
        V[...] = Vn


class HydrostaticFreeSurfaceModel(O.NonhydrostaticModel):
    """HydrostaticFreeSurfaceModel(; grid, momentum_advection, tracer_advection, free_surface = ExplicitFreeSurface(g), coriolis,
    closure, buoyancy, tracers) with the QuasiAdamsBashforth2 time stepper.  Reuses the nonhydrostatic oracle's fields and
    operators; w is diagnostic, η is the extra prognostic field (stored as a (sx, sy) array, the plane k = Nz+1 of the
    reference's reduced field)."""

    def __init__(self, grid, tracers=(), momentum_advection="Centered2", tracer_advection=None, coriolis_f=None, closure=None, coriolis_beta=None,
                 buoyancy=None, boundary_conditions=None, gravitational_acceleration=g_Earth, split_explicit_substeps=None,
                 split_explicit_timestepper="ForwardBackward", implicit_free_surface=False, timestepper="QuasiAdamsBashforth2"):
        """split_explicit_substeps = N: free_surface = SplitExplicitFreeSurface(substeps = N) with the ForwardBackwardScheme
        (split_explicit_free_surface.jl:60-97); None: ExplicitFreeSurface.  timestepper = "SplitRungeKutta3":
        SplitRungeKutta3TimeStepper (split_hydrostatic_runge_kutta_3.jl, hydrostatic_free_surface_rk3_step.jl) with the split-explicit
        free surface."""
        assert timestepper in ("QuasiAdamsBashforth2", "SplitRungeKutta3")
        self.split_rk3 = timestepper == "SplitRungeKutta3"
        assert not self.split_rk3 or split_explicit_substeps is not None, "SplitRungeKutta3 here: SplitExplicitFreeSurface only"
        assert grid.topo[2] == O.BOUNDED and grid.topo[0] == O.PERIODIC and grid.topo[1] == O.PERIODIC
        self.implicit = bool(implicit_free_surface)   # ImplicitFreeSurface(solver_method = :FastFourierTransform)
        assert not (self.implicit and split_explicit_substeps is not None)
        self.split = split_explicit_substeps
        self.ab3 = ab3_coefficients() if split_explicit_timestepper in ("AdamsBashforth3", "AB3") else None
        if self.split is not None:
            assert grid.dzc is None or True
            self.frac_dt, self.weights = weights_from_substeps(int(self.split))
            shp = (grid.Nx, grid.Ny)
            self.U, self.V = np.zeros(shp), np.zeros(shp)          # barotropic velocities (transports), interior only
            self.Ub, self.Vb, self.etab = np.zeros(shp), np.zeros(shp), np.zeros(shp)   # filtered state
            self.GU, self.GV = np.zeros(shp), np.zeros(shp)
            self.initialized = False
        self.eta = np.zeros((grid.Nx + 2 * grid.Hx, grid.Ny + 2 * grid.Hy), order="F")
        self.g_eta = np.zeros_like(self.eta)
        self.g_eta_m = np.zeros_like(self.eta)
        self.gravity = float(gravitational_acceleration)
        self.tracer_scheme = None
        self.vector_invariant = momentum_advection == "VectorInvariant"
        if self.vector_invariant and tracer_advection is None:
            tracer_advection = "Centered2"  # the reference's default tracer scheme
        super().__init__(grid, tracers=tracers, timestepper="QuasiAdamsBashforth2",
                         advection=tracer_advection if self.vector_invariant else momentum_advection, coriolis_f=coriolis_f, coriolis_beta=coriolis_beta,
                         closure=closure, buoyancy=buoyancy, boundary_conditions=boundary_conditions)
        ta = momentum_advection if tracer_advection is None else tracer_advection
        self.tracer_scheme = {"WENO5": O.ADV_WENO5, "Centered2": O.ADV_CENTERED2, "UpwindBiased5": O.ADV_UPWIND5}[ta]
        self.solver = None  # no nonhydrostatic pressure
        self.update_state(compute_tendencies=False)

    # ---- update_state! (update_hydrostatic_free_surface_model_state.jl:35-53, 74-96) -------------------------------------
    def _fill_eta(self):
        g, e = self.grid, self.eta
        Hx, Hy, Nx, Ny = g.Hx, g.Hy, g.Nx, g.Ny
        e[:Hx, :] = e[Nx:Nx + Hx, :]
        e[Nx + Hx:, :] = e[Hx:2 * Hx, :]
        e[:, :Hy] = e[:, Ny:Ny + Hy]
        e[:, Ny + Hy:] = e[:, Hy:2 * Hy]

    def update_state(self, compute_tendencies=True):
        g = self.grid
        # prognostic fields: u, v, η, tracers (w is diagnostic)
        for f, l, n in zip(self.fields, self.locs, self.names):
            if n != "w":
                O.fill_halo_regions(g, f, l, fill_boundary_normal_velocities=False, bcs=self.bcs.get(n))
        self._fill_eta()
        # compute_auxiliaries!: w from continuity, hydrostatic pressure, diffusivities
        compute_w_from_continuity(g, self.u, self.v, self.w)
        if self.pHY is not None:
            T, S = self._buoyancy_tracers()
            O.update_hydrostatic_pressure(g, self.physics, T, S, self.pHY)
        assert self.amd is None, "eddy-viscosity closures are not part of this slice"
        if compute_tendencies:
            self.compute_tendencies()

    # ---- compute_tendencies! (compute_hydrostatic_free_surface_tendencies.jl) -----------------------------------------------
    def compute_tendencies(self):
        g, ph = self.grid, self.physics
        Gu, Gv, Gw = self.Gn[0], self.Gn[1], self.Gn[2]
        if getattr(self, "vector_invariant", False):
            vector_invariant_momentum_tendencies(g, self.u, self.v, self.w, Gu, Gv)  # -U_dot_∇u, -U_dot_∇v (VectorInvariant())
        else:
            O.momentum_tendencies(g, self.u, self.v, self.w, Gu, Gv, Gw, self.scheme)  # ... (flux form)
        # - explicit_barotropic_pressure_x/y_gradient = g ∂xᶠᶜᶜ η, g ∂yᶜᶠᶜ η (explicit_free_surface.jl:36-40), the same for every k
        e, Hx, Hy, Hz = self.eta, g.Hx, g.Hy, g.Hz
        px = np.zeros_like(e)
        py = np.zeros_like(e)
        if getattr(self, "split", None) is None and not getattr(self, "implicit", False):  # Split-explicit / implicit: explicit_barotropic_pressure_*_gradient = 0 (SplitExplicitFreeSurfaces.jl:46-47, implicit_free_surface.jl:106-107)
            px[1:, :] = self.gravity * ((e[1:, :] - e[:-1, :]) / g.dx)
            py[:, 1:] = self.gravity * ((e[:, 1:] - e[:, :-1]) / g.dy)
        ii, jj = slice(Hx, Hx + g.Nx), slice(Hy, Hy + g.Ny)
        Gu[ii, jj, Hz:Hz + g.Nz] -= px[ii, jj, None]
        Gv[ii, jj, Hz:Hz + g.Nz] -= py[ii, jj, None]
        if ph.c.coriolis or ph.c.closure or ph.c.buoyancy:                                # - f x U - ∇pHY′ - ∂ⱼτᵢⱼ, in that order
            T, S = self._buoyancy_tracers()
            O.momentum_extra_tendencies(g, ph, self.u, self.v, self.w, T, S, self.pHY, Gu, Gv, Gw)
        Gw[...] = 0.0  # there is no w tendency
        scheme = self.scheme if self.tracer_scheme is None else self.tracer_scheme
        for n, c in enumerate(self.tracers):
            O.tracer_tendency(g, self.u, self.v, self.w, c, self.Gn[3 + n], scheme)
            if ph.c.closure:
                O.tracer_diffusion(g, self.kappa[self.tracer_names[n]], c, self.Gn[3 + n])
        for f, l, n, G in zip(self.fields, self.locs, self.names, self.Gn):
            if n in self.bcs and n != "w":
                O.apply_flux_bcs(g, l, f, G, self.bcs[n])

    # ---- time_step! (quasi_adams_bashforth_2.jl:74-115 with ab2_step!(::HydrostaticFreeSurfaceModel), :9-26) -------------------
    # ---- SplitExplicitFreeSurface pieces ---------------------------------------------------------------------------------
    def _dz_centres(self):
        g = self.grid
        return np.full(g.Nz, g.dz) if g.dzc is None else np.asarray(g.dzc[g.Hz:g.Hz + g.Nz])

    def _barotropic_mode(self, f):
        """integrate_barotropic_mode! (barotropic_split_explicit_corrector.jl:13-32) on a static grid (σ = 1): Σₖ Δz u σ, k upwards"""
        g, dz = self.grid, self._dz_centres()
        fi = g.interior_N(f)
        out = dz[0] * fi[:, :, 0] * 1.0
        for k in range(1, g.Nz):
            out = out + dz[k] * fi[:, :, k] * 1.0
        return out

    def initialize(self):
        """initialize_free_surface! (initialize_split_explicit_substepping.jl:15-25), once, before the first step (the reference
        calls it from run!(simulation) / first_time_step!)"""
        if self.split is not None:
            self.U[...] = self._barotropic_mode(self.u)
            self.V[...] = self._barotropic_mode(self.v)
            self.initialized = True

    def _split_explicit_step(self, dt, chi):
        g, grav, H = self.grid, self.gravity, self.grid.Lz   # column depth of a static, flat-bottom grid
        dx, dy, Az = g.dx, g.dy, g.dx * g.dy
        dz = self._dz_centres()
        # compute_free_surface_tendency!: GU = Σₖ Δz ab2_step_G (compute_slow_tendencies.jl:12-32), then the filtered state is zeroed
        C1, C2 = 3 * 1.0 / 2 + chi, 1.0 / 2 + chi
        not_euler = 1.0 if C2 != 0 else 0.0
        for G, idx in ((self.GU, 0), (self.GV, 1)):
            gn, gm = g.interior_N(self.Gn[idx]), g.interior_N(self.Gm[idx])
            acc = dz[0] * (C1 * gn[:, :, 0] - C2 * gm[:, :, 0] * not_euler)
            for k in range(1, g.Nz):
                acc = acc + dz[k] * (C1 * gn[:, :, k] - C2 * gm[:, :, k] * not_euler)
            G[...] = acc
        self.etab[...] = 0.0
        self.Ub[...] = 0.0
        self.Vb[...] = 0.0
        return dx, dy, Az, grav, H

    def _substep(self, dt):
        """iterate_split_explicit! with the ForwardBackwardScheme (step_split_explicit_free_surface.jl:3-46, 62-98) and
        _update_split_explicit_state! (:100-108); periodic wrap = the topology-aware operators δxTᶜᵃᵃ, ∂xTᶠᶜᶠ"""
        g, grav, H = self.grid, self.gravity, self.grid.Lz
        dx, dy, Az = g.dx, g.dy, g.dx * g.dy
        Hx, Hy = g.Hx, g.Hy
        eta = self.eta[Hx:Hx + g.Nx, Hy:Hy + g.Ny].copy()
        if self.ab3 is not None:
            iterate_split_explicit_ab3(eta, self.U, self.V, self.etab, self.Ub, self.Vb, self.GU, self.GV, self.frac_dt * dt, self.weights,
                                       grav, H, dx, dy, self.ab3)
        else:
            iterate_split_explicit(eta, self.U, self.V, self.etab, self.Ub, self.Vb, self.GU, self.GV, self.frac_dt * dt, self.weights,
                                   grav, H, dx, dy)
        self.eta[Hx:Hx + g.Nx, Hy:Hy + g.Ny] = self.etab
        self.U[...] = self.Ub
        self.V[...] = self.Vb

    def _barotropic_corrector(self):
        """barotropic_split_explicit_corrector! (:44-71): U̅ <- Σ Δz u of the stepped velocities, u += (U - U̅) / H at every level"""
        g, H = self.grid, self.grid.Lz
        self.Ub[...] = self._barotropic_mode(self.u)
        self.Vb[...] = self._barotropic_mode(self.v)
        ui, vi = g.interior_N(self.u), g.interior_N(self.v)
        ui[...] = ui + ((self.U - self.Ub) / H)[:, :, None]
        vi[...] = vi + ((self.V - self.Vb) / H)[:, :, None]

    def _implicit_step(self, dt):
        """step_free_surface!(::ImplicitFreeSurface) with the FFTImplicitFreeSurfaceSolver (implicit_free_surface.jl:112-145,
        fft_based_implicit_free_surface_solver.jl:76-115): Qu = Σₖ Ax u★ (sum!: k ascending from 0), rhs = (δx Qu + δy Qv - Az η / Δt) /
        (g Lz Δt Az), then solve!(η, solver, rhs, m = -1 / (g Lz Δt²)): η̂ = -rhŝ / (λx + λy - m) on the horizontal (TX, TY, Flat) grid
        (fft_based_poisson_solver.jl:95-125; no zero-mode gauge when m != 0); then pressure_correct_velocities!: u -= g Δt ∂x η
        (barotropic_pressure_correction.jl:21-47)."""
        g, grav, Lz = self.grid, self.gravity, self.grid.Lz
        dz = self._dz_centres()
        ui, vi = g.interior_N(self.u), g.interior_N(self.v)
        Qu = np.zeros((g.Nx, g.Ny))
        Qv = np.zeros((g.Nx, g.Ny))
        for k in range(g.Nz):
            Qu = Qu + (g.dy * dz[k]) * ui[:, :, k]
            Qv = Qv + (g.dx * dz[k]) * vi[:, :, k]
        Az = g.dx * g.dy
        ii, jj = slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny)
        eta = self.eta[ii, jj]
        dQ = (np.roll(Qu, -1, 0) - Qu) + (np.roll(Qv, -1, 1) - Qv)
        rhs = (dQ - Az * eta / dt) / (grav * Lz * dt * Az)
        m = -1.0 / (grav * Lz * dt ** 2)
        lx = (2 * np.sin(np.arange(g.Nx) * np.pi / g.Nx) / (g.Lx / g.Nx)) ** 2     # poisson_eigenvalues.jl:8-31, Periodic
        ly = (2 * np.sin(np.arange(g.Ny) * np.pi / g.Ny) / (g.Ly / g.Ny)) ** 2
        from scipy import fft as sfft
        hat = sfft.fft2(rhs.astype(np.complex128))
        hat = -hat / ((lx[:, None] + ly[None, :]) + 0.0 - m)
        self.eta[ii, jj] = np.real(sfft.ifft2(hat))
        self._fill_eta()
        e = self.eta
        ui[...] = ui - (grav * dt * ((e[ii, jj] - e[g.Hx - 1:g.Hx + g.Nx - 1, jj]) / g.dx))[:, :, None]
        vi[...] = vi - (grav * dt * ((e[ii, jj] - e[ii, g.Hy - 1:g.Hy + g.Ny - 1]) / g.dy))[:, :, None]

    def _time_step_split_rk3(self, dt):
        """time_step!(model::AbstractModel{<:SplitRungeKutta3TimeStepper}, Δt) (split_hydrostatic_runge_kutta_3.jl:76-133) with
        split_rk3_substep!(::HydrostaticFreeSurfaceModel) (hydrostatic_free_surface_rk3_step.jl:7-28): per stage the integrated RK3
        tendencies (compute_slow_tendencies.jl:85-108), u, v, tracers by Uᵐ⁺¹ = ζ Uⁿ + γ (Uᵐ + Δt Gᵐ) (γ², γ³ = 1/4, 2/3; ζ², ζ³ = 3/4,
        1/3; stage 1: U + Δt G), the COMPLETE barotropic substepping over Δt (stage 3 restarts from the state at step n,
        initialize_split_explicit_substepping.jl:44-63), the stage-2 average of η, U, V (rk3_average_free_surface!), the barotropic
        corrector and update_state!(compute_tendencies = true); one tick of Δt at the end.  Static grid: σ = 1."""
        g = self.grid
        if self.iteration == 0:
            if not self.initialized:
                self.initialize()
            self.update_state(compute_tendencies=True)
        dz = self._dz_centres()
        stepped = [self.u, self.v] + list(self.tracers)
        gidx = [0, 1] + [3 + n for n in range(len(self.tracers))]
        psi = [f.copy() for f in stepped]                                         # cache_previous_fields!
        psi_eta, psi_U, psi_V = self.eta.copy(), self.U.copy(), self.V.copy()
        ii, jj = slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny)
        GUm, GVm = np.zeros_like(self.GU), np.zeros_like(self.GV)                   # timestepper.G⁻.U, G⁻.V
        for stage, (gam, zet) in enumerate(((None, None), (1.0 / 4, 3.0 / 4), (2.0 / 3, 1.0 / 3)), 1):
            # compute_free_surface_tendency!: G_vertical_integral, then the stage's combination; initialize_free_surface_state!
            integ = []
            for idx in (0, 1):
                gn = g.interior_N(self.Gn[idx])
                acc = dz[0] * gn[:, :, 0]
                for k in range(1, g.Nz):
                    acc = acc + dz[k] * gn[:, :, k]
                integ.append(acc)
            if stage == 1:
                self.GU[...], self.GV[...] = integ[0], integ[1]
                GUm[...], GVm[...] = self.GU, self.GV
            elif stage == 2:
                self.GU[...], self.GV[...] = integ[0], integ[1]
                GUm[...], GVm[...] = (self.GU + GUm) / 6, (self.GV + GVm) / 6
            else:
                self.GU[...], self.GV[...] = 2 * integ[0] / 3 + GUm, 2 * integ[1] / 3 + GVm
                self.U[...], self.V[...], self.eta[...] = psi_U, psi_V, psi_eta
            self.etab[...] = 0.0
            self.Ub[...] = 0.0
            self.Vb[...] = 0.0
            # rk3_substep_velocities!, rk3_substep_tracers!
            for q, (f, P) in enumerate(zip(stepped, psi)):
                fi, Pi, G = g.interior_N(f), g.interior_N(P), g.interior_N(self.Gn[gidx[q]])
                if stage == 1:
                    fi[...] = (fi + dt * G) if q < 2 else (Pi + dt * G * 1.0)
                else:
                    fi[...] = zet * Pi + (gam if q < 2 else gam * 1.0) * (fi + dt * G)
            self._substep(dt)                                                      # step_free_surface! over the whole Δt
            if stage == 2:                                                         # rk3_average_free_surface!
                self.U[...] = zet * psi_U + gam * self.U
                self.V[...] = zet * psi_V + gam * self.V
                self.eta[ii, jj] = zet * psi_eta[ii, jj] + gam * self.eta[ii, jj]
            self._barotropic_corrector()                                           # pressure_correct_velocities!
            self.update_state(compute_tendencies=True)
        self.time += dt
        self.iteration += 1
        self.last_dt = dt

    def time_step(self, dt, euler=False):
        g = self.grid
        if getattr(self, "split_rk3", False):
            return self._time_step_split_rk3(dt)
        if self.iteration == 0:
            if self.split is not None and not self.initialized:
                self.initialize()
            self.update_state(compute_tendencies=True)
        euler = euler or (dt != self.last_dt)
        chi = -0.5 if euler else self.chi
        if getattr(self, "implicit", False):
            for idx in (0, 1):                                 # local_ab2_step!
                O.ab2_step(g, self.locs[idx], self.fields[idx], self.Gn[idx], self.Gm[idx], dt, chi)
            alpha, beta = 1.5 + chi, 0.5 + chi
            for n, c in enumerate(self.tracers):
                ci, gn, gm = g.interior_N(c), g.interior_N(self.Gn[3 + n]), g.interior_N(self.Gm[3 + n])
                ci[...] = 1.0 * ci + dt * (alpha * 1.0 * gn - beta * 1.0 * gm)
            self._implicit_step(dt)                            # step_free_surface! + pressure_correct_velocities!
            self.time += dt
            self.iteration += 1
            self.last_dt = dt
            self.cache_previous_tendencies()
            self.update_state(compute_tendencies=True)
            return
        if self.split is not None:
            self._split_explicit_step(dt, chi)                 # compute_free_surface_tendency!
            for idx in (0, 1):                                 # local_ab2_step!
                O.ab2_step(g, self.locs[idx], self.fields[idx], self.Gn[idx], self.Gm[idx], dt, chi)
            alpha, beta = 1.5 + chi, 0.5 + chi
            for n, c in enumerate(self.tracers):
                ci, gn, gm = g.interior_N(c), g.interior_N(self.Gn[3 + n]), g.interior_N(self.Gm[3 + n])
                ci[...] = 1.0 * ci + dt * (alpha * 1.0 * gn - beta * 1.0 * gm)
            self._substep(dt)                                  # step_free_surface!
            self.time += dt
            self.iteration += 1
            self.last_dt = dt
            self._barotropic_corrector()                       # pressure_correct_velocities!
            self.cache_previous_tendencies()
            self.update_state(compute_tendencies=True)
            return
        # compute_free_surface_tendency!: Gη = w[i, j, Nz+1] (explicit_free_surface.jl:126-140)
        self.g_eta[...] = self.w[:, :, g.Hz + g.Nz]
        # local_ab2_step!: velocities with ab2_step_field!, tracers with _ab2_step_tracer_field! (σ = 1)
        for idx in (0, 1):
            O.ab2_step(g, self.locs[idx], self.fields[idx], self.Gn[idx], self.Gm[idx], dt, chi)
        alpha, beta = 1.5 + chi, 0.5 + chi
        for n, c in enumerate(self.tracers):
            Gn, Gm = self.Gn[3 + n], self.Gm[3 + n]
            ci, gn, gm = g.interior_N(c), g.interior_N(Gn), g.interior_N(Gm)
            ci[...] = 1.0 * ci + dt * (alpha * 1.0 * gn - beta * 1.0 * gm)
        # step_free_surface!: _explicit_ab2_step_free_surface! (explicit_free_surface.jl:84-96)
        not_euler = 0.0 if chi == -0.5 else 1.0
        Hx, Hy = g.Hx, g.Hy
        ii, jj = slice(Hx, Hx + g.Nx), slice(Hy, Hy + g.Ny)
        G = (1.5 + chi) * self.g_eta[ii, jj] - (0.5 + chi) * self.g_eta_m[ii, jj] * not_euler
        self.eta[ii, jj] += dt * G
        self.time += dt
        self.iteration += 1
        self.last_dt = dt
        # calculate_pressure_correction! / pressure_correct_velocities!: nothing for an explicit free surface
        self.cache_previous_tendencies()
        self.g_eta_m[...] = self.g_eta
        self.update_state(compute_tendencies=True)

    def set(self, **kw):
        g = self.grid
        for name, val in kw.items():
            if name == "eta":
                self.eta[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny] = val
            elif name in ("u", "v"):
                f, l = getattr(self, name), {"u": O.LOC_U, "v": O.LOC_V}[name]
                g.interior(f)[...] = val
            else:
                g.interior(self.tracers[self.tracer_names.index(name)])[...] = val
        self.update_state(compute_tendencies=False)
