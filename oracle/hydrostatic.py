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
advection.  Not covered: split-explicit / implicit free surfaces, upwinding vector-invariant variants, z-star coordinates,
vertically implicit diffusion, immersed boundaries, forcing.  HIP counterpart: oceananigans.jl_amd/hydrostatic.py
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


class HydrostaticFreeSurfaceModel(O.NonhydrostaticModel):
    """HydrostaticFreeSurfaceModel(; grid, momentum_advection, tracer_advection, free_surface = ExplicitFreeSurface(g), coriolis,
    closure, buoyancy, tracers) with the QuasiAdamsBashforth2 time stepper.  Reuses the nonhydrostatic oracle's fields and
    operators; w is diagnostic, η is the extra prognostic field (stored as a (sx, sy) array, the plane k = Nz+1 of the
    reference's reduced field)."""

    def __init__(self, grid, tracers=(), momentum_advection="Centered2", tracer_advection=None, coriolis_f=None, closure=None,
                 buoyancy=None, boundary_conditions=None, gravitational_acceleration=g_Earth):
        assert grid.topo[2] == O.BOUNDED and grid.topo[0] == O.PERIODIC and grid.topo[1] == O.PERIODIC
        self.eta = np.zeros((grid.Nx + 2 * grid.Hx, grid.Ny + 2 * grid.Hy), order="F")
        self.g_eta = np.zeros_like(self.eta)
        self.g_eta_m = np.zeros_like(self.eta)
        self.gravity = float(gravitational_acceleration)
        self.tracer_scheme = None
        self.vector_invariant = momentum_advection == "VectorInvariant"
        if self.vector_invariant and tracer_advection is None:
            tracer_advection = "Centered2"  # the reference's default tracer scheme
        super().__init__(grid, tracers=tracers, timestepper="QuasiAdamsBashforth2",
                         advection=tracer_advection if self.vector_invariant else momentum_advection, coriolis_f=coriolis_f,
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
    def time_step(self, dt, euler=False):
        g = self.grid
        if self.iteration == 0:
            self.update_state(compute_tendencies=True)
        euler = euler or (dt != self.last_dt)
        chi = -0.5 if euler else self.chi
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
