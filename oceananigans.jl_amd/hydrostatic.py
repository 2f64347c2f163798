"""First slice of the reference's HydrostaticFreeSurfaceModel on the HIP backend: ExplicitFreeSurface,
flux-form momentum advection, QuasiAdamsBashforth2, static (Periodic, Periodic, Bounded) RectilinearGrid, one GPU.

    src/Models/HydrostaticFreeSurfaceModels/hydrostatic_free_surface_model.jl (constructor), explicit_free_surface.jl,
    compute_w_from_continuity.jl, hydrostatic_free_surface_tendency_kernel_functions.jl:29-97,
    hydrostatic_free_surface_ab2_step.jl:9-111, update_hydrostatic_free_surface_model_state.jl:35-96

The horizontal momentum and tracer tendencies are the terms of the NonhydrostaticModel path (flux-form advection, FPlane,
∂x pHY′, ScalarDiffusivity, flux boundary conditions) plus the barotropic pressure gradient g ∇η; w is diagnosed from continuity;
there is no pressure solve.  momentum_advection = VectorInvariant() (the reference's default: enstrophy-conserving vorticity flux,
energy-conserving vertical advection and kinetic-energy gradient) or a flux-form scheme.  NOT in this slice (each raises):
SplitExplicitFreeSurface / ImplicitFreeSurface, upwinding / WENO vector-invariant variants, z-star coordinates, vertically implicit diffusion, eddy-viscosity closures,
a tracer advection scheme different from the momentum one, Distributed architectures.
"""
import ctypes as C

import torch

from . import _lib
from .architectures import stream_ptr
from .fields import fill_halo_regions
from .grids import Bounded, Periodic
from .models import NonhydrostaticModel, compute_boundary_tendency_contributions, update_hydrostatic_pressure
from .physics import AnisotropicMinimumDissipation, Centered

g_Earth = 9.80665  # Oceananigans.BuoyancyFormulations.g_Earth


class VectorInvariant:
    """VectorInvariant() with its defaults (Advection/vector_invariant_advection.jl:104-130): EnstrophyConserving vorticity scheme,
    EnergyConserving vertical scheme and kinetic-energy gradient, second order."""
    buffer = 1

    def __repr__(self):
        return "VectorInvariant()"


class ExplicitFreeSurface:
    """ExplicitFreeSurface(; gravitational_acceleration = g_Earth) (explicit_free_surface.jl:9-21)"""

    def __init__(self, gravitational_acceleration=g_Earth):
        self.gravitational_acceleration = float(gravitational_acceleration)


class ImplicitFreeSurface:
    """ImplicitFreeSurface(; solver_method = :Default, gravitational_acceleration = g_Earth) (implicit_free_surface.jl:79-80): the
    reference's default free surface on an xy-regular RectilinearGrid (hydrostatic_free_surface_model.jl:51-52).  Only the
    :FastFourierTransform solver (its default there, FFTImplicitFreeSurfaceSolver) is implemented:
        (∇² - 1 / (g Lz Δt²)) ηⁿ⁺¹ = (∇ʰ·Q★ - ηⁿ / Δt) / (g Lz Δt),    u -= g Δt ∂x ηⁿ⁺¹."""

    def __init__(self, solver_method="Default", gravitational_acceleration=g_Earth):
        if str(solver_method).lstrip(":") not in ("Default", "FastFourierTransform"):
            raise NotImplementedError("ImplicitFreeSurface: only the FastFourierTransform solver is implemented")
        self.gravitational_acceleration = float(gravitational_acceleration)


def averaging_shape_function(tau, p=2, q=4, r=0.18927):
    """Shchepetkin & McWilliams (2005) averaging kernel (split_explicit_free_surface.jl:191-194)"""
    tau0 = (p + 2) * (p + q + 2) / (p + 1) / (p + q + 1)
    return (tau / tau0) ** p * (1 - (tau / tau0) ** q) - r * (tau / tau0)


def weights_from_substeps(substeps, averaging_kernel=averaging_shape_function):
    """weights_from_substeps (split_explicit_free_surface.jl:250-263): (fractional step size, normalised averaging weights truncated at
    Julia's searchsortedlast(weights, 0, rev=true)).  Julia's range(0.0, 2.0, length = N+1) is a TwicePrecision StepRangeLen whose
    elements are the correctly rounded 2k/N, so the nodes are built from exact rationals here (np.linspace can differ in the last
    bit).  Parity of the weights with a live Julia run is unpinned."""
    from fractions import Fraction

    import numpy as np
    tau = [float(Fraction(2 * k, substeps)) for k in range(substeps + 1)]
    w = np.array([averaging_kernel(t) for t in tau[1:]])
    lo, hi = 0, len(w) + 1
    while lo < hi - 1:
        mid = (lo + hi) >> 1
        if w[mid - 1] < 0:
            hi = mid
        else:
            lo = mid
    w = w[:lo]
    return float(tau[1] - tau[0]), w / w.sum()


MINIMUM_SUBSTEPS = 5  # step_split_explicit_free_surface.jl:51


class ForwardBackwardScheme:
    """ForwardBackwardScheme() (split_explicit_timesteppers.jl:13-17): η = f(U) then U = f(η)"""


class AdamsBashforth3Scheme:
    """AdamsBashforth3Scheme(; β = 0.281105, α = 1.5 + β, θ = -0.5 - 2β, γ = 0.088, δ = 0.614, ϵ = 0.013, μ = 1 - δ - γ - ϵ)
    (split_explicit_timesteppers.jl:69-70): η = f(U, Uᵐ⁻¹, Uᵐ⁻²) then U = f(η, ηᵐ, ηᵐ⁻¹, ηᵐ⁻²)."""

    def __init__(self, β=0.281105, α=None, θ=None, γ=0.088, δ=0.614, ϵ=0.013, μ=None):
        self.β = float(β)
        self.α = 1.5 + self.β if α is None else float(α)
        self.θ = -0.5 - 2 * self.β if θ is None else float(θ)
        self.γ, self.δ, self.ϵ = float(γ), float(δ), float(ϵ)
        self.μ = 1 - self.δ - self.γ - self.ϵ if μ is None else float(μ)

    def c_array(self):
        return (C.c_double * 7)(self.α, self.θ, self.β, self.δ, self.μ, self.γ, self.ϵ)


class SplitExplicitFreeSurface:
    """SplitExplicitFreeSurface(grid = nothing; substeps, cfl, fixed_Δt, gravitational_acceleration = g_Earth, averaging_kernel,
    timestepper = ForwardBackwardScheme()) (split_explicit_free_surface.jl:120-155).

      * substeps = N                      -> FixedSubstepNumber (:158-162)
      * neither substeps nor cfl          -> MINIMUM_SUBSTEPS substeps (the reference's disambiguation method, :178-179)
      * cfl (needs `grid`), no fixed_Δt   -> FixedTimeStepSize (:165-168, :217-235): Δt_barotropic = cfl Δs / sqrt(g Lz), the number of
                                             substeps max(MINIMUM_SUBSTEPS, ceil(2 Δt / Δt_barotropic)) and the weights are recomputed from
                                             the baroclinic Δt at every step (step_split_explicit_free_surface.jl:54-58)
      * cfl and fixed_Δt (needs `grid`)   -> FixedSubstepNumber with ceil(2 fixed_Δt / Δt_barotropic) substeps (:171-176)"""

    def __init__(self, grid=None, substeps=None, cfl=None, fixed_Δt=None, gravitational_acceleration=g_Earth,
                 averaging_kernel=averaging_shape_function, timestepper=None):
        import math
        self.gravitational_acceleration = float(gravitational_acceleration)
        self.averaging_kernel = averaging_kernel
        self.timestepper = ForwardBackwardScheme() if timestepper is None else timestepper
        if not isinstance(self.timestepper, (ForwardBackwardScheme, AdamsBashforth3Scheme)):
            raise TypeError("timestepper must be ForwardBackwardScheme() or AdamsBashforth3Scheme()")
        self.Δt_barotropic = None
        if cfl is not None:
            if substeps is not None:
                raise ValueError("SplitExplicitFreeSurface: give either substeps or cfl, not both")
            if grid is None:
                raise ValueError("The grid is a required positional argument to SplitExplicitFreeSurface when cfl is specified")
            inv2 = 1.0 / grid.dx ** 2 + 1.0 / grid.dy ** 2
            ds = math.sqrt(1.0 / inv2)
            self.Δt_barotropic = float(cfl) * ds / math.sqrt(self.gravitational_acceleration * grid.Lz)
            if fixed_Δt is not None:
                substeps = math.ceil(2 * float(fixed_Δt) / self.Δt_barotropic)
                self.Δt_barotropic = None
        elif substeps is None:
            substeps = MINIMUM_SUBSTEPS
        if self.Δt_barotropic is None:
            self.fractional_step_size, self.averaging_weights = weights_from_substeps(int(substeps), averaging_kernel)

    def settings(self, dt):
        """calculate_substeps / calculate_adaptive_settings (step_split_explicit_free_surface.jl:54-58): (fractional Δτ, weights)"""
        if self.Δt_barotropic is None:
            return self.fractional_step_size, self.averaging_weights
        import math
        n = max(MINIMUM_SUBSTEPS, math.ceil(2 * float(dt) / self.Δt_barotropic))
        return weights_from_substeps(n, self.averaging_kernel)


class HydrostaticFreeSurfaceModel:
    def __init__(self, grid, momentum_advection=None, tracer_advection=None, tracers=(), free_surface=None, coriolis=None,
                 closure=None, buoyancy=None, boundary_conditions=None, fused=None, timestepper="QuasiAdamsBashforth2", math_mode=None):
        """fused (default: True with VectorInvariant() momentum): one QAB2 step = one pass for the horizontal momentum (tendency, AB2
        step, barotropic forcing and mode), one launch per WENO / UpwindBiased tracer (tendency + AB2 step), the temporally blocked
        substep loop, one pass for the barotropic corrector + w, one halo launch, the hydrostatic pressure; the tendency evaluation
        that closes the reference's time_step! is deferred into the next step's fused launches (`flush_tendencies` completes it).
        fused = False keeps the reference's launch sequence.  Both are bit-identical in strict math."""
        from .grids import FullyConnected
        timestepper = str(timestepper).lstrip(":")
        if timestepper not in ("QuasiAdamsBashforth2", "SplitRungeKutta3"):
            raise ValueError(f"timestepper must be :QuasiAdamsBashforth2 or :SplitRungeKutta3, got {timestepper!r}")
        self.split_rk3 = timestepper == "SplitRungeKutta3"
        if self.split_rk3:
            # (the reference itself warns that this time stepper is experimental, split_hydrostatic_runge_kutta_3.jl:57-58)
            if not isinstance(free_surface, SplitExplicitFreeSurface):
                raise NotImplementedError("SplitRungeKutta3: with a SplitExplicitFreeSurface")
            if fused:
                raise NotImplementedError("SplitRungeKutta3 runs the reference's launch sequence (fused = False)")
            fused = False
        arch = grid.architecture
        self._dist = arch if (hasattr(arch, "partition") and arch.communicates) else None   # a slab-x rank (distributed.py)
        if tuple(grid.topology) != (FullyConnected if self._dist is not None else Periodic, Periodic, Bounded):
            raise NotImplementedError("HydrostaticFreeSurfaceModel: (Periodic, Periodic, Bounded) grids (x may be slab-partitioned)")
        if free_surface is None:
            free_surface = ImplicitFreeSurface()  # default_free_surface(grid::XYRegularRG) (hydrostatic_free_surface_model.jl:51-52)
        if not isinstance(free_surface, (ExplicitFreeSurface, SplitExplicitFreeSurface, ImplicitFreeSurface)):
            raise NotImplementedError("free_surface must be ExplicitFreeSurface(...), SplitExplicitFreeSurface(...) or ImplicitFreeSurface()")
        self.split = isinstance(free_surface, SplitExplicitFreeSurface)
        self.implicit = isinstance(free_surface, ImplicitFreeSurface)
        if momentum_advection is None:
            momentum_advection = VectorInvariant()  # the reference's default (hydrostatic_free_surface_model.jl)
        self.vector_invariant = isinstance(momentum_advection, VectorInvariant)
        if self.vector_invariant:
            # the container model below carries the tracer scheme (reference default: Centered()); momentum goes its own way
            container_advection = tracer_advection if tracer_advection is not None else Centered()
        else:
            if tracer_advection is not None and type(tracer_advection) is not type(momentum_advection):
                raise NotImplementedError("with a flux-form momentum scheme, tracer_advection must be the same scheme in this slice")
            container_advection = momentum_advection
        if isinstance(closure, AnisotropicMinimumDissipation):
            raise NotImplementedError("eddy-viscosity closures are not part of this slice")
        # fields, physics descriptors, tendency storage and the Adams-Bashforth bookkeeping of the nonhydrostatic model are reused;
        # its pressure solver and w tendency are simply not used
        self._nh = NonhydrostaticModel(grid, advection=container_advection, tracers=tracers, timestepper="QuasiAdamsBashforth2",
                                       closure=closure, buoyancy=buoyancy, coriolis=coriolis, boundary_conditions=boundary_conditions,
                                       pressure_solver=None, math_mode=math_mode)
        nh = self._nh
        grid = nh.grid  # (halo-inflated / math-mode-pinned by the container model)
        self.grid, self.architecture, self.clock = grid, grid.architecture, nh.clock
        self.free_surface = free_surface
        self.u, self.v, self.w = nh.u, nh.v, nh.w
        self.velocities, self.tracers, self.tracer_names = nh.velocities, nh.tracers, nh.tracer_names
        sx, sy = grid.Nx + 2 * grid.Hx, grid.Ny + 2 * grid.Hy
        dev = self.u.data.device
        self.eta = torch.zeros((sy, sx), dtype=torch.float64, device=dev)       # η[i, j, Nz+1], halos included, x fastest
        self._Geta = torch.zeros_like(self.eta)
        self._Geta_m = torch.zeros_like(self.eta)
        if self.split:  # barotropic velocities, filtered state, slow forcing: planes like η (interiors used)
            self.U, self.V, self._Ub, self._Vb, self._etab, self._GU, self._GV = (torch.zeros_like(self.eta) for _ in range(7))
            self._weights_key, self._weights, self._frac = None, None, None
            self._initialized = False
        self._adv_only = _lib.CModelTerms()                                     # the advective part alone, by scheme
        self._adv_only.advection = nh._terms.advection
        self.fused = (self.vector_invariant and not self.implicit) if fused is None else bool(fused)
        if self.implicit:
            if self.fused or self._dist is not None:
                raise NotImplementedError("ImplicitFreeSurface: the reference's launch sequence on one GPU (fused = False)")
            from .grids import Flat, RectilinearGrid
            from .solvers import FFTBasedPoissonSolver
            # FFTImplicitFreeSurfaceSolver (fft_based_implicit_free_surface_solver.jl:39-70): a Poisson solver on the horizontal grid
            hgrid = RectilinearGrid(grid.architecture, size=(grid.Nx, grid.Ny), x=grid._interval[0],
                                    y=grid._interval[1], topology=(Periodic, Periodic, Flat), halo=(grid.Hx, grid.Hy))
            self._fs_solver = FFTBasedPoissonSolver(hgrid)
            self._Qu, self._Qv = torch.zeros_like(self.eta), torch.zeros_like(self.eta)
            self._fs_rhs = torch.zeros((grid.Ny, grid.Nx), dtype=torch.float64, device=dev)
        if self.fused and not self.vector_invariant:
            raise NotImplementedError("fused = True needs momentum_advection = VectorInvariant()")
        self._tracer_fusable = not isinstance(container_advection, Centered)    # the tiled WENO / UpwindBiased tracer kernel has the epilogue
        self._alt = None                                                        # second storage of u, v and the tracers
        self._tendencies_current = False                                        # Gⁿ holds the tendencies of the current state
        if self._dist is not None and not (self.split and self.fused):
            raise NotImplementedError("a slab-partitioned HydrostaticFreeSurfaceModel needs the fused step with a SplitExplicitFreeSurface "
                                      "(DistributedSplitExplicitFreeSurface, distributed_split_explicit_free_surface.jl)")
        if self._dist is not None and self.split and free_surface.Δt_barotropic is not None:
            # materialize_free_surface (split_explicit_free_surface.jl:166-173): FixedTimeStepSize on a connected topology is an error
            # (every rank would derive its own substep count, and with it the width of the exchanged barotropic halos)
            raise ValueError("A variable substepping through a CFL condition is not supported for the `SplitExplicitFreeSurface` on a "
                             "partitioned grid. Provide a fixed number of substeps through the `substeps` keyword argument as: "
                             "`free_surface = SplitExplicitFreeSurface(grid; substeps = N)` where `N::Int`")
        if self._dist is not None and self.split and len(free_surface.averaging_weights) > grid.Nx:
            raise ValueError(f"SplitExplicitFreeSurface on a partitioned grid: the {len(free_surface.averaging_weights)} barotropic substeps "
                             f"need halos wider than the local x extent {grid.Nx} (use fewer ranks or fewer substeps)")
        if self.split and self.fused:
            self._Us, self._Vs = torch.zeros_like(self.eta), torch.zeros_like(self.eta)   # Σ Δz u*, Σ Δz v*
            self._work = torch.zeros((3,) + tuple(self.eta.shape), dtype=torch.float64, device=dev)
            self._dist_buffers = None
        self.update_state(compute_tendencies=False)

    # ---- helpers -------------------------------------------------------------------------------------------------------
    def field(self, name):
        return self._nh.field(name)

    def eta_interior(self):
        g = self.grid
        return self.eta[g.Hy:g.Hy + g.Ny, g.Hx:g.Hx + g.Nx]

    def _fill_eta_halos(self):
        _lib.call("ocn_fill_free_surface_halos", self.grid.cref, self.eta.data_ptr(), stream_ptr())

    # ---- update_state! (update_hydrostatic_free_surface_model_state.jl:35-53, 74-96) -------------------------------------
    def update_state(self, compute_tendencies=True):
        nh, g = self._nh, self.grid
        from .models import update_boundary_conditions
        update_boundary_conditions(nh)
        fill_halo_regions((self.u, self.v) + tuple(self.tracers), fill_boundary_normal_velocities=False)
        self._fill_eta_halos()
        _lib.call("ocn_compute_w_from_continuity", g.cref, self.u.ptr, self.v.ptr, self.w.ptr, stream_ptr())
        update_hydrostatic_pressure(nh)
        self._tendencies_current = False
        if compute_tendencies:
            self.compute_tendencies()

    # ---- compute_tendencies! (hydrostatic_free_surface_tendency_kernel_functions.jl:45-52, 86-93, 125-131) -----------------------
    def flush_tendencies(self):
        """Complete the tendency evaluation the fused time step defers (update_state!(model; compute_tendencies = true) at the end of
        the reference's time_step!): afterwards timestepper Gⁿ holds the tendencies of the current state."""
        if not self._tendencies_current:
            self.compute_tendencies()

    def compute_tendencies(self):
        nh, g, s = self._nh, self.grid, stream_ptr()
        self._tendencies_current = True
        Gn = nh.timestepper._Gn
        grav = self.free_surface.gravitational_acceleration
        # explicit_barotropic_pressure_x/y_gradient: g ∇η for the ExplicitFreeSurface, zero for the split-explicit one
        eta_ptr = None if (self.split or self.implicit) else self.eta.data_ptr()
        if self.vector_invariant:                                                         # - U_dot_∇u - g ∂x η, - U_dot_∇v - g ∂y η
            _lib.call("ocn_compute_vector_invariant_momentum_tendencies", g.cref, self.u.ptr, self.v.ptr, self.w.ptr, Gn[0].ptr,
                      Gn[1].ptr, eta_ptr, grav, s)
        else:
            _lib.call("ocn_compute_momentum_tendencies_terms", g.cref, C.byref(self._adv_only), self.u.ptr, self.v.ptr, self.w.ptr,
                      Gn[0].ptr, Gn[1].ptr, Gn[2].ptr, None, s)
            if not (self.split or self.implicit):
                _lib.call("ocn_add_barotropic_pressure_gradient", g.cref, grav, self.eta.data_ptr(), Gn[0].ptr, Gn[1].ptr, s)  # - g ∇η
        _lib.call("ocn_add_momentum_terms", g.cref, C.byref(nh._terms), self.u.ptr, self.v.ptr, self.w.ptr, Gn[0].ptr, Gn[1].ptr,
                  Gn[2].ptr, None, s)                                                               # - f x U - ∇pHY′ - ∂ⱼτᵢⱼ
        for n, c in enumerate(self.tracers):
            kappa = 0.0 if nh.closure is None else nh.closure.kappa_of(self.tracer_names[n])
            _lib.call("ocn_compute_tracer_tendency_terms", g.cref, C.byref(nh._terms), kappa, None, self.u.ptr, self.v.ptr, self.w.ptr,
                      c.ptr, Gn[3 + n].ptr, None, s)
        compute_boundary_tendency_contributions(nh)

    # ---- time_step! (quasi_adams_bashforth_2.jl:74-115 with ab2_step!(::HydrostaticFreeSurfaceModel)) -------------------------
    def _time_step_split_rk3(self, dt):
        """time_step!(model::AbstractModel{<:SplitRungeKutta3TimeStepper}, Δt) (split_hydrostatic_runge_kutta_3.jl:76-133) with
        split_rk3_substep!(::HydrostaticFreeSurfaceModel) (hydrostatic_free_surface_rk3_step.jl:7-28): see oracle/hydrostatic.py
        `_time_step_split_rk3` for the sequence.  The 3-D tendencies, the vertical integrals, the substepping, the corrector and
        update_state! are the library's kernels, and so are the stage combinations of the 3-D fields (ocn_split_rk3_substep); the combinations of
        the 2-D barotropic planes (η, U, V, the integrated tendencies) are elementwise IEEE operations in the reference's order."""
        nh, g, clock, s = self._nh, self.grid, self.clock, stream_ptr()
        if clock.iteration == 0:
            if not self._initialized:
                _lib.call("ocn_compute_barotropic_mode", g.cref, self.u.ptr, self.v.ptr, self.U.data_ptr(), self.V.data_ptr(), s)
                self._initialized = True
            self.update_state(compute_tendencies=True)
        elif not self._tendencies_current:
            self.compute_tendencies()
        stepped = [self.u, self.v] + list(self.tracers)
        gidx = [0, 1] + [3 + n for n in range(len(self.tracers))]
        psi = [f.data.clone() for f in stepped]                                    # cache_previous_fields! (halos included)
        psi_eta, psi_U, psi_V = self.eta.clone(), self.U.clone(), self.V.clone()
        ii, jj = slice(g.Hy, g.Hy + g.Ny), slice(g.Hx, g.Hx + g.Nx)               # planes are [y, x]
        kk = slice(g.Hz, g.Hz + g.Nz)
        GUi, GVi, GUm, GVm = (torch.zeros_like(self.eta) for _ in range(4))
        # (a / python_number multiplies by the rounded reciprocal on the device; a 0-dim device tensor divides)
        three, six = (torch.tensor(x, dtype=torch.float64, device=self.eta.device) for x in (3.0, 6.0))
        for stage, (gam, zet) in enumerate(((None, None), (1.0 / 4, 3.0 / 4), (2.0 / 3, 1.0 / 3)), 1):
            clock.stage = stage
            Gn = nh.timestepper._Gn
            # G_vertical_integral: the slow-forcing kernel with χ = -1/2 returns Σ Δz (1 Gⁿ - 0 G⁻)
            _lib.call("ocn_split_explicit_forcing", g.cref, Gn[0].ptr, Gn[0].ptr, Gn[1].ptr, Gn[1].ptr, -0.5, GUi.data_ptr(), GVi.data_ptr(), s)
            if stage == 1:
                self._GU.copy_(GUi); self._GV.copy_(GVi)
                GUm.copy_(GUi); GVm.copy_(GVi)
            elif stage == 2:
                self._GU.copy_(GUi); self._GV.copy_(GVi)
                GUm.copy_((GUi + GUm) / six); GVm.copy_((GVi + GVm) / six)
            else:
                self._GU.copy_(2 * GUi / three + GUm); self._GV.copy_(2 * GVi / three + GVm)
                self.U.copy_(psi_U); self.V.copy_(psi_V); self.eta.copy_(psi_eta)
            # split_rk3_substep_field! of u, v and the tracers in ONE launch: stage 1 is Uⁿ + Δt G (for the tracers Ψⁿ + Δt G·1: the same
            # bits), stages 2 and 3 ζ Ψⁿ + γ (Uᵐ + Δt Gᵐ)
            pa = _lib.ptr_array
            locs = _lib.i32_array([f.loc for f in stepped])
            Gs = pa([Gn[q].ptr for q in gidx])
            if stage == 1:
                _lib.call("ocn_rk3_substep", g.cref, len(stepped), pa([f.ptr for f in stepped]), Gs, Gs, locs, float(dt), 1.0, 0.0, 0, s)
            else:
                _lib.call("ocn_split_rk3_substep", g.cref, len(stepped), pa([f.ptr for f in stepped]), Gs, pa([P.data_ptr() for P in psi]), locs,
                          float(dt), float(gam), float(zet), s)
            self._substep_free_surface(dt, s)                                      # the complete substepping over Δt
            if stage == 2:
                self.U[ii, jj] = zet * psi_U[ii, jj] + gam * self.U[ii, jj]
                self.V[ii, jj] = zet * psi_V[ii, jj] + gam * self.V[ii, jj]
                self.eta[ii, jj] = zet * psi_eta[ii, jj] + gam * self.eta[ii, jj]
            _lib.call("ocn_barotropic_split_explicit_corrector", g.cref, self.u.ptr, self.v.ptr, self.U.data_ptr(), self.V.data_ptr(),
                      self._Ub.data_ptr(), self._Vb.data_ptr(), float(g.Lz), s)
            self.update_state(compute_tendencies=True)
        clock.stage = 1
        clock.time += dt
        clock.iteration += 1
        clock.last_dt = dt
        clock.last_stage_dt = dt

    def time_step(self, dt, euler=False):
        if self.split_rk3:
            return self._time_step_split_rk3(dt)
        if self.fused:
            return self._time_step_fused(dt, euler)
        nh, g, clock, s = self._nh, self.grid, self.clock, stream_ptr()
        if not self._tendencies_current and clock.iteration > 0:
            self.compute_tendencies()
        if clock.iteration == 0:
            if self.split and not self._initialized:  # initialize_free_surface! (the reference: run!(simulation) / first_time_step!)
                _lib.call("ocn_compute_barotropic_mode", g.cref, self.u.ptr, self.v.ptr, self.U.data_ptr(), self.V.data_ptr(), s)
                self._initialized = True
            self.update_state(compute_tendencies=True)
        euler = euler or (dt != clock.last_dt)
        chi = -0.5 if euler else nh.timestepper.chi
        Gn, Gm = nh.timestepper._Gn, nh.timestepper._Gm
        # local_ab2_step!: u, v by ab2_step_field!; tracers by _ab2_step_tracer_field!, which with σ = 1 (static grid) is the same
        # arithmetic for finite values
        idx = [0, 1] + [3 + n for n in range(len(self.tracers))]
        fields = [self.u, self.v] + list(self.tracers)
        _lib.call("ocn_ab2_step", g.cref, len(idx), _lib.ptr_array([f.ptr for f in fields]), _lib.ptr_array([Gn[q].ptr for q in idx]),
                  _lib.ptr_array([Gm[q].ptr for q in idx]), _lib.i32_array([f.loc for f in fields]), float(dt), float(chi), s)
        fs = self.free_surface
        if self.split:
            # compute_free_surface_tendency! (slow forcing from Gⁿ, G⁻ BEFORE they are cached), then step_free_surface!
            _lib.call("ocn_split_explicit_forcing", g.cref, Gn[0].ptr, Gm[0].ptr, Gn[1].ptr, Gm[1].ptr, float(chi), self._GU.data_ptr(),
                      self._GV.data_ptr(), s)
            self._substep_free_surface(dt, s)
            # pressure_correct_velocities!: the barotropic corrector
            _lib.call("ocn_barotropic_split_explicit_corrector", g.cref, self.u.ptr, self.v.ptr, self.U.data_ptr(), self.V.data_ptr(),
                      self._Ub.data_ptr(), self._Vb.data_ptr(), float(g.Lz), s)
        elif self.implicit:
            # step_free_surface!(::ImplicitFreeSurface) (implicit_free_surface.jl:112-145); the halo fills of the velocities and of the
            # volume fluxes are not needed: the kernels wrap their x / y neighbour indices
            _lib.call("ocn_implicit_free_surface_rhs", g.cref, self.u.ptr, self.v.ptr, self.eta.data_ptr(), fs.gravitational_acceleration,
                      float(dt), self._Qu.data_ptr(), self._Qv.data_ptr(), self._fs_rhs.data_ptr(), s)
            h = self._fs_solver._h
            _lib.call("ocn_poisson_set_source_term", h, self._fs_rhs.data_ptr(), s)
            _lib.call("ocn_poisson_solve_shifted", h, self.eta.data_ptr(), -1.0 / (fs.gravitational_acceleration * float(g.Lz) * float(dt) ** 2), s)
            self._fill_eta_halos()
            # pressure_correct_velocities!(model, ::ImplicitFreeSurface, Δt) (barotropic_pressure_correction.jl:21-47)
            _lib.call("ocn_barotropic_pressure_correction", g.cref, self.u.ptr, self.v.ptr, self.eta.data_ptr(), fs.gravitational_acceleration,
                      float(dt), s)
        else:
            # compute_free_surface_tendency! + step_free_surface!
            _lib.call("ocn_explicit_free_surface_ab2_step", g.cref, self.w.ptr, self.eta.data_ptr(), self._Geta.data_ptr(),
                      self._Geta_m.data_ptr(), float(dt), float(chi), s)
        clock.time += dt
        clock.iteration += 1
        clock.last_dt = dt
        clock.last_stage_dt = dt
        # (no pressure correction with an explicit free surface) cache_previous_tendencies!: role swap
        nh.timestepper._Gn, nh.timestepper._Gm = Gm, Gn
        self._Geta, self._Geta_m = self._Geta_m, self._Geta
        self.update_state(compute_tendencies=True)

    # ---- the same step with fused launches -----------------------------------------------------------------------------------
    def _substep_free_surface(self, dt, s):
        fs, g = self.free_surface, self.grid
        frac, w = fs.settings(dt)
        if self._weights_key is not w:
            self._weights_key, self._weights = w, (C.c_double * len(w))(*[float(x) for x in w])
        args = (g.cref, len(self._weights), self._weights, frac * float(dt), fs.gravitational_acceleration, float(g.Lz), self.eta.data_ptr(),
                self.U.data_ptr(), self.V.data_ptr(), self._etab.data_ptr(), self._Ub.data_ptr(), self._Vb.data_ptr(), self._GU.data_ptr(),
                self._GV.data_ptr())
        if isinstance(fs.timestepper, AdamsBashforth3Scheme):
            if self._dist is not None:
                raise NotImplementedError("AdamsBashforth3Scheme substepping on a slab-partitioned grid")
            if getattr(self, "_ab3_work", None) is None:
                self._ab3_work = torch.zeros((7,) + tuple(self.eta.shape), dtype=torch.float64, device=self.eta.device)
            _lib.call("ocn_split_explicit_substeps_ab3", *args[:6], fs.timestepper.c_array(), *args[6:], self._ab3_work.data_ptr(), s)
        elif self._dist is not None:
            # DistributedSplitExplicitFreeSurface: halos of η, U, V, Gᵁ, Gⱽ as wide as the number of substeps, ONE exchange per baroclinic
            # step, no communication while substepping (distributed_split_explicit_free_surface.jl; split_explicit_free_surface.jl:283-300)
            n = len(self._weights)
            if self._dist_buffers is None or self._dist_buffers[0] != n:
                dev = self.eta.device
                strip = 5 * n * g.Ny
                self._dist_buffers = (n, torch.zeros(11 * (g.Nx + 2 * n) * g.Ny, dtype=torch.float64, device=dev),
                                      [torch.zeros(strip, dtype=torch.float64, device=dev) for _ in range(4)])
            _, work, (sw, se, rw, re) = self._dist_buffers
            _lib.call("ocn_split_explicit_dist_begin", g.cref, n, self.eta.data_ptr(), self.U.data_ptr(), self.V.data_ptr(), self._GU.data_ptr(),
                      self._GV.data_ptr(), work.data_ptr(), sw.data_ptr(), se.data_ptr(), s)
            self._dist.exchange_strips(sw, se, rw, re)
            _lib.call("ocn_split_explicit_dist_run", g.cref, n, self._weights, frac * float(dt), fs.gravitational_acceleration, float(g.Lz),
                      self.eta.data_ptr(), self.U.data_ptr(), self.V.data_ptr(), work.data_ptr(), rw.data_ptr(), re.data_ptr(), s)
        elif self.fused:
            _lib.call("ocn_split_explicit_substeps_blocked", *args, self._work.data_ptr(), s)
        else:
            _lib.call("ocn_split_explicit_substeps", *args, s)

    def _time_step_fused(self, dt, euler=False):
        """time_step!(model, Δt) (quasi_adams_bashforth_2.jl:74-115 with ab2_step!(::HydrostaticFreeSurfaceModel),
        hydrostatic_free_surface_ab2_step.jl:9-26) re-cut at the launch boundaries that the data flow allows; every field ends up with
        the bits of the reference sequence (strict math).  State on entry and exit: halos filled, w and pHY′ consistent with u, v, T, S."""
        nh, g, clock, s = self._nh, self.grid, self.clock, stream_ptr()
        if clock.iteration == 0 and self.split and not self._initialized:
            _lib.call("ocn_compute_barotropic_mode", g.cref, self.u.ptr, self.v.ptr, self.U.data_ptr(), self.V.data_ptr(), s)
            self._initialized = True
        euler = bool(euler or (dt != clock.last_dt))
        chi = -0.5 if euler else nh.timestepper.chi
        Gn, Gm = nh.timestepper._Gn, nh.timestepper._Gm
        from .models import update_boundary_conditions
        update_boundary_conditions(nh)   # (the deferred tendencies belong to update_state! at the current clock time)
        prog = [self.u, self.v] + list(self.tracers)
        if self._alt is None:
            self._alt = [torch.zeros_like(f.data) for f in prog]
        alt = self._alt
        t = C.byref(nh._terms)
        from .models import _bcs_ref
        fs = self.free_surface
        grav = fs.gravitational_acceleration
        # compute_tendencies! (of the state the previous step left) + ab2_step_velocities! + compute_free_surface_tendency!
        sp = (self._GU.data_ptr(), self._GV.data_ptr(), self._Us.data_ptr(), self._Vs.data_ptr()) if self.split else (None,) * 4
        _lib.call("ocn_hydrostatic_momentum_ab2_step", g.cref, t, _bcs_ref(self.u, g), _bcs_ref(self.v, g), self.u.ptr, self.v.ptr, self.w.ptr,
                  Gn[0].ptr, Gn[1].ptr, Gm[0].ptr, Gm[1].ptr, alt[0].data_ptr(), alt[1].data_ptr(), float(dt), float(chi), int(euler),
                  None if self.split else self.eta.data_ptr(), grav, *sp, s)
        # tracer tendencies + ab2_step_tracers!
        gamma, zeta = 1.5 + chi, -(0.5 + chi)
        kappas = [0.0 if nh.closure is None else nh.closure.kappa_of(name) for name in self.tracer_names]
        if self._tracer_fusable and self.tracers:
            from .models import fused_tracer_launches
            fused_tracer_launches(g, t, self.u, self.v, self.w, self.tracers, kappas, [None] * len(self.tracers), Gn[3:], Gm[3:], alt[2:], dt,
                                  gamma, zeta, 0 if euler else 1, None, s)
        else:
            for n, c in enumerate(self.tracers):
                _lib.call("ocn_compute_tracer_tendency_terms", g.cref, t, kappas[n], None, self.u.ptr, self.v.ptr, self.w.ptr, c.ptr,
                          Gn[3 + n].ptr, None, s)
        if self.tracers and not self._tracer_fusable:
            self._apply_tracer_flux_bcs()
            idx = [3 + n for n in range(len(self.tracers))]
            _lib.call("ocn_ab2_step", g.cref, len(idx), _lib.ptr_array([c.ptr for c in self.tracers]), _lib.ptr_array([Gn[q].ptr for q in idx]),
                      _lib.ptr_array([Gm[q].ptr for q in idx]), _lib.i32_array([c.loc for c in self.tracers]), float(dt), float(chi), s)
        # step_free_surface!, then pressure_correct_velocities! (the barotropic corrector) + compute_w_from_continuity!
        if self.split:
            self._substep_free_surface(dt, s)
            _lib.call("ocn_barotropic_corrector_and_w", g.cref, alt[0].data_ptr(), alt[1].data_ptr(), self.u.ptr, self.v.ptr, self.w.ptr,
                      self.U.data_ptr(), self.V.data_ptr(), self._Us.data_ptr(), self._Vs.data_ptr(), float(g.Lz), s)
        else:
            _lib.call("ocn_explicit_free_surface_ab2_step", g.cref, self.w.ptr, self.eta.data_ptr(), self._Geta.data_ptr(),
                      self._Geta_m.data_ptr(), float(dt), float(chi), s)
            _lib.call("ocn_barotropic_corrector_and_w", g.cref, alt[0].data_ptr(), alt[1].data_ptr(), self.u.ptr, self.v.ptr, self.w.ptr,
                      None, None, None, None, float(g.Lz), s)
            self._Geta, self._Geta_m = self._Geta_m, self._Geta
        if self._tracer_fusable:
            for n, c in enumerate(self.tracers):
                c.data, alt[2 + n] = alt[2 + n], c.data
            nh._refresh_term_pointers()
        clock.time += dt
        clock.iteration += 1
        clock.last_dt = dt
        clock.last_stage_dt = dt
        nh.timestepper._Gn, nh.timestepper._Gm = Gm, Gn                     # cache_previous_tendencies!: role swap
        # update_state!(model; compute_tendencies = false): halos (w's halo columns are periodic images of the interior ones computed
        # above), η halos, hydrostatic pressure; the tendencies follow in the next step's fused launches
        update_boundary_conditions(nh)   # Value / Gradient conditions of the halo fill at the new clock time
        if self._dist is not None:
            # slab-x rank: the x halos of u, v, T, S come from the neighbours; w (whose edge column needed the neighbour's corrected
            # u) is then recomputed from continuity on every column, halos included, as the reference's update_state! does
            fill_halo_regions((self.u, self.v) + tuple(self.tracers), fill_boundary_normal_velocities=False)
            _lib.call("ocn_compute_w_from_continuity", g.cref, self.u.ptr, self.v.ptr, self.w.ptr, s)
        else:
            fill_halo_regions((self.u, self.v, self.w) + tuple(self.tracers), fill_boundary_normal_velocities=False)
        self._fill_eta_halos()
        update_hydrostatic_pressure(nh)
        self._tendencies_current = False

    def _apply_tracer_flux_bcs(self):
        """compute_boundary_tendency_contributions! for the tracers alone (u, v get theirs inside the fused momentum launch)"""
        nh, g = self._nh, self.grid
        tr = [c for c in self.tracers if c.boundary_conditions is not None and c.boundary_conditions.has_flux()]
        if not tr:
            return
        Gn = nh.timestepper._Gn
        Gs = [Gn[3 + self.tracers.index(c)] for c in tr]
        arr = (C.POINTER(_lib.CFieldBcs) * len(tr))(*[C.pointer(c.boundary_conditions.c_struct(g)) for c in tr])
        _lib.call("ocn_apply_flux_bcs", g.cref, _lib.ptr_array([G.ptr for G in Gs]), _lib.ptr_array([c.ptr for c in tr]),
                  _lib.i32_array([c.loc for c in tr]), arr, len(tr), stream_ptr())

    def set(self, **kwargs):
        g = self.grid
        for name, value in kwargs.items():
            if name in ("eta", "η"):
                v = torch.as_tensor(value, dtype=torch.float64)
                self.eta_interior().copy_(v.T.to(self.eta.device) if v.ndim == 2 else v)
            elif name == "w":
                raise ValueError("w is diagnostic in a HydrostaticFreeSurfaceModel")
            else:
                self.field(name).set(value)
        self.update_state(compute_tendencies=False)
