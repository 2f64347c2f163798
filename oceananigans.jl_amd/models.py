"""NonhydrostaticModel and its time steppers.

Mirrors, call for call:
  NonhydrostaticModel(; grid, advection, tracers, timestepper)   src/Models/NonhydrostaticModels/nonhydrostatic_model.jl:114-239
  set!(model; u, v, w, ...)                                       .../set_nonhydrostatic_model.jl:33-60
  update_state!(model; compute_tendencies)                        .../update_nonhydrostatic_model_state.jl:20-57
  compute_tendencies!(model)                                      .../compute_nonhydrostatic_tendencies.jl:17-139
  calculate_pressure_correction!, pressure_correct_velocities!    .../pressure_correction.jl:8-50
  time_step!(model::RungeKutta3, Δt)                              src/TimeSteppers/runge_kutta_3.jl:77-151
  time_step!(model::QuasiAdamsBashforth2, Δt)                     src/TimeSteppers/quasi_adams_bashforth_2.jl:74-115
  cache_previous_tendencies!                                      src/TimeSteppers/store_tendencies.jl:12-22
Julia's `f!` names are spelled `f` here.  forcing, stokes_drift, background_fields are `nothing`; advection is WENO()
or Centered(); coriolis = FPlane, closure = ScalarDiffusivity, buoyancy = BuoyancyTracer / SeawaterBuoyancy and bottom /
top Flux / Value / Gradient boundary conditions are the SURVEY §8(f) rank-1 terms (physics.py); anything else raises.
"""
import math

import numpy as np
import os

import torch

from . import _lib
import ctypes as C

from .advection import WENO, UpwindBiased
from .physics import (AnisotropicMinimumDissipation, BetaPlane, BuoyancyTracer, Centered, FieldBoundaryConditions, FPlane, ScalarDiffusivity,
                      SeawaterBuoyancy)
from .architectures import stream_ptr
from .fields import CenterField, Field, XFaceField, YFaceField, ZFaceField, fill_halo_regions
from .grids import Bounded, Flat
from .solvers import nonhydrostatic_pressure_solver


class Clock:
    """src/TimeSteppers/clock.jl:16-22"""

    def __init__(self):
        self.time = 0.0
        self.last_dt = math.inf
        self.last_stage_dt = math.inf
        self.iteration = 0
        self.stage = 1


class _TendencyStore:
    """Gⁿ / G⁻ tuples of a time stepper.  Reading `Gn` first completes a deferred compute_tendencies! (flush_tendencies)."""

    def __init__(self, grid, prognostic_fields, model=None):
        self._Gn = [Field(f.loc, grid) for f in prognostic_fields]
        self._Gm = [Field(f.loc, grid) for f in prognostic_fields]
        self._model = model

    @property
    def Gn(self):
        if self._model is not None:
            flush_tendencies(self._model)
        return self._Gn

    @property
    def Gm(self):
        return self._Gm


class RungeKutta3TimeStepper(_TendencyStore):
    """src/TimeSteppers/runge_kutta_3.jl:10-63; γ, ζ each rounded once to Float64."""

    def __init__(self, grid, prognostic_fields, model=None):
        super().__init__(grid, prognostic_fields, model)
        self.g1, self.g2, self.g3 = 8 / 15, 5 / 12, 3 / 4
        self.z2, self.z3 = -17 / 60, -5 / 12


class QuasiAdamsBashforth2TimeStepper(_TendencyStore):
    """src/TimeSteppers/quasi_adams_bashforth_2.jl:3-60; χ = 0.1 by default."""

    def __init__(self, grid, prognostic_fields, chi=0.1, model=None):
        super().__init__(grid, prognostic_fields, model)
        self.chi = chi


class NonhydrostaticModel:
    def __init__(self, grid, advection=None, tracers=(), timestepper="RungeKutta3", closure=None, buoyancy=None,
                 coriolis=None, forcing=None, stokes_drift=None, boundary_conditions=None,
                 hydrostatic_pressure_anomaly="default", pressure_solver="default", math_mode=None):
        """math_mode: None keeps the grid's (default: the process default, ocn.set_math_mode); MATH_STRICT / MATH_FAST pin this
        model's arithmetic variant whatever other models of the process use (ocn_grid.math)."""
        for name, val in (("forcing", forcing), ("stokes_drift", stokes_drift)):
            if val is not None:
                raise NotImplementedError(f"{name} != nothing is outside the MI355X hot-path scope (see DESIGN.md)")
        if advection is None:
            advection = Centered()  # the reference default (nonhydrostatic_model.jl:117)
        if not isinstance(advection, (WENO, Centered, UpwindBiased)):
            raise NotImplementedError("advection must be WENO(), UpwindBiased(order=5) or Centered()")
        if coriolis is not None and not isinstance(coriolis, (FPlane, BetaPlane)):
            raise NotImplementedError("only coriolis = FPlane(...) or BetaPlane(...) is implemented")
        if isinstance(coriolis, BetaPlane) and grid.topology[1] == Flat:
            raise NotImplementedError("BetaPlane needs a non-Flat y")
        if closure is not None and not isinstance(closure, (ScalarDiffusivity, AnisotropicMinimumDissipation)):
            raise NotImplementedError("only closure = ScalarDiffusivity(...) or AnisotropicMinimumDissipation(...) is implemented")
        if buoyancy is not None and not isinstance(buoyancy, (BuoyancyTracer, SeawaterBuoyancy)):
            raise NotImplementedError("only buoyancy = BuoyancyTracer() or SeawaterBuoyancy(...) is implemented")
        if isinstance(tracers, str):
            tracers = (tracers,)
        tracers = tuple(tracers)
        if buoyancy is not None:  # validate_buoyancy (BuoyancyFormulations/buoyancy_force.jl:66-76)
            for req in buoyancy.required_tracers:
                if req not in tracers:
                    raise ValueError(f"{type(buoyancy).__name__} requires tracers {buoyancy.required_tracers}, got {tracers}")
        # inflate_grid_halo_size (nonhydrostatic_model.jl:183, 243-257; test_nonhydrostatic_models.jl:34-66): the model rebuilds the
        # grid with the halo its advection scheme and closure need, never with a smaller one than the user gave.
        # adapt_advection_order (a lower-order scheme in a direction with too few cells) is not supported: that raises.
        required = max(advection.buffer, 1 if closure is not None else 0, 1)
        H = (grid.Hx, grid.Hy, grid.Hz)
        for d, (N, t) in enumerate(zip(grid.size, grid.topology)):
            if t != Flat and N < advection.buffer:
                raise NotImplementedError(f"{advection!r} needs size >= {advection.buffer} in dimension {d + 1} (got N={N}); "
                                          "adapt_advection_order is not implemented")
        if any(t != Flat and h < required for h, t in zip(H, grid.topology)):
            grid = grid.with_halo(tuple(max(h, required) for h in H))
        if math_mode is not None:
            grid = grid.with_math_mode(math_mode)
        self.grid = grid
        self.architecture = grid.architecture
        self.advection = advection
        self.coriolis, self.closure, self.buoyancy = coriolis, closure, buoyancy
        self.clock = Clock()
        bcs = dict(boundary_conditions or {})
        # boundary conditions of the diffusivity fields: {"νₑ": FieldBoundaryConditions, "κₑ": {tracer: FieldBoundaryConditions}}
        # (build_diffusivity_fields, anisotropic_minimum_dissipation.jl:333-341); ASCII aliases nu_e / kappa_e
        nu_bcs = bcs.pop("νₑ", bcs.pop("nu_e", None))
        kappa_bcs = dict(bcs.pop("κₑ", bcs.pop("kappa_e", None)) or {})
        if (nu_bcs is not None or kappa_bcs) and not isinstance(closure, AnisotropicMinimumDissipation):
            raise ValueError("νₑ / κₑ boundary conditions need closure = AnisotropicMinimumDissipation()")
        for name in bcs:
            if name not in ("u", "v", "w") + tracers:
                raise ValueError(f"boundary conditions given for unknown field {name!r}")
            if not isinstance(bcs[name], FieldBoundaryConditions):
                raise TypeError("boundary_conditions values must be FieldBoundaryConditions")
        # validate_boundary_condition_topology (boundary_condition.jl:128-136) + the impenetrable wall-normal component
        normal = {"u": ("west", "east"), "v": ("south", "north"), "w": ("bottom", "top")}
        # (a rank-local grid answers with the topology of the GLOBAL grid: the slabs of a Bounded x are RightConnected / FullyConnected /
        #  LeftConnected, every rank is given the same conditions and the slab that holds the wall applies them)
        gtopo = getattr(grid, "global_topology", None) or grid.topology
        for name, b in bcs.items():
            for d, pair in enumerate((("west", "east"), ("south", "north"), ("bottom", "top"))):
                for side in pair:
                    if b.sides[side] is None:
                        continue
                    if gtopo[d] != "Bounded":
                        raise ValueError(f"Cannot set {side} boundary condition of {name} in a `{gtopo[d]}` direction!")
                    is_open = b.sides[side].kind == _lib.BC_OPEN
                    if side in normal.get(name, ()) and not is_open:
                        raise NotImplementedError(f"{name} keeps its impenetrable {side} boundary condition, or takes OpenBoundaryCondition(value) "
                                                  "(its value on the boundary face)")
                    if is_open and side not in normal.get(name, ()):
                        raise ValueError(f"OpenBoundaryCondition on the {side} boundary applies to the velocity component normal to it, not to {name}")
        self.u, self.v, self.w = XFaceField(grid, bcs.get("u")), YFaceField(grid, bcs.get("v")), ZFaceField(grid, bcs.get("w"))
        self.velocities = (self.u, self.v, self.w)
        self.tracer_names = tracers
        self.tracers = tuple(CenterField(grid, bcs.get(n)) for n in self.tracer_names)
        self.pNHS = CenterField(grid)
        # nonhydrostatic_model.jl:143-158: pHY′ is a separate CenterField iff buoyancy is not nothing
        self.pHY = None
        if buoyancy is not None and hydrostatic_pressure_anomaly == "default":
            self.pHY = CenterField(grid)
        elif hydrostatic_pressure_anomaly not in ("default", None):
            raise ValueError("hydrostatic_pressure_anomaly must be 'default' or None")
        # build_diffusivity_fields (anisotropic_minimum_dissipation.jl:333-341): νₑ and one κₑ per tracer, default conditions
        self.diffusivity_fields = None
        if isinstance(closure, AnisotropicMinimumDissipation):
            if grid.topology[2] == Flat:
                raise NotImplementedError("AnisotropicMinimumDissipation needs a non-Flat z")
            self.diffusivity_fields = {"nu_e": CenterField(grid, nu_bcs),
                                       "kappa_e": tuple(CenterField(grid, kappa_bcs.get(n)) for n in tracers)}
        # (pressure_solver=None: no solver is built -- the HydrostaticFreeSurfaceModel reuses this class as its field container)
        self.pressure_solver = nonhydrostatic_pressure_solver(grid) if pressure_solver == "default" else pressure_solver
        prog = self.prognostic_fields()
        if timestepper in ("RungeKutta3", ":RungeKutta3"):
            self.timestepper = RungeKutta3TimeStepper(grid, prog, model=self)
        elif timestepper in ("QuasiAdamsBashforth2", ":QuasiAdamsBashforth2"):
            self.timestepper = QuasiAdamsBashforth2TimeStepper(grid, prog, model=self)
        else:
            raise ValueError(f"unknown timestepper {timestepper!r}")
        self._tuple_cache = {}
        self.copy_cached_tendencies = False
        # anything beyond plain WENO advection goes through the general (unfused) tendency entry points
        self._has_user_bcs = any(not b.is_default() for b in list(bcs.values()) + list(kappa_bcs.values()) + ([nu_bcs] if nu_bcs else []))
        self._has_flux_bcs = any(b.has_flux() for b in bcs.values())
        self.general_terms = (isinstance(advection, (Centered, UpwindBiased)) or coriolis is not None or closure is not None
                              or buoyancy is not None or self._has_user_bcs)
        if self._has_user_bcs and hasattr(grid.architecture, "partition") and any(
                s is not None and (s.values is not None or s.func is not None) for b in bcs.values() for s in b.sides.values()):
            raise NotImplementedError("array / function boundary conditions on a Distributed architecture are not implemented")
        self._terms = self._make_terms()
        # fused stage boundaries: tendencies (+ boundary fluxes) + the next substep in as few launches as possible.  Plain
        # WENO momentum uses the tiled kernel's epilogue; tracers and the §8(f) terms use the general fused entry points
        # (WENO advection only: the Centered(order=2) tracer kernel has no epilogue)
        self._general_fused = self.general_terms or bool(self.tracers)
        # grids with a Bounded or Flat x / y run the direction-generic kernels (csrc/general.hip) with the reference's launch sequence
        xy_periodic = grid.topology[0] in ("Periodic", "FullyConnected") and grid.topology[1] == "Periodic"
        # ... and since round 4 the stage boundaries of grids with walls / Flat directions in x, y fuse too on one GPU (the substep as the
        # epilogue of the tiled kernels on the interior box, one more per-cell kernel on the wall frames: csrc/general.hip) -- unless a flux
        # goes through an x / y wall (ocn_apply_flux_bcs adds those on the unfused path) or a bottom / top flux is an array on a grid
        # with x walls  [OCN_FUSE_WALLS=0: the reference's launch sequence]
        def _fusable_conditions(b):
            if b is None:
                return True
            for side, v in b.sides.items():
                if v is None or v.kind != _lib.BC_FLUX:
                    continue
                if side in ("west", "east", "south", "north"):
                    return False
                if (v.values is not None or v.func is not None) and grid.topology[0] != "Periodic":
                    return False
            return True
        walls_fusable = (not xy_periodic and not hasattr(grid.architecture, "partition") and os.environ.get("OCN_FUSE_WALLS", "1") != "0"
                         and isinstance(advection, (WENO, UpwindBiased)) and isinstance(self.timestepper, RungeKutta3TimeStepper)
                         and all(_fusable_conditions(b) for b in bcs.values()))
        self.fuse_stage_boundaries = (xy_periodic or walls_fusable) and ((not self._general_fused) or (
            isinstance(advection, (WENO, UpwindBiased)) and os.environ.get("OCN_FUSE_GENERAL", "1") != "0"))
        if not xy_periodic:
            if hasattr(grid.architecture, "partition") and not (
                    grid.topology[0] in ("Periodic", "FullyConnected", "RightConnected", "LeftConnected", "Bounded")
                    and grid.topology[1] == "Bounded" and grid.topology[2] == "Bounded"):
                # (distributed_fft_based_poisson_solver.jl:62-66: a Bounded y needs a Bounded z, a Bounded x a Bounded y)
                raise NotImplementedError("a partitioned x needs (Periodic, Periodic, *), (Periodic, Bounded, Bounded) or (Bounded, Bounded, Bounded)")
        self._alt_velocities = None
        self._alt_fields = None
        # defer the last compute_tendencies! of a step and fuse it with the first substep of the next one
        self.defer_final_tendencies = self.fuse_stage_boundaries
        # fold the pressure correction of stages 1 and 2 into the loads of the fused launch (all-periodic grids)
        self.correct_on_load = (self.fuse_stage_boundaries and not self._general_fused and all(t == "Periodic" for t in grid.topology)
                                and grid.Nx >= 16 and grid.Ny >= 8 and grid.Nz >= 4
                                and not hasattr(grid.architecture, "partition")
                                and os.environ.get("OCN_CORRECT_ON_LOAD", "1") != "0")
        self._pending_tendencies = False
        # slab-x ranks: the same folding with the pressure planes of the neighbours exchanged after the solve (distributed.py)
        sup = getattr(grid.architecture, "correct_on_load_supported", None)
        self.dist_correct_on_load = bool(sup is not None and sup(self))
        update_state(self, compute_tendencies=False)

    def _make_terms(self):
        """struct ocn_model_terms for the C ABI (pointers never change after construction)."""
        t = _lib.CModelTerms()
        t.advection = (_lib.ADVECTION_CENTERED2 if isinstance(self.advection, Centered)
                       else _lib.ADVECTION_UPWIND5 if isinstance(self.advection, UpwindBiased) else _lib.ADVECTION_WENO5)
        if self.coriolis is not None:
            if isinstance(self.coriolis, BetaPlane):
                # f = f₀ + β ynode (beta_plane.jl:43-57): the y nodes of Centers / Faces with halos, element 0 <-> j = 1 - Hy
                from .architectures import on_architecture
                g = self.grid
                self._ynodes = tuple(on_architecture(g.architecture, np.ascontiguousarray(g.nodes_1d(1, face, with_halos=True)))
                                     for face in (False, True))
                t.coriolis, t.f, t.coriolis_beta = 2, self.coriolis.f0, self.coriolis.beta
                t.yc, t.yf = self._ynodes[0].data_ptr(), self._ynodes[1].data_ptr()
            else:
                t.coriolis, t.f = 1, self.coriolis.f
        if isinstance(self.closure, AnisotropicMinimumDissipation):
            t.closure, t.nu_e = 2, self.diffusivity_fields["nu_e"].ptr
        elif self.closure is not None:
            t.closure, t.nu = 1, self.closure.nu
        b = self.buoyancy
        if isinstance(b, BuoyancyTracer):
            t.buoyancy = _lib.BUOYANCY_TRACER
            t.T = self.field("b").ptr
        elif isinstance(b, SeawaterBuoyancy):
            t.g = b.gravitational_acceleration
            t.alpha, t.beta = b.equation_of_state.thermal_expansion, b.equation_of_state.haline_contraction
            if b.constant_salinity is not None:
                t.buoyancy, t.T = _lib.BUOYANCY_SEAWATER_T, self.field("T").ptr
            elif b.constant_temperature is not None:
                t.buoyancy, t.S = _lib.BUOYANCY_SEAWATER_S, self.field("S").ptr
            else:
                t.buoyancy, t.T, t.S = _lib.BUOYANCY_SEAWATER_TS, self.field("T").ptr, self.field("S").ptr
        if self.pHY is not None:
            t.pHY = self.pHY.ptr
        return t

    def _refresh_term_pointers(self):
        """the buoyancy tracers' storage may have been swapped by a fused stage boundary"""
        t, b = self._terms, self.buoyancy
        if isinstance(b, BuoyancyTracer):
            t.T = self.field("b").ptr
        elif isinstance(b, SeawaterBuoyancy):
            if b.constant_temperature is None:
                t.T = self.field("T").ptr
            if b.constant_salinity is None:
                t.S = self.field("S").ptr

    def prognostic_fields(self):
        return self.velocities + self.tracers

    def field(self, name):
        if name in ("u", "v", "w"):
            return getattr(self, name)
        if name in self.tracer_names:
            return self.tracers[self.tracer_names.index(name)]
        raise ValueError(f"name {name} not found in model.velocities or model.tracers.")

    # cached ctypes tuples (pointers never change after construction)
    def _tuples(self):
        ts = self.timestepper
        c = self._tuple_cache.setdefault((ts._Gn[0].ptr, self.u.ptr), {})  # one cached set per role assignment of the buffers
        if not c:
            prog = self.prognostic_fields()
            c["n"] = len(prog)
            c["U"] = _lib.ptr_array([f.ptr for f in prog])
            c["Gn"] = _lib.ptr_array([f.ptr for f in ts._Gn])
            c["Gm"] = _lib.ptr_array([f.ptr for f in ts._Gm])
            c["locs"] = _lib.i32_array([f.loc for f in prog])
        return c


def flush_tendencies(model):
    """Completes a compute_tendencies! that `time_step` deferred (model.defer_final_tendencies): the reference leaves
    Gⁿ = tendencies(state) after every time_step!; this backend may postpone that launch and fuse it with the first
    substep of the next step.  Anything that reads Gⁿ or is about to modify the state calls this first."""
    finish = getattr(model.architecture, "finish_halo_exchange", None)
    if finish is not None:  # Distributed: the halo exchange started at the end of the last step may still be in flight
        finish()
    if getattr(model, "_pending_tendencies", False):
        model._pending_tendencies = False
        compute_tendencies_(model)


def set(model, enforce_incompressibility=True, **kwargs):
    """set!(model; enforce_incompressibility=true, kwargs...)"""
    flush_tendencies(model)
    for name, value in kwargs.items():
        f = model.field(name)
        f.set(value)
        fill_halo_regions(f)
    update_state(model, compute_tendencies=False)
    if enforce_incompressibility:
        calculate_pressure_correction(model, 1.0)
        pressure_correct_velocities(model, 1.0)
        update_state(model, compute_tendencies=False)


def update_boundary_conditions(model):
    """update_boundary_condition!(fields, model) at the top of update_state! (update_nonhydrostatic_model_state.jl:29-31): function-valued
    boundary conditions are evaluated at the current clock time (physics.py BoundaryCondition.refresh)."""
    if not getattr(model, "_has_user_bcs", False):
        return
    fields = list(model.prognostic_fields())
    d = getattr(model, "diffusivity_fields", None)
    if d is not None:
        fields += [d["nu_e"]] + list(d["kappa_e"])
    for f in fields:
        b = getattr(f, "boundary_conditions", None)
        if b is not None:
            b.refresh(model.grid, f.loc, model.clock.time)


def update_state(model, compute_tendencies=True, defer_exchange=False):
    """update_state!: tupled halo fill of velocities+tracers (fill_boundary_normal_velocities=false), then tendencies.
    defer_exchange (Distributed, with compute_tendencies=False at the end of a step whose tendency launch is deferred): the x-halo
    exchange is only STARTED; the next step's fused launch overlaps it with its interior range (flush_tendencies completes it)."""
    update_boundary_conditions(model)
    arch_hook = getattr(model.architecture, "update_state", None)
    if arch_hook is not None:  # Distributed: async exchange overlapped with interior tendencies
        return arch_hook(model, compute_tendencies, defer_exchange) if defer_exchange else arch_hook(model, compute_tendencies)
    fill_halo_regions(model.prognostic_fields(), fill_boundary_normal_velocities=False)
    compute_auxiliaries(model)
    if compute_tendencies:
        compute_tendencies_(model)


def compute_auxiliaries(model):
    """compute_auxiliaries! (update_nonhydrostatic_model_state.jl:59-70) + the diffusivity halo fill (:48)"""
    compute_diffusivities(model)
    update_hydrostatic_pressure(model)
    if model.diffusivity_fields is not None:
        d = model.diffusivity_fields
        fill_halo_regions((d["nu_e"],) + d["kappa_e"])


def compute_diffusivities(model, irange=None):
    """compute_diffusivities!(diffusivity_fields, closure::AnisotropicMinimumDissipation, model); irange = (i_first, i_last):
    those columns only, 0 and Nx+1 included (Distributed: interior during the halo exchange, edges and halo columns after it)."""
    d = model.diffusivity_fields
    if d is None:
        return
    g, s = model.grid, stream_ptr()
    nt = len(model.tracers)
    if nt <= 4:  # νₑ and every κₑ in one launch
        Ck = (C.c_double * max(nt, 1))(*[model.closure.Ckappa_of(n) for n in model.tracer_names])
        if irange is not None:
            _lib.call("ocn_compute_amd_diffusivities_range", g.cref, model.closure.Cnu, model.u.ptr, model.v.ptr, model.w.ptr, d["nu_e"].ptr,
                      nt, Ck, _lib.ptr_array([c.ptr for c in model.tracers] or [None]),
                      _lib.ptr_array([k.ptr for k in d["kappa_e"]] or [None]), int(irange[0]), int(irange[1]), s)
            return
        _lib.call("ocn_compute_amd_diffusivities", g.cref, model.closure.Cnu, model.u.ptr, model.v.ptr, model.w.ptr, d["nu_e"].ptr, nt,
                  Ck, _lib.ptr_array([c.ptr for c in model.tracers] or [None]), _lib.ptr_array([k.ptr for k in d["kappa_e"]] or [None]), s)
        return
    if irange is not None:
        raise NotImplementedError("column ranges of the AMD diffusivities need at most 4 tracers")
    _lib.call("ocn_compute_amd_viscosity", g.cref, model.closure.Cnu, model.u.ptr, model.v.ptr, model.w.ptr, d["nu_e"].ptr, s)
    for name, c, k in zip(model.tracer_names, model.tracers, d["kappa_e"]):
        _lib.call("ocn_compute_amd_diffusivity", g.cref, model.closure.Ckappa_of(name), model.u.ptr, model.v.ptr, model.w.ptr,
                  c.ptr, k.ptr, s)


def update_hydrostatic_pressure(model, irange=None):
    """update_hydrostatic_pressure!(model) (update_hydrostatic_pressure.jl:25-28); irange: columns i_first..i_last of 0..Nx+1"""
    if model.pHY is not None:
        if irange is not None:
            _lib.call("ocn_update_hydrostatic_pressure_range", model.grid.cref, C.byref(model._terms), model.pHY.ptr, int(irange[0]),
                      int(irange[1]), stream_ptr())
            return
        _lib.call("ocn_update_hydrostatic_pressure", model.grid.cref, C.byref(model._terms), model.pHY.ptr, stream_ptr())


def compute_tendencies_(model, rng=None, boundary_contributions=True):
    """compute_tendencies! -> compute_interior_tendency_contributions!: K1-K3 fused + K4 per tracer, then
    compute_boundary_tendency_contributions! (compute_nonhydrostatic_tendencies.jl:17-54).  A caller that covers the domain
    with several ranges passes boundary_contributions=False and calls compute_boundary_tendency_contributions once at the end."""
    g = model.grid
    model._pending_tendencies = False
    Gn = model.timestepper._Gn
    r = None if rng is None else _lib.i32_array(list(rng))
    s = stream_ptr()
    if model.general_terms:
        t = C.byref(model._terms)
        _lib.call("ocn_compute_momentum_tendencies_terms", g.cref, t, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr,
                  Gn[1].ptr, Gn[2].ptr, r, s)
        for n, c in enumerate(model.tracers):
            kappa, kappa_e = 0.0, None
            if model.diffusivity_fields is not None:
                kappa_e = model.diffusivity_fields["kappa_e"][n].ptr
            elif model.closure is not None:
                kappa = model.closure.kappa_of(model.tracer_names[n])
            _lib.call("ocn_compute_tracer_tendency_terms", g.cref, t, kappa, kappa_e, model.u.ptr, model.v.ptr, model.w.ptr, c.ptr,
                      Gn[3 + n].ptr, r, s)
        if boundary_contributions:
            compute_boundary_tendency_contributions(model)
        return
    _lib.call("ocn_compute_momentum_tendencies", g.cref, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr,
              Gn[2].ptr, r, s)
    for n, c in enumerate(model.tracers):
        _lib.call("ocn_compute_tracer_tendency", g.cref, model.u.ptr, model.v.ptr, model.w.ptr, c.ptr, Gn[3 + n].ptr, r, s)


def compute_boundary_tendency_contributions(model):
    """compute_boundary_tendency_contributions! (compute_nonhydrostatic_tendencies.jl:152-195): bottom / top flux conditions"""
    if not (model.general_terms and model._has_flux_bcs):
        return
    g, Gn, prog = model.grid, model.timestepper._Gn, model.prognostic_fields()
    arr = (C.POINTER(_lib.CFieldBcs) * len(prog))(*[
        (C.pointer(f.boundary_conditions.c_struct(g)) if f.boundary_conditions is not None and f.boundary_conditions.has_flux()
         else C.POINTER(_lib.CFieldBcs)()) for f in prog])
    _lib.call("ocn_apply_flux_bcs", g.cref, _lib.ptr_array([G.ptr for G in Gn]), _lib.ptr_array([f.ptr for f in prog]),
              _lib.i32_array([f.loc for f in prog]), arr, len(prog), stream_ptr())


compute_tendencies = compute_tendencies_


def calculate_pressure_correction(model, dt, fill_pressure_halos=True, minimal_exchange=False):
    """calculate_pressure_correction!(model, Δt) (pressure_correction.jl:8-20).

    minimal_exchange (Distributed only; used between the RK3 substeps, where update_state! refills every halo right after the
    correction): the two synchronous x exchanges move only the planes the projection reads -- u[nx+1] for the divergence and
    p[0] for ∂x p of the first interior face -- instead of all 2 Hx planes of u, v, w and p."""
    plane = getattr(model.architecture, "fill_neighbour_plane", None) if minimal_exchange else None
    if plane is not None:
        plane(tuple(model.velocities), model.u, "east")
    else:
        fill_halo_regions(model.velocities)
    solve_for_pressure(model.pNHS, model.pressure_solver, dt, model.velocities)
    if fill_pressure_halos:
        if plane is not None:
            plane((model.pNHS,), model.pNHS, "west")
        else:
            fill_halo_regions(model.pNHS)


def solve_for_pressure(pressure, solver, dt, U):
    """solve_for_pressure! (solve_for_pressure.jl:78-82)"""
    solver.compute_source_term(U[0], U[1], U[2], dt)
    solver.solve(pressure)
    return pressure


def pressure_correct_velocities(model, dt):
    """pressure_correct_velocities! (pressure_correction.jl:40-50)"""
    _lib.call("ocn_pressure_correct_velocities", model.grid.cref, model.u.ptr, model.v.ptr, model.w.ptr, model.pNHS.ptr,
              float(dt), stream_ptr())


def cache_previous_tendencies(model, copy=None):
    """cache_previous_tendencies! (store_tendencies.jl:12-22): G⁻ <- Gⁿ.

    By default the two tendency tuples swap roles instead of being copied: the next compute_tendencies! overwrites
    every entry of Gⁿ that any kernel ever reads or writes (interior incl. the never-touched wall faces, which are 0
    in both buffers), so a swap is indistinguishable from the copy and saves a 48 B/cell pass per stage.
    `copy=True` (or model.copy_cached_tendencies) runs the reference's copy kernel K7 through the C ABI."""
    if copy is None:
        copy = model.copy_cached_tendencies
    if copy:
        t = model._tuples()
        _lib.call("ocn_cache_previous_tendencies", model.grid.cref, t["n"], t["Gm"], t["Gn"], t["locs"], stream_ptr())
    else:
        ts = model.timestepper
        ts._Gn, ts._Gm = ts._Gm, ts._Gn


def rk3_substep(model, dt, gamma, zeta):
    """rk3_substep! (runge_kutta_3.jl:160-183) for every prognostic field in one launch"""
    t = model._tuples()
    _lib.call("ocn_rk3_substep", model.grid.cref, t["n"], t["U"], t["Gn"], t["Gm"], t["locs"], float(dt), float(gamma),
              0.0 if zeta is None else float(zeta), 0 if zeta is None else 1, stream_ptr())


def ab2_step(model, dt, chi):
    """ab2_step! (quasi_adams_bashforth_2.jl:128-160)"""
    t = model._tuples()
    _lib.call("ocn_ab2_step", model.grid.cref, t["n"], t["U"], t["Gn"], t["Gm"], t["locs"], float(dt), float(chi), stream_ptr())


def time_step(model, dt, euler=False):
    """time_step!(model, Δt)"""
    if isinstance(model.timestepper, RungeKutta3TimeStepper):
        return _time_step_rk3(model, dt)
    return _time_step_qab2(model, dt, euler)


def update_state_and_rk3_substep(model, dt, gamma, zeta, fill_halos=True, p_correct=None, dt_correct=0.0):
    """update_state!(model) followed by the next stage's rk3_substep!, with compute_tendencies! and the substep fused into
    one launch (ocn_compute_momentum_tendencies_rk3).  The substep result lands in a second set of velocity arrays whose
    storage is then swapped into the model's fields."""
    if model._general_fused:
        return _update_state_and_rk3_substep_general(model, dt, gamma, zeta, fill_halos)
    if model._alt_velocities is None:
        model._alt_velocities = tuple(torch.zeros_like(f.data) for f in model.velocities)
    alt = model._alt_velocities
    Gn, Gm = model.timestepper._Gn, model.timestepper._Gm
    model._pending_tendencies = False
    g = model.grid

    def launch(rng=None):
        _lib.call("ocn_compute_momentum_tendencies_rk3", g.cref, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr,
                  Gn[2].ptr, Gm[0].ptr, Gm[1].ptr, Gm[2].ptr, alt[0].data_ptr(), alt[1].data_ptr(), alt[2].data_ptr(),
                  float(dt), float(gamma), 0.0 if zeta is None else float(zeta), 0 if zeta is None else 1,
                  None if p_correct is None else p_correct.ptr, float(dt_correct),
                  None if rng is None else _lib.i32_array(list(rng)), stream_ptr())

    hook = getattr(model.architecture, "update_state_fused", None)
    if hook is not None:  # Distributed: halo exchange overlapped with the interior launch, then the two buffer strips
        hook(model, launch, fill_halos)
    else:
        if fill_halos:
            fill_halo_regions(model.prognostic_fields(), fill_boundary_normal_velocities=False)
        launch()
    old = tuple(f.data for f in model.velocities)
    for f, a in zip(model.velocities, alt):
        f.data = a
    model._alt_velocities = old


def _bcs_ref(field, grid):
    b = getattr(field, "boundary_conditions", None)
    return C.byref(b.c_struct(grid)) if b is not None and b.has_flux() else None


def fused_tracer_launches(grid, terms_ref, u, v, w, tracers, kappas, kappa_es, Gn, Gm, outs, dt, gamma, zeta, has_zeta, rng, s):
    """tendency + boundary flux + the next substep of every tracer: pairs of tracers share ONE launch
    (ocn_compute_tracer_pair_tendency_terms_rk3: u, v, w and the tile staging are read once for both), a remaining single tracer -- and
    ranges too small for the tiled kernel -- take ocn_compute_tracer_tendency_terms_rk3."""
    n, q = len(tracers), 0
    r = None if rng is None else _lib.i32_array(list(rng))
    while q < n:
        if q + 1 < n:
            pair = (q, q + 1)
            did = C.c_int32(0)
            bcs = (C.POINTER(_lib.CFieldBcs) * 2)(*[(C.pointer(tracers[t].boundary_conditions.c_struct(grid))
                                                      if tracers[t].boundary_conditions is not None and tracers[t].boundary_conditions.has_flux()
                                                      else C.POINTER(_lib.CFieldBcs)()) for t in pair])
            _lib.call("ocn_compute_tracer_pair_tendency_terms_rk3", grid.cref, terms_ref, (C.c_double * 2)(*[float(kappas[t]) for t in pair]),
                      _lib.ptr_array([kappa_es[t] for t in pair]), bcs, u.ptr, v.ptr, w.ptr, _lib.ptr_array([tracers[t].ptr for t in pair]),
                      _lib.ptr_array([Gn[t].ptr for t in pair]), _lib.ptr_array([Gm[t].ptr for t in pair]),
                      _lib.ptr_array([outs[t].data_ptr() for t in pair]), float(dt), float(gamma), float(zeta), int(has_zeta), r,
                      C.byref(did), s)
            if did.value:
                q += 2
                continue
        c = tracers[q]
        _lib.call("ocn_compute_tracer_tendency_terms_rk3", grid.cref, terms_ref, float(kappas[q]), kappa_es[q], _bcs_ref(c, grid), u.ptr, v.ptr,
                  w.ptr, c.ptr, Gn[q].ptr, Gm[q].ptr, outs[q].data_ptr(), float(dt), float(gamma), float(zeta), int(has_zeta), r, s)
        q += 1


def _update_state_and_rk3_substep_general(model, dt, gamma, zeta, fill_halos=True):
    """update_state! + the next rk3_substep! for models with tracers and / or the §8(f) terms: momentum = tiled WENO launch +
    one finishing pass (extra terms, u / v boundary fluxes, substep), or the plain fused launch when there is nothing to add;
    each tracer = ONE launch (WENO advection, diffusion, boundary flux, substep).  All substep results land in a second set
    of arrays whose storage is then swapped into the fields.  Bit-identical to the unfused sequence in strict math."""
    update_boundary_conditions(model)
    prog = model.prognostic_fields()
    if model._alt_fields is None:
        model._alt_fields = [torch.zeros_like(f.data) for f in prog]
    alt = model._alt_fields
    Gn, Gm = model.timestepper._Gn, model.timestepper._Gm
    model._pending_tendencies = False
    g = model.grid
    z, hz = (0.0, 0) if zeta is None else (float(zeta), 1)
    t = C.byref(model._terms)
    momentum_extra = (model.coriolis is not None or model.closure is not None or model.buoyancy is not None
                      or isinstance(model.advection, (Centered, UpwindBiased)) or _bcs_ref(model.u, g) is not None or _bcs_ref(model.v, g) is not None)

    def launch(rng=None):
        s = stream_ptr()  # read HERE: the Distributed hook runs the east buffer strip under torch.cuda.stream(side stream)
        r = None if rng is None else _lib.i32_array(list(rng))
        if momentum_extra:
            _lib.call("ocn_compute_momentum_tendencies_terms_rk3", g.cref, t, _bcs_ref(model.u, g), _bcs_ref(model.v, g),
                      model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr, Gn[2].ptr, Gm[0].ptr, Gm[1].ptr, Gm[2].ptr,
                      alt[0].data_ptr(), alt[1].data_ptr(), alt[2].data_ptr(), float(dt), float(gamma), z, hz, r, s)
        else:
            _lib.call("ocn_compute_momentum_tendencies_rk3", g.cref, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr,
                      Gn[2].ptr, Gm[0].ptr, Gm[1].ptr, Gm[2].ptr, alt[0].data_ptr(), alt[1].data_ptr(), alt[2].data_ptr(),
                      float(dt), float(gamma), z, hz, None, 0.0, r, s)
        nt = len(model.tracers)
        if nt:
            kappas, kappa_es = [0.0] * nt, [None] * nt
            for n in range(nt):
                if model.diffusivity_fields is not None:
                    kappa_es[n] = model.diffusivity_fields["kappa_e"][n].ptr
                elif model.closure is not None:
                    kappas[n] = model.closure.kappa_of(model.tracer_names[n])
            fused_tracer_launches(g, t, model.u, model.v, model.w, model.tracers, kappas, kappa_es, Gn[3:], Gm[3:], alt[3:], dt, gamma, z, hz, rng, s)

    hook = getattr(model.architecture, "update_state_general", None) if fill_halos else None
    if hook is not None:  # Distributed: halo exchange overlapped with the interior auxiliaries and tendencies (distributed.py)
        hook(model, launch)
    else:
        if fill_halos:
            fill_halo_regions(prog, fill_boundary_normal_velocities=False)
            compute_auxiliaries(model)
        launch()
    for n, f in enumerate(prog):
        f.data, alt[n] = alt[n], f.data
    model._refresh_term_pointers()


def _project_and_advance(model, dt, stage_dt, gamma_next, zeta_next):
    """Everything between two substeps of RK3 (runge_kutta_3.jl:103-118): pressure projection with the stage Δt,
    cache_previous_tendencies!, update_state! and the NEXT stage's rk3_substep!, with as many of those folded into one
    launch as the grid allows (results identical to the unfused sequence, tested bit for bit)."""
    if model.fuse_stage_boundaries and model.correct_on_load:
        # pressure_correct_velocities! + update_state! + rk3_substep! in ONE launch: the correction is applied to the
        # velocities as the tendency kernel loads them, with wrapped indices (the pressure halos are not even needed)
        calculate_pressure_correction(model, stage_dt, fill_pressure_halos=False)
        cache_previous_tendencies(model)
        update_state_and_rk3_substep(model, dt, gamma_next, zeta_next, fill_halos=False, p_correct=model.pNHS, dt_correct=stage_dt)
        return
    if getattr(model, "dist_correct_on_load", False):
        return model.architecture.project_and_advance(model, dt, stage_dt, gamma_next, zeta_next)
    calculate_pressure_correction(model, stage_dt, minimal_exchange=True)
    pressure_correct_velocities(model, stage_dt)
    cache_previous_tendencies(model)
    if model.fuse_stage_boundaries:
        update_state_and_rk3_substep(model, dt, gamma_next, zeta_next)
    else:
        update_state(model, compute_tendencies=True)
        rk3_substep(model, dt, gamma_next, zeta_next)


def _time_step_rk3(model, dt):
    ts, clock = model.timestepper, model.clock
    if clock.iteration == 0:
        update_state(model, compute_tendencies=True)
    first_stage_dt = ts.g1 * dt
    second_stage_dt = (ts.g2 + ts.z2) * dt
    third_stage_dt = (ts.g3 + ts.z3) * dt
    t_next = clock.time + dt  # next_time(clock, Δt)

    # ---- first stage
    if model._pending_tendencies:  # last step's deferred compute_tendencies! fused with this step's first substep
        update_state_and_rk3_substep(model, dt, ts.g1, None, fill_halos=False)
    else:
        rk3_substep(model, dt, ts.g1, None)
    clock.time += first_stage_dt
    clock.stage = 2
    clock.last_stage_dt = first_stage_dt
    _project_and_advance(model, dt, first_stage_dt, ts.g2, ts.z2)   # ... ends with the second substep
    # ---- second stage
    clock.time += second_stage_dt
    clock.stage = 3
    clock.last_stage_dt = second_stage_dt
    _project_and_advance(model, dt, second_stage_dt, ts.g3, ts.z3)  # ... ends with the third substep
    # ---- third stage
    clock.last_stage_dt = t_next - clock.time  # corrected_third_stage_Δt
    clock.time = t_next
    clock.iteration += 1
    clock.stage = 1
    clock.last_dt = dt
    calculate_pressure_correction(model, third_stage_dt)
    pressure_correct_velocities(model, third_stage_dt)
    if model.fuse_stage_boundaries and model.defer_final_tendencies:
        update_state(model, compute_tendencies=False, defer_exchange=True)  # halos now; the tendency launch is fused into the next step
        model._pending_tendencies = True
    else:
        update_state(model, compute_tendencies=True)


def _time_step_qab2(model, dt, euler=False):
    ts, clock = model.timestepper, model.clock
    if clock.iteration == 0:
        update_state(model, compute_tendencies=True)
    euler = euler or (dt != clock.last_dt)
    chi = -0.5 if euler else ts.chi
    ab2_step(model, dt, chi)
    clock.time += dt
    clock.iteration += 1
    clock.last_dt = dt
    clock.last_stage_dt = dt
    calculate_pressure_correction(model, dt)
    pressure_correct_velocities(model, dt)
    cache_previous_tendencies(model)
    update_state(model, compute_tendencies=True)


def _advance_clock_one_rk3_step(model, dt):
    """the clock after one time_step!(model::RungeKutta3, dt), as _time_step_rk3 leaves it"""
    ts, clock = model.timestepper, model.clock
    t_next = clock.time + dt
    t2 = (clock.time + ts.g1 * dt) + (ts.g2 + ts.z2) * dt
    clock.last_stage_dt = t_next - t2  # corrected_third_stage_Δt
    clock.time = t_next
    clock.iteration += 1
    clock.stage = 1
    clock.last_dt = dt


def _adopt_driver_tendencies(model, pointers):
    """G^n of a C driver (device pointers, one per prognostic field) -> the model's time stepper: the Python host's tendencies are then
    current and nothing is pending."""
    s = stream_ptr()
    for G, ptr in zip(model.timestepper._Gn, pointers):
        if ptr and ptr != G.ptr:
            _lib.call("ocn_memcpy_d2d", G.ptr, ptr, G.data.numel() * 8, s)
    model._pending_tendencies = False


class RK3Driver:
    """The whole RK3 time_step! behind ONE entry point of the C ABI (ocn_rk3_driver_*, csrc/driver.hip): what a Julia host binds
    when it wants the fused stage boundaries without managing alternating array roles itself.  Wraps the velocity and pressure
    fields of a plain WENO model (no tracers, no extra terms) on one GPU or on ONE RANK of a slab-x run with the RCCL transport
    (ocn_rk3_driver_create_distributed: the collectives are issued by the library too, no Python between the launches);
    after flush() the state is bit-identical to `time_step(model, dt)` + flush_tendencies.
    defer_correction (default: the library's, on for all-periodic grids): the third stage's pressure correction rides on the next
    step's first fused launch; between time_step() and flush() the model's velocity fields are then NOT the corrected ones."""

    def __init__(self, model, own_solver=False, defer_correction=None):
        if model.tracers or model.general_terms:
            raise NotImplementedError("RK3Driver: WENO advection only (no tracers / extra terms)")
        if not isinstance(model.timestepper, RungeKutta3TimeStepper) or not isinstance(model.advection, WENO):
            raise NotImplementedError("RK3Driver: RungeKutta3 + WENO()")
        flush_tendencies(model)
        self.model = model
        self._h = C.c_void_p()
        arch = model.grid.architecture
        if hasattr(arch, "partition"):
            comm = getattr(arch.fabric, "_h", None)
            impl = getattr(model.pressure_solver, "impl", None)
            if comm is None or getattr(impl, "_h", None) is None:
                raise NotImplementedError("RK3Driver on a Distributed architecture needs the RCCL transport (make_distributed) and the library's "
                                          "distributed Poisson handle")
            _lib.call("ocn_rk3_driver_create_distributed", C.byref(self._h), model.grid.cref, model.u.ptr, model.v.ptr, model.w.ptr,
                      model.pNHS.ptr, impl._h, comm, stream_ptr())
        else:
            _lib.call("ocn_rk3_driver_create", C.byref(self._h), model.grid.cref, model.u.ptr, model.v.ptr, model.w.ptr, model.pNHS.ptr,
                      None if own_solver else model.pressure_solver._h, stream_ptr())  # borrows the model's solver handle by default
        if defer_correction is not None:
            _lib.call("ocn_rk3_driver_configure", self._h, int(bool(defer_correction)))

    def time_step(self, dt):
        _lib.call("ocn_rk3_driver_time_step", self._h, float(dt), stream_ptr())
        _advance_clock_one_rk3_step(self.model, dt)

    def flush(self):
        """velocities back in the model's fields, deferred tendencies completed -- in the driver's own G^n arrays, which are copied into the
        model's time stepper so that the state is the one `time_step(model, dt)` + flush_tendencies leaves: a following
        ocn.time_step(model, dt) or write_checkpoint sees the current G^n, not the values from before the driver took over."""
        _lib.call("ocn_rk3_driver_flush", self._h, stream_ptr())
        _adopt_driver_tendencies(self.model, self.tendency_pointers())

    def tendency_pointers(self):
        ptrs = [C.c_void_p() for _ in range(6)]
        _lib.call("ocn_rk3_driver_fields", self._h, *[C.byref(p) for p in ptrs])
        return [p.value for p in ptrs[3:]]

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ocn_rk3_driver_destroy(h)
            except Exception:
                pass
            self._h = None


class ModelRK3Driver:
    """time_step!(model, Δt) of a model WITH tracers and the §8(f) terms (config 4's term set) behind ONE entry point of the C ABI
    (ocn_model_driver_*, csrc/model_driver.hip): halo fills, compute_auxiliaries!, the fused tendency / substep launches and the pressure
    projection are issued by the library in the order `time_step(model, dt)` issues them on the general fused path, so after flush() the
    state is bit-identical to the Python host's -- without an interpreter between the ~20 launches of a stage.  One GPU (any x / y topology the
    fused stage boundaries take: Periodic, and since round 4 walls / Flat directions without a flux through an x / y wall), or ONE RANK of a slab-x run over the RCCL transport (ocn_model_driver_create_distributed: every exchange is issued by the library too;
    synchronous, without the interior / buffer split of distributed.py); WENO or UpwindBiased advection, RungeKutta3; number- or
    array-valued boundary conditions (functions of time are refused)."""

    def __init__(self, model, own_solver=False):
        if not isinstance(model.timestepper, RungeKutta3TimeStepper):
            raise NotImplementedError("ModelRK3Driver: RungeKutta3")
        xy_periodic = model.grid.topology[0] in ("Periodic", "FullyConnected") and model.grid.topology[1] == "Periodic"
        if not (model.fuse_stage_boundaries and model._general_fused and (xy_periodic or not hasattr(model.grid.architecture, "partition"))):
            raise NotImplementedError("ModelRK3Driver: a model on the general fused path (tracers and / or extra terms, WENO / UpwindBiased "
                                      "advection; on a Distributed architecture Periodic x and y); plain WENO models take RK3Driver")
        arch = model.grid.architecture
        comm = impl = None
        if hasattr(arch, "partition"):
            comm = getattr(arch.fabric, "_h", None)
            impl = getattr(model.pressure_solver, "impl", None)
            if comm is None or getattr(impl, "_h", None) is None:
                raise NotImplementedError("ModelRK3Driver on a Distributed architecture needs the RCCL transport (make_distributed) and the "
                                          "library's distributed Poisson handle")
        nt = len(model.tracers)
        if nt > _lib.MODEL_MAX_TRACERS:
            raise NotImplementedError(f"ModelRK3Driver: at most {_lib.MODEL_MAX_TRACERS} tracers")
        prog = model.prognostic_fields()
        for f in prog:
            b = getattr(f, "boundary_conditions", None)
            if b is not None and any(s is not None and s.func is not None for s in b.sides.values()):
                raise NotImplementedError("ModelRK3Driver: boundary conditions that are functions of time need the Python host")
        d = model.diffusivity_fields
        if d is not None:
            for f in (d["nu_e"],) + tuple(d["kappa_e"]):
                b = getattr(f, "boundary_conditions", None)
                if b is not None and not b.is_default():
                    raise NotImplementedError("ModelRK3Driver: boundary conditions on the diffusivity fields need the Python host")
        flush_tendencies(model)
        self.model = model
        desc = _lib.CModelDriverDesc()
        model._refresh_term_pointers()
        C.memmove(C.byref(desc.terms), C.byref(model._terms), C.sizeof(_lib.CModelTerms))
        desc.n_tracers = nt
        desc.tracer_T = desc.tracer_S = -1
        b = model.buoyancy
        names = list(model.tracer_names)
        if isinstance(b, BuoyancyTracer):
            desc.tracer_T = names.index("b")
        elif isinstance(b, SeawaterBuoyancy):
            if b.constant_temperature is None:
                desc.tracer_T = names.index("T")
            if b.constant_salinity is None:
                desc.tracer_S = names.index("S")
        amd = isinstance(model.closure, AnisotropicMinimumDissipation)
        if amd:
            desc.C_nu = model.closure.Cnu
            desc.nu_e = d["nu_e"].ptr
        for n, name in enumerate(names):
            desc.tracers[n] = model.tracers[n].ptr
            if amd:
                desc.C_kappa[n] = model.closure.Ckappa_of(name)
                desc.kappa_e[n] = d["kappa_e"][n].ptr
            elif model.closure is not None:
                desc.kappa[n] = model.closure.kappa_of(name)
        desc.pHY = None if model.pHY is None else model.pHY.ptr
        self._bcs = []  # keep the structs alive until the library has copied them
        for n, f in enumerate(prog):
            bc = getattr(f, "boundary_conditions", None)
            if bc is not None and not bc.is_default():
                self._bcs.append(bc.c_struct(model.grid))
                desc.bcs[n] = C.pointer(self._bcs[-1])
        self._h = C.c_void_p()
        if comm is not None:
            finish = getattr(arch, "finish_halo_exchange", None)
            if finish is not None:
                finish()
            _lib.call("ocn_model_driver_create_distributed", C.byref(self._h), model.grid.cref, C.byref(desc), model.u.ptr, model.v.ptr,
                      model.w.ptr, model.pNHS.ptr, impl._h, comm, stream_ptr())
        else:
            _lib.call("ocn_model_driver_create", C.byref(self._h), model.grid.cref, C.byref(desc), model.u.ptr, model.v.ptr, model.w.ptr,
                      model.pNHS.ptr, None if own_solver else model.pressure_solver._h, stream_ptr())

    def time_step(self, dt):
        _lib.call("ocn_model_driver_time_step", self._h, float(dt), stream_ptr())
        _advance_clock_one_rk3_step(self.model, dt)

    def flush(self):
        """every prognostic field back in the model's arrays, deferred tendencies completed and copied into the model's time stepper
        (see RK3Driver.flush)"""
        _lib.call("ocn_model_driver_flush", self._h, stream_ptr())
        _adopt_driver_tendencies(self.model, [self.tendency_pointer(n) for n in range(len(self.model.prognostic_fields()))])

    def tendency_pointer(self, n):
        f, G = C.c_void_p(), C.c_void_p()
        _lib.call("ocn_model_driver_field", self._h, int(n), C.byref(f), C.byref(G))
        return G.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ocn_model_driver_destroy(h)
            except Exception:
                pass
            self._h = None
