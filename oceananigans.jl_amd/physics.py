"""Model terms beyond WENO advection, mirroring the reference's constructors:

  Centered(order=2)                               src/Advection/centered_reconstruction.jl:39-60 (the reference's default advection)
  FPlane(f=...) / FPlane(rotation_rate, latitude) src/Coriolis/f_plane.jl:8-42
  ScalarDiffusivity(ν=..., κ=...)                 src/TurbulenceClosures/turbulence_closure_implementations/scalar_diffusivity.jl
  BuoyancyTracer(), SeawaterBuoyancy(...), LinearEquationOfState(...)
                                                  src/BuoyancyFormulations/{buoyancy_tracer,seawater_buoyancy,linear_equation_of_state}.jl
  FluxBoundaryCondition, ValueBoundaryCondition, GradientBoundaryCondition, FieldBoundaryConditions, BetaPlane
                                                  src/BoundaryConditions/{boundary_condition,field_boundary_conditions}.jl

Only what the HIP kernels implement is accepted; everything else raises NotImplementedError.
"""
import math

import numpy as np

from . import _lib
from .architectures import on_architecture


def sind(x):
    """Julia's sind (base/special/trig.jl): the argument is reduced mod 360 exactly and the result is the correctly rounded
    sin(x π / 180) -- exactly 0, ±0.5, ±1 where they occur, which math.sin(math.radians(x)) misses (sind(30) == 0.5).
    Used by FPlane(latitude = φ), f = 2 Ω sind(φ) (f_plane.jl:38-40; test/test_coriolis.jl:24-27 expects f = 2 from Ω = 2, φ = 30)."""
    from decimal import Decimal, localcontext
    from fractions import Fraction
    r = Fraction(x) % 360
    sign = 1.0
    if r >= 180:
        r, sign = r - 180, -1.0
    if r > 90:
        r = 180 - r
    if r == 0:
        return 0.0
    if r == 30:
        return sign * 0.5
    if r == 90:
        return sign
    with localcontext() as ctx:
        ctx.prec = 60
        pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899")
        t = Decimal(r.numerator) / Decimal(r.denominator) * pi / 180
        term, total, n = t, t, 1
        while abs(term) > Decimal(10) ** -58:
            term = -term * t * t / ((2 * n) * (2 * n + 1))
            total += term
            n += 1
        return sign * float(total)


def cosd(x):
    """Julia's cosd: cos(x π / 180) correctly rounded, exact at the multiples of 30 / 90 where the cosine is 0, ±0.5, ±1
    (cosd(x) = sind(90 - x) holds exactly for the reduced argument)."""
    from fractions import Fraction
    return sind(Fraction(90) - Fraction(x))


class Centered:
    """Centered(order=2)"""

    def __init__(self, order=2, grid=None):
        if order % 2 != 0:
            raise ValueError("Centered reconstruction scheme is defined only for even orders")
        if order != 2 or grid is not None:
            raise NotImplementedError("the MI355X backend implements Centered(order=2) only")
        self.order = order
        self.buffer = 1  # required_halo_size

    code = _lib.ADVECTION_CENTERED2

    def __repr__(self):
        return "Centered(order=2)"


class FPlane:
    """FPlane(; f, rotation_rate=Ω_Earth, latitude) (f_plane.jl:23-42)"""
    OMEGA_EARTH = 7.292115e-5  # src/Coriolis/Coriolis.jl

    def __init__(self, f=None, rotation_rate=None, latitude=None):
        if f is not None and latitude is not None:
            raise ValueError("Either both keywords rotation_rate and latitude must be specified, or only f must be specified.")
        if f is None and latitude is None:
            raise ValueError("Either both keywords rotation_rate and latitude must be specified, or only f must be specified.")
        if f is None:
            rotation_rate = self.OMEGA_EARTH if rotation_rate is None else rotation_rate
            f = 2 * rotation_rate * sind(latitude)
        self.f = float(f)


class BetaPlane:
    """BetaPlane(; f₀, β) or BetaPlane(; rotation_rate=Ω_Earth, latitude, radius=R_Earth) (beta_plane.jl:6-41): f = f₀ + β y, evaluated at
    the y node of each velocity point (:43-57)."""
    OMEGA_EARTH = 7.292115e-5  # src/Coriolis/Coriolis.jl:19
    R_EARTH = 6371.0e3         # src/Grids/Grids.jl:40

    def __init__(self, f0=None, beta=None, rotation_rate=None, latitude=None, radius=None, **kw):
        f0 = kw.pop("f₀", f0)
        beta = kw.pop("β", beta)
        if kw:
            raise TypeError(f"unknown keyword(s) {sorted(kw)}")
        use_f_and_beta = f0 is not None and beta is not None
        use_planet = latitude is not None
        if use_f_and_beta == use_planet:
            raise ValueError("Either both keywords f₀ and β must be specified, *or* all of rotation_rate, latitude, and radius.")
        if use_planet:
            rotation_rate = self.OMEGA_EARTH if rotation_rate is None else rotation_rate
            radius = self.R_EARTH if radius is None else radius
            f0 = 2 * rotation_rate * sind(latitude)
            beta = 2 * rotation_rate * cosd(latitude) / radius
        self.f0, self.beta = float(f0), float(beta)


class ScalarDiffusivity:
    """ScalarDiffusivity(; ν=0, κ=0): ExplicitTimeDiscretization, ThreeDimensionalFormulation, constant coefficients.
    κ is a number (every tracer) or a dict {tracer name: κ}."""

    def __init__(self, nu=0.0, kappa=0.0, time_discretization="Explicit", formulation="ThreeDimensional", **kw):
        nu = kw.pop("ν", nu)
        kappa = kw.pop("κ", kappa)
        if kw:
            raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
        if time_discretization != "Explicit":
            raise NotImplementedError("VerticallyImplicitTimeDiscretization is not implemented")
        if formulation != "ThreeDimensional":
            raise NotImplementedError("only the ThreeDimensionalFormulation (isotropic) is implemented")
        if callable(nu) or callable(kappa) or np.ndim(nu) > 0:  # (numpy scalars are numbers)
            raise NotImplementedError("only constant ν, κ are implemented")
        self.nu = float(nu)
        self.kappa = kappa

    def kappa_of(self, name):
        if isinstance(self.kappa, dict):
            if name not in self.kappa:
                raise ValueError(f"no diffusivity κ given for tracer {name}")
            return float(self.kappa[name])
        return float(self.kappa)


class AnisotropicMinimumDissipation:
    """AnisotropicMinimumDissipation(; C = 1/12, Cν = nothing, Cκ = nothing, Cb = nothing)
    (anisotropic_minimum_dissipation.jl:106-115): ExplicitTimeDiscretization, number (or per-tracer dict) Poincaré constants."""

    def __init__(self, C=1 / 12, Cnu=None, Ckappa=None, Cb=None, **kw):
        Cnu = kw.pop("Cν", Cnu)
        Ckappa = kw.pop("Cκ", Ckappa)
        if kw:
            raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
        if Cb is not None:
            raise NotImplementedError("the buoyancy modification (Cb) is not implemented")
        self.Cnu = float(C if Cnu is None else Cnu)
        self.Ckappa = C if Ckappa is None else Ckappa
        if callable(self.Ckappa) or callable(Cnu):
            raise NotImplementedError("only number Poincaré constants are implemented")

    def Ckappa_of(self, name):
        if isinstance(self.Ckappa, dict):
            if name not in self.Ckappa:
                raise ValueError(f"no Poincaré constant Cκ given for tracer {name}")
            return float(self.Ckappa[name])
        return float(self.Ckappa)


class LinearEquationOfState:
    """linear_equation_of_state.jl:33-35"""

    def __init__(self, thermal_expansion=1.67e-4, haline_contraction=7.80e-4):
        self.thermal_expansion = float(thermal_expansion)
        self.haline_contraction = float(haline_contraction)


class BuoyancyTracer:
    required_tracers = ("b",)


class SeawaterBuoyancy:
    """seawater_buoyancy.jl:88-108; g_Earth = 9.80665 (BuoyancyFormulations.jl)"""

    def __init__(self, gravitational_acceleration=9.80665, equation_of_state=None, constant_temperature=None,
                 constant_salinity=None):
        eos = LinearEquationOfState() if equation_of_state is None else equation_of_state
        if not isinstance(eos, LinearEquationOfState):
            raise NotImplementedError("only LinearEquationOfState is implemented")
        if constant_temperature is not None and constant_salinity is not None:
            raise ValueError("constant_temperature and constant_salinity cannot both be set")
        self.equation_of_state = eos
        self.gravitational_acceleration = float(gravitational_acceleration)
        self.constant_temperature = constant_temperature
        self.constant_salinity = constant_salinity

    @property
    def required_tracers(self):
        if self.constant_salinity is not None:
            return ("T",)
        if self.constant_temperature is not None:
            return ("S",)
        return ("T", "S")


# --------------------------------------------------------------------------------------------------------
# Boundary conditions
# --------------------------------------------------------------------------------------------------------
class BoundaryCondition:
    """BoundaryCondition(classification, condition).  `condition` is a number, an array over the boundary's two tangential directions
    or a function of the tangential coordinates and time; `coeff` restates the
    ContinuousBoundaryFunction  f(x, y, t, c, p) = p * c  with field_dependencies = the field itself as
    condition + coeff * c[i, j, boundary-adjacent cell] (include/ocn_hip.h: struct ocn_bc)."""

    def __init__(self, kind, condition=0.0, coeff=0.0, parameters=None):
        """condition: a number, an (Nx, Ny) array, or a function f(x, y, t) [f(x, y, t, parameters) with `parameters`] of the two
        coordinates tangential to a bottom / top boundary (without those of Flat directions: f(x, t) on an x-z slice) and time -- the reference's ContinuousBoundaryFunction without field
        dependencies (continuous_boundary_function.jl:17-115).  The function is evaluated on the host at the field's own nodes (called
        once with broadcastable arrays) every time the model state is updated, with the clock time of that moment, and uploaded into
        the (Nx, Ny) device array the kernels read."""
        self.kind = kind
        self.coeff = float(coeff)
        self.values = None
        self.value = 0.0
        self.func, self.parameters = None, parameters
        if np.isscalar(condition):
            self.value = float(condition)
        elif callable(condition):
            self.func = condition
        else:
            self.values = np.ascontiguousarray(np.asarray(condition, dtype=np.float64).T)  # stored [j, i]: x fastest
        self._device_values = None

    # the two directions tangential to a side, in the order the kernels index `values` (first one fastest): csrc/kernels.hip
    # fill_halos_general_kernel / apply_flux_bcs_lateral_kernel / apply_flux_bcs_kernel -> values[(a1 - 1) + n1 (a2 - 1)]
    _TANGENTIAL = {"west": (1, 2), "east": (1, 2), "south": (0, 2), "north": (0, 2), "bottom": (0, 1), "top": (0, 1)}

    def _extents(self, grid, side):
        N = (grid.Nx, grid.Ny, grid.Nz)
        d1, d2 = self._TANGENTIAL[side]
        return N[d1], N[d2]

    def refresh(self, grid, loc, time, side="top"):
        """Re-evaluate a function-valued condition at `time` (no-op otherwise): f(ξ, η, t[, p]) of the two coordinates tangential to the
        boundary -- (x, y) on bottom / top, (y, z) on west / east, (x, z) on south / north -- at the field's own nodes there
        (continuous_boundary_function.jl:17-115: the reference's ContinuousBoundaryFunction without field dependencies)."""
        if self.func is None:
            return
        d1, d2 = self._TANGENTIAL[side]
        n1, n2 = self._extents(grid, side)
        nodes = grid.nodes(loc)
        c1 = np.asarray(nodes[d1]).reshape(-1)[:n1].reshape(-1, 1)
        c2 = np.asarray(nodes[d2]).reshape(-1)[:n2].reshape(1, -1)
        # the coordinates of Flat directions are not arguments (a top condition on a (Bounded, Flat, Bounded) grid is f(x, t[, p]),
        # examples/horizontal_convection.jl:47)
        coords = tuple(c for c, d in zip((c1, c2), (d1, d2)) if grid.topology[d] != "Flat")
        args = coords + (float(time),) if self.parameters is None else coords + (float(time), self.parameters)
        vals = np.broadcast_to(np.asarray(self.func(*args), dtype=np.float64), (n1, n2))
        self.values = np.array(vals.T, dtype=np.float64, order="C")  # (a copy: broadcast views are read-only)
        if self._device_values is None:
            self._device_values = on_architecture(grid.architecture, self.values)
        else:
            import torch
            self._device_values.copy_(torch.from_numpy(self.values))

    def c_struct(self, grid, side="top"):
        ptr = None
        if self.func is not None and self._device_values is None:
            raise RuntimeError("function-valued boundary condition used before its first evaluation (update_boundary_conditions)")
        if self.values is not None:
            n1, n2 = self._extents(grid, side)
            if self.values.shape != (n2, n1):
                raise ValueError(f"array boundary condition on the {side} boundary has shape {self.values.T.shape}, expected {(n1, n2)}")
            if self._device_values is None:
                self._device_values = on_architecture(grid.architecture, self.values)
            ptr = self._device_values.data_ptr()
        return _lib.CBc(self.kind, 0, self.value, self.coeff, ptr)


def FluxBoundaryCondition(condition=0.0, coeff=0.0, parameters=None):
    return BoundaryCondition(_lib.BC_FLUX, condition, coeff, parameters)


def ValueBoundaryCondition(condition=0.0, parameters=None):
    return BoundaryCondition(_lib.BC_VALUE, condition, parameters=parameters)


def GradientBoundaryCondition(condition=0.0, parameters=None):
    return BoundaryCondition(_lib.BC_GRADIENT, condition, parameters=parameters)


def OpenBoundaryCondition(condition=0.0, parameters=None):
    """OpenBoundaryCondition(value) on the side a velocity component is normal to: that component ON the boundary face is set to the
    value (a number, an array over the tangential directions or a function of the tangential coordinates and time) by every halo fill
    that fills boundary-normal velocities (fill_halo_regions_open.jl:9-70); the default Impenetrable condition is Open(nothing) = 0."""
    return BoundaryCondition(_lib.BC_OPEN, condition, parameters=parameters)


class FieldBoundaryConditions:
    """FieldBoundaryConditions(; west, east, south, north, bottom, top); unspecified sides keep the topology defaults
    (field_boundary_conditions.jl:15-33).  Flux, Value, Gradient on every side of a Bounded direction, each a number, condition +
    coeff * c, an array over the two tangential directions ((Ny, Nz) on west / east, (Nx, Nz) on south / north, (Nx, Ny) on bottom /
    top) or a function of the tangential coordinates and time; the fluxes of the lateral sides enter through apply_x_bcs! /
    apply_y_bcs! (apply_flux_bcs.jl:38-146)."""
    SIDES = ("west", "east", "south", "north", "bottom", "top")

    def __init__(self, **sides):
        for k in sides:
            if k not in self.SIDES:
                raise TypeError(f"unknown boundary {k!r}")
        self.sides = {k: sides.get(k) for k in self.SIDES}
        self._c = None

    def is_default(self):
        return all(v is None for v in self.sides.values())

    def refresh(self, grid, loc, time):
        """update_boundary_condition! for function-valued conditions: evaluate them at the clock time"""
        for k, v in self.sides.items():
            if v is not None:
                v.refresh(grid, loc, time, k)

    def has_flux(self):
        return any(v is not None and v.kind == _lib.BC_FLUX for v in self.sides.values())

    def c_struct(self, grid):
        if self._c is None:
            default = _lib.CBc(_lib.BC_DEFAULT, 0, 0.0, 0.0, None)
            self._c = _lib.CFieldBcs(*[(default if self.sides[k] is None else self.sides[k].c_struct(grid, k)) for k in self.SIDES])
        return self._c
