"""Advection schemes accepted by NonhydrostaticModel: mirrors src/Advection/weno_reconstruction.jl:98-123.

Only `WENO()` = WENO(order=5) without `grid` (uniform coefficients, also along a stretched z:
reconstruction_coefficients.jl:244-250) is implemented by the HIP kernels."""


class WENO:
    def __init__(self, order=5, grid=None, bounds=None):
        if order % 2 == 0:
            raise ValueError("WENO reconstruction scheme is defined only for odd orders")
        if order != 5 or grid is not None or bounds is not None:
            raise NotImplementedError("the MI355X backend implements WENO(order=5) with uniform coefficients only")
        self.order = order
        self.buffer = 3  # required_halo_size (Advection.jl:61-63)

    def __repr__(self):
        return "WENO(order=5)"


class UpwindBiased:
    """UpwindBiased(order=5) (src/Advection/upwind_biased_reconstruction.jl:41-66): fixed 5-point upwind stencils, advecting
    velocity scheme Centered(order=4), buffer schemes UpwindBiased(order=3) -> (order=1) near Bounded walls."""

    def __init__(self, order=5, grid=None):
        if order % 2 == 0:
            raise ValueError("UpwindBiased reconstruction scheme is defined only for odd orders")
        if order != 5 or grid is not None:
            raise NotImplementedError("the MI355X backend implements UpwindBiased(order=5) with uniform coefficients only")
        self.order = order
        self.buffer = 3

    def __repr__(self):
        return "UpwindBiased(order=5)"
