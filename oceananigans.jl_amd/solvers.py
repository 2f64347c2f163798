"""Solvers: mirrors src/Solvers/Solvers.jl:3-8 -- FFTBasedPoissonSolver, FourierTridiagonalPoissonSolver,
BatchedTridiagonalSolver, solve! -- on top of libocn_hip's rocFFT-based handle."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .architectures import on_architecture, stream_ptr
from .grids import Bounded, Flat, Periodic


class _PoissonHandle:
    def __init__(self, grid):
        self.grid = grid
        self._h = C.c_void_p()
        _lib.call("ocn_poisson_create", C.byref(self._h), grid.cref)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ocn_poisson_destroy(h)
            except Exception:
                pass
            self._h = None

    def info(self):
        k, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.call("ocn_poisson_info", self._h, C.byref(k), C.byref(r), C.byref(d))
        return {"kind": k.value, "r2c": bool(r.value), "direct_out": d.value, "fused_z": bool(d.value & 2), "dct_z": bool(d.value & 8)}

    def compute_source_term(self, u, v, w, dt):
        """compute_source_term!(pressure, solver, Δt, Ũ) (solve_for_pressure.jl:57-76)"""
        _lib.call("ocn_poisson_compute_source_term", self._h, u.ptr, v.ptr, w.ptr, float(dt), stream_ptr())

    def set_source_term(self, R):
        """R: halo-free array [i, j, k] (numpy) or tensor [k, j, i]."""
        g = self.grid
        if not isinstance(R, torch.Tensor):
            R = on_architecture(g.architecture, np.ascontiguousarray(np.asarray(R, dtype=np.float64).T))
        if tuple(R.shape) != (g.Nz, g.Ny, g.Nx):
            raise ValueError(f"source term shape {tuple(R.shape)} != {(g.Nz, g.Ny, g.Nx)}")
        R = R.contiguous()
        _lib.call("ocn_poisson_set_source_term", self._h, R.data_ptr(), stream_ptr())
        torch.cuda.current_stream().synchronize()  # R may be a temporary

    def solve(self, phi):
        """solve!(ϕ, solver): ϕ is a Center field whose interior receives the solution."""
        _lib.call("ocn_poisson_solve", self._h, phi.ptr, stream_ptr())
        return phi


class FFTBasedPoissonSolver(_PoissonHandle):
    """src/Solvers/fft_based_poisson_solver.jl:52-125.  (Periodic, Periodic, Periodic | Flat): the FFT pipelines of csrc/poisson.hip; any
    topology with a Bounded x or y (regular spacings): the general solver, cosine transforms along the Bounded dimensions
    (plan_transforms.jl:16-34; `general=True` forces it for the other regular topologies, Bounded z included)."""

    def __init__(self, grid, general=False):
        import os
        bounded_xy = Bounded in grid.topology[:2] or Flat in grid.topology[:2]  # (the library routes every non-Periodic x / y to the general solver)
        if grid._dzc is not None:
            raise ValueError("FFTBasedPoissonSolver requires a regular grid")
        if grid.topology[2] == Bounded and not (bounded_xy or general):
            raise NotImplementedError("(Periodic, Periodic, Bounded): FourierTridiagonalPoissonSolver in this backend (an exact solver of the "
                                      "same system); FFTBasedPoissonSolver(grid, general=True) selects the cosine-transform solver")
        if general and not bounded_xy:
            old = os.environ.get("OCN_POISSON_GENERAL")
            os.environ["OCN_POISSON_GENERAL"] = "1"
            try:
                super().__init__(grid)
            finally:
                if old is None:
                    os.environ.pop("OCN_POISSON_GENERAL")
                else:
                    os.environ["OCN_POISSON_GENERAL"] = old
        else:
            super().__init__(grid)


class FourierTridiagonalPoissonSolver(_PoissonHandle):
    """src/Solvers/fourier_tridiagonal_poisson_solver.jl:82-147, tridiagonal direction z (Bounded, regular or stretched); x and y regular
    with any topology (XYRegularRG): Periodic -> Fourier, Bounded -> cosine transforms, Flat -> none."""

    def __init__(self, grid):
        import os
        if grid.topology[2] != Bounded:
            raise ValueError("`FourierTridiagonalPoissonSolver` can only be used when the stretched direction's topology is `Bounded`.")
        if (Bounded in grid.topology[:2] or Flat in grid.topology[:2]) and grid._dzc is None:
            # a regular z next to walls in x / y: the library would pick the cosine-transform solver; ask for the Thomas sweep
            old = os.environ.get("OCN_POISSON_GENERAL_TRI")
            os.environ["OCN_POISSON_GENERAL_TRI"] = "1"
            try:
                super().__init__(grid)
            finally:
                if old is None:
                    os.environ.pop("OCN_POISSON_GENERAL_TRI")
                else:
                    os.environ["OCN_POISSON_GENERAL_TRI"] = old
        else:
            super().__init__(grid)


def nonhydrostatic_pressure_solver(grid):
    """src/Models/NonhydrostaticModels/NonhydrostaticModels.jl:25-62"""
    hook = getattr(grid.architecture, "pressure_solver", None)
    if hook is not None:
        return hook(grid)
    if Bounded in grid.topology[:2] or Flat in grid.topology[:2]:
        if grid._dzc is not None:  # XYRegularRG with a stretched z: GridWithFourierTridiagonalSolver (Solvers.jl:51-52)
            return FourierTridiagonalPoissonSolver(grid)
        return FFTBasedPoissonSolver(grid)  # XYZRegularRG (NonhydrostaticModels.jl:25-62)
    if grid.topology[2] == Bounded:
        return FourierTridiagonalPoissonSolver(grid)
    return FFTBasedPoissonSolver(grid)


class BatchedTridiagonalSolver:
    """BatchedTridiagonalSolver(grid; lower_diagonal, diagonal, upper_diagonal), z direction
    (src/Solvers/batched_tridiagonal_solver.jl:11-79).  a, c: (Nz-1,), b: [i,j,k] real."""

    def __init__(self, arch, lower_diagonal, diagonal, upper_diagonal):
        b = np.asarray(diagonal, dtype=np.float64)
        self.Nx, self.Ny, self.Nz = b.shape
        self.a = on_architecture(arch, np.ascontiguousarray(lower_diagonal, dtype=np.float64))
        self.c = on_architecture(arch, np.ascontiguousarray(upper_diagonal, dtype=np.float64))
        self.b = on_architecture(arch, np.ascontiguousarray(b.T))
        self.t = torch.zeros_like(self.b)
        self.arch = arch

    def solve(self, rhs, phi0=None):
        """solve!(ϕ, solver, rhs): rhs complex [i,j,k]; returns complex [i,j,k] (host)."""
        f = np.ascontiguousarray(np.asarray(rhs, dtype=np.complex128).T)
        fd = on_architecture(self.arch, f.view(np.float64))
        if phi0 is None:
            phi = torch.zeros_like(fd)
        else:
            phi = on_architecture(self.arch, np.ascontiguousarray(np.asarray(phi0, dtype=np.complex128).T).view(np.float64))
        _lib.call("ocn_batched_tridiagonal_solve_z", self.Nx, self.Ny, self.Nz, self.a.data_ptr(), self.b.data_ptr(),
                  self.c.data_ptr(), fd.data_ptr(), self.t.data_ptr(), phi.data_ptr(), stream_ptr())
        out = phi.cpu().numpy().view(np.complex128)
        return out.T


def solve(phi, solver, *args):
    """solve!(ϕ, solver, ...)"""
    return solver.solve(phi, *args)
