"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.

numpy/ctypes harness around oracle/libocn_oracle.so: grid construction, field storage,
halo fills, the two direct Poisson solvers and the RK3 / QAB2 `time_step!` call
sequence of Oceananigans.jl v0.96.19's NonhydrostaticModel, restated on the CPU.
Every piece cites the reference file:line it follows (relative to the reference tree).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: pinned by the reference's own property tests re-expressed in tests/
(Poisson residual, tridiagonal vs dense solve, halo identities, operator identities,
incompressibility after a step, WENO order of accuracy, reconstruction-coefficient
doctest vectors).  Bitwise parity with a live Julia run is UNPINNED (Julia is not
available in this project; FFT results are pinned only to ~sqrt(eps) by identity tests).
"""
import ctypes as C
import os
import subprocess
from fractions import Fraction

import numpy as np
import scipy.fft as sfft

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PERIODIC, BOUNDED, FLAT = 0, 1, 2
_TOPO = {"P": PERIODIC, "B": BOUNDED, "F": FLAT, "Periodic": PERIODIC, "Bounded": BOUNDED, "Flat": FLAT}

# field locations as bitmasks: bit0 x-face, bit1 y-face, bit2 z-face
LOC_U, LOC_V, LOC_W, LOC_C = 1, 2, 4, 0


class _CGrid(C.Structure):
    _fields_ = [("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32),
                ("Hx", C.c_int32), ("Hy", C.c_int32), ("Hz", C.c_int32),
                ("tx", C.c_int32), ("ty", C.c_int32), ("tz", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("dzc", C.c_void_p), ("dzf", C.c_void_p)]


def build():
    """Compile the C oracle (gcc, strict IEEE)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libocn_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.ocn_oracle_weno5.restype = C.c_double
        _LIB.ocn_oracle_weno3.restype = C.c_double
        _LIB.ocn_oracle_centered4.restype = C.c_double
        _LIB.ocn_oracle_upwind5.restype = C.c_double
        _LIB.ocn_oracle_upwind3.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------
# Grid: src/Grids/rectilinear_grid.jl:248-275, grid_generation.jl:34-155
# --------------------------------------------------------------------------------------
def _regular_spacing(c1, c2, N):
    # Δ = FT(BigFloat(c2) - BigFloat(c1)) / N) -- grid_generation.jl:105-133
    return float((Fraction(float(c2)) - Fraction(float(c1))) / N)


class Grid:
    """RectilinearGrid restated: x, y regular; z regular or stretched (z = array of Nz+1 faces)."""

    def __init__(self, size, x=None, y=None, z=None, topology=("P", "P", "P"), halo=None):
        topo = tuple(_TOPO[t] for t in topology)
        size = list(size)
        ext = [x, y, z]
        if len(size) == 3:
            full_size = [1 if topo[d] == FLAT else size[d] for d in range(3)]
        else:  # Flat dims omitted from `size`; inflated to 1 (Grids/input_validation.jl:61-95)
            it = iter(size)
            full_size = [1 if topo[d] == FLAT else next(it) for d in range(3)]
        self.Nx, self.Ny, self.Nz = full_size
        if halo is None:
            halo = [min(3, n) for n in full_size]  # default halo (input_validation.jl:71-77)
        elif len(halo) != 3:
            it = iter(halo)
            halo = [0 if topo[d] == FLAT else next(it) for d in range(3)]
        halo = [0 if topo[d] == FLAT else halo[d] for d in range(3)]
        self.Hx, self.Hy, self.Hz = halo
        self.topo = topo
        self.tx, self.ty, self.tz = topo
        N = full_size
        self.L = [1.0, 1.0, 1.0]
        self.d = [1.0, 1.0, 1.0]
        self.dzc = None
        self.dzf = None
        self.zf = None
        self._interval = [None, None, None]
        for d in range(3):
            if topo[d] == FLAT:
                continue  # Δ = L = 1 (grid_generation.jl:138-155)
            e = ext[d]
            if d == 2 and e is not None and len(e) != 2:
                self._stretched_z(np.asarray(e, dtype=np.float64))
                continue
            if d < 2 and len(e) != 2:
                raise ValueError("only z may be stretched in this restatement")
            self.L[d] = float(Fraction(float(e[1])) - Fraction(float(e[0])))
            self.d[d] = _regular_spacing(e[0], e[1], N[d])
            self._interval[d] = (float(e[0]), float(e[1]))
        self.Lx, self.Ly, self.Lz = self.L
        self.dx, self.dy, self.dz = self.d
        self.c = _CGrid(self.Nx, self.Ny, self.Nz, self.Hx, self.Hy, self.Hz, self.tx, self.ty, self.tz,
                        self.dx, self.dy, self.dz,
                        None if self.dzc is None else self.dzc.ctypes.data,
                        None if self.dzf is None else self.dzf.ctypes.data)

    def _stretched_z(self, faces):
        # generate_coordinate (grid_generation.jl:34-95), variably spaced
        N, H, topo = self.Nz, self.Hz, self.tz
        assert faces.shape == (N + 1,) and np.all(np.diff(faces) > 0)
        F = faces.copy()
        self.L[2] = float(F[N] - F[0])
        if topo == BOUNDED:
            dlo = [F[1] - F[0]] * H
            dhi = [F[-1] - F[-2]] * H
        else:
            dlo = [F[N - H + i] - F[N - H + i - 1] for i in range(1, H + 1)]  # Fi[end-H+i]-Fi[end-H+i-1]
            dhi = [F[i] - F[i - 1] for i in range(1, H + 1)]
        dhi_r = dhi[::-1]
        Fm = [F[0] - sum(dlo[i:H]) for i in range(H)]
        Fp = [F[N] + sum(dhi_r[i:H]) for i in range(H)][::-1]
        Fall = np.array(Fm + list(F) + Fp, dtype=np.float64)
        TC = N + 2 * H
        Cc = np.array([(Fall[i + 1] + Fall[i]) / 2 for i in range(TC)])
        dF0 = [Cc[i] - Cc[i - 1] for i in range(1, TC)]
        TF = N + 2 * H + (1 if topo == BOUNDED else 0)
        Fall = Fall[:TF]
        dC = np.array([Fall[i + 1] - Fall[i] for i in range(TF - 1)])  # Δᶜ, index k=1-H.. (offset -H)
        A = [dF0[0]] + dF0 + [dF0[-1]]
        for i in range(len(A) - 1, 0, -1):
            A[i] = A[i - 1]
        dFall = np.array(A)  # Δᶠ, OffsetArray(-H-1): element 0 <-> k = -H
        # store with element 0 <-> k = 1-H
        self.dzc = np.ascontiguousarray(dC, dtype=np.float64)
        self.dzf = np.ascontiguousarray(dFall[1:], dtype=np.float64)
        if self.dzc.size < N + 2 * H:  # periodic: TF-1 = N+2H-1 -> pad (never read)
            self.dzc = np.concatenate([self.dzc, self.dzc[-1:]])
        self.zf = Fall
        self.d[2] = float("nan")

    def coordinate(self, d, face):
        """The reference's coordinate array of dimension d (0 x, 1 y, 2 z) including halos, first element <-> index 1 - H:
        F = range(FT(F₋), FT(F₊), length = TF) / C = range(FT(C₋), FT(C₊), length = TC) for a regular dimension
        (grid_generation.jl:98-135, elements by Julia's TwicePrecision range: oracle/julia_base.py), the explicit arrays for a
        stretched z (:34-95)."""
        from . import julia_base
        N, H, t = (self.Nx, self.Ny, self.Nz)[d], (self.Hx, self.Hy, self.Hz)[d], self.topo[d]
        if d == 2 and self.zf is not None:
            F = np.asarray(self.zf)
            return F.copy() if face else np.array([(F[i + 1] + F[i]) / 2 for i in range(N + 2 * H)])
        c1, c2 = (Fraction(v) for v in self._interval[d])
        L = c2 - c1
        D = L / N
        Fm = c1 - H * D
        bounded = t == BOUNDED
        if face:
            lo, hi, n = Fm, Fm + (L + 2 * H * D if bounded else L + (2 * H - 1) * D), N + 2 * H + (1 if bounded else 0)
        else:
            lo = Fm + D / 2
            hi, n = lo + L + D * (2 * H - 1), N + 2 * H
        return np.array(julia_base.julia_range(float(lo), float(hi), n))

    def nodes(self, d, face, with_halos=False):
        """xnodes / ynodes / znodes (nodes_and_spacings.jl)"""
        N, H, t = (self.Nx, self.Ny, self.Nz)[d], (self.Hx, self.Hy, self.Hz)[d], self.topo[d]
        a = self.coordinate(d, face)
        return a if with_halos else a[H:H + N + (1 if face and t == BOUNDED else 0)]

    # parent extents of a field at `loc`
    def shape(self, loc):
        def e(N, H, t, face):
            return N + 2 * H + (1 if (face and t == BOUNDED) else 0)
        return (e(self.Nx, self.Hx, self.tx, loc & 1), e(self.Ny, self.Hy, self.ty, loc & 2), e(self.Nz, self.Hz, self.tz, loc & 4))

    def zeros(self, loc):
        # parent array, column-major (x fastest): numpy Fortran order
        return np.zeros(self.shape(loc), dtype=np.float64, order="F")

    def interior(self, a, loc=None):
        """View of the interior 1:N (plus the extra boundary face for Face-in-Bounded)."""
        sx, sy, sz = a.shape
        return a[self.Hx:sx - self.Hx, self.Hy:sy - self.Hy, self.Hz:sz - self.Hz]

    def interior_N(self, a):
        """View of exactly 1:Nx, 1:Ny, 1:Nz."""
        return a[self.Hx:self.Hx + self.Nx, self.Hy:self.Hy + self.Ny, self.Hz:self.Hz + self.Nz]

    @property
    def cref(self):
        return C.byref(self.c)


# --------------------------------------------------------------------------------------
# Halo fills: src/BoundaryConditions/fill_halo_regions.jl:50-196 (ordering: Flux/Open-"nothing"
# first, Periodic last), field_boundary_conditions.jl:15-33 (defaults)
# --------------------------------------------------------------------------------------
class _CBC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("value", C.c_double), ("coeff", C.c_double), ("values", C.c_void_p)]


class BC:
    """One side's boundary condition: kind in {"flux", "value", "gradient"}; the condition is a number, a 2-D array over
    the boundary's interior extent, or `value + coeff * c[interior neighbour]` (see ocn_bc in ocn_oracle.c)."""
    KINDS = {"default": 0, "flux": 1, "value": 2, "gradient": 3, "open": 4}

    def __init__(self, kind, value=0.0, coeff=0.0, values=None):
        self.kind = self.KINDS[kind]
        self.value, self.coeff = float(value), float(coeff)
        self.values = None if values is None else np.asfortranarray(values, dtype=np.float64)

    def c(self):
        return _CBC(self.kind, 0, self.value, self.coeff, None if self.values is None else self.values.ctypes.data)


def FluxBoundaryCondition(v=0.0, **kw):
    return BC("flux", v, **kw) if np.isscalar(v) else BC("flux", values=v)


def ValueBoundaryCondition(v=0.0):
    return BC("value", v) if np.isscalar(v) else BC("value", values=v)


def GradientBoundaryCondition(v=0.0):
    return BC("gradient", v) if np.isscalar(v) else BC("gradient", values=v)


def OpenBoundaryCondition(v=0.0):
    """OpenBoundaryCondition(value): the wall-normal velocity on the boundary face (boundary_condition.jl, fill_halo_regions_open.jl:65-70)"""
    return BC("open", v) if np.isscalar(v) else BC("open", values=v)


_SIDES = (("west", "east"), ("south", "north"), ("bottom", "top"))


def _cbc_ref(bc):
    return None if bc is None else C.byref(bc.c())


def fill_halo_regions(g, a, loc, fill_boundary_normal_velocities=True, bcs=None):
    """bcs: {"top": BC, "bottom": BC, ...} user boundary conditions of this field (sides not named keep the defaults)."""
    L = lib()
    N = (g.Nx, g.Ny, g.Nz)
    H = (g.Hx, g.Hy, g.Hz)
    bcs = bcs or {}
    if fill_boundary_normal_velocities:
        for d in range(3):  # fill_open_boundary_regions! (fill_halo_regions_open.jl:9-34)
            if g.topo[d] == BOUNDED and (loc >> d) & 1 and loc in (1, 2, 4):
                lo, hi = bcs.get(_SIDES[d][0]), bcs.get(_SIDES[d][1])
                clo, chi = (None if lo is None else lo.c()), (None if hi is None else hi.c())
                L.ocn_oracle_fill_open_bcs(g.cref, loc, _p(a), d, None if clo is None else C.byref(clo), None if chi is None else C.byref(chi))
    for d in range(3):  # non-periodic first
        if g.topo[d] == BOUNDED and not ((loc >> d) & 1):
            lo, hi = bcs.get(_SIDES[d][0]), bcs.get(_SIDES[d][1])
            if (lo is not None and lo.kind >= 2) or (hi is not None and hi.kind >= 2):
                clo, chi = (None if lo is None else lo.c()), (None if hi is None else hi.c())
                L.ocn_oracle_fill_value_gradient(g.cref, loc, d, None if clo is None else C.byref(clo),
                                                 None if chi is None else C.byref(chi), _p(a))
            else:
                L.ocn_oracle_fill_flux(g.cref, loc, _p(a), d)
    sx, sy, sz = a.shape
    for d in range(3):
        if g.topo[d] == PERIODIC:
            L.ocn_oracle_fill_periodic(_p(a), sx, sy, sz, d, N[d], H[d])


# --------------------------------------------------------------------------------------
# Kernels (thin wrappers)
# --------------------------------------------------------------------------------------
ADV_WENO5, ADV_CENTERED2, ADV_UPWIND5 = 0, 1, 2


def momentum_tendencies(g, u, v, w, Gu, Gv, Gw, scheme=ADV_WENO5):
    lib().ocn_oracle_momentum_tendencies_scheme(g.cref, scheme, _p(u), _p(v), _p(w), _p(Gu), _p(Gv), _p(Gw))


def tracer_tendency(g, u, v, w, c, Gc, scheme=ADV_WENO5):
    lib().ocn_oracle_tracer_tendency_scheme(g.cref, scheme, _p(u), _p(v), _p(w), _p(c), _p(Gc))


class _CPhysics(C.Structure):
    _fields_ = [("coriolis", C.c_int32), ("closure", C.c_int32), ("buoyancy", C.c_int32), ("_pad", C.c_int32),
                ("f", C.c_double), ("nu", C.c_double), ("g", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("coriolis_beta", C.c_double), ("yc", C.c_void_p), ("yf", C.c_void_p)]


class Physics:
    """coriolis = FPlane(f); closure = ScalarDiffusivity(ν, κ); buoyancy = "BuoyancyTracer" or
    ("SeawaterBuoyancy", g, α, β[, "T" | "S"]) with a LinearEquationOfState."""

    def __init__(self, f=None, nu=None, kappa=None, buoyancy=None, coriolis_beta=None, grid=None):
        """coriolis_beta (with `grid`): BetaPlane(f₀ = f, β = coriolis_beta), f = f₀ + β ynode (beta_plane.jl:43-57)"""
        self.f, self.nu, self.kappa = f, nu, kappa
        self.buoyancy = buoyancy
        kind, gg, al, be = 0, 0.0, 0.0, 0.0
        if buoyancy == "BuoyancyTracer":
            kind = 1
        elif buoyancy is not None:
            _, gg, al, be = buoyancy[:4]
            kind = {None: 2, "T": 3, "S": 4}[buoyancy[4] if len(buoyancy) > 4 else None]
        self.c = _CPhysics(0 if f is None else 1, 0 if nu is None else 1, kind, 0, 0.0 if f is None else float(f),
                           0.0 if nu is None else float(nu), float(gg), float(al), float(be), 0.0, None, None)
        if coriolis_beta is not None:
            assert f is not None and grid is not None
            # ynode = yᵃᶜᵃ[j], yᵃᶠᵃ[j] of the grid, halos included (element 0 <-> j = 1 - Hy)
            self._yc = np.ascontiguousarray(grid.nodes(1, face=False, with_halos=True), dtype=np.float64)
            self._yf = np.ascontiguousarray(grid.nodes(1, face=True, with_halos=True), dtype=np.float64)
            self.c.coriolis, self.c.coriolis_beta = 2, float(coriolis_beta)
            self.c.yc, self.c.yf = self._yc.ctypes.data, self._yf.ctypes.data

    @property
    def ref(self):
        return C.byref(self.c)


def update_hydrostatic_pressure(g, ph, T, S, pHY):
    lib().ocn_oracle_update_hydrostatic_pressure(g.cref, ph.ref, None if T is None else _p(T), None if S is None else _p(S), _p(pHY))


def momentum_extra_tendencies(g, ph, u, v, w, T, S, pHY, Gu, Gv, Gw, nu_e=None):
    lib().ocn_oracle_momentum_extra_tendencies_nu(g.cref, ph.ref, _p(u), _p(v), _p(w), None if T is None else _p(T),
                                                  None if S is None else _p(S), None if pHY is None else _p(pHY),
                                                  None if nu_e is None else _p(nu_e), _p(Gu), _p(Gv), _p(Gw))


def tracer_diffusion(g, kappa, c, Gc, kappa_e=None):
    lib().ocn_oracle_tracer_diffusion_kappa(g.cref, C.c_double(kappa), None if kappa_e is None else _p(kappa_e), _p(c), _p(Gc))


def amd_viscosity(g, Cnu, u, v, w, nu_e):
    """_compute_AMD_viscosity! over the interior (anisotropic_minimum_dissipation.jl:125-147), Cb = nothing"""
    lib().ocn_oracle_amd_viscosity(g.cref, C.c_double(Cnu), _p(u), _p(v), _p(w), _p(nu_e))


def amd_diffusivity(g, Ck, u, v, w, c, kappa_e):
    """_compute_AMD_diffusivity! (:149-169)"""
    lib().ocn_oracle_amd_diffusivity(g.cref, C.c_double(Ck), _p(u), _p(v), _p(w), _p(c), _p(kappa_e))


def apply_flux_bcs(g, loc, c, G, bcs):
    """apply_x_bcs!, apply_y_bcs!, apply_z_bcs! of one field (apply_flux_bcs.jl:13-15)"""
    for d in range(3):
        lo, hi = bcs.get(_SIDES[d][0]), bcs.get(_SIDES[d][1])
        if (lo is not None and lo.kind == 1) or (hi is not None and hi.kind == 1):
            clo, chi = (None if lo is None else lo.c()), (None if hi is None else hi.c())
            lib().ocn_oracle_apply_flux_bcs(g.cref, loc, d, None if clo is None else C.byref(clo),
                                            None if chi is None else C.byref(chi), _p(c), _p(G))


def rk3_substep(g, loc, U, Gn, Gm, dt, gamma, zeta):
    lib().ocn_oracle_rk3_substep(g.cref, loc, _p(U), _p(Gn), _p(Gm), C.c_double(dt), C.c_double(gamma),
                                 C.c_double(0.0 if zeta is None else zeta), 0 if zeta is None else 1)


def ab2_step(g, loc, U, Gn, Gm, dt, chi):
    lib().ocn_oracle_ab2_step(g.cref, loc, _p(U), _p(Gn), _p(Gm), C.c_double(dt), C.c_double(chi))


def cache_tendency(g, loc, Gm, Gn):
    lib().ocn_oracle_cache_tendency(g.cref, loc, _p(Gm), _p(Gn))


def divergence(g, u, v, w):
    out = np.zeros((g.Nx, g.Ny, g.Nz), order="F")
    lib().ocn_oracle_divergence(g.cref, _p(u), _p(v), _p(w), _p(out))
    return out


def laplacian(g, p):
    out = np.zeros((g.Nx, g.Ny, g.Nz), order="F")
    lib().ocn_oracle_laplacian(g.cref, _p(p), _p(out))
    return out


def pressure_correct(g, u, v, w, p, dt):
    lib().ocn_oracle_pressure_correct(g.cref, _p(u), _p(v), _p(w), _p(p), C.c_double(dt))


# --------------------------------------------------------------------------------------
# Poisson solvers
# --------------------------------------------------------------------------------------
def poisson_eigenvalues(N, L, topo):
    """src/Solvers/poisson_eigenvalues.jl:8-31"""
    inds = np.arange(1, N + 1, dtype=np.float64)
    if topo == PERIODIC:
        return (2 * np.sin((inds - 1) * np.pi / N) / (L / N)) ** 2
    if topo == BOUNDED:
        return (2 * np.sin((inds - 1) * np.pi / (2 * N)) / (L / N)) ** 2
    return np.zeros(N)


def _forward(a, g, dims, workers):
    # plan_transforms.jl:53-57,129-140: Bounded dims first (REDFT10), then Periodic (FFT)
    for d in dims:
        if g.topo[d] == BOUNDED:
            a = sfft.dct(a.real, type=2, axis=d, workers=workers) + 1j * sfft.dct(a.imag, type=2, axis=d, workers=workers)
    per = [d for d in dims if g.topo[d] == PERIODIC]
    if per:
        a = sfft.fftn(a, axes=per, workers=workers)
    return a


def _backward(a, g, dims, workers):
    N = (g.Nx, g.Ny, g.Nz)
    per = [d for d in dims if g.topo[d] == PERIODIC]
    if per:
        a = sfft.ifftn(a, axes=per, workers=workers)  # FFTW ifft: normalised
    for d in dims:
        if g.topo[d] == BOUNDED:  # REDFT01 x 1/(2N)  (discrete_transforms.jl:34)
            a = (sfft.dct(a.real, type=3, axis=d, workers=workers) + 1j * sfft.dct(a.imag, type=3, axis=d, workers=workers)) * (1 / (2 * N[d]))
    return a


class FFTBasedPoissonSolver:
    """src/Solvers/fft_based_poisson_solver.jl:52-125"""

    def __init__(self, g, workers=1):
        assert g.dzc is None, "FFTBasedPoissonSolver needs a regular grid"
        self.g = g
        self.workers = workers
        self.lx = poisson_eigenvalues(g.Nx, g.Lx, g.tx).reshape(-1, 1, 1)
        self.ly = poisson_eigenvalues(g.Ny, g.Ly, g.ty).reshape(1, -1, 1)
        self.lz = poisson_eigenvalues(g.Nz, g.Lz, g.tz).reshape(1, 1, -1)
        self.storage = np.zeros((g.Nx, g.Ny, g.Nz), dtype=np.complex128, order="F")

    def source_term(self, u, v, w, dt):  # compute_source_term! (solve_for_pressure.jl:70-76)
        g = self.g
        lib().ocn_oracle_source_term(g.cref, _p(u), _p(v), _p(w), C.c_double(dt), 0, _p(self.storage))

    def solve(self, p):
        g = self.g
        b = _forward(self.storage, g, (0, 1, 2), self.workers)
        with np.errstate(divide="ignore", invalid="ignore"):
            phi = -b / ((self.lx + self.ly) + self.lz)
        phi[0, 0, 0] = 0
        phi = _backward(phi, g, (0, 1, 2), self.workers)
        self.storage[...] = phi
        lib().ocn_oracle_copy_real(g.cref, _p(self.storage), _p(p))


class FourierTridiagonalPoissonSolver:
    """src/Solvers/fourier_tridiagonal_poisson_solver.jl:82-147, z stretched & Bounded"""

    def __init__(self, g, workers=1):
        assert g.tz == BOUNDED
        self.g = g
        self.workers = workers
        self.lx = poisson_eigenvalues(g.Nx, g.Lx, g.tx)
        self.ly = poisson_eigenvalues(g.Ny, g.Ly, g.ty)
        H = g.Hz
        dzf = g.dzf if g.dzf is not None else np.full(g.Nz + 2 * H, g.dz)
        # lower = upper = 1/Δzᶠ[q], q = 2..Nz  (:97-99)
        self.a = np.ascontiguousarray([1 / dzf[q + H - 1] for q in range(2, g.Nz + 1)], dtype=np.float64)
        self.D = np.zeros((g.Nx, g.Ny, g.Nz), order="F")
        self._tmpgrid = g
        if g.dzc is None:  # regular z handled by giving the C kernel explicit arrays
            self._dzc = np.full(g.Nz + 2 * H, g.dz)
            self._dzf = np.full(g.Nz + 2 * H, g.dz)
            cg = _CGrid.from_buffer_copy(g.c)
            cg.dzc = self._dzc.ctypes.data
            cg.dzf = self._dzf.ctypes.data
            self._cg = cg
        else:
            self._cg = g.c
        lib().ocn_oracle_main_diagonal_z(C.byref(self._cg), _p(self.lx), _p(self.ly), _p(self.D))
        self.source = np.zeros((g.Nx, g.Ny, g.Nz), dtype=np.complex128, order="F")
        self.storage = np.zeros((g.Nx, g.Ny, g.Nz), dtype=np.complex128, order="F")
        self.t = np.zeros((g.Nx, g.Ny, g.Nz), order="F")

    def source_term(self, u, v, w, dt):  # _fourier_tridiagonal_source_term! ZDirection (:33-38)
        lib().ocn_oracle_source_term(C.byref(self._cg), _p(u), _p(v), _p(w), C.c_double(dt), 1, _p(self.source))

    def set_source_term(self, R):  # set_source_term! (:155-161): multiply by Δzᶜ
        g = self.g
        H = g.Hz
        dzc = g.dzc[H:H + g.Nz] if g.dzc is not None else np.full(g.Nz, g.dz)
        self.source[...] = R * dzc.reshape(1, 1, -1)

    def solve(self, p):
        g = self.g
        self.source[...] = _forward(self.source, g, (0, 1), self.workers)
        lib().ocn_oracle_tridiag_solve_z(g.Nx, g.Ny, g.Nz, _p(self.a), _p(self.D), _p(self.a), _p(self.source), _p(self.t), _p(self.storage))
        phi = _backward(self.storage, g, (0, 1), self.workers)
        phi = phi - np.mean(phi)  # :142
        self.storage[...] = phi
        lib().ocn_oracle_copy_real(g.cref, _p(self.storage), _p(p))


def batched_tridiagonal_solve_z(a, b, c, f, phi0=None):
    """solve!(ϕ, ::BatchedTridiagonalSolver, rhs) for the z direction (batched_tridiagonal_solver.jl:100-123,209-235).
    a, c: (Nz-1,), b: (Nx,Ny,Nz) real, f: (Nx,Ny,Nz) complex."""
    Nx, Ny, Nz = b.shape
    b = np.asfortranarray(b, dtype=np.float64)
    f = np.asfortranarray(f, dtype=np.complex128)
    phi = np.zeros((Nx, Ny, Nz), dtype=np.complex128, order="F") if phi0 is None else np.asfortranarray(phi0, dtype=np.complex128).copy(order="F")
    t = np.zeros((Nx, Ny, Nz), order="F")
    a = np.ascontiguousarray(a, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    lib().ocn_oracle_tridiag_solve_z(Nx, Ny, Nz, _p(a), _p(b), _p(c), _p(f), _p(t), _p(phi))
    return phi


# --------------------------------------------------------------------------------------
# NonhydrostaticModel + time steppers
# --------------------------------------------------------------------------------------
class NonhydrostaticModel:
    """Restates NonhydrostaticModel(; grid, advection=WENO(), timestepper) with closure, buoyancy,
    coriolis, forcing all `nothing` (nonhydrostatic_model.jl:114-239), `set!`
    (set_nonhydrostatic_model.jl:33-60), RK3 `time_step!` (runge_kutta_3.jl:77-151) and
    QAB2 `time_step!` (quasi_adams_bashforth_2.jl:74-115)."""

    def __init__(self, grid, tracers=(), timestepper="RungeKutta3", workers=1, advection="WENO5", coriolis_f=None, coriolis_beta=None,
                 closure=None, buoyancy=None, boundary_conditions=None, hydrostatic_pressure_anomaly="default"):
        """advection: "WENO5" | "Centered2"; closure = (ν, {tracer: κ} or κ); buoyancy as in Physics;
        boundary_conditions = {"u": {"top": BC, ...}, ...} (§8(f) rank 1)."""
        g = self.grid = grid
        self.scheme = {"WENO5": ADV_WENO5, "Centered2": ADV_CENTERED2, "UpwindBiased5": ADV_UPWIND5}[advection]
        need = 1 if self.scheme == ADV_CENTERED2 else 3
        for d, (N, H) in enumerate(((g.Nx, g.Hx), (g.Ny, g.Hy), (g.Nz, g.Hz))):
            if g.topo[d] != FLAT:
                assert H >= need and N >= need, "WENO5 needs halo >= 3 (nonhydrostatic_model.jl:183,243-257) and N >= 3 (adapt_advection_order)"
        nu = kappa = None
        self.amd = None
        if closure is not None and closure[0] == "AMD":  # AnisotropicMinimumDissipation(Cν, Cκ), C = 1/12 by default
            Cnu = closure[1] if len(closure) > 1 else 1 / 12
            Ck = closure[2] if len(closure) > 2 else 1 / 12
            self.amd = (Cnu, Ck if isinstance(Ck, dict) else {n: Ck for n in tracers})
            self.nu_e = g.zeros(LOC_C)
            self.kappa_e = [g.zeros(LOC_C) for _ in tracers]
            nu, kappa = 0.0, {n: 0.0 for n in tracers}
        elif closure is not None:
            nu, kappa = closure
            if not isinstance(kappa, dict):
                kappa = {n: kappa for n in tracers}
        self.kappa = kappa
        self.physics = Physics(f=coriolis_f, nu=nu, kappa=kappa, buoyancy=buoyancy, coriolis_beta=coriolis_beta, grid=grid)
        self.bcs = boundary_conditions or {}
        # nonhydrostatic_model.jl:143-158: a separate hydrostatic pressure anomaly exists iff buoyancy is not nothing
        self.pHY = g.zeros(LOC_C) if (buoyancy is not None and hydrostatic_pressure_anomaly == "default") else None
        self.u, self.v, self.w = g.zeros(LOC_U), g.zeros(LOC_V), g.zeros(LOC_W)
        self.p = g.zeros(LOC_C)
        self.tracer_names = tuple(tracers)
        self.tracers = [g.zeros(LOC_C) for _ in tracers]
        self.locs = [LOC_U, LOC_V, LOC_W] + [LOC_C] * len(self.tracers)
        self.Gn = [g.zeros(l) for l in self.locs]
        self.Gm = [g.zeros(l) for l in self.locs]
        self.timestepper = timestepper
        if g.dzc is None:
            self.solver = FFTBasedPoissonSolver(g, workers)
        else:
            self.solver = FourierTridiagonalPoissonSolver(g, workers)
        # RungeKutta3TimeStepper constants (runge_kutta_3.jl:53-62)
        self.g1, self.g2, self.g3 = 8 / 15, 5 / 12, 3 / 4
        self.z2, self.z3 = -17 / 60, -5 / 12
        self.chi = 0.1  # QuasiAdamsBashforth2TimeStepper default χ
        self.time = 0.0
        self.iteration = 0
        self.last_dt = float("inf")
        self.update_state(compute_tendencies=False)

    @property
    def fields(self):
        return [self.u, self.v, self.w] + self.tracers

    @property
    def names(self):
        return ("u", "v", "w") + self.tracer_names

    def _buoyancy_tracers(self):
        b = self.physics.buoyancy
        t = dict(zip(self.tracer_names, self.tracers))
        if b is None:
            return None, None
        if b == "BuoyancyTracer":
            return t["b"], None
        return t.get("T"), t.get("S")

    def update_state(self, compute_tendencies=True):
        g = self.grid
        for f, l, n in zip(self.fields, self.locs, self.names):  # update_nonhydrostatic_model_state.jl:34-35
            fill_halo_regions(g, f, l, fill_boundary_normal_velocities=False, bcs=self.bcs.get(n))
        if self.amd is not None:  # compute_auxiliaries! (:59-70): compute_diffusivities! ...
            amd_viscosity(g, self.amd[0], self.u, self.v, self.w, self.nu_e)
            for n, c in enumerate(self.tracers):
                amd_diffusivity(g, self.amd[1][self.tracer_names[n]], self.u, self.v, self.w, c, self.kappa_e[n])
        if self.pHY is not None:  # ... then update_hydrostatic_pressure!
            T, S = self._buoyancy_tracers()
            update_hydrostatic_pressure(g, self.physics, T, S, self.pHY)
        if self.amd is not None:  # fill_halo_regions!(model.diffusivity_fields; only_local_halos=true) (:48)
            # user boundary conditions on the diffusivity fields: boundary_conditions = {"νₑ": {...}, "κₑ": {tracer: {...}}}
            # (build_diffusivity_fields, anisotropic_minimum_dissipation.jl:333-341)
            fill_halo_regions(g, self.nu_e, LOC_C, bcs=self.bcs.get("νₑ"))
            for n, a in enumerate(self.kappa_e):
                fill_halo_regions(g, a, LOC_C, bcs=self.bcs.get("κₑ", {}).get(self.tracer_names[n]))
        if compute_tendencies:
            self.compute_tendencies()

    def compute_tendencies(self):
        g, ph = self.grid, self.physics
        momentum_tendencies(g, self.u, self.v, self.w, self.Gn[0], self.Gn[1], self.Gn[2], self.scheme)
        if ph.c.coriolis or ph.c.closure or ph.c.buoyancy:
            T, S = self._buoyancy_tracers()
            momentum_extra_tendencies(g, ph, self.u, self.v, self.w, T, S, self.pHY, self.Gn[0], self.Gn[1], self.Gn[2],
                                      nu_e=self.nu_e if self.amd is not None else None)
        for n, c in enumerate(self.tracers):
            tracer_tendency(g, self.u, self.v, self.w, c, self.Gn[3 + n], self.scheme)
            if ph.c.closure:
                tracer_diffusion(g, self.kappa[self.tracer_names[n]], c, self.Gn[3 + n],
                                 kappa_e=self.kappa_e[n] if self.amd is not None else None)
        # compute_boundary_tendency_contributions! (compute_nonhydrostatic_tendencies.jl:204-213)
        for f, l, n, G in zip(self.fields, self.locs, self.names, self.Gn):
            if n in self.bcs:
                apply_flux_bcs(g, l, f, G, self.bcs[n])

    def calculate_pressure_correction(self, dt):  # pressure_correction.jl:8-20
        g = self.grid
        for f, l, n in zip((self.u, self.v, self.w), (LOC_U, LOC_V, LOC_W), "uvw"):
            fill_halo_regions(g, f, l, bcs=self.bcs.get(n))
        self.solver.source_term(self.u, self.v, self.w, dt)
        self.solver.solve(self.p)
        fill_halo_regions(g, self.p, LOC_C)

    def pressure_correct_velocities(self, dt):
        pressure_correct(self.grid, self.u, self.v, self.w, self.p, dt)

    def set(self, enforce_incompressibility=True, **kw):
        g = self.grid
        for name, val in kw.items():
            if name in ("u", "v", "w"):
                f, l = getattr(self, name), {"u": LOC_U, "v": LOC_V, "w": LOC_W}[name]
            else:
                f, l = self.tracers[self.tracer_names.index(name)], LOC_C
            g.interior(f)[...] = val
            fill_halo_regions(g, f, l, bcs=self.bcs.get(name))
        self.update_state(compute_tendencies=False)
        if enforce_incompressibility:
            self.calculate_pressure_correction(1.0)
            self.pressure_correct_velocities(1.0)
            self.update_state(compute_tendencies=False)

    def cache_previous_tendencies(self):
        for Gm, Gn, l in zip(self.Gm, self.Gn, self.locs):
            cache_tendency(self.grid, l, Gm, Gn)

    def time_step(self, dt, euler=False):
        if self.timestepper == "RungeKutta3":
            return self._rk3(dt)
        return self._qab2(dt, euler)

    def _rk3(self, dt):
        g = self.grid
        if self.iteration == 0:
            self.update_state(compute_tendencies=True)
        g1, g2, g3, z2, z3 = self.g1, self.g2, self.g3, self.z2, self.z3
        stages = ((g1, None, g1 * dt), (g2, z2, (g2 + z2) * dt), (g3, z3, (g3 + z3) * dt))
        t_next = self.time + dt
        for m, (gam, zet, sdt) in enumerate(stages):
            for f, l, Gn, Gm in zip(self.fields, self.locs, self.Gn, self.Gm):
                rk3_substep(g, l, f, Gn, Gm, dt, gam, zet)
            self.time = self.time + sdt if m < 2 else t_next
            self.calculate_pressure_correction(sdt)
            self.pressure_correct_velocities(sdt)
            if m < 2:
                self.cache_previous_tendencies()
            self.update_state(compute_tendencies=True)
        self.iteration += 1
        self.last_dt = dt

    def _qab2(self, dt, euler=False):
        g = self.grid
        if self.iteration == 0:
            self.update_state(compute_tendencies=True)
        euler = euler or (dt != self.last_dt)
        chi = -0.5 if euler else self.chi
        for f, l, Gn, Gm in zip(self.fields, self.locs, self.Gn, self.Gm):
            ab2_step(g, l, f, Gn, Gm, dt, chi)
        self.time += dt
        self.iteration += 1
        self.last_dt = dt
        self.calculate_pressure_correction(dt)
        self.pressure_correct_velocities(dt)
        self.cache_previous_tendencies()
        self.update_state(compute_tendencies=True)
