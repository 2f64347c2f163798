"""RectilinearGrid: mirrors src/Grids/rectilinear_grid.jl:248-275 and grid_generation.jl:34-155
for x, y regular and z regular or stretched; topologies (Periodic, Periodic, Periodic|Bounded|Flat).
"""
import ctypes as C
import math
import struct
from fractions import Fraction

import numpy as np
import torch

from . import _lib
from .architectures import child_architecture, device

Periodic, Bounded, Flat, FullyConnected = "Periodic", "Bounded", "Flat", "FullyConnected"
# the first / last slab of a grid whose partitioned x is Bounded (Grids.jl:97-108): wall on the west / on the east side only
RightConnected, LeftConnected = "RightConnected", "LeftConnected"
_CODE = {Periodic: _lib.OCN_PERIODIC, Bounded: _lib.OCN_BOUNDED, Flat: _lib.OCN_FLAT, FullyConnected: _lib.OCN_FULLY_CONNECTED,
         RightConnected: _lib.OCN_RIGHT_CONNECTED, LeftConnected: _lib.OCN_LEFT_CONNECTED}
# topologies whose Face fields carry the east boundary face: N + 1 points (BoundedTopology, grid_utils.jl:43)
_EAST_FACE = (Bounded, LeftConnected)


class Center:
    pass


class Face:
    pass


def _exact(x):
    return Fraction(float(x))


class _JuliaRange:
    """range(start, stop, length = n) of Julia's Base for Float64 -- the object the reference stores as a regular coordinate
    (`F = range(FT(F₋), FT(F₊), length = TF)`, src/Grids/grid_generation.jl:117-121): a StepRangeLen whose reference value and step are
    double-double numbers (base/twiceprecision.jl, `_linspace`).  Its elements are what xnodes / ynodes / znodes return, and they
    are NOT c₁ + (i - 1) Δ rounded once: the reference's own doctest prints `x ∈ [3.60072e-17, 6.28319)` for x = (0, 2π)
    (src/Grids/rectilinear_grid.jl:176-182), which this class reproduces (tests/test_reference_fixtures.py)."""

    def __init__(self, start, stop, n):
        self.n = n
        self.rational = None
        if n < 2 or start == stop:
            self.rational = [Fraction(start)] * n
            return
        p, q = self._rat(start), self._rat(stop)
        if p[1] and q[1]:
            den = p[1] * q[1] // math.gcd(p[1], q[1])
            if abs(den * start) <= 2.0 ** 53 and abs(den * stop) <= 2.0 ** 53:
                a, b = round(den * start), round(den * stop)
                if a / den == start and b / den == stop:  # endpoints are exact ratios: interpolate exactly, round once
                    self.rational = [Fraction(a * (n - i) + b * (i - 1), (n - 1) * den) for i in range(1, n + 1)]
                    return
        span = stop - start
        k = round(-(start / span) * (n - 1) + 1)  # index of the element of smallest magnitude
        if 1 < k < n:
            t = (k - 1) / (n - 1)
            ref = (1 - t) * start + t * stop
            step = (ref - start) / (k - 1) if k - 1 < n - k else (stop - ref) / (n - k)
        else:
            k = 1 if k <= 1 else n
            ref, step = (start, span / (n - 1)) if k == 1 else (stop, span / (n - 1))
        bits = min(27, math.ceil(math.log2(max(k - 1, n - k))) + 1)
        hi = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", step))[0] >> bits << bits))[0]
        lo_end, hi_end = self._two_sum((1 - k) * hi, ref), self._two_sum((n - k) * hi, ref)
        ea, eb = (start - lo_end[0]) - lo_end[1], (stop - hi_end[0]) - hi_end[1]
        self.k, self.ref, self.step_hi = k, ref, hi
        self.step_lo = (eb - ea) / (n - 1)
        self.ref_lo = ea - (1 - k) * self.step_lo

    @staticmethod
    def _two_sum(x, y):
        if abs(y) > abs(x):
            x, y = y, x
        h = x + y
        return h, (x - h) + y

    @staticmethod
    def _rat(x, limit=1 << 24):
        """Base.rat: the continued-fraction convergent of x with terms up to maxintfloat(Float32)"""
        y, a, b, c, d = x, 1, 0, 0, 1
        while abs(y) <= limit:
            f = math.trunc(y)
            y -= f
            a, c = f * a + c, a
            b, d = f * b + d, b
            if max(abs(a), abs(b)) > limit:
                return c, d
            if y == 0 or a / b == x:
                break
            y = 1.0 / y
        return a, b

    def __getitem__(self, i):
        """element i, 1-based like the reference's indices"""
        if self.rational is not None:
            return float(self.rational[i - 1])
        u = i - self.k
        hi, lo = self._two_sum(self.ref, u * self.step_hi)
        return hi + (lo + (u * self.step_lo + self.ref_lo))


class RectilinearGrid:
    """RectilinearGrid(arch; size, x, y, z, topology, halo).

    `size`/`halo` omit Flat dimensions like the reference (input_validation.jl:61-95).  `z` may be a
    2-tuple (regular) or an array of Nz+1 increasing face positions (stretched, needs Bounded z)."""

    def __init__(self, architecture, size, x=None, y=None, z=None, topology=(Periodic, Periodic, Bounded), halo=None,
                 extent=None, _local=False):
        if extent is not None:
            # validate_rectilinear_domain (input_validation.jl:106-131): an "oceanic" default domain x = (0, Lx), y = (0, Ly), z = (-Lz, 0)
            if x is not None or y is not None or z is not None:
                raise ValueError("Cannot specify both 'extent' and 'x, y, z' keyword arguments.")
            ext_t = tuple(extent) if np.ndim(extent) else (extent,)
            if len(ext_t) != sum(t != Flat for t in topology):
                raise ValueError(f"extent {ext_t} must have one entry per non-Flat dimension of {tuple(topology)}")
            it = iter(ext_t)
            Ls = [None if t == Flat else float(next(it)) for t in topology]
            x = None if Ls[0] is None else (0.0, Ls[0])
            y = None if Ls[1] is None else (0.0, Ls[1])
            z = None if Ls[2] is None else (-Ls[2], 0.0)
        if hasattr(architecture, "partition") and not _local:
            # RectilinearGrid(arch::Distributed, ...) returns the rank-local grid (distributed_grids.jl:75-118)
            from .distributed import distributed_rectilinear_grid
            g = distributed_rectilinear_grid(architecture, tuple(size), x=x, y=y, z=z, topology=tuple(topology), halo=halo)
            self.__dict__.update(g.__dict__)
            self._ctor = dict(size=tuple(size), x=x, y=y, z=z, topology=tuple(topology))  # the GLOBAL description (with_halo)
            self._ctor_local = False
            return
        self._ctor = dict(size=tuple(size) if np.ndim(size) else (size,), x=x, y=y, z=z, topology=tuple(topology))
        self._ctor_local = _local
        self.architecture = architecture
        topo = tuple(topology)
        for t in topo:
            if t not in _CODE:
                raise ValueError(f"unknown topology {t!r}")
        nflat = sum(t == Flat for t in topo)
        size = tuple(size) if np.ndim(size) else (size,)
        if len(size) != 3 - nflat:
            raise ValueError(f"size {size} must have {3 - nflat} elements for topology {topo}")  # validate_size
        it = iter(size)
        N = [1 if t == Flat else int(next(it)) for t in topo]
        if halo is None:
            H = [0 if t == Flat else min(3, n) for t, n in zip(topo, N)]  # default halo = 3 capped by N
        else:
            halo = tuple(halo) if np.ndim(halo) else (halo,)
            if len(halo) != 3 - nflat:
                raise ValueError(f"halo {halo} must have {3 - nflat} elements")
            it = iter(halo)
            H = [0 if t == Flat else int(next(it)) for t in topo]
        for d in range(3):
            if topo[d] != Flat and H[d] > N[d]:
                raise ValueError(f"halo {H[d]} must be <= size {N[d]} in dimension {d + 1}")  # validate_halo
        self.topology = topo
        self.Nx, self.Ny, self.Nz = N
        self.Hx, self.Hy, self.Hz = H
        ext = [x, y, z]
        L, D = [1.0] * 3, [1.0] * 3
        self.z_faces = None
        self._dzc_host = self._dzf_host = None
        self._interval = [(0.0, 1.0)] * 3  # end points of each regular dimension (for node coordinates)
        self._ranges = {}
        for d in range(3):
            if topo[d] == Flat:
                continue
            e = ext[d]
            if e is None:
                raise ValueError(f"coordinate {'xyz'[d]} must be given for a non-Flat dimension")
            if callable(e):  # a function of the face index k = 1 .. N+1 (grid_generation.jl:34-50)
                e = np.array([float(e(k)) for k in range(1, N[d] + 2)])
            if np.ndim(e) == 1 and len(e) == 2:
                c1, c2 = e
                self._interval[d] = (float(c1), float(c2))
                if not c2 > c1:
                    raise ValueError(f"{'xyz'[d]} must be an increasing interval!")
                Lx = _exact(c2) - _exact(c1)
                L[d] = float(Lx)                 # FT(L)
                D[d] = float(Lx / N[d])          # FT(BigFloat(L)/N), grid_generation.jl:105-133
            else:
                if d != 2:
                    raise NotImplementedError("only the z direction may be stretched")
                L[d] = self._generate_stretched_z(np.asarray(e, dtype=np.float64), N[d], H[d], topo[d])
                D[d] = float("nan")
        self.Lx, self.Ly, self.Lz = L
        self.dx, self.dy, self.dz = D
        self._dzc = self._dzf = None
        dev = None if architecture is None else device(child_architecture(architecture))  # None: metadata only (tests)
        if self._dzc_host is not None and dev is not None:
            self._dzc = torch.from_numpy(self._dzc_host).to(dev)
            self._dzf = torch.from_numpy(self._dzf_host).to(dev)
        self.c = _lib.CGrid(self.Nx, self.Ny, self.Nz, self.Hx, self.Hy, self.Hz,
                            _CODE[topo[0]], _CODE[topo[1]], _CODE[topo[2]], 0,
                            self.dx, self.dy, 0.0 if self._dzc is not None else self.dz, self.Lx, self.Ly, self.Lz,
                            None if self._dzc is None else self._dzc.data_ptr(),
                            None if self._dzf is None else self._dzf.data_ptr())

    def _generate_stretched_z(self, faces, N, H, topo):
        """generate_coordinate for explicit faces (grid_generation.jl:34-95)."""
        if topo != Bounded:
            raise NotImplementedError("a stretched z direction must be Bounded")
        if faces.shape != (N + 1,):
            raise ValueError(f"z must hold Nz+1 = {N + 1} face positions")
        if not np.all(np.diff(faces) > 0):
            raise ValueError("The elements of z must be increasing!")
        F = faces
        dlo = [F[1] - F[0]] * H
        dhi = [F[-1] - F[-2]] * H
        Fm = [F[0] - sum(dlo[i:]) for i in range(H)]
        Fp = [F[N] + sum(dhi[i:]) for i in range(H)][::-1]
        Fall = np.array(Fm + list(F) + Fp)
        TC = N + 2 * H
        Cc = (Fall[1:TC + 1] + Fall[:TC]) / 2
        d0 = list(Cc[1:] - Cc[:-1])
        dzc = Fall[1:] - Fall[:-1]                   # Δzᵃᵃᶜ[k], k = 1-H .. N+H
        dzf_full = np.array([d0[0], d0[0]] + d0)     # Δzᵃᵃᶠ[k], k = -H .. N+H (shifted copy, :72-75)
        self.z_faces = Fall
        self._dzc_host = np.ascontiguousarray(dzc[:N + 2 * H])
        self._dzf_host = np.ascontiguousarray(dzf_full[1:])  # element 0 <-> k = 1-H
        assert self._dzf_host.size == N + 2 * H and self._dzc_host.size == N + 2 * H
        return float(F[N] - F[0])

    # -- node coordinates: the elements of the reference's coordinate ranges (grid_generation.jl:98-135) --------------------------
    def _coordinate_range(self, d, face):
        """F = range(FT(F₋), FT(F₊), length = TF) or C = range(FT(C₋), FT(C₊), length = TC) of a regular dimension; element
        i + H of the range is the reference's ξ[i] (OffsetArray(F, -H))."""
        key = (d, bool(face))
        r = self._ranges.get(key)
        if r is None:
            N = (self.Nx, self.Ny, self.Nz)[d]
            H = (self.Hx, self.Hy, self.Hz)[d]
            bounded = self.topology[d] in _EAST_FACE
            c1, c2 = (Fraction(v) for v in self._interval[d])
            L = c2 - c1
            D = L / N                                                # BigFloat arithmetic of the reference, here exact
            Fm = c1 - H * D
            if face:
                lo, hi, n = Fm, Fm + (L + 2 * H * D if bounded else L + (2 * H - 1) * D), N + 2 * H + (1 if bounded else 0)
            else:
                lo = Fm + D / 2
                hi, n = lo + L + D * (2 * H - 1), N + 2 * H
            r = self._ranges[key] = _JuliaRange(float(lo), float(hi), n)
        return r

    def nodes_1d(self, d, face, with_halos=False):
        """xnodes / ynodes / znodes(grid, ℓ; with_halos): node coordinates along dimension d (0 x, 1 y, 2 z) at Face or Center
        location; a Face location in a Bounded dimension has N+1 interior nodes."""
        N = (self.Nx, self.Ny, self.Nz)[d]
        H = (self.Hx, self.Hy, self.Hz)[d]
        topo = self.topology[d]
        if topo == Flat:
            return np.zeros(1)
        n = N + 1 if (face and topo in _EAST_FACE) else N
        if d == 2 and self.z_faces is not None:
            Fall = np.asarray(self.z_faces)
            if with_halos:
                return Fall.copy() if face else 0.5 * (Fall[1:] + Fall[:-1])
            F = Fall[H:H + N + 1]
            return F[:n].copy() if face else 0.5 * (F[1:] + F[:-1])
        r = self._coordinate_range(d, face)
        if with_halos:
            return np.array([r[i] for i in range(1, r.n + 1)])
        return np.array([r[i + H] for i in range(1, n + 1)])

    def domain(self, d):
        """x_domain / y_domain / z_domain: (ξ[1], ξ[N+1]) of the Face coordinate (grid_utils.jl:120), what `show(grid)` prints"""
        N = (self.Nx, self.Ny, self.Nz)[d]
        H = (self.Hx, self.Hy, self.Hz)[d]
        if d == 2 and self.z_faces is not None:
            F = np.asarray(self.z_faces)
            return float(F[H]), float(F[H + N])
        r = self._coordinate_range(d, True)
        return r[1 + H], r[N + 1 + H]

    def spacing_extrema(self, d, face=False):
        """(min, max) of the interior cell spacings along d: `min(Δz)=…, max(Δz)=…` of the reference's `show` (Center spacings)"""
        if d == 2 and self._dzc_host is not None:
            a = (self._dzf_host if face else self._dzc_host)[self.Hz:self.Hz + self.Nz + (1 if face else 0)]
            return float(a.min()), float(a.max())
        D = (self.dx, self.dy, self.dz)[d]
        return D, D

    def nodes(self, loc):
        """(x, y, z) node coordinate arrays, shaped for broadcasting over [i, j, k], of a field at bitmask `loc`."""
        x = self.nodes_1d(0, loc & 1).reshape(-1, 1, 1)
        y = self.nodes_1d(1, loc & 2).reshape(1, -1, 1)
        z = self.nodes_1d(2, loc & 4).reshape(1, 1, -1)
        return x, y, z

    def with_halo(self, new_halo):
        """with_halo(new_halo, grid) (rectilinear_grid.jl:367-383): the same grid rebuilt with another halo; new_halo has one entry per
        dimension (entries of Flat dimensions are ignored, like the reference's pop_flat_elements)."""
        if self._ctor_local:
            raise NotImplementedError("with_halo of a rank-local grid: rebuild it from the global description")
        halo = tuple(int(h) for h, t in zip(new_halo, self.topology) if t != Flat)
        g = RectilinearGrid(self.architecture, halo=halo, **self._ctor)
        g.c.math = self.c.math
        return g

    # -- sizes ------------------------------------------------------------------------------------
    def parent_shape(self, loc):
        """(sx, sy, sz) of the OffsetArray parent for a field at `loc` (grid_utils.jl:66-72)."""
        N = (self.Nx, self.Ny, self.Nz)
        H = (self.Hx, self.Hy, self.Hz)
        return tuple(N[d] + 2 * H[d] + (1 if ((loc >> d) & 1 and self.topology[d] in _EAST_FACE) else 0) for d in range(3))

    @property
    def size(self):
        return (self.Nx, self.Ny, self.Nz)

    @property
    def cref(self):
        return C.byref(self.c)

    @property
    def math_mode(self):
        """None (the process default, ocn.set_math_mode) or MATH_STRICT / MATH_FAST: the arithmetic variant of every kernel launched
        for this grid (ocn_grid.math)."""
        return {_lib.GRID_MATH_DEFAULT: None, _lib.GRID_MATH_STRICT: _lib.MATH_STRICT, _lib.GRID_MATH_FAST: _lib.MATH_FAST}[self.c.math]

    def with_math_mode(self, mode):
        """The same grid (shared device vectors) whose launches use `mode` whatever the process default is: two models of one process
        may run different arithmetic variants."""
        import copy
        if mode not in (None, _lib.MATH_STRICT, _lib.MATH_FAST):
            raise ValueError(f"math_mode must be None, MATH_STRICT or MATH_FAST, got {mode!r}")
        g = copy.copy(self)
        g.c = _lib.CGrid.from_buffer_copy(self.c)
        g.c.math = {None: _lib.GRID_MATH_DEFAULT, _lib.MATH_STRICT: _lib.GRID_MATH_STRICT, _lib.MATH_FAST: _lib.GRID_MATH_FAST}[mode]
        return g

    def __repr__(self):
        return (f"{self.Nx}x{self.Ny}x{self.Nz} RectilinearGrid on {self.architecture} with "
                f"{self.Hx}x{self.Hy}x{self.Hz} halo, topology {self.topology}")
