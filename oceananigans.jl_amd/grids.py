"""RectilinearGrid: mirrors src/Grids/rectilinear_grid.jl:248-275 and grid_generation.jl:34-155
for x, y regular and z regular or stretched; topologies (Periodic, Periodic, Periodic|Bounded|Flat).
"""
import ctypes as C
from fractions import Fraction

import numpy as np
import torch

from . import _lib
from .architectures import child_architecture, device

Periodic, Bounded, Flat, FullyConnected = "Periodic", "Bounded", "Flat", "FullyConnected"
_CODE = {Periodic: _lib.OCN_PERIODIC, Bounded: _lib.OCN_BOUNDED, Flat: _lib.OCN_FLAT, FullyConnected: _lib.OCN_FULLY_CONNECTED}


class Center:
    pass


class Face:
    pass


def _exact(x):
    return Fraction(float(x))


class RectilinearGrid:
    """RectilinearGrid(arch; size, x, y, z, topology, halo).

    `size`/`halo` omit Flat dimensions like the reference (input_validation.jl:61-95).  `z` may be a
    2-tuple (regular) or an array of Nz+1 increasing face positions (stretched, needs Bounded z)."""

    def __init__(self, architecture, size, x=None, y=None, z=None, topology=(Periodic, Periodic, Bounded), halo=None,
                 _local=False):
        if hasattr(architecture, "partition") and not _local:
            # RectilinearGrid(arch::Distributed, ...) returns the rank-local grid (distributed_grids.jl:75-118)
            from .distributed import distributed_rectilinear_grid
            g = distributed_rectilinear_grid(architecture, tuple(size), x=x, y=y, z=z, topology=tuple(topology), halo=halo)
            self.__dict__.update(g.__dict__)
            return
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
        self._origin = [0.0, 0.0, 0.0]  # left end of each regular dimension (for node coordinates)
        for d in range(3):
            if topo[d] == Flat:
                continue
            e = ext[d]
            if e is None:
                raise ValueError(f"coordinate {'xyz'[d]} must be given for a non-Flat dimension")
            if np.ndim(e) == 1 and len(e) == 2:
                c1, c2 = e
                self._origin[d] = float(c1)
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

    # -- node coordinates (grid_generation.jl:34-135: faces F[i] = c1 + (i-1) Δ, centres F[i] + Δ/2; Julia builds them as
    #    twice-precision ranges, reproduced here by exact rational arithmetic rounded once) ----------------------------
    def nodes_1d(self, d, face):
        """Interior node coordinates along dimension d (0 x, 1 y, 2 z) at Face or Center location; a Face location in a
        Bounded dimension has N+1 nodes."""
        N = (self.Nx, self.Ny, self.Nz)[d]
        H = (self.Hx, self.Hy, self.Hz)[d]
        topo = self.topology[d]
        if topo == Flat:
            return np.zeros(1)
        n = N + 1 if (face and topo == Bounded) else N
        if d == 2 and self.z_faces is not None:
            F = np.asarray(self.z_faces)[H:H + N + 1]
            return F[:n].copy() if face else 0.5 * (F[1:] + F[:-1])
        from fractions import Fraction
        c1 = Fraction(self._origin[d])
        D = Fraction((self.dx, self.dy, self.dz)[d])
        off = Fraction(0) if face else Fraction(1, 2)
        return np.array([float(c1 + (i + off) * D) for i in range(n)])

    def nodes(self, loc):
        """(x, y, z) node coordinate arrays, shaped for broadcasting over [i, j, k], of a field at bitmask `loc`."""
        x = self.nodes_1d(0, loc & 1).reshape(-1, 1, 1)
        y = self.nodes_1d(1, loc & 2).reshape(1, -1, 1)
        z = self.nodes_1d(2, loc & 4).reshape(1, 1, -1)
        return x, y, z

    # -- sizes ------------------------------------------------------------------------------------
    def parent_shape(self, loc):
        """(sx, sy, sz) of the OffsetArray parent for a field at `loc` (grid_utils.jl:66-72)."""
        N = (self.Nx, self.Ny, self.Nz)
        H = (self.Hx, self.Hy, self.Hz)
        return tuple(N[d] + 2 * H[d] + (1 if ((loc >> d) & 1 and self.topology[d] == Bounded) else 0) for d in range(3))

    @property
    def size(self):
        return (self.Nx, self.Ny, self.Nz)

    @property
    def cref(self):
        return C.byref(self.c)

    def __repr__(self):
        return (f"{self.Nx}x{self.Ny}x{self.Nz} RectilinearGrid on {self.architecture} with "
                f"{self.Hx}x{self.Hy}x{self.Hz} halo, topology {self.topology}")
