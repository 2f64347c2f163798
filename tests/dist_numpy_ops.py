"""CPU stand-ins (numpy / scipy / the C oracle) for the device `ops` of oceananigans.jl_amd/distributed.py.

TEST INFRASTRUCTURE: lets the multi-rank choreography (who sends what to whom, in which order; chunk layout of the
all-to-all) run on CPU ranks over gloo.  The transposes restate src/DistributedComputations/distributed_transpose.jl:25-95
index for index; pack/unpack restate src/Fields/field_boundary_buffers.jl:276-308."""
import numpy as np
import scipy.fft as sfft
import torch

from oracle import oracle as O


class HostArch:
    """child architecture whose arrays live in host memory (tests only)."""
    device = torch.device("cpu")


def fview(f):
    """F-ordered [i,j,k] numpy view of a Field's parent tensor."""
    return f.data.numpy().T


class _NumpyDistPoisson:
    def __init__(self, grid, arch):
        self.g, self.R, self.rank = grid, arch.partition.x, arch.local_rank
        nx, Ny, Nz = grid.Nx, grid.Ny, grid.Nz
        self.ny, self.Nxg = Ny // self.R, nx * self.R
        self.y = np.zeros((nx, Ny, Nz), dtype=np.complex128, order="F")
        self.x = np.zeros((self.Nxg, self.ny, Nz), dtype=np.complex128, order="F")
        n = nx * Ny * Nz * 2
        self.send, self.recv = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        self.lx = O.poisson_eigenvalues(self.Nxg, grid.global_Lx, O.PERIODIC)
        self.ly = O.poisson_eigenvalues(Ny, grid.Ly, O.PERIODIC)
        self.lz = O.poisson_eigenvalues(Nz, grid.Lz, O.PERIODIC)
        self.og = O.Grid((nx, Ny, Nz), x=(0, grid.Lx), y=(0, grid.Ly), z=(0, grid.Lz), topology="PPP", halo=(grid.Hx, grid.Hy, grid.Hz))

    def _cbuf(self, t):
        return t.numpy().view(np.complex128)

    def source_term(self, u, v, w, dt):
        S = O.FFTBasedPoissonSolver(self.og)
        S.source_term(np.asfortranarray(fview(u)), np.asfortranarray(fview(v)), np.asfortranarray(fview(w)), dt)
        self.y[...] = S.storage

    def forward_yz(self):
        self.y[...] = sfft.fftn(self.y, axes=(1, 2))

    def pack_y_to_x(self):  # send[i + nx*(k + Nz*j)] = y[i,j,k]  (:38-42)
        self._cbuf(self.send)[...] = np.transpose(self.y, (0, 2, 1)).ravel(order="F")

    def unpack_x_from_y(self, rbuf):  # x[i,j,k] = recv[i' + nx*(k + Nz*j) + m*nx*ny*Nz]  (:51-60)
        nx, ny, Nz = self.g.Nx, self.ny, self.g.Nz
        r = self._cbuf(rbuf).reshape((nx, Nz, ny, self.R), order="F")
        for m in range(self.R):
            self.x[m * nx:(m + 1) * nx] = np.transpose(r[:, :, :, m], (0, 2, 1))

    def solve_x(self):
        xh = sfft.fft(self.x, axis=0)
        j0 = self.rank * self.ny
        lam = (self.lx[:, None, None] + self.ly[None, j0:j0 + self.ny, None]) + self.lz[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            xh = -xh / lam
        if self.rank == 0:
            xh[0, 0, 0] = 0
        self.x[...] = sfft.ifft(xh, axis=0)

    def pack_x_to_y(self):  # send[j + ny*(k + Nz*i)] = x[i,j,k]  (:31-35)
        self._cbuf(self.send)[...] = np.transpose(self.x, (1, 2, 0)).ravel(order="F")

    def unpack_y_from_x(self, rbuf):  # y[i,j,k] = recv[j' + ny*(k + Nz*i) + m*nx*ny*Nz]  (:86-95)
        nx, ny, Nz = self.g.Nx, self.ny, self.g.Nz
        r = self._cbuf(rbuf).reshape((ny, Nz, nx, self.R), order="F")
        for m in range(self.R):
            self.y[:, m * ny:(m + 1) * ny, :] = np.transpose(r[:, :, :, m], (2, 0, 1))

    def backward_yz(self, p):
        self.y[...] = sfft.ifftn(self.y, axes=(1, 2))
        g = self.g
        fview(p)[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, g.Hz:g.Hz + g.Nz] = self.y.real


class _NumpyDistTridiagonal(_NumpyDistPoisson):
    """z Bounded: FFT_y on the slab -> y->x all-to-all -> FFT_x -> Thomas sweep in z with this rank's ky range -> zero-mean
    gauge on the (0,0) column (rank 0) -> IFFT_x -> x->y all-to-all -> IFFT_y -> real part
    (distributed_fft_tridiagonal_solver.jl:260-292 with the z-local pencil transposes elided: z is local throughout)."""

    def __init__(self, grid, arch):
        self.g, self.R, self.rank = grid, arch.partition.x, arch.local_rank
        nx, Ny, Nz = grid.Nx, grid.Ny, grid.Nz
        self.ny, self.Nxg = Ny // self.R, nx * self.R
        self.y = np.zeros((nx, Ny, Nz), dtype=np.complex128, order="F")
        self.x = np.zeros((self.Nxg, self.ny, Nz), dtype=np.complex128, order="F")
        n = nx * Ny * Nz * 2
        self.send, self.recv = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        zf = getattr(grid, "z_faces", None)
        z = (0.0, grid.Lz) if zf is None else np.asarray(zf)[grid.Hz:grid.Hz + Nz + 1]
        halo = (grid.Hx, grid.Hy, grid.Hz)
        self.og = O.Grid((nx, Ny, Nz), x=(0, grid.Lx), y=(0, grid.Ly), z=z, topology="PPB", halo=halo)
        self.local = O.FourierTridiagonalPoissonSolver(self.og)
        # the x-local layout as a grid of its own: Nxg global x modes, this rank's ny stored ky
        xg = O.Grid((self.Nxg, self.ny, Nz), x=(0, grid.global_Lx), y=(0, grid.Ly), z=z, topology="PPB",
                    halo=(min(halo[0], self.Nxg), min(halo[1], self.ny), halo[2]))
        self.xs = O.FourierTridiagonalPoissonSolver(xg)
        lx = O.poisson_eigenvalues(self.Nxg, grid.global_Lx, O.PERIODIC)
        ly = np.ascontiguousarray(O.poisson_eigenvalues(Ny, grid.Ly, O.PERIODIC)[self.rank * self.ny:(self.rank + 1) * self.ny])
        O.lib().ocn_oracle_main_diagonal_z(O.C.byref(self.xs._cg), lx.ctypes.data_as(O.C.c_void_p), ly.ctypes.data_as(O.C.c_void_p),
                                           self.xs.D.ctypes.data_as(O.C.c_void_p))

    def source_term(self, u, v, w, dt):
        self.local.source_term(np.asfortranarray(fview(u)), np.asfortranarray(fview(v)), np.asfortranarray(fview(w)), dt)
        self.y[...] = self.local.source

    def forward_yz(self):
        self.y[...] = sfft.fft(self.y, axis=1)

    def solve_x(self):
        xh = np.asfortranarray(sfft.fft(self.x, axis=0))
        phi = O.batched_tridiagonal_solve_z(self.xs.a, self.xs.D, self.xs.a, xh)
        if self.rank == 0:
            phi[0, 0, :] -= np.mean(phi[0, 0, :])
        self.x[...] = sfft.ifft(phi, axis=0)

    def backward_yz(self, p):
        self.y[...] = sfft.ifft(self.y, axis=1)
        g = self.g
        fview(p)[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, g.Hz:g.Hz + g.Nz] = self.y.real


class _NumpyDistSlab(_NumpyDistPoisson):
    """The slab pipeline of libocn_hip's distributed FFT solver (ocn_hip.h, ocn_dist_poisson_pipeline == 1) restated with numpy:
    real transform along y and complex along z on the slab, the half spectrum handed to the all-to-all as
    send[d][ky + NyH (kz_l + cz xl)] (kz = d cz + kz_l), x transform + division + inverse x transform in place on
    recv[xg][kz_l][ky], the return exchange recv -> send, inverse z and inverse real y into p.  (The library keeps kz and kx in
    its column kernels' stage order; any consistent order works -- here the natural one.)"""
    fast = 1

    def __init__(self, grid, arch):
        super().__init__(grid, arch)
        nx, Ny, Nz = grid.Nx, grid.Ny, grid.Nz
        assert Nz % self.R == 0
        self.NyH, self.cz = Ny // 2 + 1, Nz // self.R
        n = self.NyH * nx * Nz * 2
        self.send, self.recv = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)

    def forward_yz(self):
        nx, Nz, R, cz, NyH = self.g.Nx, self.g.Nz, self.R, self.cz, self.NyH
        A = sfft.fft(sfft.rfft(self.y.real, axis=1), axis=2)                       # (nx, NyH, Nz)
        arr = np.transpose(A.reshape(nx, NyH, R, cz), (2, 0, 3, 1))                # (d, xl, kz_l, ky): ky fastest in C order
        self._cbuf(self.send)[...] = np.ascontiguousarray(arr).ravel()

    def solve_x(self):
        nx, R, cz, NyH = self.g.Nx, self.R, self.cz, self.NyH
        X = self._cbuf(self.recv).reshape(R * nx, cz, NyH)                          # (xg = r nx + xl, kz_l, ky), in place
        xh = sfft.fft(X, axis=0)
        lam = (self.lx[:, None, None] + self.ly[None, None, :NyH]) + self.lz[None, self.rank * cz:(self.rank + 1) * cz, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            xh = -xh / lam
        if self.rank == 0:
            xh[0, 0, 0] = 0.0
        X[...] = sfft.ifft(xh, axis=0)

    def backward_yz(self, p):
        nx, Ny, Nz, R, cz, NyH = self.g.Nx, self.g.Ny, self.g.Nz, self.R, self.cz, self.NyH
        arr = self._cbuf(self.send).reshape(R, nx, cz, NyH)                         # (d, xl, kz_l, ky)
        A = np.transpose(arr, (1, 3, 0, 2)).reshape(nx, NyH, Nz)
        y = sfft.irfft(sfft.ifft(A, axis=2), n=Ny, axis=1)
        g = self.g
        fview(p)[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, g.Hz:g.Hz + g.Nz] = y


class _NumpyDistXTri(_NumpyDistPoisson):
    """The transpose-free pipeline of libocn_hip (ocn_dist_poisson_pipeline == 3, csrc/xtri.hip) restated with numpy: real y and
    complex z transforms on the slab; per (ky, kz) mode the cyclic tridiagonal system p[i-1] - (2 + mu) p[i] + p[i+1] = dx^2 F[i],
    mu = dx^2 (ly + lz) -- the operator whose x-FFT eigenvalues the reference divides by (poisson_eigenvalues.jl:8-31) -- by local
    Thomas solves, an all-gather of the blocks' first / last values (gsend -> grecv) and the circulant interface system; the
    (0, 0) mode by prefix sums with the means removed.  Vectorised over the modes, loops over xl."""
    fast = 3

    def __init__(self, grid, arch):
        super().__init__(grid, arch)
        nx, Ny, Nz = grid.Nx, grid.Ny, grid.Nz
        self.NyH = Ny // 2 + 1
        self.M = self.NyH * Nz
        self.send = None
        self.A = np.zeros((nx, self.NyH, Nz), dtype=np.complex128)
        self.gsend = torch.zeros(2 * (2 * self.M + nx), dtype=torch.float64)
        self.grecv = torch.zeros(2 * (2 * self.M + nx) * self.R, dtype=torch.float64)
        dx = grid.dx
        self.dx2 = dx * dx
        mu = self.dx2 * (self.ly[:self.NyH, None] + self.lz[None, :])     # (NyH, Nz)
        mu[0, 0] = 1.0                                                      # singular mode: finite stand-in, its line is replaced
        self.mu = mu
        sq = np.sqrt(mu * (mu + 4))
        self.r = 2 / ((2 + mu) + sq)

    def forward_yz(self):
        nx, M = self.g.Nx, self.M
        F = self.dx2 * sfft.fft(sfft.rfft(self.y.real, axis=1), axis=2)   # (nx, NyH, Nz); the unnormalised inverse is irfft / ifft below
        gs = self._cbuf(self.gsend)
        gs[2 * M:] = F[:, 0, 0]
        b = -(2 + self.mu)
        c = np.zeros((nx,) + b.shape)
        d = np.zeros_like(F)
        c[0] = 1 / b
        d[0] = F[0] / b
        for i in range(1, nx):
            c[i] = 1 / (b - c[i - 1])
            d[i] = (F[i] - d[i - 1]) * c[i]
        x = np.zeros_like(F)
        x[nx - 1] = d[nx - 1]
        for i in range(nx - 2, -1, -1):
            x[i] = d[i] - c[i] * x[i + 1]
        self.A[...] = x
        gs[0:2 * M:2] = x[0].ravel()
        gs[1:2 * M:2] = x[nx - 1].ravel()

    def solve_x(self):
        nx, M, R, n = self.g.Nx, self.M, self.R, self.g.Nx
        gr = self._cbuf(self.grecv).reshape(R, 2 * M + nx)
        shp = self.mu.shape
        g1 = gr[:, 0:2 * M:2].reshape((R,) + shp)
        gn = gr[:, 1:2 * M:2].reshape((R,) + shp)
        r = self.r
        den = 1 - r ** (2 * (n + 1))
        v1 = -(r - r ** (2 * n + 1)) / den
        vn = -(r ** n - r ** (n + 2)) / den
        # a_s + v1 z_(s-1) + vn a_(s+1) = g1_s;  z_s + vn z_(s-1) + v1 a_(s+1) = gn_s  (cyclic): DFT over s
        g1h, gnh = sfft.fft(g1, axis=0) / R, sfft.fft(gn, axis=0) / R      # xh_k = (1/R) sum_s x_s w^(-k s)
        k = np.arange(R).reshape((R,) + (1,) * len(shp))
        w = np.exp(2j * np.pi * k / R)
        A11, A12, A21, A22 = 1 + vn * w, v1 / w, v1 * w, 1 + vn / w
        det = A11 * A22 - A12 * A21
        ah = (A22 * g1h - A12 * gnh) / det
        zh = (A11 * gnh - A21 * g1h) / det
        a = sfft.ifft(ah, axis=0) * R                                      # x_s = sum_k xh_k w^(k s)
        z = sfft.ifft(zh, axis=0) * R
        x0, xn1 = z[(self.rank - 1) % R], a[(self.rank + 1) % R]
        i = np.arange(1, n + 1).reshape((n,) + (1,) * len(shp))
        v = -(r ** i - r ** (2 * (n + 1) - i)) / den
        wsp = -(r ** (n + 1 - i) - r ** (n + 1 + i)) / den
        self.A[...] = self.A - v * x0 - wsp * xn1
        # the (0, 0) line over the global x extent
        F = gr[:, 2 * M:].ravel()
        N = F.size
        Fp = F - F.mean()
        sc = np.cumsum(Fp)
        dline = -sc.sum() / N + sc
        pl = np.concatenate([[0], np.cumsum(dline)[:-1]])
        pl = pl - pl.mean()
        self.A[:, 0, 0] = pl[self.rank * n:(self.rank + 1) * n]

    def backward_yz(self, p):
        Ny = self.g.Ny
        y = sfft.irfft(sfft.ifft(self.A, axis=2), n=Ny, axis=1)
        g = self.g
        fview(p)[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, g.Hz:g.Hz + g.Nz] = y


class _NumpyDistSlabTridiagonal(_NumpyDistTridiagonal):
    """The tridiagonal flavour of the slab pipeline (ocn_dist_poisson_pipeline == 2): the half spectrum is partitioned by ky in R
    zero-padded chunks of c = ceil(NyH / R); send[d][ky_l + c (z + Nz xl)]; after the exchange recv[xg][z][ky_l]: FFT_x, Thomas
    sweep in z with this rank's ky range, zero-mean gauge on rank 0, IFFT_x into `send`; exchanged back (send -> recv again),
    inverse real y into p."""
    fast = 2

    def __init__(self, grid, arch):
        self.g, self.R, self.rank = grid, arch.partition.x, arch.local_rank
        nx, Ny, Nz = grid.Nx, grid.Ny, grid.Nz
        self.Nxg, self.NyH = nx * self.R, Ny // 2 + 1
        self.c = -(-self.NyH // self.R)
        self.y = np.zeros((nx, Ny, Nz), dtype=np.complex128, order="F")
        n = self.c * self.R * Nz * nx * 2
        self.send, self.recv = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        zf = getattr(grid, "z_faces", None)
        z = (0.0, grid.Lz) if zf is None else np.asarray(zf)[grid.Hz:grid.Hz + Nz + 1]
        halo = (grid.Hx, grid.Hy, grid.Hz)
        self.og = O.Grid((nx, Ny, Nz), x=(0, grid.Lx), y=(0, grid.Ly), z=z, topology="PPB", halo=halo)
        self.local = O.FourierTridiagonalPoissonSolver(self.og)
        xg = O.Grid((self.Nxg, self.c, Nz), x=(0, grid.global_Lx), y=(0, grid.Ly), z=z, topology="PPB",
                    halo=(min(halo[0], self.Nxg), min(halo[1], self.c), halo[2]))
        self.xs = O.FourierTridiagonalPoissonSolver(xg)
        lx = O.poisson_eigenvalues(self.Nxg, grid.global_Lx, O.PERIODIC)
        lyp = np.ones(self.c * self.R)
        lyp[:self.NyH] = O.poisson_eigenvalues(Ny, grid.Ly, O.PERIODIC)[:self.NyH]
        ly = np.ascontiguousarray(lyp[self.rank * self.c:(self.rank + 1) * self.c])
        O.lib().ocn_oracle_main_diagonal_z(O.C.byref(self.xs._cg), lx.ctypes.data_as(O.C.c_void_p), ly.ctypes.data_as(O.C.c_void_p),
                                           self.xs.D.ctypes.data_as(O.C.c_void_p))

    def forward_yz(self):
        nx, Nz, R, c, NyH = self.g.Nx, self.g.Nz, self.R, self.c, self.NyH
        A = np.zeros((nx, R * c, Nz), dtype=np.complex128)
        A[:, :NyH] = sfft.rfft(self.y.real, axis=1)
        arr = np.transpose(A.reshape(nx, R, c, Nz), (1, 0, 3, 2))                  # (d, xl, z, ky_l): ky_l fastest in C order
        self._cbuf(self.send)[...] = np.ascontiguousarray(arr).ravel()

    def solve_x(self):
        nx, Nz, R, c = self.g.Nx, self.g.Nz, self.R, self.c
        X = self._cbuf(self.recv).reshape(R * nx, Nz, c)                            # (xg, z, ky_l)
        xh = np.asfortranarray(np.transpose(sfft.fft(X, axis=0), (0, 2, 1)))        # (kx, ky_l, z) for the oracle's sweep
        phi = O.batched_tridiagonal_solve_z(self.xs.a, self.xs.D, self.xs.a, xh)
        if self.rank == 0:
            phi[0, 0, :] -= np.mean(phi[0, 0, :])
        out = np.transpose(sfft.ifft(phi, axis=0), (0, 2, 1))                       # back to (xg, z, ky_l)
        self._cbuf(self.send)[...] = np.ascontiguousarray(out).ravel()

    def backward_yz(self, p):
        nx, Ny, Nz, R, c, NyH = self.g.Nx, self.g.Ny, self.g.Nz, self.R, self.c, self.NyH
        arr = self._cbuf(self.recv).reshape(R, nx, Nz, c)                           # (d, xl, z, ky_l)
        A = np.transpose(arr, (1, 0, 3, 2)).reshape(nx, R * c, Nz)[:, :NyH]
        y = sfft.irfft(A, n=Ny, axis=1)
        g = self.g
        fview(p)[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, g.Hz:g.Hz + g.Nz] = y


class NumpyOps:
    name = "numpy"

    def __init__(self, slab=False, xtri=False):
        self.slab = slab  # restate the library's slab pipelines instead of the transposing reference choreography
        self.xtri = xtri  # restate the transpose-free pipeline (periodic z)

    def new_buffer(self, arch, n):
        return torch.zeros(n, dtype=torch.float64)

    def local_fill(self, grid, fields, fbnv):
        bounded_z = grid.topology[2] == "Bounded"
        if bounded_z:
            og = O.Grid((grid.Nx, grid.Ny, grid.Nz), x=(0, 1), y=(0, 1), z=(0, 1), topology="PPB", halo=(grid.Hx, grid.Hy, grid.Hz))
        for f in fields:  # y, z fills over the whole parent cross-section; x is communication
            a = fview(f)
            sx, sy, sz = a.shape
            if bounded_z:  # impenetrable w (if asked) and no-flux for the rest, before the periodic fills
                tmp = np.asfortranarray(a)
                if f.loc == 4:
                    if fbnv:
                        O.lib().ocn_oracle_fill_open(og.cref, f.loc, tmp.ctypes.data_as(O.C.c_void_p), 2)
                else:
                    O.lib().ocn_oracle_fill_flux(og.cref, f.loc, tmp.ctypes.data_as(O.C.c_void_p), 2)
                a[...] = tmp
            for d, (N, H) in ((1, (grid.Ny, grid.Hy)), (2, (grid.Nz, grid.Hz))):
                if d == 2 and bounded_z:
                    continue
                tmp = np.asfortranarray(a)
                O.lib().ocn_oracle_fill_periodic(tmp.ctypes.data_as(O.C.c_void_p), sx, sy, sz, d, N, H)
                a[...] = tmp

    def pack_x(self, grid, f, west, east):
        a, H, nx = fview(f), grid.Hx, grid.Nx
        west.numpy()[...] = a[H:2 * H].ravel(order="F")        # parent[1+Hx : 2Hx, :, :]
        east.numpy()[...] = a[nx:nx + H].ravel(order="F")      # parent[1+nx : nx+Hx, :, :]

    def unpack_x(self, grid, f, west, east):
        a, H, nx = fview(f), grid.Hx, grid.Nx
        shp = (H,) + a.shape[1:]
        a[0:H] = west.numpy().reshape(shp, order="F")           # parent[1 : Hx]
        a[nx + H:nx + 2 * H] = east.numpy().reshape(shp, order="F")  # parent[1+nx+Hx : nx+2Hx]

    def plane_x(self, grid, f, which, buf, unpack):
        a, H, nx = fview(f), grid.Hx, grid.Nx
        if not unpack:   # first / last interior plane
            buf.numpy()[...] = a[H + nx - 1 if which else H].ravel(order="F")
        else:            # the halo plane next to it
            a[H + nx if which else H - 1] = buf.numpy().reshape(a.shape[1:], order="F")

    def sync(self):
        pass

    def make_dist_poisson(self, grid, arch):
        if grid.topology[2] == "Bounded":
            return (_NumpyDistSlabTridiagonal if self.slab else _NumpyDistTridiagonal)(grid, arch)
        if self.xtri:
            return _NumpyDistXTri(grid, arch)
        return (_NumpyDistSlab if self.slab and grid.Nz % arch.partition.x == 0 else _NumpyDistPoisson)(grid, arch)
