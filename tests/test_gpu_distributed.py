"""Slab-x distributed model on ONE GPU: R ranks run as threads of this process and exchange through an in-process
fabric, so the real HIP pack/unpack/transpose/FFT kernels and the real choreography (async halo exchange overlapped with
interior tendencies, buffer tendencies, distributed FFT solve) are exercised and compared with the single-rank model.
(RCCL itself needs one GPU per rank; the fabric-over-torch.distributed path is covered by tests/test_distributed_gloo.py.)"""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class ThreadWorld:
    def __init__(self, R):
        self.R = R
        self.barrier = threading.Barrier(R, timeout=120)
        self.lock = threading.Lock()
        self.mail = {}

    def put(self, key, t):
        with self.lock:
            self.mail.setdefault(key, []).append(t)

    def get(self, key):
        with self.lock:
            return self.mail[key].pop(0)


class ThreadFabric:
    def __init__(self, world, rank):
        self.world, self.rank, self.size = world, rank, world.R

    def start_exchange(self, sends, recvs):
        for t, dst in sends:
            self.world.put(("p2p", self.rank, dst), t.clone())
        return recvs

    def wait(self, recvs):
        torch.cuda.synchronize()
        self.world.barrier.wait()
        for t, src in recvs:
            t.copy_(self.world.get(("p2p", src, self.rank)))
        torch.cuda.synchronize()
        self.world.barrier.wait()

    def all_to_all(self, recv, send):
        n = send.numel() // self.size
        for m in range(self.size):
            self.world.put(("a2a", self.rank, m), send[m * n:(m + 1) * n].clone())
        torch.cuda.synchronize()
        self.world.barrier.wait()
        for m in range(self.size):
            recv[m * n:(m + 1) * n].copy_(self.world.get(("a2a", m, self.rank)))
        torch.cuda.synchronize()
        self.world.barrier.wait()


def _run_ranks(R, fn):
    world = ThreadWorld(R)
    out, errs = [None] * R, []

    def target(r):
        try:
            torch.cuda.set_device(0)
            out[r] = fn(r, ThreadFabric(world, r))
        except Exception as e:  # noqa
            import traceback
            errs.append(f"rank {r}: {traceback.format_exc()}")
            world.barrier.abort()

    threads = [threading.Thread(target=target, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errs, "\n".join(errs)
    return out


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("topo", ["PPP", "PPB", "PBB", "BBB", "PBB-wide", "BBB-wide"])
def test_distributed_steps_match_single_rank(ocn, oracle, R, topo):
    """Two RK3 steps on R slab-x ranks against the single-rank model AND, directly, against the CPU oracle; "PPB" = stretched Bounded
    z, i.e. the distributed Fourier-tridiagonal solver (config 4's solver at 1 -> 8 GPUs); "PBB" = the channel: walls in y too (cosine
    transforms in y, the direction-generic kernels on every slab; distributed_grids.jl:75-118); "BBB" = the closed box: the partitioned x is
    Bounded as well -- the first slab is RightConnected (wall on its west side), the last LeftConnected (wall on its east side, u carries
    the wall face), the ones between FullyConnected; cosine transforms along x after the transpose (the four topologies of
    test_distributed_poisson_solvers.jl:128-148)."""
    from helpers import stretched_faces
    P = "Periodic"
    N = (32, 16, 12)
    if topo.endswith("-wide"):  # slabs whose interior range (nx - 2 Hx columns) keeps an interior box: tiled kernels on a RANGE + wall frames
        if R != 2:
            pytest.skip("one wide case")
        topo, N = topo[:3], (64, 20, 10)
    if topo == "PPP":
        ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
        solver_class = ocn.DistributedFFTBasedPoissonSolver
    else:
        names = {"P": P, "B": "Bounded"}
        ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=stretched_faces(N[2], 2.0), topology=tuple(names[t] for t in topo), halo=(3, 3, 3))
        solver_class = ocn.DistributedFourierTridiagonalPoissonSolver
    rng = np.random.default_rng(1234)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    if topo != "PPP":
        init["w"] = rng.uniform(-1, 1, (N[0], N[1], N[2] + 1))
    if topo in ("PBB", "BBB"):
        init["v"] = rng.uniform(-1, 1, (N[0], N[1] + 1, N[2]))
    if topo == "BBB":
        init["u"] = rng.uniform(-1, 1, (N[0] + 1, N[1], N[2]))
    dt = 0.01
    nxl = N[0] // R

    def xs(name, r):
        """the x range of rank r's interior of field `name` in the global array: u of the last slab of a Bounded x carries the wall face"""
        return slice(r * nxl, (r + 1) * nxl + (1 if (topo == "BBB" and name in ("u", "Gu") and r == R - 1) else 0))

    ocn.set_math_mode(ocn.MATH_STRICT)
    sg = ocn.RectilinearGrid(ocn.GPU(), size=N, **ext)
    sm = ocn.NonhydrostaticModel(sg, advection=ocn.WENO())
    ocn.set(sm, **init)
    for _ in range(2):
        ocn.time_step(sm, dt)
    ocn.sync_device()
    ref = [f.interior() for f in sm.velocities] + [sm.pNHS.interior()]
    refG = [f.interior() for f in sm.timestepper.Gn]

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        want = "FullyConnected" if (topo != "BBB" or 0 < r < R - 1) else ("RightConnected" if r == 0 else "LeftConnected")
        assert g.Nx == N[0] // R and g.topology[0] == want
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
        assert isinstance(m.pressure_solver, solver_class)
        ocn.set(m, **{k: v[xs(k, r)] for k, v in init.items()})
        for _ in range(2):
            ocn.time_step(m, dt)
        ocn.sync_device()
        return [f.interior() for f in m.velocities] + [m.pNHS.interior()], [f.interior() for f in m.timestepper.Gn]

    outs = _run_ranks(R, rank_main)
    nx = N[0] // R
    scale = max(np.abs(a).max() for a in ref[:3])
    for r, (fields, G) in enumerate(outs):
        for a, b, name in zip(fields, ref, ("u", "v", "w", "p")):
            tol = 1e-11 * scale if name != "p" else 1e-10 * max(1.0, np.abs(ref[3]).max())
            assert np.abs(a - b[xs(name, r)]).max() <= tol, f"rank {r} field {name}"
        for a, b, name in zip(G, refG, "uvw"):
            b = b[xs(name, r)]
            if topo != "PPP" and name == "w":
                # the wall face k = 1 of Gw: the serial launch excludes the periphery, a KernelParameters launch (distributed)
                # writes it (kernel_launching.jl:236-240); no kernel ever reads it
                a, b = a[:, :, 1:], b[:, :, 1:]
            if topo in ("PBB", "BBB") and name == "v":
                a, b = a[:, 1:, :], b[:, 1:, :]
            if topo == "BBB" and name == "u" and r == 0:
                a, b = a[1:], b[1:]
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    # the same two steps on the CPU oracle: the distributed result is compared with it directly, not only through the single-rank model
    O = oracle
    og = O.Grid(N, x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi) if topo == "PPP" else ext["z"], topology=topo, halo=(3, 3, 3))
    om = O.NonhydrostaticModel(og)
    om.set(**init)
    for _ in range(2):
        om.time_step(dt)
    for r, (fields, _) in enumerate(outs):
        for a, b, name in zip(fields[:3], (om.u, om.v, om.w), "uvw"):
            assert np.abs(a - og.interior(b)[xs(name, r)]).max() <= 1e-11 * scale, f"rank {r} field {name} vs the oracle"


@pytest.mark.parametrize("topo", ["PPP", "PPB", "PBB", "BBB", "BBB-faces"])
@pytest.mark.parametrize("size", [(44, 44, 8), (16, 44, 8)])
def test_divergence_free_poisson_solution_on_four_ranks(ocn, oracle, size, topo):
    """test_distributed_poisson_solvers.jl:34-89, 128-136 re-expressed with its sizes, its (4, 1, 1) ranks and its four topologies (and
    :91-126, 150-154, "BBB-faces": the tridiagonal solver's own test -- z given as an array of faces, halo (2, 2, 2)): a
    random velocity on every slab (halos filled across the slabs), R = div U, solve_for_pressure! with Δt = 1, then ∇²ϕ ≈ R on every rank
    (the reference's `≈`: sqrt(eps) relative) -- and the assembled pressure equals the oracle's single-process solve to 1e-10.  (A Bounded
    z takes the Fourier-tridiagonal solver here, an exact solver of the same discrete system; 44 is not a length of the column kernels:
    rocFFT lines with the cosine-transform passes around them.)"""
    O = oracle
    R = 4
    names = {"P": "Periodic", "B": "Bounded"}
    faces = topo.endswith("-faces")
    topo = topo[:3]
    H = 2 if faces else 3
    z = np.linspace(0, 2 * np.pi, size[2] + 1) if faces else (0, 2 * np.pi)  # collect(range(0, 2π, length = Nz + 1)) (:101-103)
    ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=tuple(names[t] for t in topo), halo=(H, H, H))
    og = O.Grid(size, x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=topo, halo=(H, H, H))
    rng = np.random.default_rng(1234)
    U = []
    for loc in (1, 2, 4):
        a = og.zeros(loc)
        og.interior(a)[...] = rng.random(og.interior(a).shape)
        O.fill_halo_regions(og, a, loc)
        U.append(a)
    Rhs = O.divergence(og, *U)
    S = O.FourierTridiagonalPoissonSolver(og) if topo[2] == "B" else O.FFTBasedPoissonSolver(og)
    p0 = og.zeros(0)
    S.source_term(*U, 1.0)
    S.solve(p0)
    nxl = size[0] // R

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=size, **ext)
        fields = [ocn.XFaceField(g), ocn.YFaceField(g), ocn.ZFaceField(g)]
        for f, a, loc in zip(fields, U, (1, 2, 4)):
            n = f.interior().shape[0]  # nxl, or nxl + 1 for u on the last slab of a Bounded x
            f.set(og.interior(a)[r * nxl:r * nxl + n])
        ocn.fill_halo_regions(tuple(fields))
        solver = ocn.nonhydrostatic_pressure_solver(g)
        phi = ocn.CenterField(g)
        ocn.solve_for_pressure(phi, solver, 1.0, fields)
        ocn.fill_halo_regions(phi)
        ocn.sync_device()
        return phi.parent()

    for r, parent in enumerate(_run_ranks(R, rank_main)):
        # the Laplacian of the slab's solution from its parent array: x halos as exchanged, y / z halos as filled
        interior = parent[H:-H, H:-H, H:-H]
        sl = slice(r * nxl, (r + 1) * nxl)
        ref = og.interior(p0)[sl]
        assert np.abs(interior - ref).max() <= 1e-10 * max(1.0, np.abs(p0).max()), f"rank {r}"
        dx, dy, dz = (2 * np.pi / n for n in size)
        P = parent
        c = P[H:-H, H:-H, H:-H]
        lap = ((P[H + 1:-H + 1 or None, H:-H, H:-H] - 2 * c + P[H - 1:-H - 1, H:-H, H:-H]) / dx ** 2
               + (P[H:-H, H + 1:-H + 1 or None, H:-H] - 2 * c + P[H:-H, H - 1:-H - 1, H:-H]) / dy ** 2
               + (P[H:-H, H:-H, H + 1:-H + 1 or None] - 2 * c + P[H:-H, H:-H, H - 1:-H - 1]) / dz ** 2)
        assert np.linalg.norm(lap - Rhs[sl]) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(Rhs[sl]), f"rank {r}: the Laplacian of the solution"


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("topo", ["PPP", "PPB"])
def test_distributed_slab_pipeline_matches_single_rank(ocn, R, topo, monkeypatch):
    """Sizes the library's slab pipeline covers (real y transform, z column FFT into the exchange layout, fused x column
    kernel -- or, for a Bounded stretched z, the ky-partitioned layout with FFT_x and the Thomas sweep; csrc/colfft.hip): the
    handle must select it, and two RK3 steps must match the single-rank model, whose solver is a different code path (rocFFT
    or the row / column pipeline).  "PPB": Ny/2 + 1 = 65 is not a multiple of R, so the zero padding of the exchange is covered."""
    from helpers import stretched_faces
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "0")  # the all-to-all pipeline (the transpose-free one has its own test below)
    P = "Periodic"
    if topo == "PPP":
        N = (64, 128, 64)
        ext = dict(x=(0, 2 * np.pi), y=(0, 4 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
    else:
        N = (64, 128, 12)
        ext = dict(x=(0, 2 * np.pi), y=(0, 4 * np.pi), z=stretched_faces(N[2], 2.0), topology=(P, P, "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(4321)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    if topo == "PPB":
        init["w"] = rng.uniform(-1, 1, (N[0], N[1], N[2] + 1))
    dt = 0.005
    ocn.set_math_mode(ocn.MATH_STRICT)
    sg = ocn.RectilinearGrid(ocn.GPU(), size=N, **ext)
    sm = ocn.NonhydrostaticModel(sg, advection=ocn.WENO())
    ocn.set(sm, **init)
    for _ in range(2):
        ocn.time_step(sm, dt)
    ocn.sync_device()
    ref = [f.interior() for f in sm.velocities] + [sm.pNHS.interior()]

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
        assert m.pressure_solver.impl.fast == (1 if topo == "PPP" else 2)
        sl = slice(r * g.Nx, (r + 1) * g.Nx)
        ocn.set(m, **{k: v[sl] for k, v in init.items()})
        for _ in range(2):
            ocn.time_step(m, dt)
        ocn.sync_device()
        return [f.interior() for f in m.velocities] + [m.pNHS.interior()]

    outs = _run_ranks(R, rank_main)
    nx = N[0] // R
    scale = max(np.abs(a).max() for a in ref[:3])
    for r, fields in enumerate(outs):
        sl = slice(r * nx, (r + 1) * nx)
        for a, b, name in zip(fields, ref, ("u", "v", "w", "p")):
            tol = 1e-11 * scale if name != "p" else 1e-10 * max(1.0, np.abs(ref[3]).max())
            assert np.abs(a - b[sl]).max() <= tol, f"rank {r} field {name}: {np.abs(a - b[sl]).max()}"


@pytest.mark.parametrize("R,Nx", [(1, 64), (2, 64), (4, 64), (8, 64), (2, 6), (4, 48), (8, 24)])
def test_distributed_transpose_free_pipeline_matches_single_rank(ocn, R, Nx, monkeypatch):
    """ocn_dist_poisson_pipeline = 3 (csrc/xtri.hip): no transposes -- the x direction is the cyclic tridiagonal system the
    reference inverts with FFT_x and the eigenvalue division (distributed_fft_based_poisson_solver.jl:141-178), solved by local
    Thomas sweeps, ONE all-gather of two numbers per (ky, kz) mode and a circulant interface system.  Two RK3 steps must match the
    single-rank model (FFT in x: a different algorithm for the same operator) for 1 (a rank that is its own neighbour), 2, 4, 8
    ranks, blocks of 3 to 64 columns and an x extent that is not a power of two."""
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1")
    P = "Periodic"
    N = (Nx, 128, 64)
    ext = dict(x=(0, 2 * np.pi), y=(0, 4 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
    rng = np.random.default_rng(977 + R)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    dt = 0.005
    ocn.set_math_mode(ocn.MATH_STRICT)
    sg = ocn.RectilinearGrid(ocn.GPU(), size=N, **ext)
    sm = ocn.NonhydrostaticModel(sg, advection=ocn.WENO())
    ocn.set(sm, **init)
    for _ in range(2):
        ocn.time_step(sm, dt)
    ocn.sync_device()
    ref = [f.interior() for f in sm.velocities] + [sm.pNHS.interior()]

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric, force_communication=(R == 1))
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
        assert m.pressure_solver.impl.fast == 3
        sl = slice(r * g.Nx, (r + 1) * g.Nx)
        ocn.set(m, **{k: v[sl] for k, v in init.items()})
        for _ in range(2):
            ocn.time_step(m, dt)
        ocn.sync_device()
        return [f.interior() for f in m.velocities] + [m.pNHS.interior()]

    outs = _run_ranks(R, rank_main)
    nx = N[0] // R
    scale = max(np.abs(a).max() for a in ref[:3])
    for r, fields in enumerate(outs):
        sl = slice(r * nx, (r + 1) * nx)
        for a, b, name in zip(fields, ref, ("u", "v", "w", "p")):
            tol = 1e-11 * scale if name != "p" else 1e-10 * max(1.0, np.abs(ref[3]).max())
            assert np.abs(a - b[sl]).max() <= tol, f"rank {r} field {name}: {np.abs(a - b[sl]).max()}"


@pytest.mark.parametrize("Lx,Ly,Lz", [(2.0e3, 0.5, 3.0), (2.0e-2, 50.0, 7.0), (1.0, 1.0, 1.0e-3)])
def test_transpose_free_poisson_extreme_aspect_ratios(ocn, Lx, Ly, Lz, monkeypatch):
    """The cyclic tridiagonal x solve over the whole range of mu = dx^2 (ly + lz): mu ~ 1e7 (r^n underflows, the blocks decouple),
    mu ~ 1e-9 for the gravest modes (r -> 1, the closed forms go through expm1 / log1p), and a thin z.  solve_for_pressure! on 4 ranks
    against the single-rank FFT solver on the same velocities: relative 20 eps x (largest / smallest eigenvalue) of max|p| (the
    conditioning of the eigenvalue division itself), and the discrete Laplacian of the distributed p reproduces the source term."""
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1")
    P, R, N = "Periodic", 4, (48, 128, 64)
    ext = dict(x=(0, Lx), y=(0, Ly), z=(0, Lz), topology=(P, P, P), halo=(3, 3, 3))
    rng = np.random.default_rng(2024)
    init = [rng.uniform(-1, 1, N) for _ in range(3)]
    sg = ocn.RectilinearGrid(ocn.GPU(), size=N, **ext)
    U = [ocn.Field(loc, sg) for loc in (1, 2, 4)]
    for f, a in zip(U, init):
        f.set(a)
    ocn.fill_halo_regions(U)
    ps = ocn.CenterField(sg)
    ocn.solve_for_pressure(ps, ocn.nonhydrostatic_pressure_solver(sg), 0.7, U)
    ocn.sync_device()
    ref = ps.interior()

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        sl = slice(r * g.Nx, (r + 1) * g.Nx)
        Ul = [ocn.Field(loc, g) for loc in (1, 2, 4)]
        for f, a in zip(Ul, init):
            f.set(a[sl])
        ocn.fill_halo_regions(Ul)
        solver = ocn.nonhydrostatic_pressure_solver(g)
        assert solver.impl.fast == 3
        p = ocn.CenterField(g)
        ocn.solve_for_pressure(p, solver, 0.7, Ul)
        ocn.sync_device()
        return p.interior()

    outs = _run_ranks(R, rank_main)
    got = np.concatenate(outs, axis=0)
    assert np.isfinite(got).all()
    # both solvers divide by the same eigenvalues: what they may differ by is eps x the condition number of the operator
    dx, dy, dz = Lx / N[0], Ly / N[1], Lz / N[2]
    lam_max = 4 / dx ** 2 + 4 / dy ** 2 + 4 / dz ** 2
    lam_min = min((2 * np.sin(np.pi / n) / d) ** 2 for n, d in zip(N, (dx, dy, dz)))
    tol = max(1e-10, 20 * np.finfo(float).eps * lam_max / lam_min)
    assert tol < 1e-3
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (np.abs(got - ref).max() / np.abs(ref).max(), tol)
    # residual: the periodic 7-point Laplacian of p equals div(U) / dt
    lap = ((np.roll(got, -1, 0) - 2 * got + np.roll(got, 1, 0)) / dx ** 2 + (np.roll(got, -1, 1) - 2 * got + np.roll(got, 1, 1)) / dy ** 2
           + (np.roll(got, -1, 2) - 2 * got + np.roll(got, 1, 2)) / dz ** 2)
    u, v, w = init
    div = ((np.roll(u, -1, 0) - u) / dx + (np.roll(v, -1, 1) - v) / dy + (np.roll(w, -1, 2) - w) / dz) / 0.7
    assert np.abs(lap - div).max() <= tol * np.abs(div).max(), (np.abs(lap - div).max() / np.abs(div).max(), tol)


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("closure,stepper,ytopo", [("constant", "RungeKutta3", "Periodic"), ("AMD", "RungeKutta3", "Periodic"),
                                                   ("AMD", "QuasiAdamsBashforth2", "Periodic"), ("constant", "RungeKutta3", "Bounded"),
                                                   ("AMD", "RungeKutta3", "Bounded"), ("constant", "RungeKutta3", "box"),
                                                   ("AMD", "RungeKutta3", "box"), ("AMD", "QuasiAdamsBashforth2", "box")])
def test_distributed_ocean_mixing_physics_matches_single_rank(ocn, R, closure, stepper, ytopo):
    """Config 4's physics (buoyancy, Coriolis, diffusivity, flux / gradient boundary conditions; the LES closure as written or
    replaced by a constant ScalarDiffusivity) on R slab-x ranks against the single-rank model: 2 steps.  The ranks compute the
    interior auxiliaries and tendencies while the halo exchange is in flight and the edge / halo columns of pHY′, νₑ, κₑ from the
    exchanged halos afterwards (Distributed.update_state_general): fused RK3 stage boundaries and the plain QAB2 sequence.
    ytopo = "box": (Bounded, Bounded, Bounded) -- the partitioned x has walls on the first and on the last slab."""
    from helpers import stretched_faces
    P = "Periodic"
    N = (32, 16, 12)
    box = ytopo == "box"
    if box:
        ytopo = "Bounded"
    ext = dict(x=(0, 64.0), y=(0, 64.0), z=stretched_faces(N[2], 32.0), topology=("Bounded" if box else P, ytopo, "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(77)
    init = {n: 1e-2 * rng.uniform(-1, 1, N) for n in "uv"}
    if box:
        init["u"] = 1e-2 * rng.uniform(-1, 1, (N[0] + 1, N[1], N[2]))
    if ytopo == "Bounded":  # the channel: walls in y (the direction-generic kernels and the cosine transforms on every slab)
        init["v"] = 1e-2 * rng.uniform(-1, 1, (N[0], N[1] + 1, N[2]))
    init["w"] = 1e-2 * rng.uniform(-1, 1, (N[0], N[1], N[2] + 1))
    init["T"] = 20 + 1e-2 * rng.uniform(-1, 1, N)
    init["S"] = 35 + 1e-2 * rng.uniform(-1, 1, N)
    dt = 1.0

    def build(grid):
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-3e-4)),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
        return ocn.NonhydrostaticModel(grid, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                       closure=ocn.ScalarDiffusivity(ν=1e-3, κ=2e-3) if closure == "constant" else ocn.AnisotropicMinimumDissipation(),
                                       buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                       boundary_conditions=bcs, timestepper=stepper)

    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = build(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext))
    ocn.set(sm, **init)
    for _ in range(2):
        ocn.time_step(sm, dt)
    ocn.sync_device()
    ref = [f.interior() for f in sm.prognostic_fields()] + [sm.pHY.interior()]

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        m = build(ocn.RectilinearGrid(arch, size=N, **ext))
        nx = m.grid.Nx
        ocn.set(m, **{k: v[r * nx:(r + 1) * nx + (1 if (box and k == "u" and r == R - 1) else 0)] for k, v in init.items()})
        for _ in range(2):
            ocn.time_step(m, dt)
        ocn.sync_device()
        return [f.interior() for f in m.prognostic_fields()] + [m.pHY.interior()]

    outs = _run_ranks(R, rank_main)
    nx = N[0] // R
    vscale = max(np.abs(a).max() for a in ref[:3])
    for r, fields in enumerate(outs):
        for a, b, name in zip(fields, ref, ("u", "v", "w", "T", "S", "pHY")):
            sl = slice(r * nx, (r + 1) * nx + (1 if (box and name == "u" and r == R - 1) else 0))
            scale = vscale if name in "uvw" else np.abs(b).max()
            tol = 1e-11 if closure == "constant" else 1e-9  # the eddy diffusivities amplify the solvers' rounding differences
            assert np.abs(a - b[sl]).max() <= tol * scale, f"rank {r} field {name}: {np.abs(a - b[sl]).max()} vs {tol * scale}"


@pytest.mark.parametrize("R", [2, 4])
def test_conditions_on_the_west_and_east_walls_of_a_partitioned_box(ocn, R):
    """Every rank of a (Bounded, Bounded, Bounded) run is given the same boundary conditions; the slab that holds a wall applies the ones
    on it (a heated west wall as a Value condition, a flux through the east wall, a gradient on the south wall, a bottom flux), the slabs
    between drop them: two RK3 steps with a constant diffusivity against the single-rank model."""
    N = (32, 16, 12)
    B = "Bounded"
    ext = dict(x=(0, 2.0), y=(0, 1.0), z=(-1.0, 0.0), topology=(B, B, B), halo=(3, 3, 3))
    rng = np.random.default_rng(5)
    init = {"u": 1e-2 * rng.uniform(-1, 1, (N[0] + 1, N[1], N[2])), "v": 1e-2 * rng.uniform(-1, 1, (N[0], N[1] + 1, N[2])),
            "w": 1e-2 * rng.uniform(-1, 1, (N[0], N[1], N[2] + 1)), "T": 1 + 1e-2 * rng.uniform(-1, 1, N)}

    def build(grid):
        bcs = {"T": ocn.FieldBoundaryConditions(west=ocn.ValueBoundaryCondition(2.0), east=ocn.FluxBoundaryCondition(3e-3),
                                                south=ocn.GradientBoundaryCondition(0.5), bottom=ocn.FluxBoundaryCondition(-1e-3)),
               "v": ocn.FieldBoundaryConditions(west=ocn.ValueBoundaryCondition(0.0), east=ocn.ValueBoundaryCondition(0.0))}  # no-slip side walls
        return ocn.NonhydrostaticModel(grid, advection=ocn.WENO(), tracers=("T",), closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-2),
                                       buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4), constant_salinity=35.0),
                                       boundary_conditions=bcs)

    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = build(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext))
    ocn.set(sm, **init)
    for _ in range(2):
        ocn.time_step(sm, 5e-3)
    ocn.sync_device()
    ref = [f.interior() for f in sm.prognostic_fields()]
    nx = N[0] // R

    def xs(name, r):
        return slice(r * nx, (r + 1) * nx + (1 if (name == "u" and r == R - 1) else 0))

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        m = build(ocn.RectilinearGrid(arch, size=N, **ext))
        ocn.set(m, **{k: v[xs(k, r)] for k, v in init.items()})
        for _ in range(2):
            ocn.time_step(m, 5e-3)
        ocn.sync_device()
        return [f.interior() for f in m.prognostic_fields()]

    vscale = max(np.abs(a).max() for a in ref[:3])
    for r, fields in enumerate(_run_ranks(R, rank_main)):
        for a, b, name in zip(fields, ref, ("u", "v", "w", "T")):
            scale = vscale if name in "uvw" else np.abs(b).max()
            assert np.abs(a - b[xs(name, r)]).max() <= 1e-11 * scale, f"rank {r} field {name}"
    # the conditions did something: the first column next to the heated wall differs from a run without them
    plain = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext), advection=ocn.WENO(), tracers=("T",),
                                    closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-2),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4), constant_salinity=35.0))
    ocn.set(plain, **init)
    for _ in range(2):
        ocn.time_step(plain, 5e-3)
    ocn.sync_device()
    assert np.abs(plain.tracers[0].interior()[0] - ref[3][0]).max() > 1e-4


@pytest.mark.parametrize("topo", ["PPP", "PPB"])
def test_distributed_large_slabs_match_single_rank(ocn, topo):
    """The slab pipelines at production-like extents (256 x 256 x 128 on R = 4 ranks: 64-wide slabs as at 512^3 / 8, column
    counts and strides beyond 2^16 elements): one RK3 step in the default fast math against the single-rank model, plus
    incompressibility of the distributed result."""
    from helpers import stretched_faces
    P = "Periodic"
    R = 4
    N = (256, 256, 128)
    if topo == "PPP":
        ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
    else:
        ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=stretched_faces(N[2], 2.0), topology=(P, P, "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(99)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    if topo == "PPB":
        init["w"] = rng.uniform(-1, 1, (N[0], N[1], N[2] + 1))
    dt = 1e-3
    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        sm = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext), advection=ocn.WENO())
        ocn.set(sm, **init)
        ocn.time_step(sm, dt)
        ocn.flush_tendencies(sm)
        ocn.sync_device()
        ref = [f.interior() for f in sm.velocities] + [sm.pNHS.interior()]
        del sm

        def rank_main(r, fabric):
            arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
            g = ocn.RectilinearGrid(arch, size=N, **ext)
            m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
            assert m.pressure_solver.impl.fast == (3 if topo == "PPP" else 2)  # R > 1: the transpose-free pipeline by default
            sl = slice(r * g.Nx, (r + 1) * g.Nx)
            ocn.set(m, **{k: v[sl] for k, v in init.items()})
            ocn.time_step(m, dt)
            ocn.flush_tendencies(m)
            ocn.sync_device()
            return [f.interior() for f in m.velocities] + [m.pNHS.interior()]

        outs = _run_ranks(R, rank_main)
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)
    nx = N[0] // R
    scale = max(np.abs(a).max() for a in ref[:3])
    pscale = max(1.0, np.abs(ref[3]).max())
    for r, fields in enumerate(outs):
        sl = slice(r * nx, (r + 1) * nx)
        for a, b, name in zip(fields, ref, ("u", "v", "w", "p")):
            tol = 1e-10 * (pscale if name == "p" else scale)
            assert np.abs(a - b[sl]).max() <= tol, f"rank {r} field {name}: {np.abs(a - b[sl]).max()} > {tol}"


def test_distributed_halo_exchange_on_gpu(ocn):
    R = 2
    P = "Periodic"
    N = (12, 8, 6)
    rng = np.random.default_rng(0)
    glob = rng.random(N)

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, x=(0, 1), y=(0, 1), z=(0, 1), topology=(P, P, P), halo=(3, 3, 3))
        f = ocn.CenterField(g)
        f.set(glob[r * g.Nx:(r + 1) * g.Nx])
        ocn.fill_halo_regions(f)
        ocn.sync_device()
        return f.parent()

    outs = _run_ranks(R, rank_main)
    nx = N[0] // R
    for r, a in enumerate(outs):
        I = (np.arange(-3, nx + 3) + r * nx) % N[0]
        J = np.arange(-3, N[1] + 3) % N[1]
        K = np.arange(-3, N[2] + 3) % N[2]
        np.testing.assert_array_equal(a, glob[np.ix_(I, J, K)])


# ---- the product transport: RCCL behind the C ABI (csrc/comm.hip), on the one GPU of this box ------------------------------------------
@pytest.fixture(scope="module", params=["self-via-rccl", "self-copy"])
def rccl_arch(ocn, request):
    """Distributed(GPU(), Partition(1)) over a real RCCL communicator of world size 1 with force_communication: x is FullyConnected and
    every halo strip, plane and transpose goes through ocn_halo_exchange_* / ocn_dist_poisson_exchange (grouped ncclSend / ncclRecv to
    the rank itself on the communication stream, event-ordered against the compute stream)."""
    import os
    import socket
    import torch.distributed as dist
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=0, world_size=1)
    # (read by ocn_comm_init) "1": a rank's transfers to itself go through ncclSend / ncclRecv like those to any other peer;
    # default: device copies for the self chunk (what a production run does with its own 1 / R of every all-to-all)
    os.environ["OCN_COMM_SELF_VIA_RCCL"] = "1" if request.param == "self-via-rccl" else "0"
    arch = ocn.distributed.make_distributed(0, 1, 0, force_communication=True)
    os.environ.pop("OCN_COMM_SELF_VIA_RCCL")
    info = arch.fabric.info()
    assert info["ranks_seen_by_rccl"] == 1 and info["rank"] == 0 and info["rccl_version"] > 20000
    yield arch
    arch.fabric.close()


@pytest.mark.parametrize("N,topo", [((32, 16, 12), "PPP"), ((32, 16, 12), "PPB"), ((64, 128, 64), "PPP"), ((64, 128, 32), "PPB"),
                                    ((48, 128, 64), "PPP-xtri")])
def test_rccl_transport_world1_matches_single_rank_and_oracle(ocn, oracle, rccl_arch, N, topo, monkeypatch):
    """Two RK3 steps with every exchange executed by RCCL (self send / recv): the halo strips with the overlapped interior / buffer
    split, the one-plane exchanges of the projection, the deferred end-of-step exchange and both all-to-alls of the distributed
    solver ((64, 128, 64) selects the slab pipeline, (32, 16, 12) the transposing rocFFT path, PPB the distributed Fourier-tridiagonal
    solver).  Compared with the single-rank HIP model (1e-11 of max|u|) and, at the small sizes, with the CPU oracle."""
    from helpers import stretched_faces
    O = oracle
    P = "Periodic"
    xtri = topo.endswith("-xtri")  # the transpose-free pipeline: its one exchange is ncclAllGather (ocn_comm_all_gather)
    topo = topo[:3]
    if xtri:
        monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1")
    z = (0, 2 * np.pi) if topo == "PPP" else stretched_faces(N[2], 2.0)
    ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=(P, P, P if topo == "PPP" else "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(4321)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    if topo == "PPB":
        init["w"] = rng.uniform(-1, 1, (N[0], N[1], N[2] + 1))
    dt = 0.01 if N[0] <= 32 else 0.002
    fast_expected = 3 if xtri else {(64, 128, 64): 1, (64, 128, 32): 2}.get(N, 0)
    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext), advection=ocn.WENO())
    ocn.set(sm, **init)
    g = ocn.RectilinearGrid(rccl_arch, size=N, **ext)
    assert g.topology[0] == "FullyConnected" and g.Nx == N[0]
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    assert type(m.pressure_solver).__name__.startswith("Distributed")
    assert m.pressure_solver.impl.fast == fast_expected  # 1 / 2: the slab pipelines (periodic / tridiagonal flavour), 0: transposing rocFFT path
    ocn.set(m, **init)
    for _ in range(2):
        ocn.time_step(sm, dt)
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m)
    ocn.sync_device()
    scale = max(float(np.abs(f.interior()).max()) for f in sm.velocities)
    for a, b, name in zip(m.velocities, sm.velocities, "uvw"):
        assert np.abs(a.interior() - b.interior()).max() <= 1e-11 * scale, name
    assert np.abs(m.pNHS.interior() - sm.pNHS.interior()).max() <= 1e-10 * max(1.0, float(np.abs(sm.pNHS.interior()).max()))
    # the x halos of the state the step leaves behind were filled by the transport: periodic images of the interior
    H, nx = 3, N[0]
    u = m.u.data
    assert torch.equal(u[:, :, :H], u[:, :, nx:nx + H]) and torch.equal(u[:, :, nx + H:nx + 2 * H], u[:, :, H:2 * H])
    if N[0] <= 32:
        og = O.Grid(N, x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=z, topology=topo, halo=(3, 3, 3))
        om = O.NonhydrostaticModel(og)
        om.set(**init)
        for _ in range(2):
            om.time_step(dt)
        for a, b, name in zip(m.velocities, (om.u, om.v, om.w), "uvw"):
            assert np.abs(a.interior() - og.interior(b)).max() <= 1e-11 * scale, name


@pytest.mark.parametrize("N,pipeline", [((64, 128, 64), "alltoall"), ((48, 128, 64), "xtri"), ((16, 128, 64), "xtri"),
                                        ((48, 128, 64), "xtri-conservative"), ((48, 128, 64), "xtri-epilogue"), ((64, 128, 64), "alltoall-epilogue")])
def test_c_distributed_driver_equals_python_host_and_single_rank(ocn, rccl_arch, N, pipeline, monkeypatch):
    """ocn_rk3_driver_create_distributed: the whole RK3 step of ONE RANK of a slab-x run behind one C call -- local fills, the u plane,
    the strips of u*, v*, w* in flight under the distributed pressure solve, the pressure planes, one full-slab launch that corrects on
    load, the third stage's correction deferred into the next step -- with every collective issued by the library through RCCL (a world
    of one rank talking to itself).  After flush: bit-identical (strict math) to the Python host driving the per-call entry points on
    the same architecture, and within 1e-11 / 1e-10 of the single-rank model."""
    P = "Periodic"
    # "-conservative": the switches documented for a first multi-GPU run (OCN_DIST_CORRECT_ON_LOAD=0): the C driver then runs every
    # exchange synchronously (no strip exchange in flight while the solver's collective runs), the Python host its interior / strip split
    conservative = pipeline.endswith("-conservative")
    if pipeline.endswith("-epilogue"):  # the x strips written by the fused launch's epilogue + the wrapping unpack (off by default: no faster)
        monkeypatch.setenv("OCN_DIST_EPILOGUE_STRIPS", "1")
    pipeline = pipeline.split("-")[0]
    if conservative:
        monkeypatch.setenv("OCN_DIST_CORRECT_ON_LOAD", "0")
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1" if pipeline == "xtri" else "0")
    ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
    rng = np.random.default_rng(21)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    dt = 0.1 * (2 * np.pi / max(N))
    ocn.set_math_mode(ocn.MATH_STRICT)

    def build(arch):
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(arch, size=N, **ext), advection=ocn.WENO())
        ocn.set(m, **init)
        return m

    single = build(ocn.GPU())
    host = build(rccl_arch)
    assert host.dist_correct_on_load == (not conservative)
    for _ in range(3):
        ocn.time_step(single, dt)
        ocn.time_step(host, dt)
    ocn.flush_tendencies(single)
    ocn.flush_tendencies(host)
    m = build(rccl_arch)
    drv = ocn.RK3Driver(m)
    drv.time_step(dt)
    drv.flush()            # the deferred correction + tendencies completed between steps
    drv.time_step(dt)
    drv.time_step(dt)
    drv.flush()
    ocn.sync_device()
    for a, b, name in zip(host.velocities + (host.pNHS,), m.velocities + (m.pNHS,), ("u", "v", "w", "p")):
        np.testing.assert_array_equal(a.interior(), b.interior(), err_msg=name)
    scale = max(np.abs(f.interior()).max() for f in single.velocities)
    for a, b, name in zip(single.velocities + (single.pNHS,), m.velocities + (m.pNHS,), ("u", "v", "w", "p")):
        tol = 1e-10 * max(1.0, np.abs(a.interior()).max()) if name == "p" else 1e-11 * scale
        assert np.abs(a.interior() - b.interior()).max() <= tol, name
    del drv


@pytest.mark.parametrize("closure", ["amd", "scalar"])
def test_c_distributed_model_driver_equals_python_host_and_single_rank(ocn, rccl_arch, closure):
    """ocn_model_driver_create_distributed: the whole RK3 step of ONE RANK of a slab-x run of config 4's term set (T, S, SeawaterBuoyancy
    with pHY', FPlane, AMD or ScalarDiffusivity, flux / gradient conditions, stretched Bounded z) behind one C call, every exchange
    (prognostic strips, diffusivity strips, u / p planes, the Fourier-tridiagonal solver's transposes) issued by the library through RCCL.
    After flush: bit-identical (strict math) to the Python host on the same architecture (which overlaps the exchange with an interior /
    buffer split: same results), and within 1e-10 of the single-rank model."""
    from helpers import stretched_faces
    N = (64, 128, 32)
    ext = dict(x=(0, 64.0), y=(0, 64.0), z=stretched_faces(N[2], 32.0), topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(23)
    init = {n: 1e-2 * rng.uniform(-1, 1, N) for n in "uv"}
    init["T"] = 20 + 1e-2 * rng.uniform(-1, 1, N)
    init["S"] = 35 + 1e-2 * rng.uniform(-1, 1, N)
    ocn.set_math_mode(ocn.MATH_STRICT)

    def build(arch):
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-3e-4)),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
        cl = ocn.ScalarDiffusivity(ν=1e-3, κ={"T": 2e-3, "S": 5e-4}) if closure == "scalar" else ocn.AnisotropicMinimumDissipation()
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(arch, size=N, **ext), advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                    closure=cl, buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                    boundary_conditions=bcs)
        ocn.set(m, **init)
        return m

    single = build(ocn.GPU())
    host = build(rccl_arch)
    for _ in range(3):
        ocn.time_step(single, 1.5)
        ocn.time_step(host, 1.5)
    ocn.flush_tendencies(single)
    ocn.flush_tendencies(host)
    m = build(rccl_arch)
    drv = ocn.ModelRK3Driver(m)
    drv.time_step(1.5)
    drv.flush()
    drv.time_step(1.5)
    drv.time_step(1.5)
    drv.flush()
    ocn.sync_device()
    names = ("u", "v", "w", "T", "S", "p")
    for a, b, name in zip(host.prognostic_fields() + (host.pNHS,), m.prognostic_fields() + (m.pNHS,), names):
        np.testing.assert_array_equal(a.interior(), b.interior(), err_msg=name)
    scale = max(np.abs(f.interior()).max() for f in single.velocities)
    for a, b, name in zip(single.prognostic_fields() + (single.pNHS,), m.prognostic_fields() + (m.pNHS,), names):
        ref = np.abs(a.interior()).max()
        # (AMD's diffusivities are ratios of small numbers: they amplify the rounding differences of the two pressure solvers)
        tol = 1e-10 * max(1.0, ref) if name == "p" else 1e-10 * (scale if name in "uvw" else ref)
        assert np.abs(a.interior() - b.interior()).max() <= tol, name
    del drv


# ---- R > 1 through the LIBRARY's transport code (in-process mailboxes instead of RCCL: csrc/comm.hip, ocn_comm_init_local) -------------------
_LOCAL_KEY = [1000]


def _run_ranks_local(ocn, R, fn):
    """R threads, each with an ocn_comm_init_local communicator of the same group: every exchange goes through the C entry points the RCCL
    transport uses (schedules, pack / unpack, events), with R distinct peers."""
    _LOCAL_KEY[0] += 1
    key = _LOCAL_KEY[0]
    out, errs = [None] * R, []

    def target(r):
        try:
            torch.cuda.set_device(0)
            fabric = ocn.distributed.LocalFabric(r, R, key)
            try:
                out[r] = fn(r, fabric)
            finally:
                ocn.sync_device()
        except Exception:  # noqa
            import traceback
            errs.append(f"rank {r}: {traceback.format_exc()}")

    threads = [threading.Thread(target=target, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errs, "\n".join(errs)
    return out


@pytest.mark.parametrize("R,N,pipeline", [(2, (64, 128, 64), "xtri"), (4, (64, 128, 64), "xtri"), (2, (128, 128, 64), "alltoall"),
                                          (4, (128, 128, 64), "alltoall"), (8, (128, 128, 64), "xtri")])
def test_library_transport_ranks_match_single_rank(ocn, R, N, pipeline, monkeypatch):
    """R ranks as threads over the library's in-process transport: the correction-on-load stage (u plane, strips of u*, v*, w* begun under
    the distributed solve, pressure planes + the owner-corrected u plane after it, one full-slab launch) with every send / recv issued by
    csrc/comm.hip's schedule code between R DISTINCT peers -- Python host and C driver (ocn_rk3_driver_create_distributed): the C driver is
    bit-identical to the Python host rank by rank, both within 1e-11 / 1e-10 of the single-rank model."""
    P = "Periodic"
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1" if pipeline == "xtri" else "0")
    ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
    rng = np.random.default_rng(24)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    dt = 0.1 * (2 * np.pi / max(N))
    ocn.set_math_mode(ocn.MATH_STRICT)
    single = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext), advection=ocn.WENO())
    ocn.set(single, **init)
    for _ in range(3):
        ocn.time_step(single, dt)
    ocn.flush_tendencies(single)
    ocn.sync_device()
    ref = [f.interior() for f in single.velocities] + [single.pNHS.interior()]

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        info = fabric.info()
        assert info["ranks_seen_by_rccl"] == R and info["rccl_version"] == 0 and info["rank"] == r
        out = []
        for use_driver in (False, True):
            g = ocn.RectilinearGrid(arch, size=N, **ext)
            m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
            assert m.dist_correct_on_load
            sl = slice(r * g.Nx, (r + 1) * g.Nx)
            ocn.set(m, **{k: v[sl] for k, v in init.items()})
            if use_driver:
                drv = ocn.RK3Driver(m)
                drv.time_step(dt)
                drv.flush()
                drv.time_step(dt)
                drv.time_step(dt)
                drv.flush()
                del drv
            else:
                for _ in range(3):
                    ocn.time_step(m, dt)
                ocn.flush_tendencies(m)
            ocn.sync_device()
            out.append([f.interior() for f in m.velocities] + [m.pNHS.interior()])
        fabric.barrier()
        return out

    outs = _run_ranks_local(ocn, R, rank_main)
    nx = N[0] // R
    scale = max(np.abs(a).max() for a in ref[:3])
    for r, (host, drv) in enumerate(outs):
        sl = slice(r * nx, (r + 1) * nx)
        for a, b, c, name in zip(host, drv, ref, ("u", "v", "w", "p")):
            np.testing.assert_array_equal(a, b, err_msg=f"rank {r} {name}: C driver vs Python host")
            tol = 1e-11 * scale if name != "p" else 1e-10 * max(1.0, np.abs(ref[3]).max())
            assert np.abs(a - c[sl]).max() <= tol, f"rank {r} field {name} vs the single-rank model"


@pytest.mark.parametrize("R", [2, 4, 8])
def test_replica_transport_equals_every_rank_of_a_replicated_flow(ocn, R, monkeypatch):
    """The measurement transport (ocn_comm_init_replica: this process is rank 0 of R identical ranks, every receive a device copy from
    its own send buffer to the mirror-image peer) against the real thing: an initial condition of x-period Lx / R on R ranks as threads over
    the library's in-process transport (R DISTINCT peers, csrc/comm.hip's schedules between them).  Every one of those ranks must then
    hold what the single replica process holds -- velocities and pressure after three steps of the C driver, to 1e-12 (see below why not
    bit for bit) -- which is what makes the per-rank timings of tools/bench_dist_rank.py the timings of a rank of a real R-rank run."""
    P = "Periodic"
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1")
    nx = 32
    N = (nx * R, 128, 64)
    ext = dict(x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
    rng = np.random.default_rng(31)
    slab = {n: rng.uniform(-1, 1, (nx,) + N[1:]) for n in "uvw"}   # one slab, repeated R times along x
    dt = 0.1 * (2 * np.pi / N[0])
    ocn.set_math_mode(ocn.MATH_STRICT)

    def run(arch):
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        assert g.Nx == nx
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
        ocn.set(m, **slab)
        drv = ocn.RK3Driver(m)
        for _ in range(3):
            drv.time_step(dt)
        drv.flush()
        ocn.sync_device()
        out = [f.interior() for f in m.velocities] + [m.pNHS.interior()]
        del drv
        return out

    fabric = ocn.distributed.ReplicaFabric(R)
    info = fabric.info()
    assert info["ranks_seen_by_rccl"] == R and info["rccl_version"] == 0 and info["rank"] == 0
    replica = run(ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric, force_communication=True))
    fabric.close()

    def rank_main(r, fab):
        out = run(ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fab))
        fab.barrier()
        return out

    scale = max(np.abs(a).max() for a in replica[:3])
    assert scale > 0.1
    for r, got in enumerate(_run_ranks_local(ocn, R, rank_main)):
        for a, b, name in zip(got, replica, ("u", "v", "w", "p")):
            # not bit for bit: the R ranks of a real run are not EXACTLY replicas of each other -- the (ky, kz) = (0, 0) line of the pressure
            # solve is a prefix sum over the global line, whose rounding differs from slab to slab -- so after the first projection they
            # differ among themselves by a few ulps, and from the replica process by as much
            tol = 1e-12 * (scale if name != "p" else max(1.0, np.abs(b).max()))
            assert np.abs(a - b).max() <= tol, f"rank {r} of {R}, {name}: {np.abs(a - b).max()}"


@pytest.mark.parametrize("R,ytopo", [(2, "Periodic"), (4, "Periodic"), (2, "Bounded"), (4, "Bounded"), (2, "box"), (4, "box")])
def test_library_transport_config4_terms_match_single_rank(ocn, R, ytopo):
    """Config 4's term set (T, S, SeawaterBuoyancy + pHY', FPlane, AMD, flux / gradient conditions, stretched Bounded z, the distributed
    Fourier-tridiagonal solver) on R ranks over the library's transport: Python host (interior / buffer split, diffusivities recomputed in
    the edge and halo columns) and the C model driver (ocn_model_driver_create_distributed), rank by rank bit-identical to each other and
    within 1e-10 of the single-rank model.  ytopo = "Bounded": the same physics in a channel (walls in y; stretched z at R = 2, regular z --
    whose single-rank reference is the cosine-transform FFT solver -- at R = 4), Python host only --
    the slabs run the direction-generic kernels (interior box + wall frames) and the cosine transforms in y (Ny = 128: the column kernel
    with stage-ordered wavenumbers), every exchange through the library's transport."""
    from helpers import stretched_faces
    N = (64 * R // 2, 128, 32)
    box = ytopo == "box"  # (Bounded, Bounded, Bounded): walls on the first and the last slab of the partitioned x as well
    if box:
        ytopo = "Bounded"
    channel = ytopo == "Bounded"
    ext = dict(x=(0, 64.0), y=(0, 64.0), z=(-32.0, 0.0) if (channel and R == 4) else stretched_faces(N[2], 32.0),
               topology=("Bounded" if box else "Periodic", ytopo, "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(25)
    init = {n: 1e-2 * rng.uniform(-1, 1, N) for n in "uv"}
    if box:
        init["u"] = 1e-2 * rng.uniform(-1, 1, (N[0] + 1, N[1], N[2]))
    if channel:
        init["v"] = 1e-2 * rng.uniform(-1, 1, (N[0], N[1] + 1, N[2]))
    init["T"] = 20 + 1e-2 * rng.uniform(-1, 1, N)
    init["S"] = 35 + 1e-2 * rng.uniform(-1, 1, N)
    ocn.set_math_mode(ocn.MATH_STRICT)

    def build(arch, sl=slice(None), last=False):
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-3e-4)),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(arch, size=N, **ext), advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                    closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), boundary_conditions=bcs)
        # (u of the last slab of a Bounded x carries the wall face)
        ocn.set(m, **{k: v[slice(sl.start, sl.stop + 1) if (box and last and k == "u") else sl] for k, v in init.items()})
        return m

    single = build(ocn.GPU())
    for _ in range(3):
        ocn.time_step(single, 1.5)
    ocn.flush_tendencies(single)
    ocn.sync_device()
    ref = [f.interior() for f in single.prognostic_fields()] + [single.pNHS.interior()]

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        nx = N[0] // R
        out = []
        for use_driver in ((False,) if channel else (False, True)):
            m = build(arch, slice(r * nx, (r + 1) * nx), last=(r == R - 1))
            if use_driver:
                drv = ocn.ModelRK3Driver(m)
                for _ in range(3):
                    drv.time_step(1.5)
                drv.flush()
                del drv
            else:
                for _ in range(3):
                    ocn.time_step(m, 1.5)
                ocn.flush_tendencies(m)
            ocn.sync_device()
            out.append([f.interior() for f in m.prognostic_fields()] + [m.pNHS.interior()])
        fabric.barrier()
        return out

    outs = _run_ranks_local(ocn, R, rank_main)
    nx = N[0] // R
    scale = max(np.abs(a).max() for a in ref[:3])
    names = ("u", "v", "w", "T", "S", "p")
    for r, out in enumerate(outs):
        host, drv = out[0], out[-1]
        for a, b, c, name in zip(host, drv, ref, names):
            sl = slice(r * nx, (r + 1) * nx + (1 if (box and name == "u" and r == R - 1) else 0))
            np.testing.assert_array_equal(a, b, err_msg=f"rank {r} {name}: C model driver vs Python host")
            refmax = np.abs(c).max()
            tol = 1e-10 * max(1.0, refmax) if name == "p" else 1e-10 * (scale if name in "uvw" else refmax)
            assert np.abs(a - c[sl]).max() <= tol, f"rank {r} field {name} vs the single-rank model"


# ---- HydrostaticFreeSurfaceModel on slab-x ranks (BASELINE.json configs[4] is an 8-GPU configuration) ---------------------------------------
def _hydro_model(ocn, grid, fused=None):
    return ocn.HydrostaticFreeSurfaceModel(grid, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                           free_surface=ocn.SplitExplicitFreeSurface(substeps=12), coriolis=ocn.FPlane(f=1e-4),
                                           closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                           buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                           boundary_conditions={"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1e-4)),
                                                                "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5))},
                                           fused=fused)


def _hydro_case():
    from helpers import stretched_faces
    N = (64, 12, 7)
    ext = dict(x=(0, 8.0e3), y=(0, 1.5e3), z=stretched_faces(N[2], 40.0), topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(77)
    init = dict(u=1e-2 * rng.uniform(-1, 1, N), v=1e-2 * rng.uniform(-1, 1, N), eta=1e-2 * rng.uniform(-1, 1, N[:2]),
                T=20 + 1e-2 * rng.uniform(-1, 1, N), S=35 + 1e-2 * rng.uniform(-1, 1, N))
    return N, ext, init


def _hydro_state(m):
    g = m.grid
    ii, jj = slice(g.Hy, g.Hy + g.Ny), slice(g.Hx, g.Hx + g.Nx)
    out = {n: f.interior() for n, f in zip(("u", "v", "w"), m.velocities)}
    out.update({f"c{n}": c.interior() for n, c in enumerate(m.tracers)})
    out.update(eta=m.eta[ii, jj].cpu().numpy().T, U=m.U[ii, jj].cpu().numpy().T, V=m.V[ii, jj].cpu().numpy().T)
    return out


@pytest.mark.parametrize("R,transport", [(2, "python"), (4, "python"), (2, "library"), (4, "library")])
def test_distributed_hydrostatic_config5_matches_single_rank(ocn, R, transport):
    """configs[4]'s combination on R slab-x ranks (threads of this process, real HIP kernels): the fused QAB2 step with the
    DistributedSplitExplicitFreeSurface scheme -- η, U, V, Gᵁ, Gⱽ halos as wide as the substep count, ONE exchange per baroclinic step,
    no communication while substepping (distributed_split_explicit_free_surface.jl) -- equals the single-rank model BIT FOR BIT in
    strict math after 3 steps (u, v, w, T, S, η, U, V): every rank evaluates the same expressions on the same values."""
    N, ext, init = _hydro_case()
    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = _hydro_model(ocn, ocn.RectilinearGrid(ocn.GPU(), size=N, **ext))
    sm.set(**init)
    for _ in range(3):
        sm.time_step(20.0)
    ocn.sync_device()
    ref = _hydro_state(sm)
    assert np.abs(ref["U"]).max() > 0 and np.abs(ref["w"]).max() > 0

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        assert g.Nx == N[0] // R and g.topology[0] == "FullyConnected"
        m = _hydro_model(ocn, g)
        sl = slice(r * g.Nx, (r + 1) * g.Nx)
        m.set(**{k: v[sl] for k, v in init.items()})
        for _ in range(3):
            m.time_step(20.0)
        ocn.sync_device()
        out = _hydro_state(m)
        if transport == "library":
            fabric.barrier()
        return out

    # "library": the strips and the wide barotropic halos travel through csrc/comm.hip's entry points between R distinct peers
    # (ocn_halo_exchange_*, ocn_comm_exchange_strips over the in-process transport) instead of the tests' Python fabric
    outs = _run_ranks_local(ocn, R, rank_main) if transport == "library" else _run_ranks(R, rank_main)
    nx = N[0] // R
    for r, got in enumerate(outs):
        sl = slice(r * nx, (r + 1) * nx)
        for name, a in got.items():
            np.testing.assert_array_equal(a, ref[name][sl], err_msg=f"rank {r} field {name}")


def test_rccl_world1_hydrostatic_matches_single_rank(ocn, rccl_arch):
    """The same through the product transport (RCCL world of one rank with force_communication: the rank exchanges its strips and the
    wide split-explicit halos with itself through ocn_halo_exchange_* / ocn_comm_exchange_strips)."""
    N, ext, init = _hydro_case()
    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = _hydro_model(ocn, ocn.RectilinearGrid(ocn.GPU(), size=N, **ext))
    sm.set(**init)
    g = ocn.RectilinearGrid(rccl_arch, size=N, **ext)
    assert g.topology[0] == "FullyConnected"
    m = _hydro_model(ocn, g)
    m.set(**init)
    for _ in range(3):
        sm.time_step(20.0)
        m.time_step(20.0)
    ocn.sync_device()
    ref, got = _hydro_state(sm), _hydro_state(m)
    for name in ref:
        np.testing.assert_array_equal(got[name], ref[name], err_msg=name)


@pytest.mark.parametrize("mode", ["direct", "collective"])
def test_all_gather_forms_through_rccl(mode):
    """ocn_comm_all_gather's two forms through RCCL itself (world 1, the rank's transfer to itself issued as ncclSend / ncclRecv or as
    ncclAllGather), each form in a child process of its own (OCN_COMM_ALL_GATHER is read per call since round 4: bench.py's two legs)."""
    import subprocess
    import sys
    code = r"""
import os, socket, torch, torch.distributed as dist
import oceananigans_jl_amd as ocn
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("gloo", rank=0, world_size=1)
arch = ocn.distributed.make_distributed(0, 1, 0, force_communication=True)
send = torch.arange(1000, dtype=torch.float64, device="cuda") + 0.5
recv = torch.zeros_like(send)
arch.fabric.all_gather(recv, send)
torch.cuda.synchronize()
assert torch.equal(recv, send)
# a strip exchange in flight on the communication stream WHILE the solver's exchange runs on the compute stream, both on ONE
# communicator -- the order the distributed stage issues them in (csrc/driver.hip project_for_load): begin -> all_gather -> end
g = ocn.RectilinearGrid(arch, size=(32, 16, 12), x=(0, 1), y=(0, 1), z=(0, 1), topology=("Periodic",) * 3, halo=(3, 3, 3))
fields = [ocn.Field(l, g) for l in (1, 2, 4)]
for n, f in enumerate(fields):
    f.interior_view().copy_(torch.rand(f.interior_view().shape, device="cuda", dtype=torch.float64) + n)
for rep in range(3):
    arch.ops.local_fill(g, fields, False)
    arch.fabric.halo_exchange_begin(g, fields)
    recv.zero_()
    arch.fabric.all_gather(recv, send + rep)
    arch.fabric.halo_exchange_end(g, fields)
    torch.cuda.synchronize()
    assert torch.equal(recv, send + rep)
    for f in fields:
        d, nx, H = f.data, g.Nx, 3
        assert torch.equal(d[:, :, :H], d[:, :, nx:nx + H]) and torch.equal(d[:, :, nx + H:nx + 2 * H], d[:, :, H:2 * H])
arch.fabric.close()
print("ALL-GATHER-OK")
"""
    import os
    env = dict(os.environ, OCN_COMM_SELF_VIA_RCCL="1", OCN_COMM_ALL_GATHER=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "ALL-GATHER-OK" in p.stdout, p.stderr[-2000:]
