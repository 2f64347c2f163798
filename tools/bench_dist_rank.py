#!/usr/bin/env python3
"""What ONE rank of an R-rank slab-x run costs, measured on a single GPU: the real distributed code path (local halo fills,
pack / unpack, interior / buffer tendency split, distributed transposes and FFTs, all kernels at the rank-local size) with a
LOOPBACK fabric that emulates R identical ranks -- every message a rank would receive from a neighbour is the one it sends
itself, which is exactly what R replicas of an x-periodic flow of period Lx/R exchange (except the return all-to-all of the
pressure solve, see LoopbackFabric.all_to_all: TIMING ONLY, the pressure is not the true one).  The time it prints is therefore
the per-rank compute + host time of an R-GPU step with the communication itself replaced by device copies; compare it with
(single-GPU step) / R to see what the decomposition costs before any link time.

  tools/bench_dist_rank.py [N] [R] [steps] [workload]      workload = box (512^3-style periodic) | config4 (P,P,B stretched, advection only) | config4amd (its full physics) |
                                                          config5 (2N x 2N x N/4 HydrostaticFreeSurfaceModel as bench.py --workload config5) |
                                                          driver (one C call per rank-step through the library's replica transport, ocn_comm_init_replica: rank 0 of R
                                                          identical ranks with the R-rank schedules, pipelines and interface systems) |
                                                          driver4 (config 4's term set: a one-rank RCCL world at the local size of one rank of R)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60  # the timed region carries ~30 ms of one-off cost (final flush, first-use allocations): 20 steps overstate by 1 ms
workload = sys.argv[4] if len(sys.argv) > 4 else "box"


class LoopbackFabric:
    """R identical ranks: recv from `src` = what that rank sent to me = what I send to the mirror-image peer (-src mod R)."""

    def __init__(self, R):
        self.rank, self.size = 0, R

    def start_exchange(self, sends, recvs):
        queue = {}
        for t, dst in sends:
            queue.setdefault(dst % self.size, []).append(t)
        for t, src in recvs:
            t.copy_(queue[(-src) % self.size].pop(0))
        return ()

    @staticmethod
    def wait(reqs):
        pass

    def all_to_all(self, recv, send):
        # Forward exchange of a solve: chunk m <- rank m's chunk for me = my own chunk 0 (identical ranks).  The RETURN exchange
        # cannot be emulated (the other ranks' parts of the solution are never computed here): it delivers zeros, i.e. p = 0 and
        # no projection -- the same bytes move and the same kernels run, which is all a timing needs; the fields stay finite.
        self.calls = getattr(self, "calls", 0) + 1
        n = send.numel() // self.size
        if self.calls % 2:
            recv.view(self.size, n).copy_(send[:n].expand(self.size, n))
        else:
            recv.zero_()

    def all_gather(self, recv, send):
        recv.view(self.size, -1).copy_(send.expand(self.size, -1))  # identical ranks: everybody's message is mine

    def allreduce_max(self, t):
        return t


ocn.set_math_mode(ocn.MATH_FAST)
if workload == "driver4":
    # the same for config 4's term set (ocn_model_driver_create_distributed): a (N / R) x N x (N / 2) slab, stretched Bounded z, AMD, T, S
    # a Bounded z has no transpose-free solve yet: its all-to-all pipeline cannot run over the replica transport, so this workload is a
    # one-rank RCCL world that exchanges with itself at the local size of one rank of R
    import socket
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    arch = ocn.distributed.make_distributed(0, 1, 0, force_communication=True)
    nx, Nz = N // R, N // 2
    Lz, refinement, stretching = 32.0, 1.2, 12.0
    h = lambda k: (k - 1) / Nz
    z_faces = np.array([Lz * ((1 + (h(k) - 1) / refinement) * ((1 - np.exp(-stretching * h(k))) / (1 - np.exp(-stretching))) - 1)
                        for k in range(1, Nz + 2)])

    def build():
        g = ocn.RectilinearGrid(arch, size=(nx, N, Nz), x=(0, 64 / R), y=(0, 64), z=z_faces, topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
        Q, rho, cp, dTdz = 200.0, 1026.0, 3991.0, 0.01
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1.225 / rho * 2.5e-3 * 10 * 10)),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(Q / (rho * cp)), bottom=ocn.GradientBoundaryCondition(dTdz)),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-1e-3 / 3600))}
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4), closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), boundary_conditions=bcs)
        zc = torch.from_numpy(0.5 * (z_faces[1:] + z_faces[:-1])).to("cuda")
        T = m.field("T").interior_view()
        T.copy_(20 + dTdz * zc[:, None, None] + 1e-6 * torch.rand(T.shape, device="cuda", dtype=torch.float64))
        m.field("S").interior_view().fill_(35.0)
        gen = torch.Generator(device="cuda"); gen.manual_seed(1)
        for f in m.velocities:
            v = f.interior_view()
            v.copy_(1e-2 * (2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
        ocn.set(m)
        return m

    for name in ("python host", "C driver"):
        m = build()
        g = m.grid
        dt = 0.1 * min(g.dx, float(np.diff(z_faces).min())) / max(float(f.interior_view().abs().max()) for f in m.velocities)
        if name == "C driver":
            drv = ocn.ModelRK3Driver(m)
            stepper, flush = (lambda: drv.time_step(dt)), drv.flush
        else:
            stepper, flush = (lambda: ocn.time_step(m, dt)), (lambda: ocn.flush_tendencies(m))
        for _ in range(3):
            stepper()
        flush()
        ocn.sync_device()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper()
        flush()
        host_ms = (time.perf_counter() - t0) / steps * 1e3
        ocn.sync_device()
        ms = (time.perf_counter() - t0) / steps * 1e3
        finite = all(bool(torch.isfinite(f.interior_view()).all()) for f in m.prognostic_fields())
        print(f"driver4 N={N} R={R} [{name}]: local {g.Nx}x{g.Ny}x{g.Nz}, {ms:.2f} ms/step per rank (host enqueue {host_ms:.2f} ms), finite={finite}")
        del m
    sys.exit(0)
if workload == "driver":
    # ONE C call per step (ocn_rk3_driver_create_distributed) on rank 0 of R identical ranks (the library's replica transport): the
    # kernels, schedules, pressure pipeline (transpose-free for R > 1) and interface systems of an R-rank run, every transfer a device copy.
    arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=ocn.distributed.ReplicaFabric(R), force_communication=True)
    g = ocn.RectilinearGrid(arch, size=(N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=("Periodic",) * 3, halo=(3, 3, 3))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for f in m.velocities:
        v = f.interior_view()
        v.copy_(2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1)
    ocn.set(m)
    dt = 0.1 * g.dx / max(float(f.interior_view().abs().max()) for f in m.velocities)
    variants = (("python host", lambda: ocn.time_step(m, dt), lambda: ocn.flush_tendencies(m)), ("C driver", None, None))
    only = os.environ.get("OCN_BDR_ONLY")  # "c" / "python": one host only (kernel traces)
    if only:
        variants = tuple(v for v in variants if v[0][0].lower() == only[0].lower())
    for name, stepper, flush in variants:
        if stepper is None:
            drv = ocn.RK3Driver(m)
            stepper, flush = (lambda: drv.time_step(dt)), drv.flush
        for _ in range(5):
            stepper()
        flush()
        ocn.sync_device()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper()
        flush()
        host_ms = (time.perf_counter() - t0) / steps * 1e3
        ocn.sync_device()
        ms = (time.perf_counter() - t0) / steps * 1e3
        finite = all(bool(torch.isfinite(f.interior_view()).all()) for f in m.velocities)
        print(f"driver N={N} R={R} [{name}]: local {g.Nx}x{g.Ny}x{g.Nz}, {ms:.2f} ms/step per rank (host enqueue {host_ms:.2f} ms), finite={finite}")
    sys.exit(0)
if workload == "config5":
    # BASELINE.json configs[4] (an 8-GPU configuration): what one of R slab-x ranks costs
    arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=LoopbackFabric(R)) if R > 1 else ocn.GPU()
    Nx, Nz, H, L = 2 * N, N // 4, 1000.0, 1.0e6 * 2 * N / 1024.0
    g = ocn.RectilinearGrid(arch, size=(Nx, Nx, Nz), x=(0, L), y=(0, L), z=(-H, 0.0), topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
    m = ocn.HydrostaticFreeSurfaceModel(g, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                        free_surface=ocn.SplitExplicitFreeSurface(substeps=30), coriolis=ocn.FPlane(f=1e-4),
                                        closure=ocn.ScalarDiffusivity(ν=1e-2, κ=1e-3),
                                        buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)))
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for f in (m.u, m.v):
        v = f.interior_view()
        v.copy_(1e-2 * (2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
    zc = torch.linspace(-H + H / (2 * Nz), -H / (2 * Nz), Nz, device="cuda", dtype=torch.float64)
    m.field("T").interior_view().copy_((20 + 0.01 * zc)[:, None, None].expand(Nz, Nx, g.Nx))
    m.field("S").interior_view().fill_(35.0)
    m.update_state(compute_tendencies=False)
    dt = 2.0 * g.dx / np.sqrt(9.80665 * H)
    for _ in range(5):
        m.time_step(dt)
    m.flush_tendencies()
    ocn.sync_device()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.time_step(dt)
    m.flush_tendencies()
    host_ms = (time.perf_counter() - t0) / steps * 1e3
    ocn.sync_device()
    ms = (time.perf_counter() - t0) / steps * 1e3
    finite = all(bool(torch.isfinite(f.interior_view()).all()) for f in (m.u, m.v))
    print(f"config5 N={Nx}x{Nx}x{Nz} R={R}: local {g.Nx}x{g.Ny}x{g.Nz}, {ms:.2f} ms/step per rank (host enqueue {host_ms:.2f} ms), finite={finite}")
    sys.exit(0)
arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=LoopbackFabric(R)) if R > 1 else ocn.GPU()
P = "Periodic"
if workload.startswith("config4"):
    Nz = N // 2
    Lz, refinement, stretching = 32.0, 1.2, 12.0
    h = lambda k: (k - 1) / Nz
    z_faces = np.array([Lz * ((1 + (h(k) - 1) / refinement) * ((1 - np.exp(-stretching * h(k))) / (1 - np.exp(-stretching))) - 1)
                        for k in range(1, Nz + 2)])
    g = ocn.RectilinearGrid(arch, size=(N, N, Nz), x=(0, 64), y=(0, 64), z=z_faces, topology=(P, P, "Bounded"), halo=(3, 3, 3))
else:
    g = ocn.RectilinearGrid(arch, size=(N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=(P, P, P), halo=(3, 3, 3))
if workload == "config4amd":  # the ocean_wind_mixing_and_convection physics as written (tools/bench_config4.py, physics = 2)
    Q, rho, cp, dTdz = 200.0, 1026.0, 3991.0, 0.01
    bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1.225 / rho * 2.5e-3 * 10 * 10)),
           "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(Q / (rho * cp)), bottom=ocn.GradientBoundaryCondition(dTdz)),
           "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-1e-3 / 3600))}
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4),
                                closure=ocn.AnisotropicMinimumDissipation(),
                                buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                boundary_conditions=bcs)
    zc = torch.from_numpy(0.5 * (z_faces[1:] + z_faces[:-1])).to("cuda")
    T = m.field("T").interior_view()
    T.copy_(20 + dTdz * zc[:, None, None] + 1e-6 * torch.rand(T.shape, device="cuda", dtype=torch.float64))
    m.field("S").interior_view().fill_(35.0)
else:
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for f in m.velocities:
    v = f.interior_view()
    v.copy_((1e-2 if workload == "config4amd" else 1.0) * (2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1))
ocn.set(m)  # halos + projection
umax = max(float(f.interior_view().abs().max()) for f in m.velocities)
dt = 0.1 * (min(g.dx, float(np.diff(z_faces).min())) if workload.startswith("config4") else g.dx) / umax
if R > 1 and workload == "config4amd":
    dt *= 1e-3  # no projection in the loopback emulation: keep the unprojected noise from running away (timing does not depend on dt)
for _ in range(5):
    ocn.time_step(m, dt)
ocn.flush_tendencies(m)
ocn.sync_device()
t0 = time.perf_counter()
for _ in range(steps):
    ocn.time_step(m, dt)
ocn.flush_tendencies(m)
host_ms = (time.perf_counter() - t0) / steps * 1e3  # time to ENQUEUE the steps: close to ms/step means launch-bound
ocn.sync_device()
ms = (time.perf_counter() - t0) / steps * 1e3
finite = all(bool(torch.isfinite(f.interior_view()).all()) for f in m.velocities)
print(f"{workload} N={N} R={R}: local {g.Nx}x{g.Ny}x{g.Nz}, {ms:.2f} ms/step per rank "
      f"(host enqueue {host_ms:.2f} ms; ideal from one GPU = single-GPU step / {R}), finite={finite}")
