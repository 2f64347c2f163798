#!/usr/bin/env python3
"""bench.py -- cell-updates/s of full time steps of the reference's models on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload box|config4|config5] [--size 512] [--math fast|strict]

Workloads (config.workload):
  box      (default; the configuration the metric is quoted on, BASELINE.json configs[1]/[2] at --size 256 / 512): N^3 triply
           periodic RectilinearGrid, extent (2 pi)^3, halo 3, fp64, advection = WENO() (5th order), RungeKutta3, FFT-based pressure
           solver, no tracers / closure / buoyancy.  u, v, w ~ U(-1, 1) from a fixed seed, projected to be divergence free by set!
           (one dt = 1 pressure solve); dt = 0.1 dx / max|u|.  A "step" is one full time_step!: 3 x (substep, projection, tendencies).
  config4  configs[3]: N x N x N/2 (Periodic, Periodic, Bounded), stretched z, ocean_wind_mixing_and_convection physics.
  config5  configs[4]: 2N x 2N x N/4 (1024 x 1024 x 128 at --size 512) HydrostaticFreeSurfaceModel, VectorInvariant() momentum,
           tracer_advection = WENO(), SplitExplicitFreeSurface(substeps = 30), T / S + linear SeawaterBuoyancy + FPlane +
           ScalarDiffusivity, QuasiAdamsBashforth2.  A "step" is one time_step! (one tendency evaluation, 30 barotropic substeps).

--gpus N > 1: the same global grid is x-slab partitioned over N ranks (strong scaling), one process per GPU, RCCL halo exchange
and the pressure solve's exchange (one all-gather per solve with the transpose-free pipeline, csrc/xtri.hip) behind the C ABI
(ocn_comm_*).  Run as `python bench.py --gpus N` the script starts its N ranks itself
(torch.distributed.run as a CHILD process, before anything touches the GPU); run under torch.distributed.run it is one rank.
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# dmabuf IPC for RCCL / cross-process device memory: must be in the environment BEFORE the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ALGO_BYTES_PER_CELL_STEP = 1680.0  # 560 + 584 + 536 B per cell per RK3 step (REFERENCE kernel decomposition)
# THIS backend's compulsory traffic per cell per RK3 step on one GPU: per stage the fused tendency launch with the
# correction on load and the substep epilogue (104) + the 5-pass Poisson pipeline (104) = 208; three identical stage boundaries per step
BACKEND_BYTES_PER_CELL_STEP = 624.0
TENDENCY_BYTES_PLAIN = 48.0        # fused compute_Gu/Gv/Gw launch: read u, v, w once, write Gu, Gv, Gw (fp64)
TENDENCY_BYTES_IN_STEP = 104.0     # as the step runs it: + p (correction on load) + G- read + U_out (3) write + ...
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
# fp64 / VALU issue bound: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles (16 lanes per SIMD per cycle; the
# 78.6 TFLOP/s fp64 vector peak is 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz), at the 2.4 GHz peak engine clock
VALU_PEAK_GWAVEINSTR = 1024 * 2.4 / 4.0   # = 614.4 G wave-instructions / s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", dest="n", type=int, default=512, help="grid points per dimension (512 = the metric's config)")
    ap.add_argument("--math", choices=("fast", "strict"), default="fast")
    ap.add_argument("--workload", choices=("box", "config4", "config5"), default="box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict_ms_per_step leg (3 steps of the bit-exact build)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the event-timed launches of the dominant kernel (profiling runs: every launch then belongs to a step)")
    ap.add_argument("--driver", choices=("auto", "python", "c"), default="auto",
                    help="host orchestration of the step: the Python mirror of the reference's time_step! (per-kernel entry points), or "
                         "ocn_rk3_driver_time_step -- ONE C call per (rank-)step, collectives included, the third stage's pressure correction "
                         "deferred onto the next step's first launch (box), ocn_model_driver_time_step for config 4's term set; auto = c where it applies")
    ap.add_argument("--cpu-n", type=int, default=160, help="grid size of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--substeps", type=int, default=30, help="config5: SplitExplicitFreeSurface(substeps = ...)")
    ap.add_argument("--preflight", action="store_true",
                    help="start the --gpus N ranks, rendezvous, hand the RCCL unique id to every rank, check environment and library, "
                         "print one JSON line and exit BEFORE any GPU call (runs on a machine without GPUs)")
    ap.add_argument("--dist-poisson", choices=("xtri", "alltoall"), default=None,
                    help="pressure solve of a partitioned run: xtri = transpose-free cyclic tridiagonal x solve + one all-gather (default), "
                         "alltoall = the slab FFT pipeline with two RCCL all-to-alls per solve (north_star's pencil transpose)")
    ap.add_argument("--all-gather", choices=("direct", "collective"), default=None,
                    help="the xtri solve's one exchange: direct = R - 1 grouped point-to-point transfers, one per xGMI link (default), "
                         "collective = ncclAllGather")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a child torch.distributed.run and relay rank 0's JSON line.
    Nothing in this process has touched the GPU (no torch.cuda call, libocn_hip not loaded): the child is a plain subprocess, never an
    exec of a GPU process."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and ('"metric"' in ln or '"preflight"' in ln):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if p.returncode != 0 or line is None:
        raise SystemExit(p.returncode or 1)
    raise SystemExit(0)


def host_cores():
    """CPU threads this process may actually use: the cgroup quota (16 on a 1-GPU box) capped by the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except OSError:
        pass
    return n


def config4_faces(Nz, Lz=32.0, refinement=1.2, stretching=12.0):
    """z_faces of examples/ocean_wind_mixing_and_convection.jl:38-62"""
    import numpy as np
    k = np.arange(1, Nz + 2)
    h = (k - 1) / Nz
    zeta0 = 1 + (h - 1) / refinement
    Sigma = (1 - np.exp(-stretching * h)) / (1 - np.exp(-stretching))
    return Lz * (zeta0 * Sigma - 1)


# physical constants of examples/ocean_wind_mixing_and_convection.jl:79-110
C4 = dict(JT=200.0 / (1026.0 * 3991.0), dTdz=0.01, taux=-1.225 / 1026.0 * 2.5e-3 * 10 * 10, evap=1e-3 / 3600, f=1e-4,
          alpha=2e-4, beta=8e-4)
# config 5 (no script in the reference names it; the physics of tools/bench_hydrostatic.py): 1000 m deep, 1000 km wide ocean box
C5 = dict(H=1000.0, L=1.0e6, f=1e-4, nu=1e-2, kappa=1e-3, alpha=2e-4, beta=8e-4, amp=1e-2)


def cpu_baseline_config4(n, steps):
    """The CPU oracle with the same physics (AMD closure, SeawaterBuoyancy, FPlane, boundary conditions) on n x n x n/2."""
    import numpy as np
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    from oracle import oracle as O
    Nz = n // 2
    zf = config4_faces(Nz)
    g = O.Grid((n, n, Nz), x=(0, 64), y=(0, 64), z=zf, topology="PPB", halo=(3, 3, 3))
    bcs = {"u": {"top": O.FluxBoundaryCondition(C4["taux"])},
           "T": {"top": O.FluxBoundaryCondition(C4["JT"]), "bottom": O.GradientBoundaryCondition(C4["dTdz"])},
           "S": {"top": O.BC("flux", 0.0, -C4["evap"])}}
    m = O.NonhydrostaticModel(g, tracers=("T", "S"), coriolis_f=C4["f"], closure=("AMD",),
                              buoyancy=("SeawaterBuoyancy", 9.80665, C4["alpha"], C4["beta"]), boundary_conditions=bcs, workers=cores)
    rng = np.random.default_rng(1234)
    zc = 0.5 * (zf[1:] + zf[:-1])
    m.set(u=1e-2 * rng.uniform(-1, 1, (n, n, Nz)), v=1e-2 * rng.uniform(-1, 1, (n, n, Nz)),
          T=20 + C4["dTdz"] * zc[None, None, :] + 1e-6 * rng.uniform(-1, 1, (n, n, Nz)), S=35.0)
    dt = 0.1 * float(np.diff(zf).min()) / 1e-2
    m.time_step(dt)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.time_step(dt)
    el = time.perf_counter() - t0
    return {"value": n * n * Nz * steps / el, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{steps} RK3 steps of the same model at {n}x{n}x{Nz} (C oracle, OpenMP {cores} threads, scipy pocketfft), {el:.1f} s"}


def cpu_baseline_config5(nx, nz, steps, substeps):
    """oracle/hydrostatic.py (numpy + the C oracle's tendency kernels) with the same configuration on nx x nx x nz."""
    import numpy as np
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    from oracle import hydrostatic as Hy
    from oracle import oracle as O
    H, L = C5["H"], C5["L"] * nx / 1024.0  # same dx as the GPU workload
    g = O.Grid((nx, nx, nz), x=(0, L), y=(0, L), z=(-H, 0.0), topology="PPB", halo=(3, 3, 3))
    m = Hy.HydrostaticFreeSurfaceModel(g, tracers=("T", "S"), momentum_advection="VectorInvariant", tracer_advection="WENO5",
                                       coriolis_f=C5["f"], closure=(C5["nu"], C5["kappa"]),
                                       buoyancy=("SeawaterBuoyancy", 9.80665, C5["alpha"], C5["beta"]), split_explicit_substeps=substeps)
    rng = np.random.default_rng(1234)
    zc = -H + (np.arange(nz) + 0.5) * H / nz
    m.set(u=C5["amp"] * rng.uniform(-1, 1, (nx, nx, nz)), v=C5["amp"] * rng.uniform(-1, 1, (nx, nx, nz)),
          T=20 + 0.01 * zc[None, None, :] + np.zeros((nx, nx, 1)), S=35.0)
    dt = 2.0 * g.dx / np.sqrt(Hy.g_Earth * H)
    m.time_step(dt)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.time_step(dt)
    el = time.perf_counter() - t0
    return {"value": nx * nx * nz * steps / el, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{steps} QAB2 steps of the same model at {nx}x{nx}x{nz} (oracle/hydrostatic.py: numpy + C oracle kernels, {cores} OpenMP threads), {el:.1f} s"}


def cpu_baseline(n, steps):
    """The CPU oracle (a port of the reference algorithm, OpenMP over k-planes + pocketfft) timed on the host cores
    on a bounded sample of the same workload."""
    import numpy as np
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # before the OpenMP runtime of the oracle library is loaded
    from oracle import oracle as O
    rng = np.random.default_rng(1234)
    g = O.Grid((n, n, n), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology="PPP", halo=(3, 3, 3))
    m = O.NonhydrostaticModel(g, workers=cores)
    m.set(u=rng.uniform(-1, 1, (n, n, n)), v=rng.uniform(-1, 1, (n, n, n)), w=rng.uniform(-1, 1, (n, n, n)))
    dt = 0.1 * g.dx / max(np.abs(m.u).max(), np.abs(m.v).max(), np.abs(m.w).max())
    m.time_step(dt)  # warm-up (first step also computes the initial tendencies)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.time_step(dt)
    el = time.perf_counter() - t0
    return {"value": n ** 3 * steps / el, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{steps} RK3 steps of the same model at {n}^3 (C oracle, OpenMP {cores} threads, scipy pocketfft), {el:.1f} s"}


def latest_profile(pattern):
    """The newest committed profile summary matching profiles/<pattern> (PMC numbers cannot be collected inside a timed run)."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True)
    for path in paths:
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if isinstance(d, dict) and "kernels" in d:  # (bench lines kept under profiles/ match the same patterns)
            return os.path.relpath(path, ROOT), d
    return None, None


def event_time(fn, reps=10):
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def mapped_libraries(pattern):
    """Files matching `pattern` that are mapped into this process (/proc/self/maps): which librccl.so / libocn_hip.so actually run."""
    out = []
    try:
        for ln in open("/proc/self/maps"):
            path = ln.split(None, 5)[-1].strip() if ln.count(" ") >= 5 else ""
            if pattern in os.path.basename(path) and path not in out:
                out.append(path)
    except OSError:
        pass
    return out


def preflight(a, rank, world, local_rank, stdout_fd):
    """Everything a multi-GPU launch needs BEFORE the first GPU call, so that plumbing cannot be what fails on the 8-GPU node: the ranks
    exist and agree on the world, the gloo rendezvous on MASTER_ADDR:MASTER_PORT works, rank 0's RCCL unique id (ncclGetUniqueId needs
    no device) reaches every rank intact, the HIP library loads and exports the distributed entry points, the environment carries the
    dmabuf IPC switch, LOCAL_RANK is a valid device index when devices are visible.  Touches no GPU (torch.cuda.device_count() does not
    initialise one on this image)."""
    import ctypes as C
    import hashlib
    import torch
    import torch.distributed as dist
    import oceananigans_jl_amd as ocn
    checks = {}
    lib = ocn._lib.lib()
    need = ["ocn_comm_unique_id", "ocn_comm_init", "ocn_halo_exchange_begin", "ocn_halo_exchange_end", "ocn_halo_exchange_plane",
            "ocn_halo_exchange_pressure", "ocn_comm_all_gather", "ocn_comm_all_to_all", "ocn_dist_poisson_create", "ocn_comm_schedule"]
    checks["library_symbols"] = all(hasattr(lib, n) for n in need)
    checks["ipc_mode_env"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    # ONE librccl in the process, and the one this host always binds: libocn_hip.so names librccl.so.1 and `import torch` has mapped
    # its bundled copy first, so the loader resolves the name to that file (a plain-C host binds /opt/rocm/lib's through the rpath)
    rccl_files = mapped_libraries("librccl")
    import torch as _torch
    checks["one_librccl_mapped"] = len(rccl_files) == 1
    checks["librccl_is_torch_bundled"] = len(rccl_files) == 1 and os.path.dirname(rccl_files[0]) == os.path.join(os.path.dirname(_torch.__file__), "lib")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    checks["master_addr_is_loopback_or_set"] = bool(os.environ["MASTER_ADDR"])
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    checks["world_size"] = dist.get_world_size() == world == a.gpus
    uid = (C.c_ubyte * 128)()
    if rank == 0:
        ocn._lib.call("ocn_comm_unique_id", uid)
    box = [bytes(uid)]
    dist.broadcast_object_list(box, src=0)
    digest = hashlib.sha256(box[0]).digest()
    t = torch.tensor(list(digest), dtype=torch.int64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    checks["unique_id_identical_on_all_ranks"] = len(box[0]) == 128 and any(box[0]) and all(bool((g == gathered[0]).all()) for g in gathered)
    ranks = torch.zeros(world, dtype=torch.int64)
    ranks[rank] = local_rank + 1
    dist.all_reduce(ranks)
    ndev = torch.cuda.device_count()
    checks["local_rank_is_a_device"] = ndev == 0 or local_rank < ndev
    # the neighbour schedule every rank will issue (pure host function): its sends must meet the peers' receives
    ops = (ocn._lib.CCommOp * 8)()
    n = C.c_int32()
    ocn._lib.call("ocn_comm_schedule", ocn._lib.SCHED_STRIPS, rank, world, 0, ops, 8, C.byref(n))
    mine = torch.tensor([[ops[q].is_recv, ops[q].peer, ops[q].slot] for q in range(n.value)] or [[0, 0, 0]] * 4, dtype=torch.int64)
    allops = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allops, mine)
    ok = True
    if world > 1:
        for r in range(world):
            for is_recv, peer, slot in allops[r].tolist():
                if not is_recv:  # my send_west (0) must be received by my west neighbour as recv_east (3), send_east (1) as recv_west (2)
                    want = 3 if slot == 0 else 2
                    ok &= any(ir and pr == r and sl == want for ir, pr, sl in allops[peer].tolist())
    checks["neighbour_schedule_pairs_up"] = ok
    # the all-gather of the transpose-free solve as direct transfers: every rank sends its one chunk to every peer and receives chunk s from s
    gops = (ocn._lib.CCommOp * (2 * world))()
    ocn._lib.call("ocn_comm_schedule", ocn._lib.SCHED_ALL_GATHER, rank, world, 0, gops, 2 * world, C.byref(n))
    sends = sorted(gops[q].peer for q in range(n.value) if not gops[q].is_recv)
    recvs = sorted((gops[q].peer, gops[q].slot) for q in range(n.value) if gops[q].is_recv)
    others = [r for r in range(world) if r != rank]
    checks["all_gather_schedule_complete"] = sends == others and recvs == [(r, r) for r in others]
    all_ok = all(checks.values())
    flag = torch.tensor([1 if all_ok else 0], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    out = {"preflight": "ok" if int(flag[0]) == 1 else "FAILED", "ranks": world, "local_ranks_seen": [int(v) - 1 for v in ranks],
           "visible_devices": ndev, "checks_rank0": checks, "librccl": rccl_files, "master": f"{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}",
           "gpu_touched": False}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    raise SystemExit(0 if int(flag[0]) == 1 else 1)


def conservative_leg(a, ocn, arch, dist, rand, field_counter, N, rank, world):
    """The box workload through the conservative switches the advisor of round 3 named for a first multi-GPU run: the Python host
    (one entry point per kernel), the exchange of u, v, w AFTER the solve with a synchronous fill (OCN_DIST_CORRECT_ON_LOAD=0: one
    stream touches the communicator at a time) and ncclAllGather instead of R - 1 grouped point-to-point transfers.  Same synthetic
    initial state (the field counter is rewound afterwards), same warm-up and step counts, same barriers and max over ranks."""
    import numpy as np
    import torch
    saved = {k: os.environ.get(k) for k in ("OCN_DIST_CORRECT_ON_LOAD", "OCN_COMM_ALL_GATHER")}
    os.environ["OCN_DIST_CORRECT_ON_LOAD"] = "0"
    os.environ["OCN_COMM_ALL_GATHER"] = "collective"
    sync_timeout = float(os.environ.get("OCN_BENCH_SYNC_TIMEOUT_S", "300"))
    try:
        two_pi = 2 * np.pi
        grid = ocn.RectilinearGrid(arch, size=(N, N, N), x=(0, two_pi), y=(0, two_pi), z=(0, two_pi),
                                   topology=("Periodic", "Periodic", "Periodic"), halo=(3, 3, 3))
        model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO())
        for f in model.velocities:
            f.interior_view().copy_(rand(f.interior_view().shape))
        ocn.set(model)
        umax = torch.stack([f.interior_view().abs().max() for f in model.velocities]).max()
        umax = dist.allreduce_max(umax.reshape(1))[0]
        dt = 0.1 * grid.dx / float(umax)

        def barrier():
            ocn._lib.call("ocn_sync_timeout", ocn.architectures.stream_ptr(), sync_timeout)
            dist.barrier()

        for _ in range(a.warmup):
            ocn.time_step(model, dt)
        ocn.flush_tendencies(model)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ocn.time_step(model, dt)
        ocn.flush_tendencies(model)
        barrier()
        el = time.perf_counter() - t0
        el = float(dist.allreduce_max(torch.tensor([el], device="cuda", dtype=torch.float64))[0])
        sums = torch.stack([(f.interior_view() ** 2).sum() for f in model.prognostic_fields()])
        if hasattr(dist, "allreduce_sum"):
            sums = dist.allreduce_sum(sums)
        finite = bool(all(torch.isfinite(f.data).all() for f in model.prognostic_fields()))
        return {"ms_per_step": el / a.steps * 1e3, "value": float(N) ** 3 * a.steps / el, "steps": a.steps, "warmup": a.warmup,
                "driver": "python", "all_gather": "collective", "dist_correct_on_load": False, "finite": finite,
                "correct_on_load_model": bool(getattr(model, "dist_correct_on_load", False)),
                "sum_of_squares": [float(v) for v in sums]}
    except ocn.OcnError as e:
        print(f"[bench] rank {rank} of {world}: conservative sequence failed: {e}", file=sys.stderr, flush=True)
        os._exit(3)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        field_counter[0] = 0


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dist_poisson is not None:
        os.environ["OCN_DIST_POISSON_XTRI"] = "1" if a.dist_poisson == "xtri" else "0"
    if a.all_gather is not None:
        os.environ["OCN_COMM_ALL_GATHER"] = a.all_gather
    # stdout carries exactly ONE line (the JSON of rank 0): RCCL and gloo print banners to the C-level stdout on initialisation, so
    # file descriptor 1 points at stderr until the result is ready
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if a.preflight:
        preflight(a, rank, world, local_rank, stdout_fd)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE = {world}")
    import numpy as np
    import torch
    torch.cuda.set_device(local_rank)
    import oceananigans_jl_amd as ocn
    ocn._lib.lib()  # fail loudly if the HIP extension is missing
    ocn.set_math_mode(ocn.MATH_FAST if a.math == "fast" else ocn.MATH_STRICT)

    N = a.n
    comm_info = None
    if world > 1 or os.environ.get("OCN_FORCE_DISTRIBUTED") == "1":  # the env var exercises the RCCL path on one rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        arch = ocn.distributed.make_distributed(rank, world, local_rank)  # RCCL behind the C ABI (ocn_comm_*); gloo only bootstraps
        comm_info = arch.fabric.info()
        comm_info["library"] = mapped_libraries("librccl")  # the file(s) RCCL calls resolve to in this process (preflight asserts: one)
        comm_info["libocn_hip"] = mapped_libraries("libocn_hip")
        dist = arch.fabric
    else:
        dist = None
        arch = ocn.GPU()
    two_pi = 2 * np.pi
    gen = torch.Generator(device="cuda")
    field_counter = [0]

    def rand(shape):
        """U(-1, 1) values of this rank's slab of a GLOBAL synthetic field that does not depend on the number of ranks (the n-th field
        of a run is seeded 1234 + n and generated at its global extent, every rank keeps its x range): the checksum the line reports
        is then comparable between --gpus 1 and --gpus N."""
        gen.manual_seed(1234 + field_counter[0])
        field_counter[0] += 1
        nz_, ny_, nx_ = shape
        # (the whole global field is materialised on every rank -- a 1 GiB temporary per field at 512^3 -- and sliced: simple and rank-count independent)
        full = torch.rand((nz_, ny_, nx_ * world), generator=gen, device="cuda", dtype=torch.float64)
        out = (full[:, :, rank * nx_:(rank + 1) * nx_] * 2 - 1).contiguous()
        del full
        return out

    # ---- a partitioned run measures TWICE (box workload): first the conservative sequence -- Python host, synchronous exchange after
    # the solve (OCN_DIST_CORRECT_ON_LOAD=0), ncclAllGather for the solve's one exchange -- then the default one (C driver, strips in
    # flight on the communication stream under the solve, direct all-gather).  The line reports the default sequence with the
    # conservative one beside it (config.rccl.conservative: the A/B on real links); if the default sequence does not finish within
    # OCN_BENCH_FAST_DEADLINE_S (a collective whose peer never arrives, a host call that blocks), every rank's watchdog ends the process
    # and rank 0 prints the CONSERVATIVE measurement, labelled as such (config.rccl.fast_path says what happened).  No retry, no
    # re-exec, no new communicator: the fallback line was measured before the default sequence started.
    conservative = None
    fast_guard = {"armed": False, "line": None}
    if dist is not None and a.workload == "box" and os.environ.get("OCN_BENCH_CONSERVATIVE_FIRST", "1") != "0":
        conservative = conservative_leg(a, ocn, arch, dist, rand, field_counter, N, rank, world)
        torch.cuda.empty_cache()
        if rank == 0:
            print(f"[bench] conservative sequence: {conservative['ms_per_step']:.3f} ms per step on {world} rank(s)", file=sys.stderr, flush=True)
        import threading
        once = threading.Lock()
        fast_guard["once"] = once

        def emit_fallback(reason):
            """Ends this rank; rank 0 first prints the conservative measurement as the line of the run.  Called by the watchdog thread
            (the main thread may be blocked inside a library call: ctypes has released the GIL there) or by the main thread on an error."""
            if not once.acquire(blocking=False):
                return
            print(f"[bench] rank {rank} of {world}: default sequence {reason}; reporting the conservative measurement", file=sys.stderr, flush=True)
            if rank == 0:
                info = dict(comm_info or {})
                info["fast_path"] = reason
                info["conservative"] = conservative
                line = {"metric": "cell-updates/sec (whole node), 512^3 NonhydrostaticModel WENO5, 1/2/4/8 GPU",
                        "value": conservative["value"], "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                        "ms_per_step": conservative["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                        "dtype": "f64", "data": "synthetic", "fallback": True,
                        "config": {"workload": f"{N}^3 triply-periodic NonhydrostaticModel, WENO5, RK3, FFTBasedPoissonSolver, fp64",
                                   "grid": [N, N, N], "halo": 3, "math": a.math, "partition": f"x-slab/{world}",
                                   "finite": conservative["finite"], "rccl": info, "all_gather": "collective",
                                   "state_checksum": {"sum_of_squares": conservative["sum_of_squares"],
                                                      "fields": "prognostic fields in model order, global interior",
                                                      "after_steps": a.warmup + a.steps,
                                                      "comparable_across_n_gpus": hasattr(dist, "allreduce_sum")}},
                        "comm_stats": None, "strict_ms_per_step": None, "roofline": None, "step_roofline": None, "cpu_baseline": None,
                        "driver": "python"}
                sys.stdout.flush()
                os.dup2(stdout_fd, 1)
                print(json.dumps(line), flush=True)
            os._exit(0)  # (no teardown of a communicator that may be stuck)

        _FALLBACK[0] = emit_fallback
        deadline = float(os.environ.get("OCN_BENCH_FAST_DEADLINE_S", "240"))
        fast_guard["timer"] = threading.Timer(deadline, emit_fallback, args=(f"did not finish within {deadline:g} s",))
        fast_guard["timer"].daemon = True
        fast_guard["timer"].start()
        if os.environ.get("OCN_BENCH_INJECT_FAST_HANG") == "1":  # test hook: the default sequence never returns
            time.sleep(deadline + 30)

    hydro = a.workload == "config5"
    if a.workload == "config4":
        Nx, Nz = N, N // 2
        zf = config4_faces(Nz)
        grid = ocn.RectilinearGrid(arch, size=(N, N, Nz), x=(0, 64), y=(0, 64), z=zf,
                                   topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(C4["taux"])),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(C4["JT"]), bottom=ocn.GradientBoundaryCondition(C4["dTdz"])),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-C4["evap"]))}
        model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=C4["f"]),
                                        closure=ocn.AnisotropicMinimumDissipation(),
                                        buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(C4["alpha"], C4["beta"])),
                                        boundary_conditions=bcs)
        zc = torch.from_numpy(0.5 * (zf[1:] + zf[:-1])).to("cuda")
        T = model.field("T").interior_view()
        T.copy_(20 + C4["dTdz"] * zc[:, None, None] + 0.5e-6 * (rand(T.shape) + 1))
        model.field("S").interior_view().fill_(35.0)
        amp, dmin = 1e-2, float(np.diff(zf).min())
    elif hydro:
        Nx, Nz = 2 * N, N // 4
        H, L = C5["H"], C5["L"] * Nx / 1024.0
        grid = ocn.RectilinearGrid(arch, size=(Nx, Nx, Nz), x=(0, L), y=(0, L), z=(-H, 0.0), topology=("Periodic", "Periodic", "Bounded"),
                                   halo=(3, 3, 3))
        model = ocn.HydrostaticFreeSurfaceModel(grid, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                                free_surface=ocn.SplitExplicitFreeSurface(substeps=a.substeps), coriolis=ocn.FPlane(f=C5["f"]),
                                                closure=ocn.ScalarDiffusivity(ν=C5["nu"], κ=C5["kappa"]),
                                                buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(C5["alpha"], C5["beta"])))
        zc = torch.linspace(-H + H / (2 * Nz), -H / (2 * Nz), Nz, device="cuda", dtype=torch.float64)
        model.field("T").interior_view().copy_((20 + 0.01 * zc)[:, None, None].expand(Nz, Nx, Nx))
        model.field("S").interior_view().fill_(35.0)
        amp = C5["amp"]
    else:
        Nx, Nz = N, N
        grid = ocn.RectilinearGrid(arch, size=(N, N, N), x=(0, two_pi), y=(0, two_pi), z=(0, two_pi),
                                   topology=("Periodic", "Periodic", "Periodic"), halo=(3, 3, 3))
        model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO())
        amp, dmin = 1.0, grid.dx

    # synthetic initial condition generated on the device (fixed seed per rank)
    if hydro:
        for f in (model.u, model.v):
            f.interior_view().copy_(amp * rand(f.interior_view().shape))
        model.update_state(compute_tendencies=False)
        dt = 2.0 * grid.dx / np.sqrt(9.80665 * C5["H"])      # baroclinic gravity-wave CFL 2, barotropic 2 x 2 / 30
        step, flush = (lambda: model.time_step(dt)), model.flush_tendencies
        prognostic = [model.u, model.v] + list(model.tracers)
    else:
        for f in model.velocities:
            f.interior_view().copy_(amp * rand(f.interior_view().shape))
        ocn.set(model)  # halo fills + the dt = 1 projection of set!
        umax = torch.stack([f.interior_view().abs().max() for f in model.velocities]).max()
        if dist is not None:
            umax = dist.allreduce_max(umax.reshape(1))[0]
        dt = 0.1 * min(grid.dx, dmin) / float(umax)
        prognostic = model.prognostic_fields()
        drv = None
        if a.driver in ("c", "auto"):
            try:
                drv = ocn.RK3Driver(model) if a.workload == "box" else ocn.ModelRK3Driver(model)
            except (NotImplementedError, ocn.OcnError) as e:  # e.g. a slab size outside the library's slab pipelines
                if a.driver == "c":
                    raise
                print(f"[bench] C driver not applicable ({e}); using the Python host", file=sys.stderr)
        a.driver = "c" if drv is not None else "python"
        if drv is not None:
            step, flush = (lambda: drv.time_step(dt)), drv.flush
        else:
            step, flush = (lambda: ocn.time_step(model, dt)), (lambda: ocn.flush_tendencies(model))

    sync_timeout = float(os.environ.get("OCN_BENCH_SYNC_TIMEOUT_S", "300"))

    def die(msg):
        """A rank that cannot finish (a collective whose peer never arrived) reports and exits non-zero: no retry, no re-exec."""
        print(f"[bench] rank {rank} of {world}: {msg}", file=sys.stderr, flush=True)
        if _FALLBACK[0] is not None:
            _FALLBACK[0](f"failed: {msg}")
        os._exit(3)

    def barrier():
        try:
            # host waits with a deadline (ocn_sync_timeout polls an event; ocn_comm_barrier's all-reduce is waited for the same way)
            ocn._lib.call("ocn_sync_timeout", ocn.architectures.stream_ptr(), sync_timeout)  # (torch's current stream: every launch of a step)
            if dist is not None:
                dist.barrier()
        except ocn.OcnError as e:
            die(str(e))

    def timed(nsteps, mark=False):
        barrier()
        if mark:  # two empty marker launches delimit the timed steps in rocprofv3's per-dispatch output (tools/summarize_profile.py)
            ocn._lib.call("ocn_profile_marker", ocn.architectures.stream_ptr())
            barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        flush()  # the deferred last compute_tendencies! belongs to the timed steps
        barrier()
        el_ = time.perf_counter() - t0
        if mark:
            ocn._lib.call("ocn_profile_marker", ocn.architectures.stream_ptr())
        return el_

    for _ in range(a.warmup):
        step()
    flush()
    el = timed(a.steps, mark=True)
    if dist is not None:
        el = float(dist.allreduce_max(torch.tensor([el], device="cuda", dtype=torch.float64))[0])
    finite = bool(all(torch.isfinite(f.data).all() for f in prognostic))
    # state checksum after warm-up + timed steps: sum of squares of every prognostic field over the global interior.  The synthetic
    # initial fields do not depend on the number of ranks, so a --gpus N line can be checked against the --gpus 1 line of the same
    # command (agreement to ~1e-9 relative: the pressure solvers of the two paths differ in rounding).
    sums = torch.stack([(f.interior_view() ** 2).sum() for f in prognostic])
    checksum_global = True
    if dist is not None:
        if hasattr(dist, "allreduce_sum"):
            sums = dist.allreduce_sum(sums)
        else:
            checksum_global = False  # a transport without a sum all-reduce: the sums are this rank's only
    checksum = [float(v) for v in sums]
    legs_rel_diff = None
    if conservative is not None:
        # the two sequences started from the same state and took the same steps: their global sums agree to rounding (~1e-12 measured on
        # the one-rank RCCL world).  A default sequence that finishes with a DIFFERENT state (an exchange that raced on real links) is not
        # a measurement: the conservative line is reported instead.
        legs_rel_diff = max(abs(x - y) / max(abs(y), 1e-300) for x, y in zip(checksum, conservative["sum_of_squares"]))
        if not finite or not (legs_rel_diff <= float(os.environ.get("OCN_BENCH_LEGS_RTOL", "1e-7"))):
            _FALLBACK[0](f"finished with a different state than the conservative sequence (finite = {finite}, relative difference of the "
                         f"global sums of squares {legs_rel_diff:.3e}, sums {checksum})")

    # where a multi-GPU step spends its exchange time, measured on the device by the library (ocn_comm_enable_stats): 3 more steps
    comm_stats = None
    if dist is not None and hasattr(dist, "_h"):
        try:
            import ctypes as C
            ocn._lib.call("ocn_comm_enable_stats", dist._h, 1)
            nstat = 3
            for _ in range(nstat):
                step()
            flush()
            ocn._lib.call("ocn_sync_timeout", ocn.architectures.stream_ptr(), sync_timeout)
            ms = (C.c_double * 8)()
            ocn._lib.call("ocn_comm_stats", dist._h, ms)
            ocn._lib.call("ocn_comm_enable_stats", dist._h, 0)
            mine = torch.tensor(list(ms)[:5], device="cuda", dtype=torch.float64) / nstat
            worst = dist.allreduce_max(mine.clone())
            comm_stats = {"steps": nstat, "what": "device-side ms per step, rank 0 / max over ranks (events around each exchange on the stream it runs on)",
                          "strip_exchange_on_comm_stream_ms": [float(mine[0]), float(worst[0])],
                          "halo_wait_on_compute_stream_ms": [float(mine[1]), float(worst[1])],
                          "pressure_solve_exchange_ms": [float(mine[2]), float(worst[2])],
                          "pressure_plane_exchange_ms": [float(mine[3]), float(worst[3])],
                          "single_plane_exchange_ms": [float(mine[4]), float(worst[4])],
                          "strip_exchanges_per_step": ms[5] / nstat, "solve_exchanges_per_step": ms[6] / nstat}
        except Exception as e:  # diagnostics must never cost the measurement: the metric line is printed regardless
            comm_stats = {"error": f"{type(e).__name__}: {e}"}

    # the bit-exact (strict IEEE, reference operand order) build of the same step, driver-visible
    strict_ms = None
    if a.math == "fast" and not a.no_strict and world == 1:
        # (one GPU only: on a node the extra leg would put more collectives behind the measurement for a number the N = 1 line carries)
        try:
            ocn.set_math_mode(ocn.MATH_STRICT)
            step()
            flush()
            strict_ms = timed(3) / 3 * 1e3
        finally:
            ocn.set_math_mode(ocn.MATH_FAST)

    local_cells = grid.Nx * grid.Ny * grid.Nz
    cells = Nx * (Nx if hydro else N) * Nz
    value = cells * a.steps / el
    roofline, step_roofline = None, None
    if a.no_kernel_timing:
        pass
    elif not hydro:
        # dominant kernel: the fused WENO5 momentum-tendency launch, timed live with events on the launching stream, (a) plain
        # (compute_Gu!/Gv!/Gw! alone) and (b) as the step runs it on one rank (correction on load + next substep as epilogue)
        Gn = model.timestepper.Gn  # (completes a deferred tendency launch)

        def plain():
            ocn._lib.call("ocn_compute_momentum_tendencies", grid.cref, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr,
                          Gn[2].ptr, None, 0)

        plain_ms = event_time(plain)
        in_step_ms, in_step_bytes = None, None
        if a.workload == "box" and world == 1 and getattr(model, "correct_on_load", False):
            Gm = model.timestepper._Gm
            alt = [torch.zeros_like(f.data) for f in model.velocities]

            def in_step():
                ocn._lib.call("ocn_compute_momentum_tendencies_rk3", grid.cref, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr,
                              Gn[2].ptr, Gm[0].ptr, Gm[1].ptr, Gm[2].ptr, alt[0].data_ptr(), alt[1].data_ptr(), alt[2].data_ptr(),
                              float(dt), 5.0 / 12.0, -17.0 / 60.0, 1, model.pNHS.ptr, float(dt) * 8 / 15, None, 0)

            in_step_ms, in_step_bytes = event_time(in_step), TENDENCY_BYTES_IN_STEP
            del alt
        prof_path, prof = latest_profile(f"r*_{N}.json") if a.workload == "box" else latest_profile("r*_config4.json")
        kern = {}
        if prof is not None and world == 1:
            for k in prof["kernels"]:
                if "momentum_tendencies_tiled" in k["name"]:
                    # template arguments <TZ, TX, TY, W, PC[, OB]>: PC = pressure correction on load = the in-step variant
                    targs = [x.strip() for x in k["name"].split("<", 1)[1].rstrip(">").split(",")] if "<" in k["name"] else []
                    kern["in_step" if len(targs) > 4 and targs[4] == "true" else "plain"] = k
        pk = kern.get("in_step") or kern.get("plain") or {}
        ref_ms = in_step_ms if in_step_ms is not None else plain_ms
        instr = pk.get("SQ_INSTS_VALU")  # VALU wave-instructions per launch of the profiled variant (committed PMC pass)
        achieved = None if instr is None else instr * (local_cells / prof.get("cells", local_cells)) / (ref_ms * 1e-3) / 1e9
        # achieved = ALGORITHMIC HBM bytes of the launch / its live duration, against the 8 TB/s roofline.  The kernel is
        # bound by fp64 VALU issue (9 WENO5 reconstructions per cell), so this fraction is low by construction; `valu` carries the issue-
        # rate figures (VALU wave-instructions per launch from the committed --pmc pass / the live launch time).
        algo_bytes = in_step_bytes if in_step_ms is not None else TENDENCY_BYTES_PLAIN
        hbm_achieved = algo_bytes * local_cells / (ref_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": "momentum_tendencies_tiled (fused compute_Gu/Gv/Gw, WENO5)",
            "achieved": hbm_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBPS,
            "traffic": pk.get("traffic_bytes"),  # measured HBM bytes per launch (FETCH_SIZE + WRITE_SIZE of the committed --pmc passes, calibrated)
            "algorithmic_bytes_per_cell": algo_bytes, "cells_per_launch": local_cells,
            "kernel_ms": ref_ms, "variant": "in-step (correction on load + substep epilogue)" if in_step_ms is not None else "plain",
            "frac_hbm": hbm_achieved / HBM_PEAK_GBPS,  # (same number; kept under the name earlier records use)
            "valu": {"bound": "valu_fp64", "achieved": achieved, "peak": VALU_PEAK_GWAVEINSTR, "unit": "G wave-instr/s",
                     "frac": None if achieved is None else achieved / VALU_PEAK_GWAVEINSTR,
                     "sustained_GHz": pk.get("clock_GHz_under_pmc"),  # GRBM_GUI_ACTIVE / 8 XCDs / launch time of the committed PMC pass
                     "valu_wave_instr_per_launch": instr, "valu_busy_frac_pmc": pk.get("valu_busy"),
                     "note": "the limiter of this kernel: VALU wave-instructions per launch (SQ_INSTS_VALU of the committed rocprofv3 --pmc pass "
                             "named in pmc_source) / the launch time measured live here; peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"},
            "hbm": {"in_step": None if in_step_ms is None else {
                        "algorithmic_bytes_per_cell": in_step_bytes, "kernel_ms": in_step_ms,
                        "achieved_GBps": in_step_bytes * local_cells / (in_step_ms * 1e-3) / 1e9,
                        "frac": in_step_bytes * local_cells / (in_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "measured_bytes_per_cell": kern.get("in_step", {}).get("traffic_bytes_per_cell")},
                    "plain": {"algorithmic_bytes_per_cell": TENDENCY_BYTES_PLAIN, "kernel_ms": plain_ms,
                              "achieved_GBps": TENDENCY_BYTES_PLAIN * local_cells / (plain_ms * 1e-3) / 1e9,
                              "frac": TENDENCY_BYTES_PLAIN * local_cells / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "measured_bytes_per_cell": kern.get("plain", {}).get("traffic_bytes_per_cell")},
                    "peak_GBps": HBM_PEAK_GBPS},
            "pmc_source": prof_path}
        if a.workload == "config4":
            measured = None if prof is None else prof.get("step_bytes_per_cell")
            step_roofline = {"measured_bytes_per_cell_step": measured,  # committed PMC passes of this workload (set! and warm-up launches included)
                             "measured_GBps": None if measured is None else measured * value / 1e9,
                             "frac_of_8TBps": None if measured is None else measured * value / 1e9 / (HBM_PEAK_GBPS * world),
                             "pmc_source": prof_path}
        if a.workload == "box":
            measured = None if prof is None else prof.get("step_bytes_per_cell")
            step_roofline = {
                "measured_bytes_per_cell_step": measured,  # sum over the TIMED steps' launches of FETCH_SIZE + WRITE_SIZE (committed PMC passes, marker-delimited)
                "measured_GBps": None if measured is None else measured * value / 1e9,
                "frac_of_8TBps": None if measured is None else measured * value / 1e9 / (HBM_PEAK_GBPS * world),
                "algorithmic_bytes_per_cell_step": BACKEND_BYTES_PER_CELL_STEP,  # THIS backend's compulsory traffic (3 x (104 tendency launch + 104 Poisson pipeline))
                "algorithmic_GBps": BACKEND_BYTES_PER_CELL_STEP * value / 1e9,
                "algorithmic_frac_of_8TBps": BACKEND_BYTES_PER_CELL_STEP * value / 1e9 / (HBM_PEAK_GBPS * world),
                "measured_over_algorithmic": None if measured is None else measured / BACKEND_BYTES_PER_CELL_STEP,
                "measured_window": None if prof is None else prof.get("window"),
                "reference_decomposition_bytes_per_cell_step": ALGO_BYTES_PER_CELL_STEP,
                "reference_equivalent_GBps": ALGO_BYTES_PER_CELL_STEP * value / 1e9,  # what the REFERENCE's launch sequence would have to move at this step rate; not HBM traffic of this backend
                "pmc_source": prof_path}
    else:
        # config 5: dominant kernel = the WENO tracer launch (tendency + diffusion + AB2 step), timed live; 56 B per cell:
        # read c, u, v, w, G-; write G, c_out
        nh = model._nh
        Gn, Gm = nh.timestepper._Gn, nh.timestepper._Gm
        alt = torch.zeros_like(model.tracers[0].data)
        import ctypes as C

        def tracer_launch():
            ocn._lib.call("ocn_compute_tracer_tendency_terms_rk3", grid.cref, C.byref(nh._terms), C5["kappa"], None, None, model.u.ptr, model.v.ptr,
                          model.w.ptr, model.tracers[0].ptr, Gn[3].ptr, Gm[3].ptr, alt.data_ptr(), float(dt), 1.6, -0.6, 1, None, 0)

        k_ms = event_time(tracer_launch)
        prof_path, prof = latest_profile("r*_config5.json")
        pk = {}
        if prof is not None:
            for k in prof["kernels"]:
                if "tracer_tendency_tiled" in k["name"]:
                    pk = k
        bytes_cell = 56.0
        ach = bytes_cell * local_cells / (k_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "tracer_tendency_tiled (WENO5 div_Uc + diffusion + AB2 step of one tracer)", "achieved": ach,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": pk.get("traffic_bytes"),
                    "kernel_ms": k_ms, "algorithmic_bytes_per_cell": bytes_cell, "launches_per_step": 2,
                    "valu_busy_frac_pmc": pk.get("valu_busy"), "pmc_source": prof_path}
        measured = None if prof is None else prof.get("step_bytes_per_cell")
        step_roofline = {"algorithmic_bytes_per_cell_step": 256.0,  # momentum pass 80 + 2 tracers x 56 + corrector/w 40 + pHY 24
                         "achieved_GBps": 256.0 * value / 1e9, "frac_of_8TBps": 256.0 * value / 1e9 / HBM_PEAK_GBPS,
                         "measured_bytes_per_cell_step": measured,
                         "measured_GBps": None if measured is None else measured * value / 1e9, "pmc_source": prof_path}

    if a.workload == "config4":
        workload = (f"{N}x{N}x{Nz} (Periodic, Periodic, Bounded) stretched z, ocean_wind_mixing_and_convection setup "
                    "(WENO5, AnisotropicMinimumDissipation, SeawaterBuoyancy, FPlane, flux/gradient BCs, T and S), RK3, "
                    "FourierTridiagonalPoissonSolver, fp64")
    elif hydro:
        workload = (f"{Nx}x{Nx}x{Nz} (Periodic, Periodic, Bounded) HydrostaticFreeSurfaceModel, VectorInvariant momentum, WENO5 tracer advection "
                    f"(T, S), SplitExplicitFreeSurface(substeps={a.substeps}), linear SeawaterBuoyancy, FPlane, ScalarDiffusivity, QAB2, fp64")
    else:
        workload = f"{N}^3 triply-periodic NonhydrostaticModel, WENO5, RK3, FFTBasedPoissonSolver, fp64"
    # which exchange pattern the pressure solve of a partitioned run uses (ocn_dist_poisson_pipeline): 3 = no transposes, one all-gather
    pipeline = getattr(getattr(getattr(model, "pressure_solver", None), "impl", None), "fast", None)
    out = {
        "metric": "cell-updates/sec (whole node), 512^3 NonhydrostaticModel WENO5, 1/2/4/8 GPU",
        "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "grid": [Nx, Nx if hydro else N, Nz], "halo": 3, "math": a.math, "partition": f"x-slab/{world}",
                   "finite": finite, "rccl": comm_info, "dist_pressure_pipeline": pipeline,
                   "all_gather": os.environ.get("OCN_COMM_ALL_GATHER", "direct") if world > 1 else None,
                   "state_checksum": {"sum_of_squares": checksum, "fields": "prognostic fields in model order, global interior",
                                      "after_steps": a.warmup + a.steps,
                                      "comparable_across_n_gpus": checksum_global}},
        "comm_stats": comm_stats,
        "strict_ms_per_step": strict_ms,
        "roofline": roofline,
        "step_roofline": step_roofline,
        "driver": a.driver,
    }
    if a.workload == "box" and step_roofline is not None:
        # north_star's unit, "the WENO5 tendency + pressure-correction step" = one RK3 substage (substep + projection + tendencies):
        # a third of the step; bytes from the committed PMC passes, against the 8 TB/s HBM roofline (north_star asks >= 0.40)
        mb = step_roofline.get("measured_bytes_per_cell_step")
        sub_ms = el / a.steps * 1e3 / 3
        out["substage"] = {"what": "one RK3 substage: rk3_substep + pressure projection (FFT solve) + WENO5 tendencies, i.e. north_star's "
                                   "'WENO5 tendency + pressure-correction step'",
                           "ms": sub_ms, "measured_bytes_per_cell": None if mb is None else mb / 3,
                           "measured_GBps": None if mb is None else mb / 3 * cells / (sub_ms * 1e-3) / 1e9,
                           "frac_of_8TBps": None if mb is None else mb / 3 * cells / (sub_ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world),
                           "reference_decomposition_bytes_per_cell": ALGO_BYTES_PER_CELL_STEP / 3,
                           "reference_decomposition_frac_of_8TBps": ALGO_BYTES_PER_CELL_STEP / 3 * cells / (sub_ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world),
                           "pmc_source": step_roofline.get("pmc_source")}
    if conservative is not None:
        if not fast_guard["once"].acquire(blocking=False):  # the watchdog is already printing the fallback line
            time.sleep(3600)
        fast_guard["timer"].cancel()
        _FALLBACK[0] = None
        out["config"]["rccl"] = dict(comm_info or {}, fast_path="ok", conservative=conservative, legs_relative_difference=legs_rel_diff)
        # (a teardown that hangs after the line is out ends silently)
        import threading
        end = threading.Timer(90.0, lambda: os._exit(0))
        end.daemon = True
        end.start()
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            if a.workload == "box":
                out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.cpu_steps)
            elif hydro:
                out["cpu_baseline"] = cpu_baseline_config5(256, 32, 12, a.substeps)
            else:
                out["cpu_baseline"] = cpu_baseline_config4(min(a.cpu_n, 128), a.cpu_steps)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)  # anything the teardown prints goes to stderr again
    if dist is not None:
        dist.barrier()
        dist.close()


_FALLBACK = [None]  # set by main() once a conservative measurement of a partitioned run exists

if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except Exception as exc:
        if _FALLBACK[0] is not None:
            import traceback
            traceback.print_exc()
            _FALLBACK[0](f"failed: {type(exc).__name__}: {exc}")
        raise
