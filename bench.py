#!/usr/bin/env python3
"""bench.py -- cell-updates/s of full RK3 time steps of the WENO5 NonhydrostaticModel (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--size 512] [--math fast|strict]

Workload (config.workload): N^3 triply periodic RectilinearGrid, extent (2 pi)^3, halo 3, fp64,
advection = WENO() (5th order), RungeKutta3, FFT-based pressure solver, no tracers/closure/buoyancy
(BASELINE.json configs[1]/[2] at --size 256 / 512).  Inputs are synthetic: u, v, w ~ U(-1, 1) from a fixed seed,
projected to be divergence free by set! (one dt = 1 pressure solve); dt = 0.1 dx / max|u|.
A "step" is one full time_step!: 3 x (substep, pressure projection, tendencies).  With --gpus N > 1 the same global
grid is x-slab partitioned over N ranks (strong scaling), one process per GPU, RCCL halo exchange + all-to-all
transposes.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

ALGO_BYTES_PER_CELL_STEP = 1680.0  # 560 + 584 + 536 B per cell per RK3 step (reference decomposition)
TENDENCY_BYTES_PER_CELL = 48.0     # fused compute_Gu/Gv/Gw launch: read u, v, w once, write Gu, Gv, Gw (fp64)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", dest="n", type=int, default=512, help="grid points per dimension (512 = the metric's config)")
    ap.add_argument("--math", choices=("fast", "strict"), default="fast")
    ap.add_argument("--workload", choices=("box", "config4"), default="box",
                    help="box: the metric's triply-periodic N^3 box (default); config4: BASELINE.json configs[3], the "
                         "ocean_wind_mixing_and_convection setup on N x N x N/2 (Periodic, Periodic, Bounded) with stretched z")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--driver", choices=("python", "c"), default="python",
                    help="host orchestration of the step: the Python mirror of the reference's time_step!, or ocn_rk3_driver_time_step (one C call per step; box workload, one GPU)")
    ap.add_argument("--cpu-n", type=int, default=160, help="grid size of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=12)
    return ap.parse_args()


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
    k = np.arange(1, Nz + 2)
    h = (k - 1) / Nz
    zeta0 = 1 + (h - 1) / refinement
    Sigma = (1 - np.exp(-stretching * h)) / (1 - np.exp(-stretching))
    return Lz * (zeta0 * Sigma - 1)


# physical constants of examples/ocean_wind_mixing_and_convection.jl:79-110
C4 = dict(JT=200.0 / (1026.0 * 3991.0), dTdz=0.01, taux=-1.225 / 1026.0 * 2.5e-3 * 10 * 10, evap=1e-3 / 3600, f=1e-4,
          alpha=2e-4, beta=8e-4)


def cpu_baseline_config4(n, steps):
    """The CPU oracle with the same physics (AMD closure, SeawaterBuoyancy, FPlane, boundary conditions) on n x n x n/2."""
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


def cpu_baseline(n, steps):
    """The CPU oracle (a port of the reference algorithm, OpenMP over k-planes + pocketfft) timed on the host cores
    on a bounded sample of the same workload."""
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


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    import oceananigans_jl_amd as ocn
    ocn._lib.lib()  # fail loudly if the HIP extension is missing
    ocn.set_math_mode(ocn.MATH_FAST if a.math == "fast" else ocn.MATH_STRICT)

    N = a.n
    if world > 1 or os.environ.get("OCN_FORCE_DISTRIBUTED") == "1":  # the env var exercises the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(world))
    else:
        dist = None
        arch = ocn.GPU()
    two_pi = 2 * np.pi
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    if a.workload == "config4":
        Nz = N // 2
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
        T.copy_(20 + C4["dTdz"] * zc[:, None, None] + 1e-6 * torch.rand(T.shape, generator=gen, device="cuda", dtype=torch.float64))
        model.field("S").interior_view().fill_(35.0)
        amp, dmin = 1e-2, float(np.diff(zf).min())
    else:
        Nz = N
        grid = ocn.RectilinearGrid(arch, size=(N, N, N), x=(0, two_pi), y=(0, two_pi), z=(0, two_pi),
                                   topology=("Periodic", "Periodic", "Periodic"), halo=(3, 3, 3))
        model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO())
        amp, dmin = 1.0, grid.dx

    # synthetic initial condition generated on the device (fixed seed per rank)
    for f in model.velocities:
        iv = f.interior_view()
        iv.copy_(amp * (torch.rand(iv.shape, generator=gen, device=iv.device, dtype=torch.float64) * 2 - 1))
    ocn.set(model)  # halo fills + the dt = 1 projection of set!
    umax = torch.stack([f.interior_view().abs().max() for f in model.velocities]).max()
    if dist is not None:
        dist.all_reduce(umax, op=dist.ReduceOp.MAX)
    dt = 0.1 * min(grid.dx, dmin) / float(umax)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.driver == "c":
        if world > 1 or a.workload != "box":
            raise SystemExit("--driver c: box workload on one GPU")
        drv = ocn.RK3Driver(model)
        step, flush = (lambda: drv.time_step(dt)), drv.flush
    else:
        step, flush = (lambda: ocn.time_step(model, dt)), (lambda: ocn.flush_tendencies(model))
    for _ in range(a.warmup):
        step()
    flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    flush()  # the deferred last compute_tendencies! belongs to the timed steps
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    finite = bool(all(torch.isfinite(f.data).all() for f in model.prognostic_fields()))

    # dominant kernel: the fused WENO5 momentum-tendency launch, timed live with events on the launching stream
    local_cells = grid.Nx * grid.Ny * grid.Nz
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    Gn = model.timestepper.Gn  # (completes a deferred tendency launch)

    def weno_launch():  # compute_Gu!/Gv!/Gw! of the advection term alone: the fused WENO5 kernel
        ocn._lib.call("ocn_compute_momentum_tendencies", grid.cref, model.u.ptr, model.v.ptr, model.w.ptr, Gn[0].ptr, Gn[1].ptr,
                      Gn[2].ptr, None, 0)

    weno_launch()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        weno_launch()
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / reps
    achieved = TENDENCY_BYTES_PER_CELL * local_cells / (kern_ms * 1e-3) / 1e9

    # HBM traffic of that kernel from the committed PMC profile of this workload (separate rocprofv3 --pmc passes,
    # calibrated as MI355X_MICROARCH.md prescribes; tools/profile_gpu.sh + tools/summarize_profile.py), per launch
    traffic = None
    try:
        import glob
        for path in (sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{N}.json")), reverse=True) if a.workload == "box" else []):
            prof = json.load(open(path))
            k = [k for k in prof["kernels"] if "momentum_tendencies" in k["name"] and "traffic_bytes" in k]
            if k and world == 1:
                traffic = k[0]["traffic_bytes"]
                break
    except Exception:
        traffic = None

    value = N * N * Nz * a.steps / el
    if a.workload == "config4":
        workload = (f"{N}x{N}x{Nz} (Periodic, Periodic, Bounded) stretched z, ocean_wind_mixing_and_convection setup "
                    "(WENO5, AnisotropicMinimumDissipation, SeawaterBuoyancy, FPlane, flux/gradient BCs, T and S), RK3, "
                    "FourierTridiagonalPoissonSolver, fp64")
    else:
        workload = f"{N}^3 triply-periodic NonhydrostaticModel, WENO5, RK3, FFTBasedPoissonSolver, fp64"
    out = {
        "metric": "cell-updates/sec (whole node), 512^3 NonhydrostaticModel WENO5, 1/2/4/8 GPU",
        "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "grid": [N, N, Nz], "halo": 3, "math": a.math, "partition": f"x-slab/{world}", "finite": finite},
        "roofline": {"bound": "hbm", "kernel": "momentum_tendencies (fused compute_Gu/Gv/Gw, WENO5)",
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "kernel_ms": kern_ms, "algorithmic_bytes_per_cell": TENDENCY_BYTES_PER_CELL,
                     # SURVEY 8(d) books the reference's three tendency kernels at 3 x (3 r + 1 w) x 8 = 96 B/cell; against that
                     # accounting (the one step_roofline's 1680 B uses) the same launch reaches:
                     "frac_at_reference_accounting_96B": 96.0 * local_cells / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "note": "fp64-VALU-bound kernel (~690 fp64 VALU of ~1190 instructions per cell, 77% VALU issue utilisation): see DESIGN.md section 4"},
        "step_roofline": {"algorithmic_bytes_per_cell_step": ALGO_BYTES_PER_CELL_STEP,
                          "achieved_GBps": ALGO_BYTES_PER_CELL_STEP * value / 1e9,
                          "frac_of_8TBps": ALGO_BYTES_PER_CELL_STEP * value / 1e9 / (HBM_PEAK_GBPS * world)},
    }
    if a.workload == "config4":
        out["step_roofline"] = None  # SURVEY 8(d)'s 1680 B/cell/step is the accounting of the advection-only periodic box
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline(a.cpu_n, a.cpu_steps) if a.workload == "box"
                                   else cpu_baseline_config4(min(a.cpu_n, 128), a.cpu_steps))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
