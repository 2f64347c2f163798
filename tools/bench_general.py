#!/usr/bin/env python3
"""Step time of NonhydrostaticModel(advection = WENO()) on grids with walls (the direction-generic kernels + the cosine-transform Poisson
solver) next to the periodic box of the same size:  tools/bench_general.py [N] [steps] [topologies, e.g. PBB,BBB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ocn.set_math_mode(ocn.MATH_FAST)
T = {"P": "Periodic", "B": "Bounded"}
for topo in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("PPP", "PPB", "PBB", "BBB")):
    g = ocn.RectilinearGrid(ocn.GPU(), size=(N, N, N), x=(0, 1), y=(0, 1), z=(0, 1), topology=tuple(T[t] for t in topo))
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for f in m.velocities:
        v = f.interior_view()
        v.copy_(2 * torch.rand(v.shape, generator=gen, device="cuda", dtype=torch.float64) - 1)
    ocn.set(m)
    dt = 0.1 * g.dx / max(float(f.interior_view().abs().max()) for f in m.velocities)
    for _ in range(3):
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m); ocn.sync_device()
    t0 = time.perf_counter()
    for _ in range(steps):
        ocn.time_step(m, dt)
    ocn.flush_tendencies(m); ocn.sync_device()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the pressure solve alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ocn.solve_for_pressure(m.pNHS, m.pressure_solver, dt, m.velocities)
    e1.record(); torch.cuda.synchronize()
    # the momentum tendency launch alone (box + frames on grids with walls; OCN_GENERAL_TILED=0: the per-cell kernel over everything)
    Gn = m.timestepper.Gn
    t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call = lambda: ocn._lib.call("ocn_compute_momentum_tendencies", g.cref, m.u.ptr, m.v.ptr, m.w.ptr, Gn[0].ptr, Gn[1].ptr, Gn[2].ptr, None, 0)
    call(); torch.cuda.synchronize()
    t0e.record()
    for _ in range(10):
        call()
    t1e.record(); torch.cuda.synchronize()
    tend_ms = t0e.elapsed_time(t1e) / 10
    print(f"{topo} N={N}: momentum tendencies {tend_ms:.3f} ms per launch (OCN_GENERAL_TILED={os.environ.get('OCN_GENERAL_TILED', '1')})", flush=True)
    print(f"{topo} N={N}: {ms:.2f} ms/step ({N**3/ms/1e6:.2f} Gcell-updates/s), solve_for_pressure {e0.elapsed_time(e1)/5:.2f} ms, fused={m.fuse_stage_boundaries}", flush=True)
    del m, g
    torch.cuda.empty_cache()
