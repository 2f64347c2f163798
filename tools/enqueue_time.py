"""Host enqueue time of one time step (no synchronisation inside the measured region) for the box and config 4, Python host against
the C drivers.  Sizes are reduced so that the launch queue never fills: the host cost does not depend on the grid size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oceananigans_jl_amd as ocn
from bench import C4, config4_faces


def measure(step, flush, n=8):
    for _ in range(3):
        step()
    flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    flush()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


def box(n=128):
    g = ocn.RectilinearGrid(ocn.GPU(), size=(n, n, n), x=(0, 1), y=(0, 1), z=(0, 1), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
    rng = np.random.default_rng(0)
    ocn.set(m, u=rng.uniform(-1, 1, (n, n, n)), v=rng.uniform(-1, 1, (n, n, n)), w=rng.uniform(-1, 1, (n, n, n)))
    return m


def config4(n=(128, 128, 64)):
    z = config4_faces(n[2])
    g = ocn.RectilinearGrid(ocn.GPU(), size=n, x=(0, 64.0), y=(0, 64.0), z=z, topology=("Periodic", "Periodic", "Bounded"))
    bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-3e-4)),
           "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
           "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-2.8e-7))}
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=1e-4), closure=ocn.AnisotropicMinimumDissipation(),
                                buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)), boundary_conditions=bcs)
    rng = np.random.default_rng(0)
    ocn.set(m, u=1e-2 * rng.uniform(-1, 1, n), v=1e-2 * rng.uniform(-1, 1, n), T=20 + 1e-2 * rng.uniform(-1, 1, n), S=35 + 1e-2 * rng.uniform(-1, 1, n))
    return m


if __name__ == "__main__":
    ocn.set_math_mode(ocn.MATH_FAST)
    for name, build, Drv, dt in (("box", box, ocn.RK3Driver, 1e-3), ("config4", config4, ocn.ModelRK3Driver, 1.0)):
        m = build()
        e, t = measure(lambda: ocn.time_step(m, dt), lambda: ocn.flush_tendencies(m))
        print(f"{name}: python host enqueue {e:.3f} ms / step (step incl. GPU {t:.3f} ms)")
        m = build()
        d = Drv(m)
        e, t = measure(lambda: d.time_step(dt), d.flush)
        print(f"{name}: C driver   enqueue {e:.3f} ms / step (step incl. GPU {t:.3f} ms)")
