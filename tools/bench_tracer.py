#!/usr/bin/env python3
"""Micro-benchmark of the tracer tendency kernel (plain WENO advection, and with everything folded in).
   OCN_TRACER_KERNEL=direct selects the direct kernel.  tools/bench_tracer.py [Nx] [Nz]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oceananigans_jl_amd as ocn
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ocn.set_math_mode(ocn.MATH_FAST)
zf = -32.0 * (1 - np.linspace(0, 1, Nz + 1)) ** 1.5
g = ocn.RectilinearGrid(ocn.GPU(), size=(Nx, Nx, Nz), x=(0, 64), y=(0, 64), z=zf, topology=("Periodic", "Periodic", "Bounded"), halo=(3, 3, 3))
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
F = {}
for name, loc in (("u", 1), ("v", 2), ("w", 4), ("c", 0), ("k", 0), ("Gm", 0)):
    f = ocn.Field(loc, g); f.data.copy_(torch.rand(f.data.shape, generator=gen, device="cuda", dtype=torch.float64) - (0.5 if loc else 0)); F[name] = f
G, out = ocn.Field(0, g), ocn.Field(0, g)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
plain = lambda: ocn._lib.call("ocn_compute_tracer_tendency", g.cref, F["u"].ptr, F["v"].ptr, F["w"].ptr, F["c"].ptr, G.ptr, None, 0)
t = ocn._lib.CModelTerms(); t.closure = 1; t.nu = 1e-3
fb = ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(1e-4))
fused_k = lambda: ocn._lib.call("ocn_compute_tracer_tendency_terms_rk3", g.cref, C.byref(t), 1e-3, None, C.byref(fb.c_struct(g)), F["u"].ptr, F["v"].ptr, F["w"].ptr,
                                F["c"].ptr, G.ptr, F["Gm"].ptr, out.ptr, 0.1, 0.4, -0.3, 1, None, 0)
t2 = ocn._lib.CModelTerms(); t2.closure = 2; t2.nu_e = F["k"].ptr
fused_ke = lambda: ocn._lib.call("ocn_compute_tracer_tendency_terms_rk3", g.cref, C.byref(t2), 0.0, F["k"].ptr, C.byref(fb.c_struct(g)), F["u"].ptr, F["v"].ptr, F["w"].ptr,
                                 F["c"].ptr, G.ptr, F["Gm"].ptr, out.ptr, 0.1, 0.4, -0.3, 1, None, 0)
print(f"kernel={os.environ.get('OCN_TRACER_KERNEL', 'tiled')} {Nx}x{Nx}x{Nz}: plain {timeit(plain):.3f} ms, fused(kappa const) {timeit(fused_k):.3f} ms, fused(kappa field) {timeit(fused_ke):.3f} ms")
