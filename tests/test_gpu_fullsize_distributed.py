"""BASELINE.json's multi-GPU configurations at their REAL per-rank shapes, on one GPU: R = 8 slab-x ranks run as threads of this
process (tests/test_gpu_distributed.py: ThreadFabric) with the real HIP kernels and the real choreography, against the single-rank
model on the same GPU.

  configs[2]  512^3 triply periodic box: 64 x 512 x 512 per rank, BOTH pressure pipelines (the transpose-free x-tridiagonal /
              all-gather one, csrc/xtri.hip, and the all-to-all slab pipeline north_star names)
  configs[3]  512 x 512 x 256 (Periodic, Periodic, Bounded) stretched z, ocean_wind_mixing_and_convection physics: 64 x 512 x 256 per rank
  configs[4]  1024 x 1024 x 128 HydrostaticFreeSurfaceModel, split-explicit free surface with 30 substeps: 128-wide slabs,
              30-wide barotropic halos, bit-identical to one rank

Reference: distributed_architectures.jl:167-297, test_distributed_models.jl:335-353, test_distributed_poisson_solvers.jl:70-148."""
import gc

import numpy as np
import pytest
import torch

from test_gpu_distributed import _run_ranks, _run_ranks_local

pytestmark = pytest.mark.gpu
P = "Periodic"
R = 8


def _release():
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pipeline", ["xtri", "alltoall", "xtri-library-transport-c-driver"])
def test_config3_box_512_on_8_ranks_matches_single_rank(ocn, pipeline, monkeypatch):
    """512^3 periodic box, WENO5, RK3: two steps on 8 ranks (64 x 512 x 512 each) against the single-rank model, strict math:
    u, v, w within 1e-11 max|u| and the pressure within 1e-10 (north_star's tolerance), with either distributed pressure pipeline.
    Third flavour: what `bench.py --gpus 8` runs apart from RCCL itself -- ocn_rk3_driver_create_distributed (one C call per rank-step)
    over the library's own transport code (csrc/comm.hip with in-process mailboxes between 8 distinct peers)."""
    library = pipeline.endswith("c-driver")
    pipeline = pipeline.split("-")[0]
    monkeypatch.setenv("OCN_DIST_POISSON_XTRI", "1" if pipeline == "xtri" else "0")
    N = (512, 512, 512)
    two_pi = 2 * np.pi
    ext = dict(x=(0, two_pi), y=(0, two_pi), z=(0, two_pi), topology=(P, P, P), halo=(3, 3, 3))
    rng = np.random.default_rng(1234)
    init = {n: rng.uniform(-1, 1, N) for n in "uvw"}
    dt = 0.1 * (two_pi / 512)
    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext), advection=ocn.WENO())
    ocn.set(sm, **init)
    for _ in range(2):
        ocn.time_step(sm, dt)
    ocn.flush_tendencies(sm)
    ocn.sync_device()
    ref = [f.interior() for f in sm.velocities] + [sm.pNHS.interior()]
    del sm
    _release()

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        assert (g.Nx, g.Ny, g.Nz) == (64, 512, 512)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO())
        assert m.pressure_solver.impl.fast == (3 if pipeline == "xtri" else 1)
        sl = slice(r * g.Nx, (r + 1) * g.Nx)
        ocn.set(m, **{k: v[sl] for k, v in init.items()})
        if library:
            assert m.dist_correct_on_load
            drv = ocn.RK3Driver(m)
            for _ in range(2):
                drv.time_step(dt)
            drv.flush()
            del drv
        else:
            for _ in range(2):
                ocn.time_step(m, dt)
            ocn.flush_tendencies(m)
        ocn.sync_device()
        out = [f.interior() for f in m.velocities] + [m.pNHS.interior()]
        if library:
            fabric.barrier()
        return out

    outs = _run_ranks_local(ocn, R, rank_main) if library else _run_ranks(R, rank_main)
    _release()
    scale = max(np.abs(a).max() for a in ref[:3])
    pscale = max(1.0, np.abs(ref[3]).max())
    for r, fields in enumerate(outs):
        sl = slice(r * 64, (r + 1) * 64)
        for a, b, name in zip(fields, ref, ("u", "v", "w", "p")):
            tol = 1e-10 * pscale if name == "p" else 1e-11 * scale
            err = np.abs(a - b[sl]).max()
            assert err <= tol, f"rank {r} field {name}: {err} > {tol}"


def test_config4_ocean_mixing_512x512x256_on_8_ranks_matches_single_rank(ocn):
    """configs[3] as written (WENO5, AnisotropicMinimumDissipation, T, S, SeawaterBuoyancy, FPlane, flux / gradient conditions, stretched
    Bounded z, distributed Fourier-tridiagonal solver): one RK3 step on 8 ranks (64 x 512 x 256 each) in the default fast math against
    the single-rank model.  The eddy diffusivities amplify the solvers' rounding differences: 1e-9 relative (as the small-size test)."""
    import bench
    C4 = bench.C4
    N = (512, 512, 256)
    zf = bench.config4_faces(N[2])
    ext = dict(x=(0, 64), y=(0, 64), z=zf, topology=(P, P, "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(4)
    zc = 0.5 * (zf[1:] + zf[:-1])
    init = {"u": 1e-2 * rng.uniform(-1, 1, N), "v": 1e-2 * rng.uniform(-1, 1, N),
            "T": 20 + C4["dTdz"] * zc[None, None, :] + 1e-6 * rng.uniform(-1, 1, N), "S": np.full(N, 35.0)}
    dt = 0.1 * float(np.diff(zf).min()) / 1e-2

    def build(grid):
        bcs = {"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(C4["taux"])),
               "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(C4["JT"]), bottom=ocn.GradientBoundaryCondition(C4["dTdz"])),
               "S": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(0.0, coeff=-C4["evap"]))}
        return ocn.NonhydrostaticModel(grid, advection=ocn.WENO(), tracers=("T", "S"), coriolis=ocn.FPlane(f=C4["f"]),
                                       closure=ocn.AnisotropicMinimumDissipation(),
                                       buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(C4["alpha"], C4["beta"])),
                                       boundary_conditions=bcs)

    ocn.set_math_mode(ocn.MATH_FAST)
    try:
        sm = build(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext))
        ocn.set(sm, **init)
        ocn.time_step(sm, dt)
        ocn.flush_tendencies(sm)
        ocn.sync_device()
        ref = [f.interior() for f in sm.prognostic_fields()]
        del sm
        _release()

        def rank_main(r, fabric):
            arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
            m = build(ocn.RectilinearGrid(arch, size=N, **ext))
            nx = m.grid.Nx
            assert (nx, m.grid.Ny, m.grid.Nz) == (64, 512, 256) and m.pressure_solver.impl.fast == 2
            ocn.set(m, **{k: v[r * nx:(r + 1) * nx] for k, v in init.items()})
            ocn.time_step(m, dt)
            ocn.flush_tendencies(m)
            ocn.sync_device()
            return [f.interior() for f in m.prognostic_fields()]

        outs = _run_ranks(R, rank_main)
    finally:
        ocn.set_math_mode(ocn.MATH_STRICT)
    _release()
    vscale = max(np.abs(a).max() for a in ref[:3])
    for r, fields in enumerate(outs):
        sl = slice(r * 64, (r + 1) * 64)
        for a, b, name in zip(fields, ref, ("u", "v", "w", "T", "S")):
            scale = vscale if name in "uvw" else np.abs(b).max()
            err = np.abs(a - b[sl]).max()
            assert np.isfinite(a).all() and err <= 1e-9 * scale, f"rank {r} field {name}: {err} vs {1e-9 * scale}"


def test_config5_hydrostatic_1024x1024x128_on_8_ranks_is_bit_identical(ocn):
    """configs[4]: 1024 x 1024 x 128 HydrostaticFreeSurfaceModel (VectorInvariant momentum, WENO tracers, SplitExplicitFreeSurface with
    30 substeps, T / S, linear SeawaterBuoyancy, FPlane, ScalarDiffusivity): two QAB2 steps on 8 ranks -- 128-wide slabs, 30-wide
    barotropic halos exchanged once per step -- equal the single-rank model BIT FOR BIT in strict math (u, v, w, T, S, eta, U, V)."""
    import bench
    from test_gpu_distributed import _hydro_state
    C5 = bench.C5
    Nx, Nz = 1024, 128
    N = (Nx, Nx, Nz)
    H, L = C5["H"], C5["L"]
    ext = dict(x=(0, L), y=(0, L), z=(-H, 0.0), topology=(P, P, "Bounded"), halo=(3, 3, 3))
    rng = np.random.default_rng(5)
    zc = -H + (np.arange(Nz) + 0.5) * H / Nz
    init = dict(u=C5["amp"] * rng.uniform(-1, 1, N), v=C5["amp"] * rng.uniform(-1, 1, N),
                T=20 + 0.01 * zc[None, None, :] + np.zeros((Nx, Nx, 1)), S=np.full(N, 35.0))
    dt = 2.0 * (L / Nx) / np.sqrt(9.80665 * H)

    def build(grid):
        return ocn.HydrostaticFreeSurfaceModel(grid, momentum_advection=ocn.VectorInvariant(), tracer_advection=ocn.WENO(), tracers=("T", "S"),
                                               free_surface=ocn.SplitExplicitFreeSurface(substeps=30), coriolis=ocn.FPlane(f=C5["f"]),
                                               closure=ocn.ScalarDiffusivity(ν=C5["nu"], κ=C5["kappa"]),
                                               buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(C5["alpha"], C5["beta"])))

    ocn.set_math_mode(ocn.MATH_STRICT)
    sm = build(ocn.RectilinearGrid(ocn.GPU(), size=N, **ext))
    sm.set(**init)
    for _ in range(2):
        sm.time_step(dt)
    ocn.sync_device()
    ref = _hydro_state(sm)
    assert np.abs(ref["U"]).max() > 0 and np.abs(ref["w"]).max() > 0
    del sm
    _release()

    def rank_main(r, fabric):
        arch = ocn.Distributed(ocn.GPU(), partition=ocn.Partition(R), fabric=fabric)
        g = ocn.RectilinearGrid(arch, size=N, **ext)
        assert (g.Nx, g.Ny, g.Nz) == (128, 1024, 128)
        m = build(g)
        sl = slice(r * g.Nx, (r + 1) * g.Nx)
        m.set(**{k: v[sl] for k, v in init.items()})
        for _ in range(2):
            m.time_step(dt)
        ocn.sync_device()
        return _hydro_state(m)

    outs = _run_ranks(R, rank_main)
    _release()
    for r, got in enumerate(outs):
        sl = slice(r * 128, (r + 1) * 128)
        for name, a in got.items():
            np.testing.assert_array_equal(a, ref[name][sl], err_msg=f"rank {r} field {name}")
