"""HIP side of the first HydrostaticFreeSurfaceModel slice against the CPU oracle (oracle/hydrostatic.py),
through the C ABI: ocn_compute_w_from_continuity, ocn_add_barotropic_pressure_gradient, ocn_add_momentum_terms,
ocn_explicit_free_surface_ab2_step and the model built from them.  Strict math: bit for bit."""
import numpy as np
import pytest

from helpers import from_dev, make_pair, stretched_faces, to_dev

pytestmark = pytest.mark.gpu


def _pair(O, ocn, size, stretched):
    z = stretched_faces(size[2], 40.0) if stretched else (-40.0, 0.0)
    return make_pair(O, ocn, size, "PPB", x=(0, 2.0e3), y=(0, 1.5e3), z=z)


@pytest.mark.parametrize("stretched", [False, True])
def test_w_from_continuity_bitwise(oracle, ocn, stretched):
    from oracle import hydrostatic as Hy
    O = oracle
    og, pg = _pair(O, ocn, (16, 12, 7), stretched)
    rng = np.random.default_rng(11)
    u, v, w = og.zeros(1), og.zeros(2), og.zeros(4)
    u[...] = rng.uniform(-1, 1, u.shape)
    v[...] = rng.uniform(-1, 1, v.shape)
    Hy.compute_w_from_continuity(og, u, v, w)
    du, dv, dw = to_dev(ocn, pg, 1, u), to_dev(ocn, pg, 2, v), ocn.Field(4, pg)
    ocn._lib.call("ocn_compute_w_from_continuity", pg.cref, du.ptr, dv.ptr, dw.ptr, 0)
    ocn.sync_device()
    got = from_dev(dw)
    # every column with east / north neighbours in the parent
    np.testing.assert_array_equal(got[:-1, :-1, og.Hz:og.Hz + og.Nz + 1], w[:-1, :-1, og.Hz:og.Hz + og.Nz + 1])


@pytest.mark.parametrize("advection", ["Centered2", "WENO5", "VectorInvariant"])
@pytest.mark.parametrize("physics", [False, True])
def test_hydrostatic_model_steps_match_oracle(oracle, ocn, advection, physics):
    """3 QAB2 steps (the first one Euler) of the explicit-free-surface model: u, v, w, η and the tracers equal the oracle's bit
    for bit in strict math."""
    from oracle import hydrostatic as Hy
    O = oracle
    size = (16, 12, 7)
    og, pg = _pair(O, ocn, size, stretched=True)
    rng = np.random.default_rng(12)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]))
    kw_o, kw_p, tracers = {}, {}, ()
    if physics:
        tracers = ("T", "S")
        init["T"] = 20 + 1e-2 * rng.uniform(-1, 1, size)
        init["S"] = 35 + 1e-2 * rng.uniform(-1, 1, size)
        kw_o = dict(coriolis_f=1e-4, closure=(1e-2, 2e-3), buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4),
                    boundary_conditions={"u": {"top": O.FluxBoundaryCondition(-1e-4)}, "T": {"top": O.FluxBoundaryCondition(5e-5)}})
        kw_p = dict(coriolis=ocn.FPlane(f=1e-4), closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                    buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                    boundary_conditions={"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1e-4)),
                                         "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5))})
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=tracers, momentum_advection=advection, **kw_o)
    om.set(**init)
    ocn.set_math_mode(ocn.MATH_STRICT)
    scheme = {"Centered2": ocn.Centered, "WENO5": ocn.WENO, "VectorInvariant": ocn.VectorInvariant}[advection]()
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=scheme, tracers=tracers, free_surface=ocn.ExplicitFreeSurface(), **kw_p)
    pm.set(**init)
    for dt in (2.0, 2.0, 2.0):
        om.time_step(dt)
        pm.time_step(dt)
    ocn.sync_device()
    for name, a, d in zip(("u", "v", "w"), (om.u, om.v, om.w), pm.velocities):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a), err_msg=name)
    eta = pm.eta_interior().cpu().numpy().T
    np.testing.assert_array_equal(eta, om.eta[og.Hx:og.Hx + og.Nx, og.Hy:og.Hy + og.Ny])
    for a, d in zip(om.tracers, pm.tracers):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a))
    assert np.abs(eta).max() > 0 and np.isfinite(eta).all()


def test_split_explicit_free_surface_model_steps_match_oracle(oracle, ocn):
    """SplitExplicitFreeSurface(substeps = 12), ForwardBackwardScheme: slow forcing, substepping with the averaging weights, barotropic
    corrector -- 3 QAB2 steps of the default configuration (VectorInvariant momentum, Centered tracers) with Coriolis,
    diffusivity, buoyancy and flux conditions equal the oracle's bit for bit, barotropic transports included."""
    from oracle import hydrostatic as Hy
    O = oracle
    size = (16, 12, 7)
    og, pg = _pair(O, ocn, size, stretched=True)
    rng = np.random.default_rng(21)
    init = dict(u=1e-2 * rng.uniform(-1, 1, size), v=1e-2 * rng.uniform(-1, 1, size), eta=1e-2 * rng.uniform(-1, 1, size[:2]),
                T=20 + 1e-2 * rng.uniform(-1, 1, size), S=35 + 1e-2 * rng.uniform(-1, 1, size))
    om = Hy.HydrostaticFreeSurfaceModel(og, tracers=("T", "S"), momentum_advection="VectorInvariant", coriolis_f=1e-4, closure=(1e-2, 2e-3),
                                        buoyancy=("SeawaterBuoyancy", 9.80665, 2e-4, 8e-4), split_explicit_substeps=12,
                                        boundary_conditions={"u": {"top": O.FluxBoundaryCondition(-1e-4)}, "T": {"top": O.FluxBoundaryCondition(5e-5)}})
    om.set(**init)
    ocn.set_math_mode(ocn.MATH_STRICT)
    pm = ocn.HydrostaticFreeSurfaceModel(pg, momentum_advection=ocn.VectorInvariant(), tracers=("T", "S"),
                                         free_surface=ocn.SplitExplicitFreeSurface(substeps=12), coriolis=ocn.FPlane(f=1e-4),
                                         closure=ocn.ScalarDiffusivity(ν=1e-2, κ=2e-3),
                                         buoyancy=ocn.SeawaterBuoyancy(equation_of_state=ocn.LinearEquationOfState(2e-4, 8e-4)),
                                         boundary_conditions={"u": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(-1e-4)),
                                                              "T": ocn.FieldBoundaryConditions(top=ocn.FluxBoundaryCondition(5e-5))})
    pm.set(**init)
    for dt in (20.0, 20.0, 20.0):
        om.time_step(dt)
        pm.time_step(dt)
    ocn.sync_device()
    for name, a, d in zip(("u", "v", "w"), (om.u, om.v, om.w), pm.velocities):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a), err_msg=name)
    ii, jj = slice(og.Hy, og.Hy + og.Ny), slice(og.Hx, og.Hx + og.Nx)
    np.testing.assert_array_equal(pm.eta[ii, jj].cpu().numpy().T, om.eta[og.Hx:og.Hx + og.Nx, og.Hy:og.Hy + og.Ny])
    np.testing.assert_array_equal(pm.U[ii, jj].cpu().numpy().T, om.U)
    np.testing.assert_array_equal(pm.V[ii, jj].cpu().numpy().T, om.V)
    for a, d in zip(om.tracers, pm.tracers):
        np.testing.assert_array_equal(og.interior(from_dev(d)), og.interior(a))
    assert np.abs(om.U).max() > 0
