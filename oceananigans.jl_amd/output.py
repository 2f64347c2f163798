"""NaN check and checkpoint state I/O.

  hasnan(field | model)                 src/Models/nan_checker.jl:33-34 (device reduction, ocn_hasnan)
  NaNChecker(fields, erroring)          src/Models/nan_checker.jl:3-52
  write_checkpoint(model, filepath)     src/OutputWriters/checkpointer.jl:177-203: prognostic fields, Gⁿ, G⁻ (parent arrays,
                                        halos included) and the clock
  set_from_checkpoint(model, filepath)  set!(model, filepath) (:227-288)

The reference's container is JLD2 (HDF5); no HDF5 library exists in this image, so the same addresses
("NonhydrostaticModel/u/data", "NonhydrostaticModel/timestepper/Gⁿ/u/data", "NonhydrostaticModel/clock/...") are the keys
of a numpy .npz archive instead.  Arrays are stored as the reference stores them: the OffsetArray parent, indexed [i, j, k].
"""
import numpy as np
import torch

from . import _lib
from .architectures import stream_ptr
from .fields import Field

ADDR = "NonhydrostaticModel"  # checkpointer_address(::NonhydrostaticModel)


def hasnan(x):
    """hasnan(field) = any(isnan, parent(field)); hasnan(model) = hasnan(first(fields(model)))."""
    f = x if isinstance(x, Field) else x.prognostic_fields()[0]
    flag = torch.zeros(1, dtype=torch.int32, device=f.data.device)
    _lib.call("ocn_hasnan", f.ptr, f.data.numel(), flag.data_ptr(), stream_ptr())
    return bool(flag.item())


class NaNChecker:
    """NaNChecker(; fields, erroring=false): call with the model; returns the name of the first field holding a NaN (or None)."""

    def __init__(self, fields, erroring=False):
        self.fields, self.erroring = dict(fields), erroring

    def __call__(self, model):
        flags = torch.zeros(len(self.fields), dtype=torch.int32, device=next(iter(self.fields.values())).data.device)
        for n, f in enumerate(self.fields.values()):  # all scans queued, one host read
            _lib.call("ocn_hasnan", f.ptr, f.data.numel(), flags[n:].data_ptr(), stream_ptr())
        bad = flags.cpu().numpy()
        for name, b in zip(self.fields, bad):
            if b:
                msg = f"time = {model.clock.time}, iteration = {model.clock.iteration}: NaN found in field {name}."
                if self.erroring:
                    raise RuntimeError(msg + " Aborting simulation.")
                return name
        return None


def cell_advection_timescale(model):
    """cell_advection_timescale(model) (src/Advection/cell_advection_timescale.jl:13-35); all-reduced over the ranks of a
    Distributed architecture (time_step_wizard.jl:112)."""
    g = model.grid
    out = torch.zeros(1, dtype=torch.float64, device=model.u.data.device)
    _lib.call("ocn_cell_advection_timescale", g.cref, model.u.ptr, model.v.ptr, model.w.ptr, out.data_ptr(), stream_ptr())
    reduce = getattr(getattr(g.architecture, "fabric", None), "allreduce_max", None)
    if reduce is not None and getattr(g.architecture, "partition", None) is not None and g.architecture.partition.x > 1:
        out = -reduce(-out)
    return float(out.item())


def cell_diffusion_timescale(model):
    """cell_diffusion_timescale(model) (src/TurbulenceClosures/turbulence_closure_diagnostics.jl:20-74): min(Δ² / max ν, Δ² / max κ) with
    Δ = the smallest cell spacing (1 along Flat directions); ScalarDiffusivity: the numbers ν, κ; AnisotropicMinimumDissipation: the maxima
    of the eddy-diffusivity fields' parents.  Inf without a closure."""
    from .physics import AnisotropicMinimumDissipation
    g, cl = model.grid, model.closure
    if cl is None:
        return float("inf")
    delta = min(g.spacing_extrema(d)[0] if g.topology[d] != "Flat" else 1.0 for d in range(3))
    div = lambda a, b: a / b if b != 0 else float("inf")
    if isinstance(cl, AnisotropicMinimumDissipation):
        d = model.diffusivity_fields
        max_nu = float(d["nu_e"].data.max())
        max_kappa = max((float(k.data.max()) for k in d["kappa_e"]), default=float("inf"))
        return min(div(delta ** 2, max_nu), div(delta ** 2, max_kappa))
    max_kappa = max((cl.kappa_of(n) for n in model.tracer_names), default=0.0)
    return min(div(delta ** 2, cl.nu), div(delta ** 2, max_kappa))


class DiffusiveCFL:
    """DiffusiveCFL(Δt)(model) = Δt / cell_diffusion_timescale(model) (src/Diagnostics/cfl.jl)"""

    def __init__(self, dt):
        self.dt = dt

    def __call__(self, model):
        return self.dt / cell_diffusion_timescale(model)


class AdvectiveCFL:
    """AdvectiveCFL(Δt)(model) = Δt / cell_advection_timescale(model) (src/Diagnostics/cfl.jl)"""

    def __init__(self, dt):
        self.dt = dt

    def __call__(self, model):
        return self.dt / cell_advection_timescale(model)


class TimeStepWizard:
    """TimeStepWizard(; cfl=0.2, max_change=1.1, min_change=0.5, max_Δt=Inf, min_Δt=0) (src/Simulations/time_step_wizard.jl:3-115)
    for the advective CFL: new_Δt = clamp(min(max_change Δt, max(min_change Δt, cfl τ)), min_Δt, max_Δt)."""

    def __init__(self, cfl=0.2, max_change=1.1, min_change=0.5, max_dt=float("inf"), min_dt=0.0):
        self.cfl, self.max_change, self.min_change, self.max_dt, self.min_dt = cfl, max_change, min_change, max_dt, min_dt

    def __call__(self, model, old_dt):
        new_dt = self.cfl * cell_advection_timescale(model)
        new_dt = min(self.max_change * old_dt, new_dt)
        new_dt = max(self.min_change * old_dt, new_dt)
        return min(max(new_dt, self.min_dt), self.max_dt)


def _names(model):
    return ("u", "v", "w") + tuple(model.tracer_names)


def _rank_path(model, filepath):
    """Distributed architectures write one file per rank: `name_rank$r` (output_writer_utils.jl:223-224)."""
    arch = model.grid.architecture
    filepath = str(filepath)
    if hasattr(arch, "partition"):
        root, ext = (filepath[:-4], ".npz") if filepath.endswith(".npz") else (filepath, "")
        if not root.endswith(f"_rank{arch.local_rank}"):
            filepath = f"{root}_rank{arch.local_rank}{ext}"
    return filepath


def _grid_signature(g):
    """What set!(model, filepath) compares (checkpointer.jl:241-246 compares the whole grid): sizes, halos, topology, extents, z faces."""
    zf = g._dzc_host if getattr(g, "_dzc_host", None) is not None else np.array([g.dz])  # the cell spacings pin the z faces
    return {"size_halo": np.array([g.Nx, g.Ny, g.Nz, g.Hx, g.Hy, g.Hz], dtype=np.int64),
            "topology": np.array([str(t) for t in g.topology]),
            "extent": np.array([g.dx, g.dy, g.Lx, g.Ly, g.Lz], dtype=np.float64),
            "z_spacings": np.asarray(zf, dtype=np.float64)}


def _check_grid(z, g, filepath):
    sig = _grid_signature(g)
    for k, v in sig.items():
        key = f"{ADDR}/grid/{k}"
        if key not in z.files:
            if k == "size_halo":
                raise ValueError(f"{filepath} holds no grid")
            continue  # a checkpoint of round 1 (sizes and halos only)
        same = (v.shape == z[key].shape) and (np.array_equal(v, z[key]) if v.dtype.kind in "iUS" else np.array_equal(v, z[key]))
        if not same:
            raise ValueError(f"The grid associated with {filepath} and model.grid are not the same! ({k} differs)")


def _is_hydrostatic(model):
    return hasattr(model, "free_surface")


def write_checkpoint(model, filepath):
    """write_output!(::Checkpointer, model): the reference's checkpointed properties (checkpointer.jl:10-18, 177-201) -- prognostic
    fields (parents, halos included), timestepper Gⁿ / G⁻, clock -- plus the grid signature.  HydrostaticFreeSurfaceModel: also η, the
    barotropic velocities U, V and Gηⁿ / Gη⁻ (its prognostic_fields and timestepper tendencies, hydrostatic_free_surface_model.jl)."""
    filepath = _rank_path(model, filepath)
    out = {}
    if _is_hydrostatic(model):
        model.flush_tendencies()
        nh = model._nh
        ts, names, fields = nh.timestepper, ("u", "v") + tuple(model.tracer_names), [model.u, model.v] + list(model.tracers)
        idx = [0, 1] + [3 + n for n in range(len(model.tracers))]
        Gn, Gm = [ts._Gn[q] for q in idx], [ts._Gm[q] for q in idx]
        out[f"{ADDR}/η/data"] = model.eta.cpu().numpy().T
        out[f"{ADDR}/timestepper/Gⁿ/η/data"] = model._Geta.cpu().numpy().T
        out[f"{ADDR}/timestepper/G⁻/η/data"] = model._Geta_m.cpu().numpy().T
        if model.split:
            out[f"{ADDR}/U/data"] = model.U.cpu().numpy().T
            out[f"{ADDR}/V/data"] = model.V.cpu().numpy().T
            out[f"{ADDR}/free_surface/initialized"] = np.int64(model._initialized)
    else:
        from .models import flush_tendencies
        flush_tendencies(model)  # Gⁿ must hold the tendencies of the current state, as after the reference's time_step!
        ts, names, fields = model.timestepper, _names(model), model.prognostic_fields()
        Gn, Gm = ts._Gn, ts._Gm
    for name, f, gn, gm in zip(names, fields, Gn, Gm):
        out[f"{ADDR}/{name}/data"] = f.parent()
        out[f"{ADDR}/timestepper/Gⁿ/{name}/data"] = gn.parent()
        out[f"{ADDR}/timestepper/G⁻/{name}/data"] = gm.parent()
    c = model.clock
    out[f"{ADDR}/clock/time"] = np.float64(c.time)
    out[f"{ADDR}/clock/iteration"] = np.int64(c.iteration)
    out[f"{ADDR}/clock/last_Δt"] = np.float64(c.last_dt)
    for k, v in _grid_signature(model.grid).items():
        out[f"{ADDR}/grid/{k}"] = v
    np.savez(filepath, **out)
    return filepath


def set_from_checkpoint(model, filepath):
    """set!(model, filepath) (checkpointer.jl:220-288): restores parents (halos included), Gⁿ / G⁻ (both steppers; RK3 ignores them in
    the reference because it is self-starting, restoring them is harmless) and the clock, then update_state!."""
    filepath = _rank_path(model, filepath)
    hyd = _is_hydrostatic(model)
    if not hyd:
        from .models import flush_tendencies, update_state
        flush_tendencies(model)
    with np.load(filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz") as z:
        _check_grid(z, model.grid, filepath)

        def put_tensor(t, key):
            if key in z.files:
                t.copy_(torch.from_numpy(np.ascontiguousarray(z[key].T)))
                return True
            return False

        def put(field, key):
            return put_tensor(field.data, key)

        if hyd:
            nh = model._nh
            ts, names, fields = nh.timestepper, ("u", "v") + tuple(model.tracer_names), [model.u, model.v] + list(model.tracers)
            idx = [0, 1] + [3 + n for n in range(len(model.tracers))]
            Gns, Gms = [ts._Gn[q] for q in idx], [ts._Gm[q] for q in idx]
            if not put_tensor(model.eta, f"{ADDR}/η/data"):
                raise KeyError("Field η does not exist in checkpoint and could not be restored.")
            put_tensor(model._Geta, f"{ADDR}/timestepper/Gⁿ/η/data")
            put_tensor(model._Geta_m, f"{ADDR}/timestepper/G⁻/η/data")
            if model.split:
                put_tensor(model.U, f"{ADDR}/U/data")
                put_tensor(model.V, f"{ADDR}/V/data")
                if f"{ADDR}/free_surface/initialized" in z.files:
                    model._initialized = bool(int(z[f"{ADDR}/free_surface/initialized"]))
        else:
            ts, names, fields = model.timestepper, _names(model), model.prognostic_fields()
            Gns, Gms = ts._Gn, ts._Gm
        for name, f, Gn, Gm in zip(names, fields, Gns, Gms):
            if not put(f, f"{ADDR}/{name}/data"):
                raise KeyError(f"Field {name} does not exist in checkpoint and could not be restored.")
            put(Gn, f"{ADDR}/timestepper/Gⁿ/{name}/data")
            put(Gm, f"{ADDR}/timestepper/G⁻/{name}/data")
        model.clock.time = float(z[f"{ADDR}/clock/time"])
        model.clock.iteration = int(z[f"{ADDR}/clock/iteration"])
        model.clock.last_dt = float(z[f"{ADDR}/clock/last_Δt"])
    if hyd:
        model.update_state(compute_tendencies=False)
    else:
        update_state(model, compute_tendencies=False)
    return model
