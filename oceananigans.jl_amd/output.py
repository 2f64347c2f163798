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


def write_checkpoint(model, filepath):
    """write_output!(::Checkpointer, model): prognostic fields, tendencies and clock."""
    from .models import flush_tendencies
    flush_tendencies(model)  # Gⁿ must hold the tendencies of the current state, as after the reference's time_step!
    out = {}
    ts = model.timestepper
    for name, f, Gn, Gm in zip(_names(model), model.prognostic_fields(), ts._Gn, ts._Gm):
        out[f"{ADDR}/{name}/data"] = f.parent()
        out[f"{ADDR}/timestepper/Gⁿ/{name}/data"] = Gn.parent()
        out[f"{ADDR}/timestepper/G⁻/{name}/data"] = Gm.parent()
    c = model.clock
    out[f"{ADDR}/clock/time"] = np.float64(c.time)
    out[f"{ADDR}/clock/iteration"] = np.int64(c.iteration)
    out[f"{ADDR}/clock/last_Δt"] = np.float64(c.last_dt)
    g = model.grid
    out[f"{ADDR}/grid/size_halo"] = np.array([g.Nx, g.Ny, g.Nz, g.Hx, g.Hy, g.Hz], dtype=np.int64)
    np.savez(filepath, **out)
    return filepath


def set_from_checkpoint(model, filepath):
    """set!(model, filepath): restores parents (halos included), Gⁿ/G⁻ (both steppers; RK3 ignores them in the reference
    because it is self-starting, restoring them is harmless) and the clock, then update_state!."""
    from .models import flush_tendencies, update_state
    flush_tendencies(model)
    with np.load(filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz") as z:
        g = model.grid
        sh = z[f"{ADDR}/grid/size_halo"]
        if tuple(sh) != (g.Nx, g.Ny, g.Nz, g.Hx, g.Hy, g.Hz):
            raise ValueError(f"The grid associated with {filepath} and model.grid are not the same!")
        ts = model.timestepper

        def put(field, key):
            if key in z.files:
                field.data.copy_(torch.from_numpy(np.ascontiguousarray(z[key].T)))
                return True
            return False

        for name, f, Gn, Gm in zip(_names(model), model.prognostic_fields(), ts._Gn, ts._Gm):
            if not put(f, f"{ADDR}/{name}/data"):
                raise KeyError(f"Field {name} does not exist in checkpoint and could not be restored.")
            put(Gn, f"{ADDR}/timestepper/Gⁿ/{name}/data")
            put(Gm, f"{ADDR}/timestepper/G⁻/{name}/data")
        model.clock.time = float(z[f"{ADDR}/clock/time"])
        model.clock.iteration = int(z[f"{ADDR}/clock/iteration"])
        model.clock.last_dt = float(z[f"{ADDR}/clock/last_Δt"])
    update_state(model, compute_tendencies=False)
    return model
