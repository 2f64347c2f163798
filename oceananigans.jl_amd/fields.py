"""Fields: mirrors src/Fields/field.jl:19-33 (grid + OffsetArray parent + boundary conditions),
set!, interior, fill_halo_regions! (single and tupled: src/Fields/field_tuples.jl:56-101) with the
default boundary conditions of src/BoundaryConditions/field_boundary_conditions.jl:15-33.
"""
import numpy as np
import torch

from . import _lib
from .architectures import on_architecture, stream_ptr, zeros
from .grids import Bounded, Flat

LOC = {"u": _lib.LOC_FCC, "v": _lib.LOC_CFC, "w": _lib.LOC_CCF, "c": _lib.LOC_CCC}


class Field:
    """A field at `loc` (bitmask: bit0 x-Face, bit1 y-Face, bit2 z-Face) on `grid`.

    `.data` is the parent array (halos included) as a tensor of shape (sz, sy, sx): the same bytes as the
    reference's column-major (sx, sy, sz) OffsetArray parent (src/Grids/new_data.jl:36-70), zero-initialised."""

    def __init__(self, loc, grid, data=None, boundary_conditions=None):
        self.boundary_conditions = boundary_conditions  # FieldBoundaryConditions or None (= defaults)
        if loc not in (_lib.LOC_CCC, _lib.LOC_FCC, _lib.LOC_CFC, _lib.LOC_CCF):
            raise ValueError(f"unsupported field location mask {loc}")
        self.loc = loc
        self.grid = grid
        shape = grid.parent_shape(loc)
        if data is None:
            data = zeros(grid.architecture, shape)
        elif tuple(data.shape) != tuple(reversed(shape)):
            raise ValueError(f"data shape {tuple(data.shape)} != parent shape {tuple(reversed(shape))}")
        self.data = data

    @property
    def ptr(self):
        return self.data.data_ptr()

    def interior_view(self):
        """interior(field): tensor view indexed [k, j, i] over 1:N (plus the extra boundary face where Face & Bounded)."""
        g = self.grid
        sz, sy, sx = self.data.shape
        return self.data[g.Hz:sz - g.Hz, g.Hy:sy - g.Hy, g.Hx:sx - g.Hx]

    def interior(self):
        """interior as a numpy array indexed [i, j, k] (host copy)."""
        return self.interior_view().cpu().numpy().T

    def parent(self):
        """parent(field) as a numpy array indexed [i, j, k] with halos (host copy)."""
        return self.data.cpu().numpy().T

    def set(self, value):
        """set!(field, value): an array indexed [i, j, k] over the interior, a scalar, or a function f(x, y, z) evaluated at
        the field's nodes (called once with broadcastable coordinate arrays; falls back to element-wise calls).
        Halos are left untouched (src/Fields/set!.jl)."""
        iv = self.interior_view()
        if np.isscalar(value):
            iv.fill_(float(value))
        elif callable(value):  # set!(field, f::Function): f(x, y, z) at the field's nodes (src/Fields/set!.jl:43-76)
            x, y, z = self.grid.nodes(self.loc)
            try:
                a = np.asarray(value(x, y, z), dtype=np.float64)
            except (TypeError, ValueError):
                a = np.vectorize(value, otypes=[np.float64])(x, y, z)
            a = np.broadcast_to(a, tuple(reversed(iv.shape)))
            iv.copy_(on_architecture(self.grid.architecture, np.array(a.T, dtype=np.float64, order="C")))  # (a copy: broadcast views are read-only)
        else:
            if isinstance(value, torch.Tensor):
                t = value.to(self.data.device, dtype=torch.float64)
                if tuple(t.shape) != tuple(iv.shape):
                    raise ValueError(f"tensor shape {tuple(t.shape)} != interior shape [k,j,i] {tuple(iv.shape)}")
            else:
                a = np.asarray(value, dtype=np.float64)
                if a.shape != tuple(reversed(iv.shape)):
                    raise ValueError(f"array shape {a.shape} != interior shape {tuple(reversed(iv.shape))}")
                t = on_architecture(self.grid.architecture, np.ascontiguousarray(a.T))
            iv.copy_(t)
        return self


def XFaceField(grid, boundary_conditions=None):
    return Field(_lib.LOC_FCC, grid, boundary_conditions=boundary_conditions)


def YFaceField(grid, boundary_conditions=None):
    return Field(_lib.LOC_CFC, grid, boundary_conditions=boundary_conditions)


def ZFaceField(grid, boundary_conditions=None):
    return Field(_lib.LOC_CCF, grid, boundary_conditions=boundary_conditions)


def CenterField(grid, boundary_conditions=None):
    return Field(_lib.LOC_CCC, grid, boundary_conditions=boundary_conditions)


def local_fill_halo_regions(grid, fields, fill_boundary_normal_velocities=True):
    """The rank-local part of fill_halo_regions! (one launch for the whole tuple): Periodic copies, no-flux / impenetrable
    walls, and the fields' own bottom / top Value / Gradient conditions."""
    bcs = [getattr(f, "boundary_conditions", None) for f in fields]
    if any(b is not None and not b.is_default() for b in bcs):
        import ctypes as C
        arr = (C.POINTER(_lib.CFieldBcs) * len(fields))(*[
            (C.pointer(b.c_struct(grid)) if b is not None and not b.is_default() else C.POINTER(_lib.CFieldBcs)()) for b in bcs])
        _lib.call("ocn_fill_halo_regions_bcs", grid.cref, _lib.ptr_array([f.ptr for f in fields]),
                  _lib.i32_array([f.loc for f in fields]), arr, len(fields), int(bool(fill_boundary_normal_velocities)),
                  stream_ptr())
        return
    _lib.call("ocn_fill_halo_regions", grid.cref, _lib.ptr_array([f.ptr for f in fields]),
              _lib.i32_array([f.loc for f in fields]), len(fields), int(bool(fill_boundary_normal_velocities)), stream_ptr())


def fill_halo_regions(fields, fill_boundary_normal_velocities=True, **_):
    """fill_halo_regions!(field or tuple of fields): one launch for the whole tuple."""
    if isinstance(fields, Field):
        fields = (fields,)
    fields = tuple(fields)
    if not fields:
        return
    grid = fields[0].grid
    hook = getattr(grid.architecture, "fill_halo_regions", None)
    if hook is not None:  # Distributed: local fills + x-halo exchange (distributed.py)
        return hook(fields, fill_boundary_normal_velocities)
    local_fill_halo_regions(grid, fields, fill_boundary_normal_velocities)
