"""The C-ABI library loads without a GPU and exports every symbol include/ocn_hip.h declares;
argument validation (which runs before any HIP call) reports errors through the status/ocn_last_error contract."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import oceananigans_jl_amd as ocn
    return ocn


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ocn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(ocn_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.lib()
    declared = _declared_functions()
    assert len(declared) >= 40
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"declared in ocn_hip.h but not exported: {missing}"
    # and the Python binding table covers the header
    unbound = [n for n in declared if n not in pkg._lib.EXPORTED_SYMBOLS]
    assert not unbound, f"declared but not bound in _lib.py: {unbound}"


def test_struct_layout_matches_header(pkg):
    # 10 int32 + 6 double + 2 pointers, natural alignment
    assert C.sizeof(pkg._lib.CGrid) == 10 * 4 + 6 * 8 + 2 * 8
    assert pkg._lib.CGrid.dx.offset == 40 and pkg._lib.CGrid.dzc.offset == 88


def test_version_and_math_mode(pkg):
    lib = pkg._lib.lib()
    assert b"gfx950" in lib.ocn_version()
    pkg.set_math_mode(pkg.MATH_FAST)
    assert lib.ocn_get_math_mode() == pkg.MATH_FAST
    pkg.set_math_mode(pkg.MATH_STRICT)
    with pytest.raises(pkg.OcnError, match="unknown math mode"):
        pkg._lib.call("ocn_set_math_mode", 7)


def _grid(pkg, **kw):
    base = dict(Nx=8, Ny=8, Nz=8, Hx=3, Hy=3, Hz=3, tx=0, ty=0, tz=0, math=0, dx=1.0, dy=1.0, dz=1.0, Lx=8.0, Ly=8.0, Lz=8.0)
    base.update(kw)
    return pkg._lib.CGrid(**base)


def test_argument_validation_needs_no_gpu(pkg):
    call, pa, ia = pkg._lib.call, pkg._lib.ptr_array, pkg._lib.i32_array
    fake = pa([0x1000])
    with pytest.raises(pkg.OcnError, match="unsupported topology"):   # an entry point without a direction-generic path: Periodic x, y only
        call("ocn_compute_w_from_continuity", C.byref(_grid(pkg, tx=1)), 1, 1, 1, None)
    with pytest.raises(pkg.OcnError, match="only x is ever partitioned"):
        call("ocn_fill_halo_regions", C.byref(_grid(pkg, ty=3)), fake, ia([0]), 1, 1, None)
    # (grids with walls take the fused stage boundary since round 4 -- but neither the correction on load nor a range)
    with pytest.raises(pkg.OcnError, match="pressure correction on load needs Periodic x and y"):
        call("ocn_compute_momentum_tendencies_rk3", C.byref(_grid(pkg, ty=1)), *range(16, 16 * 13, 16), 0.1, 0.5, 0.0, 0, 0x2000, 0.0, None, None)
    with pytest.raises(pkg.OcnError, match="ranges need Periodic x and y"):
        call("ocn_compute_momentum_tendencies_rk3", C.byref(_grid(pkg, ty=1)), *range(16, 16 * 13, 16), 0.1, 0.5, 0.0, 0, None, 0.0,
             ia([1, 2, 1, 2, 1, 2]), None)
    with pytest.raises(pkg.OcnError, match="Flat dimension"):
        call("ocn_fill_halo_regions", C.byref(_grid(pkg, tz=2)), fake, ia([0]), 1, 1, None)
    # a slab (FullyConnected x) of a channel is a grid the direction-generic entry points take: the walls in y pass the grid check (the
    # call then stops at its next check), the correction on load and a Flat y on a slab do not
    with pytest.raises(pkg.OcnError, match="null field pointer"):
        call("ocn_compute_momentum_tendencies", C.byref(_grid(pkg, tx=3, ty=1, tz=1)), None, 1, 1, 1, 1, 1, None, None)
    with pytest.raises(pkg.OcnError, match="pressure correction on load needs Periodic x and y"):
        call("ocn_compute_momentum_tendencies_rk3", C.byref(_grid(pkg, tx=3, ty=1, tz=1)), *range(16, 16 * 13, 16), 0.1, 0.5, 0.0, 0, 0x2000, 0.0, None, None)
    with pytest.raises(pkg.OcnError, match="Periodic or Bounded y"):
        call("ocn_compute_momentum_tendencies", C.byref(_grid(pkg, tx=3, ty=2, Ny=1, Hy=0)), 1, 1, 1, 1, 1, 1, None, None)
    # the first / last slab of a Bounded partitioned x (RightConnected = 4, LeftConnected = 5): x only; the distributed solver checks that
    # the slab's topology is the one its rank has in the global grid
    for tx in (4, 5):
        with pytest.raises(pkg.OcnError, match="null field pointer"):
            call("ocn_compute_momentum_tendencies", C.byref(_grid(pkg, tx=tx, ty=1, tz=1)), None, 1, 1, 1, 1, 1, None, None)
    with pytest.raises(pkg.OcnError, match="unknown topology code"):
        call("ocn_fill_halo_regions", C.byref(_grid(pkg, ty=4)), fake, ia([0]), 1, 1, None)
    h = C.c_void_p()
    with pytest.raises(pkg.OcnError, match="must hold a slab of x topology 4"):
        call("ocn_dist_poisson_create_global", C.byref(h), C.byref(_grid(pkg, tx=3, ty=1, tz=1)), 0, 4, 32.0, 1)
    with pytest.raises(pkg.OcnError, match="a Bounded x needs Bounded y and z"):
        call("ocn_dist_poisson_create_global", C.byref(h), C.byref(_grid(pkg, tx=5, ty=0, tz=1)), 3, 4, 32.0, 1)
    with pytest.raises(pkg.OcnError, match="number of fields"):
        call("ocn_fill_halo_regions", C.byref(_grid(pkg)), fake, ia([0]), 0, 1, None)
    with pytest.raises(pkg.OcnError, match="location mask"):
        call("ocn_fill_halo_regions", C.byref(_grid(pkg)), fake, ia([3]), 1, 1, None)
    with pytest.raises(pkg.OcnError, match="halo >= 3"):
        call("ocn_compute_momentum_tendencies", C.byref(_grid(pkg, Hx=2)), 1, 1, 1, 1, 1, 1, None, None)
    with pytest.raises(pkg.OcnError, match="null field pointer"):
        call("ocn_compute_momentum_tendencies", C.byref(_grid(pkg)), None, 1, 1, 1, 1, 1, None, None)
    with pytest.raises(pkg.OcnError, match="larger than size"):
        call("ocn_compute_momentum_tendencies", C.byref(_grid(pkg, Nx=2)), 1, 1, 1, 1, 1, 1, None, None)
    with pytest.raises(pkg.OcnError, match="divisible"):
        call("ocn_transpose_pack_y_to_x", 4, 10, 4, 3, 0x1000, 0x1000, None)
    with pytest.raises(pkg.OcnError, match="is not Periodic"):
        call("ocn_fill_halo_periodic", C.byref(_grid(pkg, tz=1)), fake, ia([0]), 1, 2, None)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libocn_hip.so")
    with pytest.raises(pkg.OcnError, match="no CPU fallback"):
        pkg._lib.lib()


def test_gpu_architecture_refuses_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.GPU()
    with pytest.raises(NotImplementedError):
        pkg.CPU()
