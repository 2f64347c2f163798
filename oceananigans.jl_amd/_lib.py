"""ctypes binding of libocn_hip.so (include/ocn_hip.h).

The HIP library is the only compute path of this package: if it is missing or a call fails
we raise -- there is no CPU / PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCN_LIB_PATH selects another build of the same library (same-box A/B runs of kernel variants, tools/ab_bench.sh): never a fallback --
# the file must exist and export every symbol of include/ocn_hip.h, or loading raises as for the default path
LIB_PATH = os.environ.get("OCN_LIB_PATH") or os.path.join(_HERE, "lib", "libocn_hip.so")

OCN_PERIODIC, OCN_BOUNDED, OCN_FLAT, OCN_FULLY_CONNECTED, OCN_RIGHT_CONNECTED, OCN_LEFT_CONNECTED = 0, 1, 2, 3, 4, 5
LOC_CCC, LOC_FCC, LOC_CFC, LOC_CCF = 0, 1, 2, 4
MATH_STRICT, MATH_FAST = 0, 1
GRID_MATH_DEFAULT, GRID_MATH_STRICT, GRID_MATH_FAST = 0, 1, 2  # ocn_grid.math


class OcnError(RuntimeError):
    pass


class CGrid(C.Structure):
    """struct ocn_grid (include/ocn_hip.h)"""
    _fields_ = [("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32),
                ("Hx", C.c_int32), ("Hy", C.c_int32), ("Hz", C.c_int32),
                ("tx", C.c_int32), ("ty", C.c_int32), ("tz", C.c_int32), ("math", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("Lx", C.c_double), ("Ly", C.c_double), ("Lz", C.c_double),
                ("dzc", C.c_void_p), ("dzf", C.c_void_p)]


class CModelTerms(C.Structure):
    """struct ocn_model_terms"""
    _fields_ = [("advection", C.c_int32), ("coriolis", C.c_int32), ("closure", C.c_int32), ("buoyancy", C.c_int32),
                ("f", C.c_double), ("nu", C.c_double), ("g", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("T", C.c_void_p), ("S", C.c_void_p), ("pHY", C.c_void_p), ("nu_e", C.c_void_p),
                ("coriolis_beta", C.c_double), ("yc", C.c_void_p), ("yf", C.c_void_p)]


class CBc(C.Structure):
    """struct ocn_bc"""
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("value", C.c_double), ("coeff", C.c_double), ("values", C.c_void_p)]


class CFieldBcs(C.Structure):
    """struct ocn_field_bcs"""
    _fields_ = [(n, CBc) for n in ("west", "east", "south", "north", "bottom", "top")]


MODEL_MAX_TRACERS = 4


class CModelDriverDesc(C.Structure):
    """struct ocn_model_driver_desc"""
    _fields_ = [("terms", CModelTerms), ("n_tracers", C.c_int32), ("tracer_T", C.c_int32), ("tracer_S", C.c_int32), ("_pad", C.c_int32),
                ("kappa", C.c_double * MODEL_MAX_TRACERS), ("C_nu", C.c_double), ("C_kappa", C.c_double * MODEL_MAX_TRACERS),
                ("tracers", C.c_void_p * MODEL_MAX_TRACERS), ("nu_e", C.c_void_p), ("kappa_e", C.c_void_p * MODEL_MAX_TRACERS),
                ("pHY", C.c_void_p), ("bcs", C.POINTER(CFieldBcs) * (3 + MODEL_MAX_TRACERS))]


class CCommOp(C.Structure):
    """struct ocn_comm_op"""
    _fields_ = [("is_recv", C.c_int32), ("peer", C.c_int32), ("slot", C.c_int32)]


SCHED_STRIPS, SCHED_PLANE_EAST, SCHED_PLANE_WEST, SCHED_ALL_TO_ALL, SCHED_ALL_GATHER = 0, 1, 2, 3, 4
ADVECTION_WENO5, ADVECTION_CENTERED2, ADVECTION_UPWIND5 = 0, 1, 2
BUOYANCY_NONE, BUOYANCY_TRACER, BUOYANCY_SEAWATER_TS, BUOYANCY_SEAWATER_T, BUOYANCY_SEAWATER_S = 0, 1, 2, 3, 4
BC_DEFAULT, BC_FLUX, BC_VALUE, BC_GRADIENT, BC_OPEN = 0, 1, 2, 3, 4

_lib = None

_vp, _i32, _dbl = C.c_void_p, C.c_int32, C.c_double
_SIGS = {
    "ocn_device_count": [C.POINTER(C.c_int)],
    "ocn_set_device": [C.c_int],
    "ocn_malloc": [C.POINTER(_vp), C.c_size_t],
    "ocn_free": [_vp],
    "ocn_memcpy_h2d": [_vp, _vp, C.c_size_t, _vp],
    "ocn_memcpy_d2h": [_vp, _vp, C.c_size_t, _vp],
    "ocn_memcpy_d2d": [_vp, _vp, C.c_size_t, _vp],
    "ocn_memset": [_vp, C.c_int, C.c_size_t, _vp],
    "ocn_sync": [_vp],
    "ocn_set_math_mode": [C.c_int],
    "ocn_fill_halo_regions": [C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _i32, _vp],
    "ocn_fill_halo_periodic": [C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _i32, _vp],
    "ocn_compute_momentum_tendencies": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "ocn_compute_momentum_tendencies_rk3": [C.POINTER(CGrid)] + [_vp] * 12 + [_dbl, _dbl, _dbl, _i32, _vp, _dbl, C.POINTER(_i32), _vp],
    "ocn_compute_momentum_tendencies_rk3_strips": [C.POINTER(CGrid)] + [_vp] * 12 + [_dbl, _dbl, _dbl, _i32, _vp, _dbl, _vp, _vp, C.c_int64, _vp],
    "ocn_compute_tracer_tendency": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "ocn_compute_momentum_tendencies_terms": [C.POINTER(CGrid), C.POINTER(CModelTerms), _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "ocn_compute_tracer_tendency_terms": [C.POINTER(CGrid), C.POINTER(CModelTerms), _dbl, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "ocn_compute_amd_viscosity": [C.POINTER(CGrid), _dbl, _vp, _vp, _vp, _vp, _vp],
    "ocn_compute_amd_diffusivities": [C.POINTER(CGrid), _dbl, _vp, _vp, _vp, _vp, _i32, C.POINTER(_dbl), C.POINTER(_vp), C.POINTER(_vp), _vp],
    "ocn_compute_amd_diffusivity": [C.POINTER(CGrid), _dbl, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_compute_momentum_tendencies_terms_rk3": [C.POINTER(CGrid), C.POINTER(CModelTerms), C.POINTER(CFieldBcs), C.POINTER(CFieldBcs)]
                                                 + [_vp] * 12 + [_dbl, _dbl, _dbl, _i32, C.POINTER(_i32), _vp],
    "ocn_compute_tracer_tendency_terms_rk3": [C.POINTER(CGrid), C.POINTER(CModelTerms), _dbl, _vp, C.POINTER(CFieldBcs)]
                                             + [_vp] * 7 + [_dbl, _dbl, _dbl, _i32, C.POINTER(_i32), _vp],
    "ocn_compute_tracer_pair_tendency_terms_rk3": [C.POINTER(CGrid), C.POINTER(CModelTerms), C.POINTER(_dbl), C.POINTER(_vp),
                                                   C.POINTER(C.POINTER(CFieldBcs)), _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                                   C.POINTER(_vp), _dbl, _dbl, _dbl, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp],
    "ocn_update_hydrostatic_pressure": [C.POINTER(CGrid), C.POINTER(CModelTerms), _vp, _vp],
    "ocn_update_hydrostatic_pressure_range": [C.POINTER(CGrid), C.POINTER(CModelTerms), _vp, _i32, _i32, _vp],
    "ocn_compute_amd_diffusivities_range": [C.POINTER(CGrid), _dbl, _vp, _vp, _vp, _vp, _i32, C.POINTER(_dbl), C.POINTER(_vp), C.POINTER(_vp), _i32, _i32, _vp],
    "ocn_fill_halo_regions_bcs": [C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), C.POINTER(C.POINTER(CFieldBcs)), _i32, _i32, _vp],
    "ocn_apply_flux_bcs": [C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), C.POINTER(C.POINTER(CFieldBcs)), _i32, _vp],
    "ocn_cell_advection_timescale": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp],
    "ocn_hasnan": [_vp, C.c_int64, _vp, _vp],
    "ocn_rk3_substep": [C.POINTER(CGrid), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _dbl, _dbl, _dbl, _i32, _vp],
    "ocn_split_rk3_substep": [C.POINTER(CGrid), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _dbl, _dbl, _dbl, _vp],
    "ocn_ab2_step": [C.POINTER(CGrid), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _dbl, _dbl, _vp],
    "ocn_cache_previous_tendencies": [C.POINTER(CGrid), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _vp],
    "ocn_pressure_correct_velocities": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _dbl, _vp],
    "ocn_divergence": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp],
    "ocn_poisson_create": [C.POINTER(_vp), C.POINTER(CGrid)],
    "ocn_poisson_destroy": [_vp],
    "ocn_poisson_info": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)],
    "ocn_poisson_compute_source_term": [_vp, _vp, _vp, _vp, _dbl, _vp],
    "ocn_poisson_set_source_term": [_vp, _vp, _vp],
    "ocn_poisson_solve": [_vp, _vp, _vp],
    "ocn_solve_for_pressure": [_vp, _vp, _vp, _vp, _vp, _dbl, _vp],
    "ocn_poisson_solve_shifted": [_vp, _vp, _dbl, _vp],
    "ocn_implicit_free_surface_rhs": [C.POINTER(CGrid), _vp, _vp, _vp, _dbl, _dbl, _vp, _vp, _vp, _vp],
    "ocn_barotropic_pressure_correction": [C.POINTER(CGrid), _vp, _vp, _vp, _dbl, _dbl, _vp],
    "ocn_batched_tridiagonal_solve_z": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_halo_pack_x": [C.POINTER(CGrid), _vp, _i32, _vp, _vp, _vp],
    "ocn_halo_unpack_x": [C.POINTER(CGrid), _vp, _i32, _vp, _vp, _vp],
    "ocn_add_momentum_terms": [C.POINTER(CGrid), C.POINTER(CModelTerms), _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "ocn_compute_vector_invariant_momentum_tendencies": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp],
    "ocn_split_explicit_forcing": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp],
    "ocn_split_explicit_substeps": [C.POINTER(CGrid), _i32, C.POINTER(_dbl), _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_compute_barotropic_mode": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp],
    "ocn_split_explicit_substeps_blocked": [C.POINTER(CGrid), _i32, C.POINTER(_dbl), _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_split_explicit_substeps_ab3": [C.POINTER(CGrid), _i32, C.POINTER(_dbl), _dbl, _dbl, _dbl, C.POINTER(_dbl)] + [_vp] * 10,
    "ocn_hydrostatic_momentum_ab2_step": [C.POINTER(CGrid), C.POINTER(CModelTerms), C.POINTER(CFieldBcs), C.POINTER(CFieldBcs)] + [_vp] * 9
                                         + [_dbl, _dbl, _i32, _vp, _dbl, _vp, _vp, _vp, _vp, _vp],
    "ocn_barotropic_corrector_and_w": [C.POINTER(CGrid)] + [_vp] * 9 + [_dbl, _vp],
    "ocn_barotropic_split_explicit_corrector": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp],
    "ocn_fill_free_surface_halos": [C.POINTER(CGrid), _vp, _vp],
    "ocn_compute_w_from_continuity": [C.POINTER(CGrid), _vp, _vp, _vp, _vp],
    "ocn_add_barotropic_pressure_gradient": [C.POINTER(CGrid), _dbl, _vp, _vp, _vp, _vp],
    "ocn_explicit_free_surface_ab2_step": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _dbl, _dbl, _vp],
    "ocn_rk3_driver_create": [C.POINTER(_vp), C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_rk3_driver_create_distributed": [C.POINTER(_vp), C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_rk3_driver_configure": [_vp, _i32],
    "ocn_rk3_driver_destroy": [_vp],
    "ocn_rk3_driver_time_step": [_vp, _dbl, _vp],
    "ocn_rk3_driver_flush": [_vp, _vp],
    "ocn_rk3_driver_fields": [_vp] + [C.POINTER(_vp)] * 6,
    "ocn_model_driver_create": [C.POINTER(_vp), C.POINTER(CGrid), C.POINTER(CModelDriverDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_model_driver_create_distributed": [C.POINTER(_vp), C.POINTER(CGrid), C.POINTER(CModelDriverDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_model_driver_destroy": [_vp],
    "ocn_model_driver_time_step": [_vp, _dbl, _vp],
    "ocn_model_driver_flush": [_vp, _vp],
    "ocn_model_driver_field": [_vp, _i32, C.POINTER(_vp), C.POINTER(_vp)],
    "ocn_halo_plane_x": [C.POINTER(CGrid), _vp, _i32, _i32, _vp, _i32, _vp],
    "ocn_halo_pack_pressure": [C.POINTER(CGrid), _vp, _vp, _dbl, _vp, _vp, _vp],
    "ocn_halo_unpack_pressure": [C.POINTER(CGrid), _vp, _vp, _vp, _vp, _vp],
    "ocn_halo_exchange_pressure": [_vp, C.POINTER(CGrid), _vp, _vp, _dbl, _vp],
    "ocn_halo_pack_x_fields": [C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _vp, _vp, _vp],
    "ocn_halo_unpack_x_fields": [C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _vp, _vp, _vp],
    "ocn_transpose_pack_y_to_x": [_i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "ocn_transpose_unpack_x_from_y": [_i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "ocn_transpose_pack_x_to_y": [_i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "ocn_transpose_unpack_y_from_x": [_i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "ocn_dist_poisson_create": [C.POINTER(_vp), C.POINTER(CGrid), _i32, _i32, _dbl],
    "ocn_dist_poisson_create_global": [C.POINTER(_vp), C.POINTER(CGrid), _i32, _i32, _dbl, _i32],
    "ocn_dist_poisson_destroy": [_vp],
    "ocn_dist_poisson_buffers": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "ocn_dist_poisson_layout": [_vp, C.POINTER(_i32), C.POINTER(C.c_int64), C.POINTER(_i32)],
    "ocn_dist_poisson_pipeline": [_vp, C.POINTER(_i32)],
    "ocn_dist_poisson_gather_buffers": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int64)],
    "ocn_dist_poisson_source_term": [_vp, _vp, _vp, _vp, _dbl, _vp],
    "ocn_dist_poisson_forward_yz": [_vp, _vp],
    "ocn_dist_poisson_solve_x": [_vp, _vp],
    "ocn_dist_poisson_backward_yz": [_vp, _vp, _vp],
    "ocn_comm_schedule": [_i32, _i32, _i32, _i32, C.POINTER(CCommOp), _i32, C.POINTER(_i32)],
    "ocn_comm_unique_id": [_vp],
    "ocn_comm_init": [C.POINTER(_vp), _i32, _i32, _vp],
    "ocn_comm_init_local": [C.POINTER(_vp), _i32, _i32, C.c_int64],
    "ocn_comm_init_replica": [C.POINTER(_vp), _i32],
    "ocn_halo_exchange_buffers": [_vp, C.POINTER(CGrid), C.POINTER(_i32), _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int64)],
    "ocn_halo_exchange_begin_packed": [_vp, C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _vp],
    "ocn_comm_enable_stats": [_vp, _i32],
    "ocn_comm_stats": [_vp, C.POINTER(C.c_double)],
    "ocn_comm_wait": [_vp, C.c_double],
    "ocn_sync_timeout": [_vp, C.c_double],
    "ocn_profile_marker": [_vp],
    "ocn_comm_destroy": [_vp],
    "ocn_comm_info": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)],
    "ocn_halo_exchange_begin": [_vp, C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _vp],
    "ocn_halo_exchange_end": [_vp, C.POINTER(CGrid), C.POINTER(_vp), C.POINTER(_i32), _i32, _vp],
    "ocn_halo_exchange_plane": [_vp, C.POINTER(CGrid), _vp, _i32, _i32, _vp],
    "ocn_comm_all_to_all": [_vp, _vp, _vp, C.c_size_t, _vp],
    "ocn_comm_all_gather": [_vp, _vp, _vp, C.c_size_t, _vp],
    "ocn_comm_exchange_strips": [_vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp],
    "ocn_split_explicit_dist_begin": [C.POINTER(CGrid), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_split_explicit_dist_run": [C.POINTER(CGrid), _i32, C.POINTER(_dbl), _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ocn_dist_poisson_exchange": [_vp, _vp, _i32, _vp],
    "ocn_comm_allreduce": [_vp, _vp, C.c_size_t, _i32, _vp],
    "ocn_comm_barrier": [_vp],
}
EXPORTED_SYMBOLS = sorted(list(_SIGS) + ["ocn_last_error", "ocn_version", "ocn_get_math_mode"])


def lib():
    """Load libocn_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OcnError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C oceananigans.jl_amd/csrc`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.ocn_last_error.restype = C.c_char_p
        L.ocn_version.restype = C.c_char_p
        L.ocn_get_math_mode.restype = C.c_int
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise OcnError(f"libocn_hip error {status}: {lib().ocn_last_error().decode()}")


def call(name, *args):
    check(getattr(lib(), name)(*args))


def ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


def i32_array(vals):
    return (C.c_int32 * len(vals))(*vals)
