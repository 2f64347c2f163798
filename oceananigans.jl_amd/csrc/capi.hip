// capi.hip -- extern "C" entry points of libocn_hip.so (see include/ocn_hip.h for the contract).
#include <atomic>
#include <cstring>
#include <string>

#include "ocn_internal.h"

namespace ocn {

static thread_local std::string g_last_error;
// process default of the arithmetic variant; a grid that carries its own (ocn_grid.math != OCN_GRID_MATH_DEFAULT) overrides it, so two
// models in one process may differ and a host thread changing the default does not touch another handle's launches
static std::atomic<int> g_math_mode{OCN_MATH_STRICT};
static inline bool strict_math(const ocn_grid *grid)
{
    const int m = grid ? grid->math : OCN_GRID_MATH_DEFAULT;
    if (m == OCN_GRID_MATH_STRICT) return true;
    if (m == OCN_GRID_MATH_FAST) return false;
    return g_math_mode.load(std::memory_order_relaxed) == OCN_MATH_STRICT;
}

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// any_xy: the entry point has a direction-generic path (csrc/general.hip, fill_halos_general_kernel, the per-location layouts of the
// steppers and pressure kernels) and accepts Bounded / Flat x and y; the others keep the Periodic (or partitioned) x, Periodic y
// their kernels are written for
static int validate_grid_impl(const ocn_grid *g, bool any_xy)
{
    OCN_REQUIRE(g != nullptr, "grid is NULL");
    OCN_REQUIRE(g->Nx >= 1 && g->Ny >= 1 && g->Nz >= 1, "grid size must be positive, got (%d, %d, %d)", g->Nx, g->Ny, g->Nz);
    OCN_REQUIRE(g->Hx >= 0 && g->Hy >= 0 && g->Hz >= 0, "negative halo");
    const int t[3] = {g->tx, g->ty, g->tz};
    const int N[3] = {g->Nx, g->Ny, g->Nz};
    const int H[3] = {g->Hx, g->Hy, g->Hz};
    for (int d = 0; d < 3; ++d) {
        OCN_REQUIRE(t[d] >= OCN_PERIODIC && t[d] <= (d == 0 ? OCN_LEFT_CONNECTED : OCN_FULLY_CONNECTED), "unknown topology code %d", t[d]);
        if (t[d] == OCN_FLAT) OCN_REQUIRE(N[d] == 1 && H[d] == 0, "Flat dimension %d must have N = 1, H = 0", d);
        // halo must not exceed the interior (Grids/input_validation.jl: halo <= size)
        if (t[d] != OCN_FLAT) OCN_REQUIRE(H[d] <= N[d], "halo %d larger than size %d in dimension %d", H[d], N[d], d);
    }
    if (!any_xy && (!(g->tx == OCN_PERIODIC || g->tx == OCN_FULLY_CONNECTED) || g->ty != OCN_PERIODIC)) {
        set_error("unsupported topology (%d, %d, %d): x must be Periodic (or FullyConnected), y Periodic", g->tx, g->ty, g->tz);
        return OCN_ERR_UNSUPPORTED;
    }
    if (g->ty == OCN_FULLY_CONNECTED) {
        set_error("only x is ever partitioned (slab decomposition)");
        return OCN_ERR_UNSUPPORTED;
    }
    // a slab of a (Periodic, Bounded, *) grid: the direction-generic kernels read the exchanged x halos like periodic images and treat
    // the y walls locally (distributed_grids.jl:75-118)
    const bool part_x = g->tx == OCN_FULLY_CONNECTED || g->tx == OCN_RIGHT_CONNECTED || g->tx == OCN_LEFT_CONNECTED;
    if (any_xy && g->ty == OCN_FLAT && part_x) {
        set_error("a partitioned x needs a Periodic or Bounded y");
        return OCN_ERR_UNSUPPORTED;
    }
    // the first / last slab of a Bounded x (RightConnected / LeftConnected): the interior must reach past the wall stencils
    if (g->tx == OCN_RIGHT_CONNECTED || g->tx == OCN_LEFT_CONNECTED)
        OCN_REQUIRE(g->Nx >= 2, "a half-Bounded slab needs Nx >= 2 (got %d)", g->Nx);
    if (g->tz == OCN_FULLY_CONNECTED) {
        set_error("z is never partitioned (distributed_architectures.jl:223-225)");
        return OCN_ERR_UNSUPPORTED;
    }
    OCN_REQUIRE((g->dzc == nullptr) == (g->dzf == nullptr), "dzc and dzf must both be set or both be NULL");
    OCN_REQUIRE(g->dx > 0 && g->dy > 0 && (g->dzc || g->dz > 0), "spacings must be positive");
    return OCN_SUCCESS;
}
int validate_grid(const ocn_grid *g) { return validate_grid_impl(g, false); }
int validate_grid_any(const ocn_grid *g) { return validate_grid_impl(g, true); }
// x and y Periodic (x possibly partitioned): the tiled / shared-layout kernels apply; otherwise the direction-generic ones
// west / east conditions on a slab of a partitioned x: the slab that holds the wall takes them, the others drop them (the host checks the
// GLOBAL topology: a slab between the walls cannot tell a Bounded from a Periodic x)
static bool partitioned_x(const ocn_grid *g) { return g->tx == OCN_FULLY_CONNECTED || g->tx == OCN_RIGHT_CONNECTED || g->tx == OCN_LEFT_CONNECTED; }
static bool side_has_wall(const ocn_grid *g, int q)
{
    if (q == 0) return x_wall_west(*g);
    if (q == 1) return x_wall_east(*g);
    return (q < 4 ? g->ty : g->tz) == OCN_BOUNDED;
}
static bool xy_periodic(const ocn_grid *g) { return (g->tx == OCN_PERIODIC || g->tx == OCN_FULLY_CONNECTED) && g->ty == OCN_PERIODIC; }

// WENO5 reads 3 halo cells (nonhydrostatic_model.jl:183, 243-257 inflates the halo to >= 3)
static int validate_weno(const ocn_grid *g)
{
    int st = validate_grid_any(g);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE((g->tx == OCN_FLAT || g->Hx >= 3) && (g->ty == OCN_FLAT || g->Hy >= 3) && (g->tz == OCN_FLAT || g->Hz >= 3),
                "WENO(order=5) needs halo >= 3, got (%d, %d, %d)", g->Hx, g->Hy, g->Hz);
    OCN_REQUIRE((g->tx == OCN_FLAT || g->Nx >= 3) && (g->ty == OCN_FLAT || g->Ny >= 3) && (g->tz == OCN_FLAT || g->Nz >= 3),
                "grid too small for WENO(order=5): adapt_advection_order would lower the order (adapt_advection_order.jl:101-108)");
    return OCN_SUCCESS;
}

}  // namespace ocn

using namespace ocn;

extern "C" {

const char *ocn_last_error(void) { return g_last_error.c_str(); }
const char *ocn_version(void) { return "libocn_hip 0.1 (gfx950)"; }

int ocn_device_count(int *count)
{
    OCN_REQUIRE(count, "count is NULL");
    OCN_CHECK_HIP(hipGetDeviceCount(count));
    return OCN_SUCCESS;
}
int ocn_set_device(int device)
{
    OCN_CHECK_HIP(hipSetDevice(device));
    return OCN_SUCCESS;
}
int ocn_malloc(void **ptr, size_t bytes)
{
    OCN_REQUIRE(ptr, "ptr is NULL");
    *ptr = nullptr;
    if (bytes == 0) return OCN_SUCCESS;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return OCN_ERR_ALLOC;
    }
    OCN_CHECK_HIP(hipMemset(*ptr, 0, bytes));
    return OCN_SUCCESS;
}
int ocn_free(void *ptr)
{
    if (ptr) OCN_CHECK_HIP(hipFree(ptr));
    return OCN_SUCCESS;
}
int ocn_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    OCN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    OCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return OCN_SUCCESS;
}
int ocn_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    OCN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    OCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return OCN_SUCCESS;
}
int ocn_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    OCN_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return OCN_SUCCESS;
}
int ocn_memset(void *ptr, int value, size_t bytes, void *stream)
{
    OCN_CHECK_HIP(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return OCN_SUCCESS;
}
int ocn_sync(void *stream)
{
    OCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return OCN_SUCCESS;
}

int ocn_sync_timeout(void *stream, double seconds)
{
    return ocn::wait_stream(as_stream(stream), seconds, "ocn_sync_timeout");
}

int ocn_profile_marker(void *stream) { return ocn::launch_profile_marker(as_stream(stream)); }

int ocn_set_math_mode(int mode)
{
    OCN_REQUIRE(mode == OCN_MATH_STRICT || mode == OCN_MATH_FAST, "unknown math mode %d", mode);
    g_math_mode = mode;
    return OCN_SUCCESS;
}
int ocn_get_math_mode(void) { return g_math_mode; }

static int make_field_tuple(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, FieldTuple &ft)
{
    OCN_REQUIRE(fields && locs, "fields/locs is NULL");
    OCN_REQUIRE(n >= 1 && n <= MAX_TUPLE, "number of fields %d outside 1..%d", n, MAX_TUPLE);
    ft.n = n;
    for (int f = 0; f < n; ++f) {
        OCN_REQUIRE(fields[f] != nullptr, "field %d is NULL", f);
        OCN_REQUIRE(locs[f] == OCN_LOC_CCC || locs[f] == OCN_LOC_FCC || locs[f] == OCN_LOC_CFC || locs[f] == OCN_LOC_CCF,
                    "field %d has unsupported location mask %d", f, locs[f]);
        ft.f[f] = fields[f];
        ft.loc[f] = locs[f];
    }
    (void)grid;
    return OCN_SUCCESS;
}

int ocn_fill_halo_regions(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n,
                          int32_t fill_boundary_normal_velocities, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    FieldTuple ft;
    st = make_field_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    if (!xy_periodic(grid)) return launch_fill_halos_general(grid, ft, fill_boundary_normal_velocities, as_stream(stream));
    return launch_fill_halos(grid, ft, fill_boundary_normal_velocities, -1, as_stream(stream));
}

int ocn_fill_halo_periodic(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, int32_t dir, void *stream)
{
    int st = validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(dir >= 0 && dir <= 2, "dir must be 0, 1 or 2");
    const int t = dir == 0 ? grid->tx : dir == 1 ? grid->ty : grid->tz;
    OCN_REQUIRE(t == OCN_PERIODIC, "direction %d is not Periodic", dir);
    FieldTuple ft;
    st = make_field_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    return launch_fill_halos(grid, ft, 0, dir, as_stream(stream));
}

int ocn_compute_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                    double *Gv, double *Gw, const int32_t *range, void *stream)
{
    int st = validate_weno(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && Gu && Gv && Gw, "ocn_compute_momentum_tendencies: null field pointer");
    if (!xy_periodic(grid))
        return strict_math(grid) ? ocn_strict::launch_momentum_tendencies_general(grid, 0, u, v, w, Gu, Gv, Gw, range, as_stream(stream))
                                              : ocn_fast::launch_momentum_tendencies_general(grid, 0, u, v, w, Gu, Gv, Gw, range, as_stream(stream));
    if (strict_math(grid)) return ocn_strict::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, nullptr, as_stream(stream));
    return ocn_fast::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, nullptr, as_stream(stream));
}

int ocn_compute_momentum_tendencies_rk3(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                        double *Gv, double *Gw, const double *Gmu, const double *Gmv, const double *Gmw,
                                        double *u_out, double *v_out, double *w_out, double dt, double gamma, double zeta,
                                        int32_t has_zeta, const double *p_correct, double dt_correct, const int32_t *range,
                                        void *stream)
{
    int st = validate_weno(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && Gu && Gv && Gw && u_out && v_out && w_out, "ocn_compute_momentum_tendencies_rk3: null field pointer");
    OCN_REQUIRE(!has_zeta || (Gmu && Gmv && Gmw), "ocn_compute_momentum_tendencies_rk3: G⁻ pointers are required when has_zeta != 0");
    OCN_REQUIRE(u_out != u && v_out != v && w_out != w, "ocn_compute_momentum_tendencies_rk3: outputs must not alias the inputs");
    if (!xy_periodic(grid)) {
        // walls in x / y: the substep as the epilogue of the tiled kernel on the interior box, as one more per-cell kernel on the wall frames
        // (wall faces carried over); no correction on load (its wrapped indices are the Periodic grids')
        OCN_REQUIRE(!p_correct, "ocn_compute_momentum_tendencies_rk3: the pressure correction on load needs Periodic x and y");
        OCN_REQUIRE(!range, "ocn_compute_momentum_tendencies_rk3: ranges need Periodic x and y");
        MomentumFinal mf{};
        mf.sub[0] = SubstepDev{Gmu, u_out};
        mf.sub[1] = SubstepDev{Gmv, v_out};
        mf.sub[2] = SubstepDev{Gmw, w_out};
        mf.sc = SubstepCoef{dt, gamma, zeta, 1, has_zeta ? 1 : 0};
        return strict_math(grid) ? ocn_strict::launch_momentum_tendencies_general(grid, 0, u, v, w, Gu, Gv, Gw, nullptr, as_stream(stream), &mf)
                                 : ocn_fast::launch_momentum_tendencies_general(grid, 0, u, v, w, Gu, Gv, Gw, nullptr, as_stream(stream), &mf);
    }
    FuseArgs fz{};
    fz.Gm[0] = Gmu; fz.Gm[1] = Gmv; fz.Gm[2] = Gmw;
    fz.Uo[0] = u_out; fz.Uo[1] = v_out; fz.Uo[2] = w_out;
    fz.dt = dt; fz.gamma = gamma; fz.zeta = zeta; fz.on = 1; fz.has_zeta = has_zeta ? 1 : 0;
    fz.pc_p = p_correct; fz.pc_dt = dt_correct; fz.pc_on = p_correct ? 1 : 0;
    if (strict_math(grid)) return ocn_strict::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, &fz, as_stream(stream));
    return ocn_fast::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, &fz, as_stream(stream));
}

int ocn_compute_momentum_tendencies_rk3_strips(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                               double *Gv, double *Gw, const double *Gmu, const double *Gmv, const double *Gmw,
                                               double *u_out, double *v_out, double *w_out, double dt, double gamma, double zeta,
                                               int32_t has_zeta, const double *p_correct, double dt_correct, double *strip_west,
                                               double *strip_east, int64_t field_doubles, void *stream)
{
    int st = validate_weno(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && Gu && Gv && Gw && u_out && v_out && w_out, "ocn_compute_momentum_tendencies_rk3_strips: null field pointer");
    OCN_REQUIRE(!has_zeta || (Gmu && Gmv && Gmw), "ocn_compute_momentum_tendencies_rk3_strips: G⁻ pointers are required when has_zeta != 0");
    OCN_REQUIRE(u_out != u && v_out != v && w_out != w, "ocn_compute_momentum_tendencies_rk3_strips: outputs must not alias the inputs");
    OCN_REQUIRE(p_correct && strip_west && strip_east && field_doubles > 0,
                "ocn_compute_momentum_tendencies_rk3_strips: the correction-on-load stage of a slab (p_correct) and both strip buffers");
    OCN_REQUIRE(grid->tx == OCN_FULLY_CONNECTED && grid->ty == OCN_PERIODIC && grid->tz == OCN_PERIODIC,
                "ocn_compute_momentum_tendencies_rk3_strips: a (FullyConnected, Periodic, Periodic) local grid");
    FuseArgs fz{};
    fz.Gm[0] = Gmu; fz.Gm[1] = Gmv; fz.Gm[2] = Gmw;
    fz.Uo[0] = u_out; fz.Uo[1] = v_out; fz.Uo[2] = w_out;
    fz.dt = dt; fz.gamma = gamma; fz.zeta = zeta; fz.on = 1; fz.has_zeta = has_zeta ? 1 : 0;
    fz.pc_p = p_correct; fz.pc_dt = dt_correct; fz.pc_on = 1;
    fz.strip_w = strip_west; fz.strip_e = strip_east; fz.strip_field = field_doubles;
    if (strict_math(grid)) return ocn_strict::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, nullptr, &fz, as_stream(stream));
    return ocn_fast::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, nullptr, &fz, as_stream(stream));
}

int ocn_cell_advection_timescale(const ocn_grid *grid, const double *u, const double *v, const double *w, double *result_device,
                                 void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && result_device, "ocn_cell_advection_timescale: null pointer");
    return launch_advection_timescale(grid, u, v, w, result_device, as_stream(stream));
}

int ocn_hasnan(const double *field, int64_t n_elements, int32_t *flag_device, void *stream)
{
    OCN_REQUIRE(field && flag_device, "ocn_hasnan: null pointer");
    OCN_REQUIRE(n_elements >= 0, "ocn_hasnan: negative element count");
    OCN_REQUIRE((reinterpret_cast<uintptr_t>(field) & 15) == 0, "ocn_hasnan: field must be 16-byte aligned");
    return launch_hasnan(field, n_elements, flag_device, as_stream(stream));
}

// ---- SURVEY §8(f) rank 1 ------------------------------------------------------------------------------
static int validate_terms(const ocn_grid *grid, const ocn_model_terms *t)
{
    OCN_REQUIRE(t != nullptr, "terms is NULL");
    OCN_REQUIRE(t->advection >= OCN_ADVECTION_WENO5 && t->advection <= OCN_ADVECTION_UPWIND5, "unknown advection scheme %d", t->advection);
    OCN_REQUIRE(t->coriolis >= 0 && t->coriolis <= 2, "unknown coriolis code %d", t->coriolis);
    OCN_REQUIRE(t->coriolis != 2 || (t->yc && t->yf && grid->ty != OCN_FLAT), "BetaPlane needs the y node vectors yc, yf (and a non-Flat y)");
    OCN_REQUIRE(t->closure >= 0 && t->closure <= 2, "unknown closure code %d", t->closure);
    OCN_REQUIRE((t->closure == 2) == (t->nu_e != nullptr), "nu_e must be given exactly when closure == 2 (AnisotropicMinimumDissipation)");
    OCN_REQUIRE(t->closure != 2 || grid->tz != OCN_FLAT, "AnisotropicMinimumDissipation needs a non-Flat z");
    OCN_REQUIRE(t->buoyancy >= OCN_BUOYANCY_NONE && t->buoyancy <= OCN_BUOYANCY_SEAWATER_S, "unknown buoyancy code %d", t->buoyancy);
    if (t->buoyancy == OCN_BUOYANCY_TRACER || t->buoyancy == OCN_BUOYANCY_SEAWATER_TS || t->buoyancy == OCN_BUOYANCY_SEAWATER_T)
        OCN_REQUIRE(t->T != nullptr, "buoyancy formulation %d needs the T (or b) tracer", t->buoyancy);
    if (t->buoyancy == OCN_BUOYANCY_SEAWATER_TS || t->buoyancy == OCN_BUOYANCY_SEAWATER_S)
        OCN_REQUIRE(t->S != nullptr, "buoyancy formulation %d needs the S tracer", t->buoyancy);
    OCN_REQUIRE(!t->pHY || t->buoyancy != OCN_BUOYANCY_NONE, "a hydrostatic pressure anomaly exists only with buoyancy (nonhydrostatic_model.jl:143-158)");
    if (t->advection != OCN_ADVECTION_CENTERED2) return validate_weno(grid);  // UpwindBiased(order=5): same halo / size needs
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE((grid->tx == OCN_FLAT || grid->Hx >= 1) && (grid->ty == OCN_FLAT || grid->Hy >= 1) && (grid->tz == OCN_FLAT || grid->Hz >= 1),
                "Centered(order=2) and the closure stencils need halo >= 1");
    return OCN_SUCCESS;
}

// advective part by scheme and math mode
static int launch_advective_momentum(int advection, const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                     double *Gv, double *Gw, const int32_t *range, hipStream_t s)
{
    const bool strict = strict_math(grid);
    if (!xy_periodic(grid)) {  // direction-generic kernels (csrc/general.hip)
        const int c2 = advection == OCN_ADVECTION_CENTERED2;
        if (advection == OCN_ADVECTION_UPWIND5)
            return strict ? ocn_strict_up::launch_momentum_tendencies_general(grid, 0, u, v, w, Gu, Gv, Gw, range, s)
                          : ocn_fast_up::launch_momentum_tendencies_general(grid, 0, u, v, w, Gu, Gv, Gw, range, s);
        return strict ? ocn_strict::launch_momentum_tendencies_general(grid, c2, u, v, w, Gu, Gv, Gw, range, s)
                      : ocn_fast::launch_momentum_tendencies_general(grid, c2, u, v, w, Gu, Gv, Gw, range, s);
    }
    switch (advection) {
        case OCN_ADVECTION_WENO5:
            return strict ? ocn_strict::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, nullptr, s)
                          : ocn_fast::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, nullptr, s);
        case OCN_ADVECTION_UPWIND5:
            return strict ? ocn_strict_up::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, nullptr, s)
                          : ocn_fast_up::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, nullptr, s);
        default:
            return strict ? ocn_strict::launch_momentum_centered2(grid, u, v, w, Gu, Gv, Gw, range, s)
                          : ocn_fast::launch_momentum_centered2(grid, u, v, w, Gu, Gv, Gw, range, s);
    }
}
static int launch_advective_tracer(int advection, const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                                   double *Gc, const int32_t *range, hipStream_t s, const TracerFuse *tf)
{
    const bool strict = strict_math(grid);
    if (!xy_periodic(grid)) {  // (tf: diffusion / bottom and top fluxes / the next substep ride along, general.hip)
        const int c2 = advection == OCN_ADVECTION_CENTERED2;
        if (advection == OCN_ADVECTION_UPWIND5)
            return strict ? ocn_strict_up::launch_tracer_tendency_general(grid, 0, u, v, w, c, Gc, range, s, tf)
                          : ocn_fast_up::launch_tracer_tendency_general(grid, 0, u, v, w, c, Gc, range, s, tf);
        return strict ? ocn_strict::launch_tracer_tendency_general(grid, c2, u, v, w, c, Gc, range, s, tf)
                      : ocn_fast::launch_tracer_tendency_general(grid, c2, u, v, w, c, Gc, range, s, tf);
    }
    switch (advection) {
        case OCN_ADVECTION_WENO5:
            return strict ? ocn_strict::launch_tracer_tendency(grid, u, v, w, c, Gc, range, s, tf) : ocn_fast::launch_tracer_tendency(grid, u, v, w, c, Gc, range, s, tf);
        case OCN_ADVECTION_UPWIND5:
            return strict ? ocn_strict_up::launch_tracer_tendency(grid, u, v, w, c, Gc, range, s, tf) : ocn_fast_up::launch_tracer_tendency(grid, u, v, w, c, Gc, range, s, tf);
        default:
            return strict ? ocn_strict::launch_tracer_centered2(grid, u, v, w, c, Gc, range, s) : ocn_fast::launch_tracer_centered2(grid, u, v, w, c, Gc, range, s);
    }
}

int ocn_compute_momentum_tendencies_terms(const ocn_grid *grid, const ocn_model_terms *terms, const double *u, const double *v,
                                          const double *w, double *Gu, double *Gv, double *Gw, const int32_t *range,
                                          void *stream)
{
    int st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && Gu && Gv && Gw, "ocn_compute_momentum_tendencies_terms: null field pointer");
    OCN_REQUIRE((grid->tx == OCN_FLAT || grid->Hx >= 1) && (grid->ty == OCN_FLAT || grid->Hy >= 1) && (grid->tz == OCN_FLAT || grid->Hz >= 1), "halo >= 1 required");
    const bool strict = strict_math(grid);
    hipStream_t s = as_stream(stream);
    st = launch_advective_momentum(terms->advection, grid, u, v, w, Gu, Gv, Gw, range, s);
    if (st != OCN_SUCCESS) return st;
    if (!(terms->coriolis || terms->closure || terms->buoyancy)) return OCN_SUCCESS;
    TermsDev t = to_dev(*terms);
    if (!xy_periodic(grid))
        return strict ? ocn_strict::launch_momentum_extra_general(grid, t, u, v, w, Gu, Gv, Gw, range, s)
                      : ocn_fast::launch_momentum_extra_general(grid, t, u, v, w, Gu, Gv, Gw, range, s);
    return strict ? ocn_strict::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, s)
                  : ocn_fast::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, s);
}

// the extra terms alone, added to a G that already holds what precedes them in the reference's sum (used by the hydrostatic
// tendencies, where - g ∂x η comes between the advection and the Coriolis term)
int ocn_add_momentum_terms(const ocn_grid *grid, const ocn_model_terms *terms, const double *u, const double *v, const double *w,
                           double *Gu, double *Gv, double *Gw, const int32_t *range, void *stream)
{
    int st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && Gu && Gv && Gw, "ocn_add_momentum_terms: null field pointer");
    OCN_REQUIRE(grid->Hx >= 1 && grid->Hy >= 1 && (grid->tz == OCN_FLAT || grid->Hz >= 1), "halo >= 1 required");
    if (!(terms->coriolis || terms->closure || terms->buoyancy)) return OCN_SUCCESS;
    TermsDev t = to_dev(*terms);
    return strict_math(grid) ? ocn_strict::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, as_stream(stream))
                                          : ocn_fast::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, as_stream(stream));
}

static int validate_hydrostatic(const ocn_grid *grid, const char *who)
{
    int st = validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    // x may be the partitioned direction of a slab-x rank (FullyConnected: halos come from the neighbours, interior arithmetic as Periodic)
    OCN_REQUIRE((grid->tx == OCN_PERIODIC || grid->tx == OCN_FULLY_CONNECTED) && grid->ty == OCN_PERIODIC && grid->tz == OCN_BOUNDED,
                "%s: the hydrostatic slice supports (Periodic, Periodic, Bounded) grids", who);
    OCN_REQUIRE(grid->Hx >= 1 && grid->Hy >= 1, "%s: needs x, y halos", who);
    return OCN_SUCCESS;
}
#define OCN_REQUIRE_PERIODIC_X(who) OCN_REQUIRE(grid->tx == OCN_PERIODIC, "%s wraps x periodically: not for a partitioned (FullyConnected) x", who)
int ocn_compute_vector_invariant_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                                     double *Gv, const double *eta, double gravitational_acceleration, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_compute_vector_invariant_momentum_tendencies");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && Gu && Gv, "ocn_compute_vector_invariant_momentum_tendencies: null field pointer");
    OCN_REQUIRE(grid->Hz >= 1, "ocn_compute_vector_invariant_momentum_tendencies: needs a z halo");
    return launch_vector_invariant(grid, u, v, w, Gu, Gv, as_stream(stream), eta, gravitational_acceleration);
}
int ocn_split_explicit_forcing(const ocn_grid *grid, const double *Gu, const double *Gu_previous, const double *Gv, const double *Gv_previous,
                               double chi, double *GU, double *GV, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_split_explicit_forcing");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(Gu && Gu_previous && Gv && Gv_previous && GU && GV, "ocn_split_explicit_forcing: null pointer");
    return launch_split_explicit_forcing(grid, Gu, Gu_previous, Gv, Gv_previous, chi, GU, GV, as_stream(stream));
}
int ocn_split_explicit_substeps(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                double column_depth, double *eta, double *U, double *V, double *eta_filtered, double *U_filtered,
                                double *V_filtered, const double *GU, const double *GV, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_split_explicit_substeps");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE_PERIODIC_X("ocn_split_explicit_substeps");
    OCN_REQUIRE(n >= 1 && weights, "ocn_split_explicit_substeps: needs n >= 1 averaging weights (host array)");
    OCN_REQUIRE(eta && U && V && eta_filtered && U_filtered && V_filtered && GU && GV, "ocn_split_explicit_substeps: null pointer");
    return launch_split_explicit_substeps(grid, n, weights, dtau, gravitational_acceleration, column_depth, eta, U, V, eta_filtered, U_filtered,
                                          V_filtered, GU, GV, as_stream(stream));
}
int ocn_compute_barotropic_mode(const ocn_grid *grid, const double *u, const double *v, double *U, double *V, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_compute_barotropic_mode");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && U && V, "ocn_compute_barotropic_mode: null pointer");
    return launch_barotropic_mode(grid, u, v, U, V, as_stream(stream));
}
int ocn_barotropic_split_explicit_corrector(const ocn_grid *grid, double *u, double *v, const double *U, const double *V, double *U_filtered,
                                            double *V_filtered, double column_depth, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_barotropic_split_explicit_corrector");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && U && V && U_filtered && V_filtered, "ocn_barotropic_split_explicit_corrector: null pointer");
    return launch_barotropic_corrector(grid, u, v, U, V, U_filtered, V_filtered, column_depth, as_stream(stream));
}
int ocn_fill_free_surface_halos(const ocn_grid *grid, double *eta, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_fill_free_surface_halos");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(eta, "ocn_fill_free_surface_halos: null pointer");
    return launch_plane_halo(grid, eta, as_stream(stream));
}
int ocn_compute_w_from_continuity(const ocn_grid *grid, const double *u, const double *v, double *w, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_compute_w_from_continuity");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w, "ocn_compute_w_from_continuity: null field pointer");
    return launch_w_from_continuity(grid, u, v, w, as_stream(stream));
}
int ocn_add_barotropic_pressure_gradient(const ocn_grid *grid, double gravitational_acceleration, const double *eta, double *Gu, double *Gv,
                                         void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_add_barotropic_pressure_gradient");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(eta && Gu && Gv, "ocn_add_barotropic_pressure_gradient: null pointer");
    return launch_barotropic_gradient(grid, gravitational_acceleration, eta, Gu, Gv, as_stream(stream));
}
int ocn_explicit_free_surface_ab2_step(const ocn_grid *grid, const double *w, double *eta, double *G_eta, const double *G_eta_previous,
                                       double dt, double chi, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_explicit_free_surface_ab2_step");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(w && eta && G_eta && G_eta_previous, "ocn_explicit_free_surface_ab2_step: null pointer");
    return launch_free_surface_ab2(grid, w, eta, G_eta, G_eta_previous, dt, chi, as_stream(stream));
}

int ocn_compute_tracer_tendency_terms(const ocn_grid *grid, const ocn_model_terms *terms, double kappa, const double *kappa_e,
                                      const double *u, const double *v, const double *w, const double *c, double *Gc,
                                      const int32_t *range, void *stream)
{
    int st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && c && Gc, "ocn_compute_tracer_tendency_terms: null field pointer");
    const bool strict = strict_math(grid);
    hipStream_t s = as_stream(stream);
    OCN_REQUIRE(!kappa_e || terms->closure == 2, "kappa_e is only meaningful with closure == 2");
    if (!xy_periodic(grid) && terms->closure) {  // advection and diffusion in one call: the interior box adds both before its store
        TracerFuse tf{};
        tf.diffusion = 1;
        tf.kappa = kappa;
        tf.kappa_e = kappa_e;
        return launch_advective_tracer(terms->advection, grid, u, v, w, c, Gc, range, s, &tf);
    }
    st = launch_advective_tracer(terms->advection, grid, u, v, w, c, Gc, range, s, nullptr);
    if (st != OCN_SUCCESS || !terms->closure) return st;
    return strict ? ocn_strict::launch_tracer_diffusion(grid, kappa, kappa_e, c, Gc, range, s)
                  : ocn_fast::launch_tracer_diffusion(grid, kappa, kappa_e, c, Gc, range, s);
}

static int validate_amd(const ocn_grid *grid)
{
    int st = validate_grid_any(grid);  // Bounded x / y: the kernel takes per-field parent layouts
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->tz != OCN_FLAT, "AnisotropicMinimumDissipation needs a non-Flat z");
    OCN_REQUIRE((grid->tx == OCN_FLAT || grid->Hx >= 1) && (grid->ty == OCN_FLAT || grid->Hy >= 1) && grid->Hz >= 1,
                "AnisotropicMinimumDissipation needs halo >= 1");
    return OCN_SUCCESS;
}

int ocn_compute_amd_viscosity(const ocn_grid *grid, double C_nu, const double *u, const double *v, const double *w, double *nu_e,
                              void *stream)
{
    int st = validate_amd(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && nu_e, "ocn_compute_amd_viscosity: null field pointer");
    return strict_math(grid) ? ocn_strict::launch_amd_viscosity(grid, C_nu, u, v, w, nu_e, as_stream(stream))
                                          : ocn_fast::launch_amd_viscosity(grid, C_nu, u, v, w, nu_e, as_stream(stream));
}

int ocn_compute_amd_diffusivities(const ocn_grid *grid, double C_nu, const double *u, const double *v, const double *w,
                                  double *nu_e, int32_t n_tracers, const double *C_kappa, const double *const *tracers,
                                  double *const *kappa_e, void *stream)
{
    int st = validate_amd(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && nu_e, "ocn_compute_amd_diffusivities: null field pointer");
    OCN_REQUIRE(n_tracers >= 0 && n_tracers <= 4, "ocn_compute_amd_diffusivities: n_tracers = %d outside 0..4", n_tracers);
    OCN_REQUIRE(n_tracers == 0 || (C_kappa && tracers && kappa_e), "ocn_compute_amd_diffusivities: null tracer arrays");
    for (int n = 0; n < n_tracers; ++n) OCN_REQUIRE(tracers[n] && kappa_e[n], "ocn_compute_amd_diffusivities: tracer %d is NULL", n);
    return strict_math(grid)
               ? ocn_strict::launch_amd_fused(grid, C_nu, u, v, w, nu_e, n_tracers, C_kappa, tracers, kappa_e, as_stream(stream))
               : ocn_fast::launch_amd_fused(grid, C_nu, u, v, w, nu_e, n_tracers, C_kappa, tracers, kappa_e, as_stream(stream));
}

int ocn_compute_amd_diffusivities_range(const ocn_grid *grid, double C_nu, const double *u, const double *v, const double *w,
                                        double *nu_e, int32_t n_tracers, const double *C_kappa, const double *const *tracers,
                                        double *const *kappa_e, int32_t i_first, int32_t i_last, void *stream)
{
    int st = validate_amd(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && nu_e, "ocn_compute_amd_diffusivities_range: null field pointer");
    OCN_REQUIRE(n_tracers >= 0 && n_tracers <= 4, "ocn_compute_amd_diffusivities_range: n_tracers = %d outside 0..4", n_tracers);
    OCN_REQUIRE(n_tracers == 0 || (C_kappa && tracers && kappa_e), "ocn_compute_amd_diffusivities_range: null tracer arrays");
    for (int n = 0; n < n_tracers; ++n) OCN_REQUIRE(tracers[n] && kappa_e[n], "ocn_compute_amd_diffusivities_range: tracer %d is NULL", n);
    OCN_REQUIRE(i_first >= 0 && i_last <= grid->Nx + 1, "ocn_compute_amd_diffusivities_range: i range %d:%d outside 0:%d", i_first, i_last, grid->Nx + 1);
    OCN_REQUIRE((i_first >= 1 && i_last <= grid->Nx) || grid->Hx >= 2, "the halo columns 0 and Nx+1 need Hx >= 2");
    const int32_t ir[2] = {i_first, i_last};
    return strict_math(grid)
               ? ocn_strict::launch_amd_fused(grid, C_nu, u, v, w, nu_e, n_tracers, C_kappa, tracers, kappa_e, as_stream(stream), ir)
               : ocn_fast::launch_amd_fused(grid, C_nu, u, v, w, nu_e, n_tracers, C_kappa, tracers, kappa_e, as_stream(stream), ir);
}

int ocn_compute_amd_diffusivity(const ocn_grid *grid, double C_kappa, const double *u, const double *v, const double *w,
                                const double *c, double *kappa_e, void *stream)
{
    int st = validate_amd(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && c && kappa_e, "ocn_compute_amd_diffusivity: null field pointer");
    return strict_math(grid) ? ocn_strict::launch_amd_diffusivity(grid, C_kappa, u, v, w, c, kappa_e, as_stream(stream))
                                          : ocn_fast::launch_amd_diffusivity(grid, C_kappa, u, v, w, c, kappa_e, as_stream(stream));
}

int ocn_update_hydrostatic_pressure(const ocn_grid *grid, const ocn_model_terms *terms, double *pHY, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(terms != nullptr && pHY != nullptr, "ocn_update_hydrostatic_pressure: null argument");
    OCN_REQUIRE(terms->buoyancy != OCN_BUOYANCY_NONE, "ocn_update_hydrostatic_pressure: buoyancy is nothing");
    if (terms->buoyancy != OCN_BUOYANCY_SEAWATER_S) OCN_REQUIRE(terms->T != nullptr, "T (or b) tracer is NULL");
    if (terms->buoyancy == OCN_BUOYANCY_SEAWATER_TS || terms->buoyancy == OCN_BUOYANCY_SEAWATER_S) OCN_REQUIRE(terms->S != nullptr, "S tracer is NULL");
    OCN_REQUIRE((grid->tx == OCN_FLAT || grid->Hx >= 1) && (grid->ty == OCN_FLAT || grid->Hy >= 1) && (grid->tz == OCN_FLAT || grid->Hz >= 1),
                "halo >= 1 required");
    return launch_hydrostatic_pressure(grid, to_dev(*terms), pHY, as_stream(stream));
}

int ocn_update_hydrostatic_pressure_range(const ocn_grid *grid, const ocn_model_terms *terms, double *pHY, int32_t i_first, int32_t i_last,
                                          void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(terms != nullptr && pHY != nullptr, "ocn_update_hydrostatic_pressure_range: null argument");
    OCN_REQUIRE(terms->buoyancy != OCN_BUOYANCY_NONE, "ocn_update_hydrostatic_pressure_range: buoyancy is nothing");
    if (terms->buoyancy != OCN_BUOYANCY_SEAWATER_S) OCN_REQUIRE(terms->T != nullptr, "T (or b) tracer is NULL");
    if (terms->buoyancy == OCN_BUOYANCY_SEAWATER_TS || terms->buoyancy == OCN_BUOYANCY_SEAWATER_S) OCN_REQUIRE(terms->S != nullptr, "S tracer is NULL");
    OCN_REQUIRE(grid->Hx >= 1 && grid->Hy >= 1 && (grid->tz == OCN_FLAT || grid->Hz >= 1), "halo >= 1 required");
    OCN_REQUIRE(i_first >= 0 && i_last <= grid->Nx + 1, "ocn_update_hydrostatic_pressure_range: i range %d:%d outside 0:%d", i_first, i_last, grid->Nx + 1);
    const int32_t ir[2] = {i_first, i_last};
    return launch_hydrostatic_pressure(grid, to_dev(*terms), pHY, as_stream(stream), ir);
}

static int make_zbc_tuple(const ocn_grid *grid, const int32_t *locs, const ocn_field_bcs *const *bcs, int32_t n, ZBcTuple &z,
                          bool flux_only, bool &any)
{
    any = false;
    for (int f = 0; f < n; ++f) {
        z.bottom[f] = ZBc{OCN_BC_DEFAULT, 0.0, 0.0, nullptr};
        z.top[f] = ZBc{OCN_BC_DEFAULT, 0.0, 0.0, nullptr};
        if (!bcs || !bcs[f]) continue;
        const ocn_field_bcs &b = *bcs[f];
        OCN_REQUIRE(b.west.kind == OCN_BC_DEFAULT && b.east.kind == OCN_BC_DEFAULT && b.south.kind == OCN_BC_DEFAULT &&
                        b.north.kind == OCN_BC_DEFAULT,
                    "field %d: only bottom / top boundary conditions are supported (x, y are Periodic)", f);
        const ocn_bc *side[2] = {&b.bottom, &b.top};
        for (int sd = 0; sd < 2; ++sd) {
            const ocn_bc &c = *side[sd];
            OCN_REQUIRE(c.kind >= OCN_BC_DEFAULT && c.kind <= OCN_BC_OPEN, "field %d: unknown boundary condition kind %d", f, c.kind);
            if (c.kind == OCN_BC_DEFAULT) continue;
            OCN_REQUIRE(grid->tz == OCN_BOUNDED, "bottom / top boundary conditions need a Bounded z (topology %d)", grid->tz);
            OCN_REQUIRE(((locs[f] & 4) != 0) == (c.kind == OCN_BC_OPEN),
                        "field %d: the wall-normal velocity takes an Open condition (its value on the boundary face), the other fields Flux / Value / Gradient", f);
            OCN_REQUIRE(!c.values || grid->tx == OCN_PERIODIC, "array boundary conditions are not supported on a partitioned grid");
            if (flux_only && c.kind != OCN_BC_FLUX) continue;
            ZBc &d = sd ? z.top[f] : z.bottom[f];
            d.kind = c.kind; d.value = c.value; d.coeff = c.coeff; d.values = c.values;
            any = true;
        }
    }
    return OCN_SUCCESS;
}

static int flux_side(const ocn_grid *grid, const ocn_field_bcs *b, const char *name, ZBc &bottom, ZBc &top)
{
    bottom = ZBc{OCN_BC_DEFAULT, 0.0, 0.0, nullptr};
    top = ZBc{OCN_BC_DEFAULT, 0.0, 0.0, nullptr};
    if (!b) return OCN_SUCCESS;
    // (Value / Gradient / Open conditions on x / y walls live in the halo fills; FLUXES through them are added by ocn_apply_flux_bcs on the
    //  unfused path only)
    OCN_REQUIRE(b->west.kind != OCN_BC_FLUX && b->east.kind != OCN_BC_FLUX && b->south.kind != OCN_BC_FLUX && b->north.kind != OCN_BC_FLUX,
                "%s: the fused stage boundaries take flux conditions at the bottom / top only", name);
    const ocn_bc *side[2] = {&b->bottom, &b->top};
    for (int sd = 0; sd < 2; ++sd) {
        const ocn_bc &c = *side[sd];
        OCN_REQUIRE(c.kind >= OCN_BC_DEFAULT && c.kind <= OCN_BC_OPEN, "%s: unknown boundary condition kind %d", name, c.kind);
        if (c.kind != OCN_BC_FLUX) continue;
        OCN_REQUIRE(grid->tz == OCN_BOUNDED, "bottom / top boundary conditions need a Bounded z (topology %d)", grid->tz);
        OCN_REQUIRE(!c.values || grid->tx == OCN_PERIODIC, "array boundary conditions need a Periodic x (not partitioned, no x walls)");
        ZBc &d = sd ? top : bottom;
        d.kind = c.kind; d.value = c.value; d.coeff = c.coeff; d.values = c.values;
    }
    return OCN_SUCCESS;
}

int ocn_compute_momentum_tendencies_terms_rk3(const ocn_grid *grid, const ocn_model_terms *terms, const ocn_field_bcs *bcs_u,
                                              const ocn_field_bcs *bcs_v, const double *u, const double *v, const double *w,
                                              double *Gu, double *Gv, double *Gw, const double *Gmu, const double *Gmv,
                                              const double *Gmw, double *u_out, double *v_out, double *w_out, double dt,
                                              double gamma, double zeta, int32_t has_zeta, const int32_t *range, void *stream)
{
    int st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(xy_periodic(grid) || !range, "ocn_compute_momentum_tendencies_terms_rk3: ranges need Periodic x and y");
    OCN_REQUIRE(u && v && w && Gu && Gv && Gw && u_out && v_out && w_out, "ocn_compute_momentum_tendencies_terms_rk3: null field pointer");
    OCN_REQUIRE(!has_zeta || (Gmu && Gmv && Gmw), "ocn_compute_momentum_tendencies_terms_rk3: G⁻ pointers are required when has_zeta != 0");
    OCN_REQUIRE(u_out != u && v_out != v && w_out != w, "ocn_compute_momentum_tendencies_terms_rk3: outputs must not alias the inputs");
    MomentumFinal mf{};
    st = flux_side(grid, bcs_u, "u", mf.bottom[0], mf.top[0]);
    if (st != OCN_SUCCESS) return st;
    st = flux_side(grid, bcs_v, "v", mf.bottom[1], mf.top[1]);
    if (st != OCN_SUCCESS) return st;
    mf.sub[0] = SubstepDev{Gmu, u_out};
    mf.sub[1] = SubstepDev{Gmv, v_out};
    mf.sub[2] = SubstepDev{Gmw, w_out};
    mf.sc = SubstepCoef{dt, gamma, zeta, 1, has_zeta ? 1 : 0};
    const bool strict = strict_math(grid);
    hipStream_t s = as_stream(stream);
    TermsDev t = to_dev(*terms);
    if (!xy_periodic(grid)) {
        // walls in x / y: advection (box + frames), then the finishing pass in the reference's order with the boundary fluxes and the substep --
        // inside the tiled kernel on the interior box, as per-cell kernels on the frames (general.hip)
        st = launch_advective_momentum(terms->advection, grid, u, v, w, Gu, Gv, Gw, nullptr, s);
        if (st != OCN_SUCCESS) return st;
        return strict ? ocn_strict::launch_momentum_extra_general(grid, t, u, v, w, Gu, Gv, Gw, nullptr, s, &mf)
                      : ocn_fast::launch_momentum_extra_general(grid, t, u, v, w, Gu, Gv, Gw, nullptr, s, &mf);
    }
    static const bool extra_first = !(std::getenv("OCN_EXTRA_FIRST") && std::getenv("OCN_EXTRA_FIRST")[0] == '0');
    if (!strict && extra_first && terms->advection != OCN_ADVECTION_CENTERED2) {
        // Fast math: the finishing pass runs FIRST and leaves the non-advective terms (and the u / v boundary fluxes) in G; the WENO
        // launch -- VALU-bound, with HBM bandwidth to spare -- adds its advective part, stores G and does the substep.  The HBM-bound
        // finishing pass then moves 64 instead of 136 B / cell (no G read-modify-write, no G^-, no second storage); the sum is the
        // reference's with the advective term added last instead of first (a reassociation: ~1 ulp of max|G|; the strict build keeps
        // the reference's order).
        MomentumFinal pre = mf;
        pre.pre = 1;
        pre.sc.on = 0;
        st = ocn_fast::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, s, &pre);
        if (st != OCN_SUCCESS) return st;
        FuseArgs fz{};
        fz.Gm[0] = Gmu; fz.Gm[1] = Gmv; fz.Gm[2] = Gmw;
        fz.Uo[0] = u_out; fz.Uo[1] = v_out; fz.Uo[2] = w_out;
        fz.dt = dt; fz.gamma = gamma; fz.zeta = zeta; fz.on = 1; fz.has_zeta = has_zeta ? 1 : 0;
        fz.acc = 1;
        return terms->advection == OCN_ADVECTION_WENO5 ? ocn_fast::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, &fz, s)
                                                       : ocn_fast_up::launch_momentum_tendencies(grid, u, v, w, Gu, Gv, Gw, range, &fz, s);
    }
    st = launch_advective_momentum(terms->advection, grid, u, v, w, Gu, Gv, Gw, range, s);
    if (st != OCN_SUCCESS) return st;
    return strict ? ocn_strict::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, s, &mf)
                  : ocn_fast::launch_momentum_extra(grid, t, u, v, w, Gu, Gv, Gw, range, s, &mf);
}

int ocn_compute_tracer_tendency_terms_rk3(const ocn_grid *grid, const ocn_model_terms *terms, double kappa, const double *kappa_e,
                                          const ocn_field_bcs *bcs_c, const double *u, const double *v, const double *w,
                                          const double *c, double *Gc, const double *Gmc, double *c_out, double dt, double gamma,
                                          double zeta, int32_t has_zeta, const int32_t *range, void *stream)
{
    int st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(xy_periodic(grid) || !range, "ocn_compute_tracer_tendency_terms_rk3: ranges need Periodic x and y");
    OCN_REQUIRE(terms->advection != OCN_ADVECTION_CENTERED2, "ocn_compute_tracer_tendency_terms_rk3: advection must be WENO5 or UpwindBiased5");
    OCN_REQUIRE(u && v && w && c && Gc && c_out, "ocn_compute_tracer_tendency_terms_rk3: null field pointer");
    OCN_REQUIRE(!has_zeta || Gmc, "ocn_compute_tracer_tendency_terms_rk3: G⁻ is required when has_zeta != 0");
    OCN_REQUIRE(c_out != c, "ocn_compute_tracer_tendency_terms_rk3: the output must not alias the input");
    OCN_REQUIRE(!kappa_e || terms->closure == 2, "kappa_e is only meaningful with closure == 2");
    TracerFuse tf{};
    tf.diffusion = terms->closure != 0;
    tf.kappa = kappa;
    tf.kappa_e = kappa_e;
    st = flux_side(grid, bcs_c, "tracer", tf.bottom, tf.top);
    if (st != OCN_SUCCESS) return st;
    tf.sub = SubstepDev{Gmc, c_out};
    tf.sc = SubstepCoef{dt, gamma, zeta, 1, has_zeta ? 1 : 0};
    return launch_advective_tracer(terms->advection, grid, u, v, w, c, Gc, range, as_stream(stream), &tf);
}

int ocn_compute_tracer_pair_tendency_terms_rk3(const ocn_grid *grid, const ocn_model_terms *terms, const double *kappa, const double *const *kappa_e,
                                               const ocn_field_bcs *const *bcs_c, const double *u, const double *v, const double *w,
                                               const double *const *c, double *const *Gc, const double *const *Gc_previous, double *const *c_out,
                                               double dt, double gamma, double zeta, int32_t has_zeta, const int32_t *range,
                                               int32_t *launched, void *stream)
{
    int st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(launched, "ocn_compute_tracer_pair_tendency_terms_rk3: launched is NULL");
    *launched = 0;
    if (!xy_periodic(grid)) return OCN_SUCCESS;  // walls in x / y: one tracer per launch (ocn_compute_tracer_tendency_terms_rk3)
    OCN_REQUIRE(terms->advection != OCN_ADVECTION_CENTERED2, "ocn_compute_tracer_pair_tendency_terms_rk3: advection must be WENO5 or UpwindBiased5");
    OCN_REQUIRE(u && v && w && c && Gc && c_out && kappa, "ocn_compute_tracer_pair_tendency_terms_rk3: null pointer");
    TracerFuse tf[2];
    for (int t = 0; t < 2; ++t) {
        OCN_REQUIRE(c[t] && Gc[t] && c_out[t] && c_out[t] != c[t], "ocn_compute_tracer_pair_tendency_terms_rk3: null or aliased field pointer");
        OCN_REQUIRE(!has_zeta || (Gc_previous && Gc_previous[t]), "ocn_compute_tracer_pair_tendency_terms_rk3: G⁻ is required when has_zeta != 0");
        OCN_REQUIRE(!(kappa_e && kappa_e[t]) || terms->closure == 2, "kappa_e is only meaningful with closure == 2");
        tf[t] = TracerFuse{};
        tf[t].diffusion = terms->closure != 0;
        tf[t].kappa = kappa[t];
        tf[t].kappa_e = kappa_e ? kappa_e[t] : nullptr;
        st = flux_side(grid, bcs_c ? bcs_c[t] : nullptr, "tracer", tf[t].bottom, tf[t].top);
        if (st != OCN_SUCCESS) return st;
        tf[t].sub = SubstepDev{Gc_previous ? Gc_previous[t] : nullptr, c_out[t]};
        tf[t].sc = SubstepCoef{dt, gamma, zeta, 1, has_zeta ? 1 : 0};
    }
    int did = 0;
    hipStream_t s = as_stream(stream);
    const bool strict = strict_math(grid);
    if (terms->advection == OCN_ADVECTION_WENO5)
        st = strict ? ocn_strict::launch_tracer_pair_tendency(grid, u, v, w, c, Gc, range, s, tf, &did)
                    : ocn_fast::launch_tracer_pair_tendency(grid, u, v, w, c, Gc, range, s, tf, &did);
    else
        st = strict ? ocn_strict_up::launch_tracer_pair_tendency(grid, u, v, w, c, Gc, range, s, tf, &did)
                    : ocn_fast_up::launch_tracer_pair_tendency(grid, u, v, w, c, Gc, range, s, tf, &did);
    *launched = did;
    return st;
}

int ocn_split_explicit_substeps_blocked(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                        double column_depth, double *eta, double *U, double *V, double *eta_filtered, double *U_filtered,
                                        double *V_filtered, const double *GU, const double *GV, double *work, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_split_explicit_substeps_blocked");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE_PERIODIC_X("ocn_split_explicit_substeps_blocked");
    OCN_REQUIRE(n >= 1 && weights, "ocn_split_explicit_substeps_blocked: needs n >= 1 averaging weights (host array)");
    OCN_REQUIRE(eta && U && V && eta_filtered && U_filtered && V_filtered && GU && GV && work, "ocn_split_explicit_substeps_blocked: null pointer");
    return launch_split_explicit_substeps_blocked(grid, n, weights, dtau, gravitational_acceleration, column_depth, eta, U, V, eta_filtered,
                                                  U_filtered, V_filtered, GU, GV, work, as_stream(stream));
}

int ocn_implicit_free_surface_rhs(const ocn_grid *grid, const double *u, const double *v, const double *eta, double gravitational_acceleration,
                                  double dt, double *Qu, double *Qv, double *rhs, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_implicit_free_surface_rhs");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE_PERIODIC_X("ocn_implicit_free_surface_rhs");
    OCN_REQUIRE(u && v && eta && Qu && Qv && rhs, "ocn_implicit_free_surface_rhs: null pointer");
    OCN_REQUIRE(dt > 0, "ocn_implicit_free_surface_rhs: dt must be positive");
    return launch_implicit_free_surface_rhs(grid, u, v, eta, gravitational_acceleration, dt, Qu, Qv, rhs, as_stream(stream));
}
int ocn_barotropic_pressure_correction(const ocn_grid *grid, double *u, double *v, const double *eta, double gravitational_acceleration,
                                       double dt, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_barotropic_pressure_correction");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && eta, "ocn_barotropic_pressure_correction: null pointer");
    return launch_barotropic_pressure_correction(grid, u, v, eta, gravitational_acceleration, dt, as_stream(stream));
}

int ocn_split_explicit_substeps_ab3(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                    double column_depth, const double *coefficients, double *eta, double *U, double *V, double *eta_filtered,
                                    double *U_filtered, double *V_filtered, const double *GU, const double *GV, double *work, void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_split_explicit_substeps_ab3");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE_PERIODIC_X("ocn_split_explicit_substeps_ab3");
    OCN_REQUIRE(n >= 1 && weights && coefficients, "ocn_split_explicit_substeps_ab3: needs n >= 1 weights and the 7 AB3 coefficients (host arrays)");
    OCN_REQUIRE(eta && U && V && eta_filtered && U_filtered && V_filtered && GU && GV && work, "ocn_split_explicit_substeps_ab3: null pointer");
    return launch_split_explicit_substeps_ab3(grid, n, weights, dtau, gravitational_acceleration, column_depth, coefficients, eta, U, V,
                                              eta_filtered, U_filtered, V_filtered, GU, GV, work, as_stream(stream));
}

static int validate_hydrostatic_slab(const ocn_grid *grid, const char *who)
{
    int st = validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE((grid->tx == OCN_PERIODIC || grid->tx == OCN_FULLY_CONNECTED) && grid->ty == OCN_PERIODIC && grid->tz == OCN_BOUNDED,
                "%s: (Periodic | FullyConnected, Periodic, Bounded) grids", who);
    return OCN_SUCCESS;
}
int ocn_split_explicit_dist_begin(const ocn_grid *grid, int32_t n, const double *eta, const double *U, const double *V, const double *GU,
                                  const double *GV, double *work, double *send_west, double *send_east, void *stream)
{
    int st = validate_hydrostatic_slab(grid, "ocn_split_explicit_dist_begin");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(n >= 1 && n <= grid->Nx, "ocn_split_explicit_dist_begin: the number of substeps (%d) must not exceed the slab's Nx (%d): the extended halo "
                "would reach past the neighbouring rank", n, grid->Nx);
    OCN_REQUIRE(eta && U && V && GU && GV && work && send_west && send_east, "ocn_split_explicit_dist_begin: null pointer");
    return launch_split_explicit_dist_begin(grid, n, eta, U, V, GU, GV, work, send_west, send_east, as_stream(stream));
}
int ocn_split_explicit_dist_run(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                double column_depth, double *eta, double *U, double *V, double *work, const double *recv_west,
                                const double *recv_east, void *stream)
{
    int st = validate_hydrostatic_slab(grid, "ocn_split_explicit_dist_run");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(n >= 1 && n <= grid->Nx && weights, "ocn_split_explicit_dist_run: needs 1 <= n <= Nx averaging weights (host array)");
    OCN_REQUIRE(eta && U && V && work && recv_west && recv_east, "ocn_split_explicit_dist_run: null pointer");
    return launch_split_explicit_dist_run(grid, n, n, weights, dtau, gravitational_acceleration, column_depth, eta, U, V, work, recv_west,
                                          recv_east, as_stream(stream));
}

int ocn_hydrostatic_momentum_ab2_step(const ocn_grid *grid, const ocn_model_terms *terms, const ocn_field_bcs *bcs_u,
                                      const ocn_field_bcs *bcs_v, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                                      const double *Gu_previous, const double *Gv_previous, double *u_out, double *v_out, double dt,
                                      double chi, int32_t euler, const double *eta, double gravitational_acceleration, double *GU,
                                      double *GV, double *U_star, double *V_star, void *stream)
{
    int st = validate_hydrostatic_slab(grid, "ocn_hydrostatic_momentum_ab2_step");
    if (st != OCN_SUCCESS) return st;
    st = validate_terms(grid, terms);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->Hz >= 1, "ocn_hydrostatic_momentum_ab2_step: needs a z halo");
    OCN_REQUIRE(terms->closure != 2, "ocn_hydrostatic_momentum_ab2_step: eddy-viscosity closures are not supported");
    OCN_REQUIRE(u && v && w && Gu && Gv && u_out && v_out, "ocn_hydrostatic_momentum_ab2_step: null field pointer");
    OCN_REQUIRE(euler || (Gu_previous && Gv_previous), "ocn_hydrostatic_momentum_ab2_step: G_previous is required unless euler != 0");
    OCN_REQUIRE(u_out != u && v_out != v, "ocn_hydrostatic_momentum_ab2_step: the outputs must not alias the inputs");
    OCN_REQUIRE((GU && GV && U_star && V_star) || (!GU && !GV && !U_star && !V_star),
                "ocn_hydrostatic_momentum_ab2_step: GU, GV, U_star, V_star go together (all or none)");
    MomentumFinal mf{};
    st = flux_side(grid, bcs_u, "u", mf.bottom[0], mf.top[0]);
    if (st != OCN_SUCCESS) return st;
    st = flux_side(grid, bcs_v, "v", mf.bottom[1], mf.top[1]);
    if (st != OCN_SUCCESS) return st;
    mf.sub[0] = SubstepDev{Gu_previous, u_out};
    mf.sub[1] = SubstepDev{Gv_previous, v_out};
    if (euler) chi = -0.5;
    mf.sc = SubstepCoef{dt, 1.5 + chi, -(0.5 + chi), 1, euler ? 0 : 1};
    ocn::HydroFuse hf{eta, gravitational_acceleration, GU, GV, U_star, V_star};
    TermsDev t = to_dev(*terms);
    return strict_math(grid) ? ocn_strict::launch_hydrostatic_momentum(grid, t, u, v, w, Gu, Gv, mf, hf, as_stream(stream))
                                          : ocn_fast::launch_hydrostatic_momentum(grid, t, u, v, w, Gu, Gv, mf, hf, as_stream(stream));
}

int ocn_barotropic_corrector_and_w(const ocn_grid *grid, const double *u_star, const double *v_star, double *u, double *v, double *w,
                                   const double *U, const double *V, const double *U_star, const double *V_star, double column_depth,
                                   void *stream)
{
    int st = validate_hydrostatic(grid, "ocn_barotropic_corrector_and_w");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u_star && v_star && u && v && w, "ocn_barotropic_corrector_and_w: null field pointer");
    OCN_REQUIRE(u != u_star && v != v_star, "ocn_barotropic_corrector_and_w: the outputs must not alias the inputs");
    OCN_REQUIRE((U && V && U_star && V_star) || (!U && !V), "ocn_barotropic_corrector_and_w: U, V, U_star, V_star go together");
    return launch_barotropic_correct_w(grid, u_star, v_star, u, v, w, U, V, U_star, V_star, column_depth, as_stream(stream));
}

int ocn_fill_halo_regions_bcs(const ocn_grid *grid, double *const *fields, const int32_t *locs,
                              const ocn_field_bcs *const *bcs, int32_t n, int32_t fill_boundary_normal_velocities,
                              void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    FieldTuple ft;
    st = make_field_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    if (!xy_periodic(grid)) {  // every side may carry a condition (fill_halo_regions_value_gradient.jl for west / east / south / north too)
        SideBcTuple sb{};
        bool any = false;
        const int T[3] = {grid->tx, grid->ty, grid->tz};
        for (int f = 0; f < n; ++f) {
            if (!bcs || !bcs[f]) continue;
            const ocn_bc *side[6] = {&bcs[f]->west, &bcs[f]->east, &bcs[f]->south, &bcs[f]->north, &bcs[f]->bottom, &bcs[f]->top};
            for (int q = 0; q < 6; ++q) {
                const ocn_bc &c = *side[q];
                OCN_REQUIRE(c.kind >= OCN_BC_DEFAULT && c.kind <= OCN_BC_OPEN, "field %d: unknown boundary condition kind %d", f, c.kind);
                if (c.kind == OCN_BC_DEFAULT) continue;
                if (q < 2 && partitioned_x(grid) && !(q == 0 ? x_wall_west(*grid) : x_wall_east(*grid))) continue;  // another slab's wall
                OCN_REQUIRE(side_has_wall(grid, q), "field %d: a boundary condition on side %d needs a Bounded direction (topology %d)", f, q, T[q / 2]);
                OCN_REQUIRE((((locs[f] >> (q / 2)) & 1) != 0) == (c.kind == OCN_BC_OPEN),
                            "field %d: the wall-normal velocity takes an Open condition (its value on the boundary face), the other fields Flux / Value / Gradient", f);
                sb.side[q][f] = ZBc{c.kind, c.value, c.coeff, c.values};
                any = true;
            }
        }
        return launch_fill_halos_general(grid, ft, fill_boundary_normal_velocities, as_stream(stream), any ? &sb : nullptr);
    }
    ZBcTuple z;
    bool any;
    st = make_zbc_tuple(grid, locs, bcs, n, z, false, any);
    if (st != OCN_SUCCESS) return st;
    return launch_fill_halos(grid, ft, fill_boundary_normal_velocities, -1, as_stream(stream), any ? &z : nullptr);
}

int ocn_apply_flux_bcs(const ocn_grid *grid, double *const *G, const double *const *fields, const int32_t *locs,
                       const ocn_field_bcs *const *bcs, int32_t n, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    FieldTuple gt, ft;
    st = make_field_tuple(grid, G, locs, n, gt);
    if (st != OCN_SUCCESS) return st;
    st = make_field_tuple(grid, const_cast<double *const *>(fields), locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    if (!xy_periodic(grid)) {  // apply_x_bcs!, apply_y_bcs!, then apply_z_bcs! (compute_nonhydrostatic_tendencies.jl:204-213)
        SideBcTuple sb{};
        ZBcTuple z{};
        bool any_xy = false, any_z = false;
        const int T[3] = {grid->tx, grid->ty, grid->tz};
        for (int f = 0; f < n; ++f) {
            if (!bcs || !bcs[f]) continue;
            const ocn_bc *side[6] = {&bcs[f]->west, &bcs[f]->east, &bcs[f]->south, &bcs[f]->north, &bcs[f]->bottom, &bcs[f]->top};
            for (int q = 0; q < 6; ++q) {
                const ocn_bc &c = *side[q];
                OCN_REQUIRE(c.kind >= OCN_BC_DEFAULT && c.kind <= OCN_BC_OPEN, "field %d: unknown boundary condition kind %d", f, c.kind);
                if (c.kind != OCN_BC_FLUX) continue;
                if (q < 2 && partitioned_x(grid) && !(q == 0 ? x_wall_west(*grid) : x_wall_east(*grid))) continue;  // another slab's wall
                OCN_REQUIRE(side_has_wall(grid, q), "field %d: a boundary condition on side %d needs a Bounded direction (topology %d)", f, q, T[q / 2]);
                OCN_REQUIRE(!((locs[f] >> (q / 2)) & 1), "field %d: the wall-normal velocity keeps its impenetrable condition", f);
                if (q < 4) {
                    sb.side[q][f] = ZBc{c.kind, c.value, c.coeff, c.values};
                    any_xy = true;
                } else {
                    (q == 4 ? z.bottom[f] : z.top[f]) = ZBc{c.kind, c.value, c.coeff, c.values};
                    any_z = true;
                }
            }
        }
        if (any_xy) {
            st = launch_apply_flux_bcs_lateral(grid, gt, ft, sb, as_stream(stream));
            if (st != OCN_SUCCESS) return st;
        }
        return any_z ? launch_apply_flux_bcs(grid, gt, ft, z, as_stream(stream)) : OCN_SUCCESS;
    }
    ZBcTuple z;
    bool any;
    st = make_zbc_tuple(grid, locs, bcs, n, z, true, any);
    if (st != OCN_SUCCESS) return st;
    if (!any) return OCN_SUCCESS;
    return launch_apply_flux_bcs(grid, gt, ft, z, as_stream(stream));
}

int ocn_compute_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                                double *Gc, const int32_t *range, void *stream)
{
    int st = validate_weno(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && c && Gc, "ocn_compute_tracer_tendency: null field pointer");
    if (!xy_periodic(grid))
        return strict_math(grid) ? ocn_strict::launch_tracer_tendency_general(grid, 0, u, v, w, c, Gc, range, as_stream(stream))
                                              : ocn_fast::launch_tracer_tendency_general(grid, 0, u, v, w, c, Gc, range, as_stream(stream));
    if (strict_math(grid)) return ocn_strict::launch_tracer_tendency(grid, u, v, w, c, Gc, range, as_stream(stream));
    return ocn_fast::launch_tracer_tendency(grid, u, v, w, c, Gc, range, as_stream(stream));
}

static int make_step_tuple(int32_t n, double *const *U, const double *const *Gn, double *const *Gm, const int32_t *locs,
                           bool needU, StepTuple &t)
{
    OCN_REQUIRE(n >= 1 && n <= MAX_TUPLE, "number of fields %d outside 1..%d", n, MAX_TUPLE);
    OCN_REQUIRE(Gn && Gm && locs && (U || !needU), "null tuple pointer");
    t.n = n;
    for (int f = 0; f < n; ++f) {
        OCN_REQUIRE(Gn[f] && Gm[f] && (!needU || U[f]), "field %d: null pointer", f);
        t.U[f] = needU ? U[f] : nullptr;
        t.Gn[f] = Gn[f];
        t.Gm[f] = Gm[f];
        t.loc[f] = locs[f];
    }
    return OCN_SUCCESS;
}

int ocn_rk3_substep(const ocn_grid *grid, int32_t n, double *const *U, const double *const *Gn, const double *const *Gm,
                    const int32_t *locs, double dt, double gamma, double zeta, int32_t has_zeta, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    StepTuple t;
    st = make_step_tuple(n, U, Gn, const_cast<double *const *>(reinterpret_cast<const double *const *>(Gm)), locs, true, t);
    if (st != OCN_SUCCESS) return st;
    return launch_stepper(grid, t, has_zeta ? 1 : 0, dt, gamma, zeta, as_stream(stream));
}

int ocn_split_rk3_substep(const ocn_grid *grid, int32_t n, double *const *U, const double *const *G, const double *const *Psi,
                          const int32_t *locs, double dt, double gamma, double zeta, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    StepTuple t;
    st = make_step_tuple(n, U, G, const_cast<double *const *>(reinterpret_cast<const double *const *>(Psi)), locs, true, t);
    if (st != OCN_SUCCESS) return st;
    for (int f = 0; f < n; ++f) OCN_REQUIRE(Psi[f] != U[f], "ocn_split_rk3_substep: field %d: Psi must not alias U", f);
    return launch_stepper(grid, t, 4, dt, gamma, zeta, as_stream(stream));
}

int ocn_ab2_step(const ocn_grid *grid, int32_t n, double *const *U, const double *const *Gn, const double *const *Gm,
                 const int32_t *locs, double dt, double chi, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    StepTuple t;
    st = make_step_tuple(n, U, Gn, const_cast<double *const *>(reinterpret_cast<const double *const *>(Gm)), locs, true, t);
    if (st != OCN_SUCCESS) return st;
    const double not_euler = (chi != -0.5) ? 1.0 : 0.0;
    return launch_stepper(grid, t, 2, dt, chi, not_euler, as_stream(stream));
}

int ocn_cache_previous_tendencies(const ocn_grid *grid, int32_t n, double *const *Gm, const double *const *Gn,
                                  const int32_t *locs, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    StepTuple t;
    st = make_step_tuple(n, nullptr, Gn, Gm, locs, false, t);
    if (st != OCN_SUCCESS) return st;
    return launch_stepper(grid, t, 3, 0.0, 0.0, 0.0, as_stream(stream));
}

int ocn_pressure_correct_velocities(const ocn_grid *grid, double *u, double *v, double *w, const double *p, double dt, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && p, "ocn_pressure_correct_velocities: null field pointer");
    OCN_REQUIRE((grid->tx == OCN_FLAT || grid->Hx >= 1) && (grid->ty == OCN_FLAT || grid->Hy >= 1) && (grid->tz == OCN_FLAT || grid->Hz >= 1),
                "pressure correction needs halo >= 1");
    return launch_pressure_correct(grid, u, v, w, p, dt, as_stream(stream));
}

int ocn_divergence(const ocn_grid *grid, const double *u, const double *v, const double *w, double *div, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(u && v && w && div, "ocn_divergence: null pointer");
    return launch_source_term(grid, u, v, w, 1.0, 0, div, grid->Nx, (long long)grid->Nx * grid->Ny, as_stream(stream));
}

int ocn_batched_tridiagonal_solve_z(int32_t Nx, int32_t Ny, int32_t Nz, const double *a, const double *b, const double *c,
                                    const double *f, double *t, double *phi, void *stream)
{
    OCN_REQUIRE(Nx >= 1 && Ny >= 1 && Nz >= 1, "bad sizes (%d, %d, %d)", Nx, Ny, Nz);
    OCN_REQUIRE(a && b && c && f && t && phi, "ocn_batched_tridiagonal_solve_z: null pointer");
    // (where the pivot is not definitely diagonally dominant the caller's ϕ stays: batched_tridiagonal_solver.jl:224-228)
    return launch_tridiag_z(Nx, Ny, Nz, a, b, c, f, t, phi, as_stream(stream), /*keep_storage=*/1);
}

int ocn_halo_pack_x(const ocn_grid *grid, const double *field, int32_t loc, double *send_west, double *send_east, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(field && send_west && send_east, "ocn_halo_pack_x: null pointer");
    return launch_halo_pack_x(grid, field, loc, send_west, send_east, 0, as_stream(stream));
}
int ocn_halo_unpack_x(const ocn_grid *grid, double *field, int32_t loc, const double *recv_west, const double *recv_east, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(field && recv_west && recv_east, "ocn_halo_unpack_x: null pointer");
    return launch_halo_pack_x(grid, field, loc, const_cast<double *>(recv_west), const_cast<double *>(recv_east), 1, as_stream(stream));
}

int ocn_halo_plane_x(const ocn_grid *grid, double *field, int32_t loc, int32_t which, double *buffer, int32_t unpack, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(field && buffer, "ocn_halo_plane_x: null pointer");
    OCN_REQUIRE(which == 0 || which == 1, "ocn_halo_plane_x: which must be 0 (west) or 1 (east)");
    OCN_REQUIRE(grid->Hx >= 1, "ocn_halo_plane_x: needs an x halo");
    return launch_halo_plane_x(grid, field, loc, which, buffer, unpack ? 1 : 0, as_stream(stream));
}
static int validate_pressure_planes(const ocn_grid *grid, const char *who)
{
    int st = validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->ty == OCN_PERIODIC && grid->tz == OCN_PERIODIC, "%s: y and z must be Periodic", who);
    OCN_REQUIRE(grid->Hx >= 1 && grid->Nx >= grid->Hx + 1, "%s: needs nx >= Hx + 1 (got nx = %d, Hx = %d)", who, grid->Nx, grid->Hx);
    return OCN_SUCCESS;
}
int ocn_halo_pack_pressure(const ocn_grid *grid, const double *p, const double *u, double dt_correct, double *send_west,
                           double *send_east, void *stream)
{
    int st = validate_pressure_planes(grid, "ocn_halo_pack_pressure");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(p && u && send_west && send_east, "ocn_halo_pack_pressure: null pointer");
    if (strict_math(grid))
        return ocn_strict::launch_pressure_planes(grid, const_cast<double *>(p), const_cast<double *>(u), dt_correct, send_west, send_east, 0, as_stream(stream));
    return ocn_fast::launch_pressure_planes(grid, const_cast<double *>(p), const_cast<double *>(u), dt_correct, send_west, send_east, 0, as_stream(stream));
}
int ocn_halo_unpack_pressure(const ocn_grid *grid, double *p, double *u, const double *recv_west, const double *recv_east, void *stream)
{
    int st = validate_pressure_planes(grid, "ocn_halo_unpack_pressure");
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(p && u && recv_west && recv_east, "ocn_halo_unpack_pressure: null pointer");
    return ocn_strict::launch_pressure_planes(grid, p, u, 0.0, const_cast<double *>(recv_west), const_cast<double *>(recv_east), 1, as_stream(stream));
}
int ocn_halo_pack_x_fields(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, double *send_west,
                           double *send_east, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(send_west && send_east, "ocn_halo_pack_x_fields: null buffer");
    FieldTuple ft;
    st = make_field_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    return launch_halo_pack_x_fields(grid, ft, send_west, send_east, 0, as_stream(stream));
}
int ocn_halo_unpack_x_fields(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, const double *recv_west,
                             const double *recv_east, void *stream)
{
    int st = validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(recv_west && recv_east, "ocn_halo_unpack_x_fields: null buffer");
    FieldTuple ft;
    st = make_field_tuple(grid, fields, locs, n, ft);
    if (st != OCN_SUCCESS) return st;
    return launch_halo_pack_x_fields(grid, ft, const_cast<double *>(recv_west), const_cast<double *>(recv_east), 1, as_stream(stream));
}

static int check_transpose(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const void *a, const void *b)
{
    OCN_REQUIRE(nx >= 1 && Ny >= 1 && Nz >= 1 && R >= 1, "bad transpose sizes");
    OCN_REQUIRE(Ny % R == 0, "Ny = %d must be divisible by the number of ranks %d", Ny, R);
    OCN_REQUIRE(a && b, "null transpose buffer");
    return OCN_SUCCESS;
}
int ocn_transpose_pack_y_to_x(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *yfield, double *send, void *stream)
{
    int st = check_transpose(nx, Ny, Nz, R, yfield, send);
    if (st != OCN_SUCCESS) return st;
    return launch_transpose(0, nx, Ny, Nz, R, yfield, send, as_stream(stream));
}
int ocn_transpose_unpack_x_from_y(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *recv, double *xfield, void *stream)
{
    int st = check_transpose(nx, Ny, Nz, R, recv, xfield);
    if (st != OCN_SUCCESS) return st;
    return launch_transpose(1, nx, Ny, Nz, R, recv, xfield, as_stream(stream));
}
int ocn_transpose_pack_x_to_y(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *xfield, double *send, void *stream)
{
    int st = check_transpose(nx, Ny, Nz, R, xfield, send);
    if (st != OCN_SUCCESS) return st;
    return launch_transpose(2, nx, Ny, Nz, R, xfield, send, as_stream(stream));
}
int ocn_transpose_unpack_y_from_x(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *recv, double *yfield, void *stream)
{
    int st = check_transpose(nx, Ny, Nz, R, recv, yfield);
    if (st != OCN_SUCCESS) return st;
    return launch_transpose(3, nx, Ny, Nz, R, recv, yfield, as_stream(stream));
}

}  // extern "C"
