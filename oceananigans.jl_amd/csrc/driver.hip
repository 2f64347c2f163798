// driver.hip -- time_step!(model::NonhydrostaticModel{<:RungeKutta3TimeStepper}, Δt) as ONE call of the C ABI
// (src/TimeSteppers/runge_kutta_3.jl:77-151), for the model of the north-star benchmark: WENO5 advection, no tracers, no
// extra terms, (Periodic, Periodic, Periodic | Bounded | Flat), one GPU.
//
// Why it exists: the per-kernel entry points are what a Julia backend binds one by one, but the stage-boundary fusion that
// makes the step fast (tendencies + the next substep in one launch writing a SECOND set of velocity
// arrays, G^n / G^- swapped instead of copied, the last tendency launch of a step deferred into the next step, the pressure
// correction folded into the tendency loads on all-periodic grids) needs arrays whose roles alternate -- which a host that
// cannot swap array storage cannot express.  This handle owns the second set and the tendency buffers and does the
// alternation itself; the caller's arrays hold the velocities whenever it asks (ocn_rk3_driver_flush).  Host code only: every
// device operation is one of the public entry points, in the order the Python host (models.py) issues them, so the two are
// bit-identical (tests/test_gpu_model.py::test_c_driver_equals_host_orchestration).
#include <cstring>

#include "ocn_internal.h"

using ocn::GridDev;
using ocn::Lay;

struct ocn_rk3_driver {
    ocn_grid grid{};
    ocn_poisson_t solver = nullptr;
    bool owns_solver = true;
    double *user[3] = {nullptr, nullptr, nullptr};  // the caller's u, v, w parents
    double *own[3] = {nullptr, nullptr, nullptr};   // second set
    double *U[3] = {nullptr, nullptr, nullptr};     // where the velocities are now
    double *A[3] = {nullptr, nullptr, nullptr};     // where the next fused launch writes
    double *p = nullptr;                            // the caller's pressure parent
    double *Gn[3] = {nullptr, nullptr, nullptr}, *Gm[3] = {nullptr, nullptr, nullptr};
    size_t bytes[3] = {0, 0, 0};
    bool pending = false;         // the last compute_tendencies! of the previous step is still due
    bool started = false;         // iteration 0 done
    bool correct_on_load = false;
    long long iteration = 0;
};

namespace {
const int32_t LOCS[3] = {OCN_LOC_FCC, OCN_LOC_CFC, OCN_LOC_CCF};

int fill_velocities(ocn_rk3_driver *d, int fbnv, void *stream)
{
    return ocn_fill_halo_regions(&d->grid, d->U, LOCS, 3, fbnv, stream);
}

// update_state! + the next rk3_substep! in one launch, then the two velocity sets trade places
int fused_launch(ocn_rk3_driver *d, double dt, double gamma, double zeta, int has_zeta, const double *p_correct, double dt_correct,
                 void *stream)
{
    int st = ocn_compute_momentum_tendencies_rk3(&d->grid, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], d->Gm[0], d->Gm[1],
                                                 d->Gm[2], d->A[0], d->A[1], d->A[2], dt, gamma, zeta, has_zeta, p_correct, dt_correct,
                                                 nullptr, stream);
    if (st != OCN_SUCCESS) return st;
    for (int f = 0; f < 3; ++f) std::swap(d->U[f], d->A[f]);
    d->pending = false;
    return OCN_SUCCESS;
}

void swap_tendencies(ocn_rk3_driver *d)  // cache_previous_tendencies! (store_tendencies.jl:12-22) as a role swap
{
    for (int f = 0; f < 3; ++f) std::swap(d->Gn[f], d->Gm[f]);
}

// everything between two substeps (runge_kutta_3.jl:103-118)
int project_and_advance(ocn_rk3_driver *d, double dt, double stage_dt, double gamma_next, double zeta_next, void *stream)
{
    int st = fill_velocities(d, 1, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_solve_for_pressure(d->solver, d->p, d->U[0], d->U[1], d->U[2], stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    if (d->correct_on_load) {
        swap_tendencies(d);
        return fused_launch(d, dt, gamma_next, zeta_next, 1, d->p, stage_dt, stream);
    }
    const int32_t ploc = OCN_LOC_CCC;
    double *pf[1] = {d->p};
    st = ocn_fill_halo_regions(&d->grid, pf, &ploc, 1, 1, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_pressure_correct_velocities(&d->grid, d->U[0], d->U[1], d->U[2], d->p, stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    swap_tendencies(d);
    st = fill_velocities(d, 0, stream);
    if (st != OCN_SUCCESS) return st;
    return fused_launch(d, dt, gamma_next, zeta_next, 1, nullptr, 0.0, stream);
}
}  // namespace

extern "C" int ocn_rk3_driver_destroy(ocn_rk3_driver_t d)
{
    if (!d) return OCN_SUCCESS;
    if (d->solver && d->owns_solver) ocn_poisson_destroy(d->solver);
    for (int f = 0; f < 3; ++f) {
        if (d->own[f]) (void)hipFree(d->own[f]);
        if (d->Gn[f]) (void)hipFree(d->Gn[f]);
        if (d->Gm[f]) (void)hipFree(d->Gm[f]);
    }
    delete d;
    return OCN_SUCCESS;
}

extern "C" int ocn_rk3_driver_create(ocn_rk3_driver_t *out, const ocn_grid *grid, double *u, double *v, double *w, double *p,
                                     ocn_poisson_t solver, void *stream)
{
    OCN_REQUIRE(out && grid && u && v && w && p, "ocn_rk3_driver_create: null argument");
    int st = ocn::validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->tx == OCN_PERIODIC && grid->ty == OCN_PERIODIC, "ocn_rk3_driver_create: x and y must be Periodic (one GPU)");
    ocn_rk3_driver *d = new ocn_rk3_driver();
    d->grid = *grid;
    d->user[0] = u; d->user[1] = v; d->user[2] = w;
    d->p = p;
    GridDev g = ocn::to_dev(*grid);
    for (int f = 0; f < 3; ++f) {
        const Lay L = ocn::make_lay(g, LOCS[f]);
        d->bytes[f] = (size_t)L.sx * L.sy * L.sz * sizeof(double);
        for (double **buf : {&d->own[f], &d->Gn[f], &d->Gm[f]}) {
            if (hipMalloc((void **)buf, d->bytes[f]) != hipSuccess || hipMemset(*buf, 0, d->bytes[f]) != hipSuccess) {
                ocn::set_error("ocn_rk3_driver_create: device allocation of %zu bytes failed", d->bytes[f]);
                ocn_rk3_driver_destroy(d);
                return OCN_ERR_ALLOC;
            }
        }
        d->U[f] = d->user[f];
        d->A[f] = d->own[f];
    }
    if (solver) {  // the caller's pressure solver for this grid (borrowed)
        d->solver = solver;
        d->owns_solver = false;
    } else {
        st = ocn_poisson_create(&d->solver, grid);
        if (st != OCN_SUCCESS) {
            ocn_rk3_driver_destroy(d);
            return st;
        }
    }
    // fold the pressure correction of stages 1 and 2 into the loads of the fused launch: all-periodic grids the tiled kernel covers
    const char *e = std::getenv("OCN_CORRECT_ON_LOAD");
    d->correct_on_load = grid->tz == OCN_PERIODIC && grid->Nx >= 16 && grid->Ny >= 8 && grid->Nz >= 4 && !(e && e[0] == '0');
    st = fill_velocities(d, 0, stream);  // update_state!(model; compute_tendencies = false) of the constructor
    if (st != OCN_SUCCESS) {
        ocn_rk3_driver_destroy(d);
        return st;
    }
    *out = d;
    return OCN_SUCCESS;
}

extern "C" int ocn_rk3_driver_time_step(ocn_rk3_driver_t d, double dt, void *stream)
{
    OCN_REQUIRE(d, "ocn_rk3_driver_time_step: null driver");
    // γ, ζ each rounded once to Float64 (runge_kutta_3.jl:53-62)
    const double g1 = 8.0 / 15, g2 = 5.0 / 12, g3 = 3.0 / 4, z2 = -17.0 / 60, z3 = -5.0 / 12;
    int st;
    if (!d->started) {  // iteration 0: update_state!(model) with the tendencies
        st = fill_velocities(d, 0, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn_compute_momentum_tendencies(&d->grid, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], nullptr, stream);
        if (st != OCN_SUCCESS) return st;
        d->started = true;
    }
    const double first_stage_dt = g1 * dt, second_stage_dt = (g2 + z2) * dt, third_stage_dt = (g3 + z3) * dt;
    // ---- first stage
    if (d->pending) {
        st = fused_launch(d, dt, g1, 0.0, 0, nullptr, 0.0, stream);
    } else {
        st = ocn_rk3_substep(&d->grid, 3, d->U, d->Gn, d->Gm, LOCS, dt, g1, 0.0, 0, stream);
    }
    if (st != OCN_SUCCESS) return st;
    st = project_and_advance(d, dt, first_stage_dt, g2, z2, stream);   // ... ends with the second substep
    if (st != OCN_SUCCESS) return st;
    st = project_and_advance(d, dt, second_stage_dt, g3, z3, stream);  // ... ends with the third substep
    if (st != OCN_SUCCESS) return st;
    // ---- third stage: projection, then the halos; its compute_tendencies! is fused into the next step's first substep
    st = fill_velocities(d, 1, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_solve_for_pressure(d->solver, d->p, d->U[0], d->U[1], d->U[2], third_stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    const int32_t ploc = OCN_LOC_CCC;
    double *pf[1] = {d->p};
    st = ocn_fill_halo_regions(&d->grid, pf, &ploc, 1, 1, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_pressure_correct_velocities(&d->grid, d->U[0], d->U[1], d->U[2], d->p, third_stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    st = fill_velocities(d, 0, stream);
    if (st != OCN_SUCCESS) return st;
    d->pending = true;
    d->iteration += 1;
    return OCN_SUCCESS;
}

extern "C" int ocn_rk3_driver_flush(ocn_rk3_driver_t d, void *stream)
{
    OCN_REQUIRE(d, "ocn_rk3_driver_flush: null driver");
    if (d->pending) {  // complete the deferred compute_tendencies!: G^n of the current state
        int st = ocn_compute_momentum_tendencies(&d->grid, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], nullptr, stream);
        if (st != OCN_SUCCESS) return st;
        d->pending = false;
    }
    if (d->U[0] != d->user[0]) {  // bring the velocities home (an odd number of fused launches since the last flush)
        for (int f = 0; f < 3; ++f) {
            OCN_CHECK_HIP(hipMemcpyAsync(d->user[f], d->U[f], d->bytes[f], hipMemcpyDeviceToDevice, ocn::as_stream(stream)));
            std::swap(d->U[f], d->A[f]);
        }
    }
    return OCN_SUCCESS;
}

extern "C" int ocn_rk3_driver_fields(ocn_rk3_driver_t d, double **u, double **v, double **w, double **Gu, double **Gv, double **Gw)
{
    OCN_REQUIRE(d, "ocn_rk3_driver_fields: null driver");
    if (u) *u = d->U[0];
    if (v) *v = d->U[1];
    if (w) *w = d->U[2];
    if (Gu) *Gu = d->Gn[0];
    if (Gv) *Gv = d->Gn[1];
    if (Gw) *Gw = d->Gn[2];
    return OCN_SUCCESS;
}
