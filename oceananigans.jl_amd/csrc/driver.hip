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
    // the third stage's pressure correction is left to the next step's first fused launch (applied on load like stages 1 and 2): the
    // velocities in U are then the UNCORRECTED u*, v*, w* with valid halos, p holds the pressure, ocn_rk3_driver_flush applies it
    bool defer_correction = false;
    bool correction_pending = false;
    double correction_dt = 0.0;
    // slab-x rank (ocn_rk3_driver_create_distributed): RCCL communicator + distributed Poisson handle, both borrowed
    ocn_comm_t comm = nullptr;
    ocn_dist_poisson_t dsolver = nullptr;
    // the last fused launch also wrote the x strips of its stepped velocities into the communicator's send buffers
    // (ocn_compute_momentum_tendencies_rk3_strips): the next exchange of U is posted without a pack launch
    bool use_strips = false, strips_ready = false;
};

namespace {
const int32_t LOCS[3] = {OCN_LOC_FCC, OCN_LOC_CFC, OCN_LOC_CCF};

// fill_halo_regions!(velocities): local periodic / wall fills, then -- on a slab -- the x exchange with the neighbours (synchronous)
int fill_velocities(ocn_rk3_driver *d, int fbnv, void *stream)
{
    d->strips_ready = false;  // (this exchange packs into the same send buffers)
    int st = ocn_fill_halo_regions(&d->grid, d->U, LOCS, 3, fbnv, stream);
    if (st != OCN_SUCCESS || !d->comm) return st;
    st = ocn_halo_exchange_begin(d->comm, &d->grid, d->U, LOCS, 3, stream);
    if (st != OCN_SUCCESS) return st;
    return ocn_halo_exchange_end(d->comm, &d->grid, d->U, LOCS, 3, stream);
}

int fill_pressure(ocn_rk3_driver *d, void *stream)
{
    const int32_t ploc = OCN_LOC_CCC;
    double *pf[1] = {d->p};
    int st = ocn_fill_halo_regions(&d->grid, pf, &ploc, 1, 1, stream);
    if (st != OCN_SUCCESS || !d->comm) return st;
    st = ocn_halo_exchange_begin(d->comm, &d->grid, pf, &ploc, 1, stream);
    if (st != OCN_SUCCESS) return st;
    return ocn_halo_exchange_end(d->comm, &d->grid, pf, &ploc, 1, stream);
}

// solve_for_pressure!(pNHS, solver, Δt, U) (solve_for_pressure.jl:78-82) on one GPU or on a slab (the slab pipelines of the handle:
// distributed_fft_based_poisson_solver.jl:141-178 without pack / unpack passes)
int solve(ocn_rk3_driver *d, double stage_dt, void *stream)
{
    if (!d->dsolver) return ocn_solve_for_pressure(d->solver, d->p, d->U[0], d->U[1], d->U[2], stage_dt, stream);
    int st = ocn_dist_poisson_source_term(d->dsolver, d->U[0], d->U[1], d->U[2], stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_dist_poisson_forward_yz(d->dsolver, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_dist_poisson_exchange(d->dsolver, d->comm, 0, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_dist_poisson_solve_x(d->dsolver, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_dist_poisson_exchange(d->dsolver, d->comm, 1, stream);
    if (st != OCN_SUCCESS) return st;
    return ocn_dist_poisson_backward_yz(d->dsolver, d->p, stream);
}

// update_state! + the next rk3_substep! in one launch, then the two velocity sets trade places
int fused_launch(ocn_rk3_driver *d, double dt, double gamma, double zeta, int has_zeta, const double *p_correct, double dt_correct,
                 void *stream)
{
    int st;
    d->strips_ready = false;
    if (d->use_strips && p_correct) {
        double *sw = nullptr, *se = nullptr;
        int64_t per_field = 0;
        st = ocn_halo_exchange_buffers(d->comm, &d->grid, LOCS, 3, &sw, &se, &per_field);
        if (st != OCN_SUCCESS) return st;
        st = ocn_compute_momentum_tendencies_rk3_strips(&d->grid, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], d->Gm[0], d->Gm[1],
                                                        d->Gm[2], d->A[0], d->A[1], d->A[2], dt, gamma, zeta, has_zeta, p_correct, dt_correct,
                                                        sw, se, per_field, stream);
        d->strips_ready = st == OCN_SUCCESS;
    } else {
        st = ocn_compute_momentum_tendencies_rk3(&d->grid, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], d->Gm[0], d->Gm[1],
                                                 d->Gm[2], d->A[0], d->A[1], d->A[2], dt, gamma, zeta, has_zeta, p_correct, dt_correct,
                                                 nullptr, stream);
    }
    if (st != OCN_SUCCESS) return st;
    for (int f = 0; f < 3; ++f) std::swap(d->U[f], d->A[f]);
    d->pending = false;
    return OCN_SUCCESS;
}

void swap_tendencies(ocn_rk3_driver *d)  // cache_previous_tendencies! (store_tendencies.jl:12-22) as a role swap
{
    for (int f = 0; f < 3; ++f) std::swap(d->Gn[f], d->Gm[f]);
}

// The projection of a stage up to (not including) the correction, for a launch that corrects on load: halos of the uncorrected
// velocities, pressure solve and -- on a slab -- the pressure planes of the neighbours.  On a slab the exchange of u*, v*, w* is posted
// BEFORE the solve (only the plane the divergence reads is waited for) and flies under it on the communication stream.
int project_for_load(ocn_rk3_driver *d, double stage_dt, void *stream)
{
    int st = ocn_fill_halo_regions(&d->grid, d->U, LOCS, 3, 1, stream);
    if (st != OCN_SUCCESS) return st;
    if (d->comm) {
        st = ocn_halo_exchange_plane(d->comm, &d->grid, d->U[0], OCN_LOC_FCC, 0, stream);  // u[nx+1] <- east neighbour's u[1]
        if (st != OCN_SUCCESS) return st;
        // the strips of u*, v*, w*: already in the send buffers when the last fused launch wrote them (no pack launch), packed here otherwise
        st = d->strips_ready ? ocn_halo_exchange_begin_packed(d->comm, &d->grid, d->U, LOCS, 3, stream)
                             : ocn_halo_exchange_begin(d->comm, &d->grid, d->U, LOCS, 3, stream);
        d->strips_ready = false;
        if (st != OCN_SUCCESS) return st;
    }
    st = solve(d, stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    if (d->comm) {
        st = ocn_halo_exchange_end(d->comm, &d->grid, d->U, LOCS, 3, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn_halo_exchange_pressure(d->comm, &d->grid, d->p, d->U[0], stage_dt, stream);
        if (st != OCN_SUCCESS) return st;
    }
    return OCN_SUCCESS;
}

// calculate_pressure_correction! + pressure_correct_velocities! + the halo fill of update_state! (pressure_correction.jl:8-50)
int project_and_correct(ocn_rk3_driver *d, double stage_dt, bool solve_too, void *stream)
{
    int st;
    if (solve_too) {
        st = fill_velocities(d, 1, stream);
        if (st != OCN_SUCCESS) return st;
        st = solve(d, stage_dt, stream);
        if (st != OCN_SUCCESS) return st;
    }
    st = fill_pressure(d, stream);
    if (st != OCN_SUCCESS) return st;
    st = ocn_pressure_correct_velocities(&d->grid, d->U[0], d->U[1], d->U[2], d->p, stage_dt, stream);
    if (st != OCN_SUCCESS) return st;
    return fill_velocities(d, 0, stream);
}

// everything between two substeps (runge_kutta_3.jl:103-118)
int project_and_advance(ocn_rk3_driver *d, double dt, double stage_dt, double gamma_next, double zeta_next, void *stream)
{
    int st;
    if (d->correct_on_load) {
        st = project_for_load(d, stage_dt, stream);
        if (st != OCN_SUCCESS) return st;
        swap_tendencies(d);
        return fused_launch(d, dt, gamma_next, zeta_next, 1, d->p, stage_dt, stream);
    }
    st = project_and_correct(d, stage_dt, true, stream);
    if (st != OCN_SUCCESS) return st;
    swap_tendencies(d);
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

static int driver_create(ocn_rk3_driver_t *out, const ocn_grid *grid, double *u, double *v, double *w, double *p, ocn_poisson_t solver,
                         ocn_dist_poisson_t dsolver, ocn_comm_t comm, void *stream)
{
    OCN_REQUIRE(out && grid && u && v && w && p, "ocn_rk3_driver_create: null argument");
    // one GPU: any topology the per-call entry points take -- on grids with walls / Flat directions in x, y the fused launch runs the tiled
    // epilogue on the interior box and the finishing kernel on the wall frames (general.hip), without the correction on load
    int st = comm ? ocn::validate_grid(grid) : ocn::validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    if (comm) {
        OCN_REQUIRE(grid->tx == OCN_FULLY_CONNECTED && grid->ty == OCN_PERIODIC && grid->tz == OCN_PERIODIC && dsolver,
                    "ocn_rk3_driver_create_distributed: a (FullyConnected, Periodic, Periodic) local grid and its distributed Poisson handle");
        int32_t fast = 0;
        st = ocn_dist_poisson_pipeline(dsolver, &fast);
        if (st != OCN_SUCCESS) return st;
        if (fast != 1 && fast != 3) {
            ocn::set_error("ocn_rk3_driver_create_distributed: the Poisson handle runs the transposing path (sizes outside the slab "
                           "pipelines): drive it through the per-call entry points");
            return OCN_ERR_UNSUPPORTED;
        }
        OCN_REQUIRE(grid->Nx >= 16 && grid->Nx >= grid->Hx + 1 && grid->Ny >= 8 && grid->Nz >= 4,
                    "ocn_rk3_driver_create_distributed: the local slab must be at least 16 x 8 x 4");
    } else {
        OCN_REQUIRE(grid->tx == OCN_PERIODIC || grid->tx == OCN_BOUNDED || grid->tx == OCN_FLAT,
                    "ocn_rk3_driver_create: a partitioned x (topology %d) needs ocn_rk3_driver_create_distributed", grid->tx);
    }
    ocn_rk3_driver *d = new ocn_rk3_driver();
    d->grid = *grid;
    d->comm = comm;
    d->dsolver = dsolver;
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
    if (comm) {
        d->owns_solver = false;
    } else if (solver) {  // the caller's pressure solver for this grid (borrowed)
        d->solver = solver;
        d->owns_solver = false;
    } else {
        st = ocn_poisson_create(&d->solver, grid);
        if (st != OCN_SUCCESS) {
            ocn_rk3_driver_destroy(d);
            return st;
        }
    }
    // fold the pressure correction into the loads of the fused launch: all-periodic grids the tiled kernel covers, one GPU or a slab.
    // OCN_CORRECT_ON_LOAD=0 (on a slab also OCN_DIST_CORRECT_ON_LOAD=0, the Python host's switch) is the conservative path for a first
    // multi-GPU run: every stage = synchronous halo exchange -> solve -> synchronous pressure exchange -> pressure_correct_velocities!
    // -> synchronous exchange -> one plain fused launch; no exchange is in flight while the solver's collective runs.
    const char *e = std::getenv("OCN_CORRECT_ON_LOAD"), *ed = std::getenv("OCN_DIST_CORRECT_ON_LOAD");
    const bool off = (e && e[0] == '0') || (comm && ed && ed[0] == '0');
    const bool all_periodic = grid->tx == OCN_PERIODIC && grid->ty == OCN_PERIODIC && grid->tz == OCN_PERIODIC;
    d->correct_on_load = (comm || (all_periodic && grid->Nx >= 16 && grid->Ny >= 8 && grid->Nz >= 4)) && !off;
    // strips written by the fused launch's epilogue instead of a pack launch: OCN_DIST_EPILOGUE_STRIPS=1.  Off by default -- measured at
    // the local sizes of one rank of 2 / of 8 (512^3, replica transport): 14.79 / 4.09 ms per rank-step with, 14.61 / 4.05 without: the
    // scattered 8-byte stores of the edge tiles and the wrapping unpack cost what the 40-us pack launch saves.
    const char *es = std::getenv("OCN_DIST_EPILOGUE_STRIPS");
    d->use_strips = comm && d->correct_on_load && grid->Nx >= 2 * grid->Hx && es && es[0] == '1';
    const char *dc = std::getenv("OCN_DRIVER_DEFER_CORRECTION");
    d->defer_correction = d->correct_on_load && !(dc && dc[0] == '0');
    st = fill_velocities(d, 0, stream);  // update_state!(model; compute_tendencies = false) of the constructor
    if (st != OCN_SUCCESS) {
        ocn_rk3_driver_destroy(d);
        return st;
    }
    *out = d;
    return OCN_SUCCESS;
}

extern "C" int ocn_rk3_driver_create(ocn_rk3_driver_t *out, const ocn_grid *grid, double *u, double *v, double *w, double *p,
                                     ocn_poisson_t solver, void *stream)
{
    return driver_create(out, grid, u, v, w, p, solver, nullptr, nullptr, stream);
}

extern "C" int ocn_rk3_driver_create_distributed(ocn_rk3_driver_t *out, const ocn_grid *local_grid, double *u, double *v, double *w,
                                                 double *p, ocn_dist_poisson_t solver, ocn_comm_t comm, void *stream)
{
    OCN_REQUIRE(solver && comm, "ocn_rk3_driver_create_distributed: null solver / communicator");
    return driver_create(out, local_grid, u, v, w, p, nullptr, solver, comm, stream);
}

extern "C" int ocn_rk3_driver_configure(ocn_rk3_driver_t d, int32_t defer_correction)
{
    OCN_REQUIRE(d, "ocn_rk3_driver_configure: null driver");
    OCN_REQUIRE(!d->correction_pending, "ocn_rk3_driver_configure: flush first (a deferred pressure correction is pending)");
    OCN_REQUIRE(!defer_correction || d->correct_on_load, "ocn_rk3_driver_configure: deferring the correction needs correction on load (all-periodic grid)");
    d->defer_correction = defer_correction != 0;
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
        const bool pc = d->correction_pending;  // the previous step's third-stage correction rides on this launch's loads
        st = fused_launch(d, dt, g1, 0.0, 0, pc ? d->p : nullptr, pc ? d->correction_dt : 0.0, stream);
        d->correction_pending = false;
    } else {
        st = ocn_rk3_substep(&d->grid, 3, d->U, d->Gn, d->Gm, LOCS, dt, g1, 0.0, 0, stream);
    }
    if (st != OCN_SUCCESS) return st;
    st = project_and_advance(d, dt, first_stage_dt, g2, z2, stream);   // ... ends with the second substep
    if (st != OCN_SUCCESS) return st;
    st = project_and_advance(d, dt, second_stage_dt, g3, z3, stream);  // ... ends with the third substep
    if (st != OCN_SUCCESS) return st;
    // ---- third stage: projection; its compute_tendencies! is fused into the next step's first substep, and so is -- when the
    //      correction is deferred -- pressure_correct_velocities! itself
    if (d->defer_correction) {
        st = project_for_load(d, third_stage_dt, stream);
        if (st != OCN_SUCCESS) return st;
        d->correction_pending = true;
        d->correction_dt = third_stage_dt;
    } else {
        st = project_and_correct(d, third_stage_dt, true, stream);
        if (st != OCN_SUCCESS) return st;
    }
    d->pending = true;
    d->iteration += 1;
    return OCN_SUCCESS;
}

extern "C" int ocn_rk3_driver_flush(ocn_rk3_driver_t d, void *stream)
{
    OCN_REQUIRE(d, "ocn_rk3_driver_flush: null driver");
    if (d->correction_pending) {  // the deferred third-stage correction: p halos, pressure_correct_velocities!, velocity halos
        int st = project_and_correct(d, d->correction_dt, false, stream);
        if (st != OCN_SUCCESS) return st;
        d->correction_pending = false;
    }
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
