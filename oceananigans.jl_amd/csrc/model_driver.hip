// model_driver.hip -- time_step!(model::NonhydrostaticModel{<:RungeKutta3TimeStepper}, Δt) as ONE call of the C ABI for a model with
// tracers and the SURVEY §8(f) terms (config 4's term set), one GPU with Periodic x and y, or ONE RANK of a slab-x run:
//   runge_kutta_3.jl:77-151, update_nonhydrostatic_model_state.jl:20-70, compute_nonhydrostatic_tendencies.jl:17-54, 204-213,
//   pressure_correction.jl:8-50, store_tendencies.jl:12-22.
// Host code only: every device operation is one of the public entry points, issued in the order the Python host issues them on its
// general fused path (models.py::_time_step_rk3, _update_state_and_rk3_substep_general) -- so the two are bit-identical
// (tests/test_gpu_physics.py::test_c_model_driver_equals_python_host) -- without an interpreter between the launches:
//   per stage boundary:  halo fill of every prognostic field (one launch) -> AMD diffusivities (one launch) -> pHY' -> halo fill of
//   nu_e, kappa_e -> momentum: tiled advection launch + one finishing pass (extra terms, u / v boundary fluxes, next substep) ->
//   one launch per tracer (advection, diffusion, boundary flux, next substep) -> velocity halos -> Poisson solve -> pressure halos ->
//   pressure correction.  Substep results land in a second set of arrays whose roles then alternate; G^n / G^- swap.
#include <cstdlib>
#include <cstring>

#include "ocn_internal.h"

using ocn::GridDev;
using ocn::Lay;

namespace {
constexpr int NF = 3 + OCN_MODEL_MAX_TRACERS;
}

struct ocn_model_driver {
    ocn_grid grid{};
    ocn_model_terms terms{};
    ocn_poisson_t solver = nullptr;
    bool owns_solver = true;
    int n = 3, nt = 0;            // prognostic fields, tracers
    int tT = -1, tS = -1;
    int32_t locs[NF] = {OCN_LOC_FCC, OCN_LOC_CFC, OCN_LOC_CCF, OCN_LOC_CCC, OCN_LOC_CCC, OCN_LOC_CCC, OCN_LOC_CCC};
    double *user[NF] = {}, *own[NF] = {}, *U[NF] = {}, *A[NF] = {};
    double *Gn[NF] = {}, *Gm[NF] = {};
    size_t bytes[NF] = {};
    double *p = nullptr, *pHY = nullptr, *nu_e = nullptr, *kappa_e[OCN_MODEL_MAX_TRACERS] = {};
    double kappa[OCN_MODEL_MAX_TRACERS] = {}, Cnu = 0.0, Ck[OCN_MODEL_MAX_TRACERS] = {};
    ocn_field_bcs bcs_store[NF];
    const ocn_field_bcs *bcs[NF] = {};  // NULL = defaults
    bool any_bcs = false, any_flux = false, momentum_extra = false;
    bool pending = false, started = false;
    long long iteration = 0;
    // slab-x rank (ocn_model_driver_create_distributed): RCCL communicator + distributed Poisson handle, both borrowed
    ocn_comm_t comm = nullptr;
    ocn_dist_poisson_t dsolver = nullptr;
    // interior / buffer split of a slab's stage boundary (update_state_split): the east buffer strip runs on a side stream
    bool split = false;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

namespace {
bool has_flux(const ocn_field_bcs *b)
{
    if (!b) return false;
    const ocn_bc *s[6] = {&b->west, &b->east, &b->south, &b->north, &b->bottom, &b->top};
    for (const ocn_bc *c : s)
        if (c->kind == OCN_BC_FLUX) return true;
    return false;
}
bool is_default(const ocn_field_bcs *b)
{
    const ocn_bc *s[6] = {&b->west, &b->east, &b->south, &b->north, &b->bottom, &b->top};
    for (const ocn_bc *c : s)
        if (c->kind != OCN_BC_DEFAULT) return false;
    return true;
}
const ocn_field_bcs *flux_bcs(const ocn_model_driver *d, int f) { return has_flux(d->bcs[f]) ? d->bcs[f] : nullptr; }

void refresh_terms(ocn_model_driver *d)  // the buoyancy tracers' storage alternates with every fused stage boundary
{
    d->terms.T = d->tT >= 0 ? d->U[3 + d->tT] : nullptr;
    d->terms.S = d->tS >= 0 ? d->U[3 + d->tS] : nullptr;
}

// fill_halo_regions!(fields): one launch for the tuple, the fields' own bottom / top Value / Gradient conditions included
int fill_local(ocn_model_driver *d, double *const *fields, const int32_t *locs, const ocn_field_bcs *const *bcs, int n, int fbnv, void *stream)
{
    bool any = false;
    for (int f = 0; f < n && bcs; ++f) any = any || bcs[f];
    return any ? ocn_fill_halo_regions_bcs(&d->grid, fields, locs, bcs, n, fbnv, stream) : ocn_fill_halo_regions(&d->grid, fields, locs, n, fbnv, stream);
}
int fill(ocn_model_driver *d, double *const *fields, const int32_t *locs, const ocn_field_bcs *const *bcs, int n, int fbnv, void *stream)
{
    int st = fill_local(d, fields, locs, bcs, n, fbnv, stream);
    if (st != OCN_SUCCESS || !d->comm) return st;
    // a slab: local (y, z) fills first, the x exchange with the neighbours last (fill_halo_regions.jl:148-196); synchronous
    st = ocn_halo_exchange_begin(d->comm, &d->grid, fields, locs, n, stream);
    if (st != OCN_SUCCESS) return st;
    return ocn_halo_exchange_end(d->comm, &d->grid, fields, locs, n, stream);
}

// solve_for_pressure! on one GPU or on a slab (the slab pipelines of the handle, csrc/poisson.hip)
int solve(ocn_model_driver *d, double stage_dt, void *stream)
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

// compute_auxiliaries! (update_nonhydrostatic_model_state.jl:59-70) + the diffusivity halo fill (:48)
int compute_auxiliaries(ocn_model_driver *d, void *stream)
{
    int st;
    if (d->terms.closure == 2) {
        st = ocn_compute_amd_diffusivities(&d->grid, d->Cnu, d->U[0], d->U[1], d->U[2], d->nu_e, d->nt, d->Ck, d->U + 3, d->kappa_e, stream);
        if (st != OCN_SUCCESS) return st;
    }
    if (d->pHY) {
        st = ocn_update_hydrostatic_pressure(&d->grid, &d->terms, d->pHY, stream);
        if (st != OCN_SUCCESS) return st;
    }
    if (d->terms.closure == 2) {
        double *f[1 + OCN_MODEL_MAX_TRACERS];
        int32_t l[1 + OCN_MODEL_MAX_TRACERS];
        f[0] = d->nu_e;
        l[0] = OCN_LOC_CCC;
        for (int t = 0; t < d->nt; ++t) {
            f[1 + t] = d->kappa_e[t];
            l[1 + t] = OCN_LOC_CCC;
        }
        st = fill(d, f, l, nullptr, 1 + d->nt, 1, stream);
        if (st != OCN_SUCCESS) return st;
    }
    return OCN_SUCCESS;
}

// update_state!(model; compute_tendencies = false): tupled halo fill (fill_boundary_normal_velocities = false) + auxiliaries
int update_state(ocn_model_driver *d, void *stream)
{
    int st = fill(d, d->U, d->locs, d->bcs, d->n, 0, stream);
    if (st != OCN_SUCCESS) return st;
    return compute_auxiliaries(d, stream);
}

// compute_tendencies! (compute_nonhydrostatic_tendencies.jl:17-54) with the boundary contributions (:204-213)
int compute_tendencies(ocn_model_driver *d, void *stream)
{
    int st = ocn_compute_momentum_tendencies_terms(&d->grid, &d->terms, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], nullptr, stream);
    if (st != OCN_SUCCESS) return st;
    for (int t = 0; t < d->nt; ++t) {
        st = ocn_compute_tracer_tendency_terms(&d->grid, &d->terms, d->terms.closure == 1 ? d->kappa[t] : 0.0,
                                               d->terms.closure == 2 ? d->kappa_e[t] : nullptr, d->U[0], d->U[1], d->U[2], d->U[3 + t],
                                               d->Gn[3 + t], nullptr, stream);
        if (st != OCN_SUCCESS) return st;
    }
    if (d->any_flux) {
        const ocn_field_bcs *fb[NF];
        for (int f = 0; f < d->n; ++f) fb[f] = flux_bcs(d, f);
        st = ocn_apply_flux_bcs(&d->grid, d->Gn, d->U, d->locs, fb, d->n, stream);
        if (st != OCN_SUCCESS) return st;
    }
    d->pending = false;
    return OCN_SUCCESS;
}

// compute_tendencies! + the next rk3_substep! of every prognostic field, then the two sets of arrays trade places
// the launches of one stage boundary over `range` (NULL = the whole slab): tendencies + next substep of every prognostic field
int launch_tendencies(ocn_model_driver *d, double dt, double gamma, double zeta, int has_zeta, const int32_t *range, void *stream)
{
    int st;
    if (d->momentum_extra)
        st = ocn_compute_momentum_tendencies_terms_rk3(&d->grid, &d->terms, flux_bcs(d, 0), flux_bcs(d, 1), d->U[0], d->U[1], d->U[2], d->Gn[0],
                                                       d->Gn[1], d->Gn[2], d->Gm[0], d->Gm[1], d->Gm[2], d->A[0], d->A[1], d->A[2], dt, gamma, zeta,
                                                       has_zeta, range, stream);
    else
        st = ocn_compute_momentum_tendencies_rk3(&d->grid, d->U[0], d->U[1], d->U[2], d->Gn[0], d->Gn[1], d->Gn[2], d->Gm[0], d->Gm[1], d->Gm[2],
                                                 d->A[0], d->A[1], d->A[2], dt, gamma, zeta, has_zeta, nullptr, 0.0, range, stream);
    if (st != OCN_SUCCESS) return st;
    int q = 0;
    while (q < d->nt) {
        const double kap[2] = {d->terms.closure == 1 ? d->kappa[q] : 0.0, (q + 1 < d->nt && d->terms.closure == 1) ? d->kappa[q + 1] : 0.0};
        if (q + 1 < d->nt) {  // pairs of tracers may share one launch (off by default in the library: *launched says)
            const double *ke[2] = {d->terms.closure == 2 ? d->kappa_e[q] : nullptr, d->terms.closure == 2 ? d->kappa_e[q + 1] : nullptr};
            const ocn_field_bcs *fb[2] = {flux_bcs(d, 3 + q), flux_bcs(d, 4 + q)};
            const double *c[2] = {d->U[3 + q], d->U[4 + q]};
            double *G[2] = {d->Gn[3 + q], d->Gn[4 + q]};
            const double *Gp[2] = {d->Gm[3 + q], d->Gm[4 + q]};
            double *out[2] = {d->A[3 + q], d->A[4 + q]};
            int32_t did = 0;
            st = ocn_compute_tracer_pair_tendency_terms_rk3(&d->grid, &d->terms, kap, ke, fb, d->U[0], d->U[1], d->U[2], c, G, Gp, out, dt, gamma,
                                                            zeta, has_zeta, range, &did, stream);
            if (st != OCN_SUCCESS) return st;
            if (did) {
                q += 2;
                continue;
            }
        }
        st = ocn_compute_tracer_tendency_terms_rk3(&d->grid, &d->terms, kap[0], d->terms.closure == 2 ? d->kappa_e[q] : nullptr, flux_bcs(d, 3 + q),
                                                   d->U[0], d->U[1], d->U[2], d->U[3 + q], d->Gn[3 + q], d->Gm[3 + q], d->A[3 + q], dt, gamma, zeta,
                                                   has_zeta, range, stream);
        if (st != OCN_SUCCESS) return st;
        q += 1;
    }
    return OCN_SUCCESS;
}

void swap_sets(ocn_model_driver *d)
{
    for (int f = 0; f < d->n; ++f) std::swap(d->U[f], d->A[f]);
    refresh_terms(d);
    d->pending = false;
}

// compute_tendencies! + the next rk3_substep! of every prognostic field, then the two sets of arrays trade places
int fused_launch(ocn_model_driver *d, double dt, double gamma, double zeta, int has_zeta, void *stream)
{
    int st = launch_tendencies(d, dt, gamma, zeta, has_zeta, nullptr, stream);
    if (st != OCN_SUCCESS) return st;
    swap_sets(d);
    return OCN_SUCCESS;
}

// update_state! + compute_tendencies! + the next substep on a slab WITH the interior / buffer split of
// interleave_communication_and_computation.jl:29-67 and compute_nonhydrostatic_buffer_tendencies.jl:10-83 (the choreography of
// distributed.py::update_state_general, launch for launch): the x exchange of the prognostic fields flies under the auxiliaries and
// tendencies of the columns that read no x halo; the diffusivities and pHY' of the edge and halo columns are RECOMPUTED from the
// exchanged halos (:55-68: what the neighbour computes for its own edge column -- same inputs, same arithmetic -- instead of a
// second exchange), then the two Hx-wide buffer strips, the east one on a side stream.
int update_state_split(ocn_model_driver *d, double dt, double gamma, double zeta, int has_zeta, void *stream)
{
    const int nx = d->grid.Nx, Hx = d->grid.Hx, Ny = d->grid.Ny, Nz = d->grid.Nz;
    const bool amd = d->terms.closure == 2;
    double *aux[1 + OCN_MODEL_MAX_TRACERS];
    int32_t auxl[1 + OCN_MODEL_MAX_TRACERS];
    aux[0] = d->nu_e; auxl[0] = OCN_LOC_CCC;
    for (int t = 0; t < d->nt; ++t) { aux[1 + t] = d->kappa_e[t]; auxl[1 + t] = OCN_LOC_CCC; }
    auto diffusivities = [&](int i0, int i1, void *s) {
        return amd ? ocn_compute_amd_diffusivities_range(&d->grid, d->Cnu, d->U[0], d->U[1], d->U[2], d->nu_e, d->nt, d->Ck, d->U + 3, d->kappa_e, i0, i1, s)
                   : OCN_SUCCESS;
    };
    auto hydrostatic = [&](int i0, int i1, void *s) {
        return d->pHY ? ocn_update_hydrostatic_pressure_range(&d->grid, &d->terms, d->pHY, i0, i1, s) : OCN_SUCCESS;
    };
#define OCN_TRY(expr) do { int st_ = (expr); if (st_ != OCN_SUCCESS) return st_; } while (0)
    OCN_TRY(fill_local(d, d->U, d->locs, d->bcs, d->n, 0, stream));
    OCN_TRY(ocn_halo_exchange_begin(d->comm, &d->grid, d->U, d->locs, d->n, stream));
    // interior: the diffusivities of columns 2 .. nx-1 read u, v, w, c at i-1 .. i+1 (local); pHY' of a column reads that column only
    OCN_TRY(diffusivities(2, nx - 1, stream));
    OCN_TRY(hydrostatic(1, nx, stream));
    if (amd) OCN_TRY(ocn_fill_halo_regions(&d->grid, aux, auxl, 1 + d->nt, 1, stream));  // their y / z halos
    // The reference's buffers are Hx wide (compute_nonhydrostatic_buffer_tendencies.jl:28-39); any width >= Hx gives the same result
    // cell for cell (OCN_DIST_BUFFER_WIDTH).  Measured at the R = 8 local size of config 4 (64 x 512 x 256, no link time): no split 5.96
    // ms per rank-step, Hx-wide buffers 7.24, 16-wide buffers (full tile columns) 7.75 -- the wider strips serialise more work behind
    // the exchange than their better tiles save.
    static const int forced = std::getenv("OCN_DIST_BUFFER_WIDTH") ? std::atoi(std::getenv("OCN_DIST_BUFFER_WIDTH")) : 0;
    int W = forced > 0 ? forced : Hx;
    if (W < Hx) W = Hx;
    if (nx - 2 * W < 1) W = Hx;
    const int32_t interior[6] = {W + 1, nx - W, 1, Ny, 1, Nz};
    OCN_TRY(launch_tendencies(d, dt, gamma, zeta, has_zeta, interior, stream));
    OCN_TRY(ocn_halo_exchange_end(d->comm, &d->grid, d->U, d->locs, d->n, stream));
    OCN_TRY(diffusivities(0, 1, stream));
    OCN_TRY(diffusivities(nx, nx + 1, stream));
    OCN_TRY(hydrostatic(0, 0, stream));
    OCN_TRY(hydrostatic(nx + 1, nx + 1, stream));
    if (amd) OCN_TRY(ocn_fill_halo_regions(&d->grid, aux, auxl, 1 + d->nt, 1, stream));
    const int w1 = W < nx ? W : nx, e0 = (nx - W + 1 > w1 + 1) ? nx - W + 1 : w1 + 1;
    const int32_t west[6] = {1, w1, 1, Ny, 1, Nz}, east[6] = {e0, nx, 1, Ny, 1, Nz};
    if (e0 <= nx) {  // the two strips are independent and each too small to fill the chip: side by side
        hipStream_t cur = ocn::as_stream(stream);
        OCN_CHECK_HIP(hipEventRecord(d->fork, cur));
        OCN_CHECK_HIP(hipStreamWaitEvent(d->side, d->fork, 0));
        OCN_TRY(launch_tendencies(d, dt, gamma, zeta, has_zeta, east, d->side));
        OCN_CHECK_HIP(hipEventRecord(d->join, d->side));
        OCN_TRY(launch_tendencies(d, dt, gamma, zeta, has_zeta, west, stream));
        OCN_CHECK_HIP(hipStreamWaitEvent(cur, d->join, 0));
    } else {
        OCN_TRY(launch_tendencies(d, dt, gamma, zeta, has_zeta, west, stream));
    }
#undef OCN_TRY
    swap_sets(d);
    return OCN_SUCCESS;
}

// calculate_pressure_correction! + pressure_correct_velocities! (pressure_correction.jl:8-50).  minimal (a slab, stages 1 and 2, whose
// update_state! refills every halo right afterwards): the two synchronous x exchanges move only the planes the projection reads --
// u[nx+1] for the divergence and p[0] for the pressure gradient at the first face -- instead of all 2 Hx planes of u, v, w and p
int project(ocn_model_driver *d, double stage_dt, bool minimal, void *stream)
{
    const int32_t ploc = OCN_LOC_CCC;
    double *pf[1] = {d->p};
    int st;
    if (d->comm && minimal) {
        const bool any = d->bcs[0] || d->bcs[1] || d->bcs[2];
        st = any ? ocn_fill_halo_regions_bcs(&d->grid, d->U, d->locs, d->bcs, 3, 1, stream) : ocn_fill_halo_regions(&d->grid, d->U, d->locs, 3, 1, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn_halo_exchange_plane(d->comm, &d->grid, d->U[0], OCN_LOC_FCC, 0, stream);
        if (st != OCN_SUCCESS) return st;
        st = solve(d, stage_dt, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn_fill_halo_regions(&d->grid, pf, &ploc, 1, 1, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn_halo_exchange_plane(d->comm, &d->grid, d->p, OCN_LOC_CCC, 1, stream);
        if (st != OCN_SUCCESS) return st;
    } else {
        st = fill(d, d->U, d->locs, d->bcs, 3, 1, stream);
        if (st != OCN_SUCCESS) return st;
        st = solve(d, stage_dt, stream);
        if (st != OCN_SUCCESS) return st;
        st = fill(d, pf, &ploc, nullptr, 1, 1, stream);
        if (st != OCN_SUCCESS) return st;
    }
    return ocn_pressure_correct_velocities(&d->grid, d->U[0], d->U[1], d->U[2], d->p, stage_dt, stream);
}

// everything between two substeps (runge_kutta_3.jl:103-118)
int project_and_advance(ocn_model_driver *d, double dt, double stage_dt, double gamma_next, double zeta_next, void *stream)
{
    int st = project(d, stage_dt, true, stream);
    if (st != OCN_SUCCESS) return st;
    for (int f = 0; f < d->n; ++f) std::swap(d->Gn[f], d->Gm[f]);  // cache_previous_tendencies! as a role swap
    if (d->split) return update_state_split(d, dt, gamma_next, zeta_next, 1, stream);
    st = update_state(d, stream);
    if (st != OCN_SUCCESS) return st;
    return fused_launch(d, dt, gamma_next, zeta_next, 1, stream);
}
}  // namespace

extern "C" int ocn_model_driver_destroy(ocn_model_driver_t d)
{
    if (!d) return OCN_SUCCESS;
    if (d->solver && d->owns_solver) ocn_poisson_destroy(d->solver);
    if (d->side) { (void)hipStreamSynchronize(d->side); (void)hipStreamDestroy(d->side); }
    if (d->fork) (void)hipEventDestroy(d->fork);
    if (d->join) (void)hipEventDestroy(d->join);
    for (int f = 0; f < NF; ++f) {
        if (d->own[f]) (void)hipFree(d->own[f]);
        if (d->Gn[f]) (void)hipFree(d->Gn[f]);
        if (d->Gm[f]) (void)hipFree(d->Gm[f]);
    }
    delete d;
    return OCN_SUCCESS;
}

static int model_driver_create(ocn_model_driver_t *out, const ocn_grid *grid, const ocn_model_driver_desc *desc, double *u, double *v, double *w,
                               double *p, ocn_poisson_t solver, ocn_dist_poisson_t dsolver, ocn_comm_t comm, void *stream)
{
    OCN_REQUIRE(out && grid && desc && u && v && w && p, "ocn_model_driver_create: null argument");
    // one GPU: any topology the per-call entry points take (walls / Flat directions in x, y since round 4: the fused stage boundaries of
    // general.hip -- tiled epilogues on the interior box, finishing kernels on the wall frames)
    int st = comm ? ocn::validate_grid(grid) : ocn::validate_grid_any(grid);
    if (st != OCN_SUCCESS) return st;
    if (comm) {
        OCN_REQUIRE(grid->tx == OCN_FULLY_CONNECTED && grid->ty == OCN_PERIODIC && dsolver,
                    "ocn_model_driver_create_distributed: a (FullyConnected, Periodic, *) local grid and its distributed Poisson handle");
        int32_t fast = 0;
        st = ocn_dist_poisson_pipeline(dsolver, &fast);
        if (st != OCN_SUCCESS) return st;
        if (fast < 1 || fast > 3) {
            ocn::set_error("ocn_model_driver_create_distributed: the Poisson handle runs the transposing path (sizes outside the slab "
                           "pipelines): drive it through the per-call entry points");
            return OCN_ERR_UNSUPPORTED;
        }
        OCN_REQUIRE(grid->Nx >= grid->Hx, "ocn_model_driver_create_distributed: the local slab must be at least a halo wide");
    } else {
        OCN_REQUIRE(grid->tx == OCN_PERIODIC || grid->tx == OCN_BOUNDED || grid->tx == OCN_FLAT,
                    "ocn_model_driver_create: a partitioned x (topology %d) needs ocn_model_driver_create_distributed", grid->tx);
        if (grid->tx != OCN_PERIODIC || grid->ty != OCN_PERIODIC)
            for (int f = 0; f < 3 + desc->n_tracers && f < 3 + OCN_MODEL_MAX_TRACERS; ++f) {
                const ocn_field_bcs *b = desc->bcs[f];
                if (!b) continue;
                OCN_REQUIRE(b->west.kind != OCN_BC_FLUX && b->east.kind != OCN_BC_FLUX && b->south.kind != OCN_BC_FLUX && b->north.kind != OCN_BC_FLUX,
                            "ocn_model_driver_create: field %d: a flux through an x / y wall needs the unfused sequence (ocn_apply_flux_bcs)", f);
                OCN_REQUIRE(grid->tx == OCN_PERIODIC || ((b->bottom.kind != OCN_BC_FLUX || !b->bottom.values) && (b->top.kind != OCN_BC_FLUX || !b->top.values)),
                            "ocn_model_driver_create: field %d: array-valued bottom / top fluxes need a Periodic x", f);
            }
    }
    OCN_REQUIRE(desc->n_tracers >= 0 && desc->n_tracers <= OCN_MODEL_MAX_TRACERS, "ocn_model_driver_create: n_tracers %d outside 0..%d",
                desc->n_tracers, OCN_MODEL_MAX_TRACERS);
    const ocn_model_terms &t = desc->terms;
    OCN_REQUIRE(t.advection == OCN_ADVECTION_WENO5 || t.advection == OCN_ADVECTION_UPWIND5,
                "ocn_model_driver_create: the fused stage boundaries need WENO5 or UpwindBiased5 advection (got %d)", t.advection);
    OCN_REQUIRE(t.closure >= 0 && t.closure <= 2 && t.buoyancy >= OCN_BUOYANCY_NONE && t.buoyancy <= OCN_BUOYANCY_SEAWATER_S,
                "ocn_model_driver_create: unknown closure %d / buoyancy %d", t.closure, t.buoyancy);
    OCN_REQUIRE(desc->tracer_T >= -1 && desc->tracer_T < desc->n_tracers && desc->tracer_S >= -1 && desc->tracer_S < desc->n_tracers,
                "ocn_model_driver_create: tracer_T / tracer_S outside the tracer list");
    const bool needT = t.buoyancy == OCN_BUOYANCY_TRACER || t.buoyancy == OCN_BUOYANCY_SEAWATER_TS || t.buoyancy == OCN_BUOYANCY_SEAWATER_T;
    const bool needS = t.buoyancy == OCN_BUOYANCY_SEAWATER_TS || t.buoyancy == OCN_BUOYANCY_SEAWATER_S;
    OCN_REQUIRE((!needT || desc->tracer_T >= 0) && (!needS || desc->tracer_S >= 0), "ocn_model_driver_create: buoyancy %d needs its tracer(s)",
                t.buoyancy);
    OCN_REQUIRE(desc->pHY == t.pHY, "ocn_model_driver_create: desc->pHY must equal desc->terms.pHY");
    OCN_REQUIRE(t.closure != 2 || (desc->nu_e && desc->nu_e == t.nu_e), "ocn_model_driver_create: closure 2 needs nu_e (== terms.nu_e)");
    ocn_model_driver *d = new ocn_model_driver();
    d->grid = *grid;
    d->terms = t;
    d->nt = desc->n_tracers;
    d->n = 3 + d->nt;
    d->tT = needT ? desc->tracer_T : -1;
    d->tS = needS ? desc->tracer_S : -1;
    d->p = p;
    d->comm = comm;
    d->dsolver = dsolver;
    d->pHY = desc->pHY;
    d->nu_e = desc->nu_e;
    d->Cnu = desc->C_nu;
    d->user[0] = u; d->user[1] = v; d->user[2] = w;
    bool ok = true;
    for (int q = 0; q < d->nt; ++q) {
        d->user[3 + q] = desc->tracers[q];
        d->kappa[q] = desc->kappa[q];
        d->Ck[q] = desc->C_kappa[q];
        d->kappa_e[q] = desc->kappa_e[q];
        ok = ok && desc->tracers[q] && (t.closure != 2 || desc->kappa_e[q]);
    }
    if (!ok) {
        delete d;
        ocn::set_error("ocn_model_driver_create: null tracer / kappa_e pointer");
        return OCN_ERR_INVALID_ARGUMENT;
    }
    for (int f = 0; f < d->n; ++f) {
        if (desc->bcs[f] && !is_default(desc->bcs[f])) {
            d->bcs_store[f] = *desc->bcs[f];
            d->bcs[f] = &d->bcs_store[f];
            d->any_bcs = true;
            d->any_flux = d->any_flux || has_flux(d->bcs[f]);
        }
    }
    d->momentum_extra = t.coriolis != 0 || t.closure != 0 || t.buoyancy != OCN_BUOYANCY_NONE || t.advection != OCN_ADVECTION_WENO5 ||
                        has_flux(d->bcs[0]) || has_flux(d->bcs[1]);
    GridDev g = ocn::to_dev(*grid);
    for (int f = 0; f < d->n; ++f) {
        const Lay L = ocn::make_lay(g, d->locs[f]);
        d->bytes[f] = (size_t)L.sx * L.sy * L.sz * sizeof(double);
        for (double **buf : {&d->own[f], &d->Gn[f], &d->Gm[f]}) {
            if (hipMalloc((void **)buf, d->bytes[f]) != hipSuccess || hipMemset(*buf, 0, d->bytes[f]) != hipSuccess) {
                ocn::set_error("ocn_model_driver_create: device allocation of %zu bytes failed", d->bytes[f]);
                ocn_model_driver_destroy(d);
                return OCN_ERR_ALLOC;
            }
        }
        d->U[f] = d->user[f];
        d->A[f] = d->own[f];
    }
    refresh_terms(d);
    if (comm) {
        d->owns_solver = false;
        // interior / buffer split of the stage boundaries (OCN_DIST_GENERAL_OVERLAP=0: every exchange synchronous, as before round 4)
        const char *e = std::getenv("OCN_DIST_GENERAL_OVERLAP");
        d->split = grid->Nx - 2 * grid->Hx >= 1 && grid->Hx >= 2 && !(e && e[0] == '0');
        if (d->split && (hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking) != hipSuccess ||
                         hipEventCreateWithFlags(&d->fork, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&d->join, hipEventDisableTiming) != hipSuccess)) {
            ocn::set_error("ocn_model_driver_create_distributed: stream / event creation failed");
            ocn_model_driver_destroy(d);
            return OCN_ERR_HIP;
        }
    } else if (solver) {
        d->solver = solver;
        d->owns_solver = false;
    } else {
        st = ocn_poisson_create(&d->solver, grid);
        if (st != OCN_SUCCESS) {
            ocn_model_driver_destroy(d);
            return st;
        }
    }
    st = update_state(d, stream);  // update_state!(model; compute_tendencies = false) of the constructor
    if (st != OCN_SUCCESS) {
        ocn_model_driver_destroy(d);
        return st;
    }
    *out = d;
    return OCN_SUCCESS;
}

extern "C" int ocn_model_driver_create(ocn_model_driver_t *out, const ocn_grid *grid, const ocn_model_driver_desc *desc, double *u, double *v,
                                       double *w, double *p, ocn_poisson_t solver, void *stream)
{
    return model_driver_create(out, grid, desc, u, v, w, p, solver, nullptr, nullptr, stream);
}

extern "C" int ocn_model_driver_create_distributed(ocn_model_driver_t *out, const ocn_grid *local_grid, const ocn_model_driver_desc *desc, double *u,
                                                   double *v, double *w, double *p, ocn_dist_poisson_t solver, ocn_comm_t comm, void *stream)
{
    OCN_REQUIRE(solver && comm, "ocn_model_driver_create_distributed: null solver / communicator");
    return model_driver_create(out, local_grid, desc, u, v, w, p, nullptr, solver, comm, stream);
}

extern "C" int ocn_model_driver_time_step(ocn_model_driver_t d, double dt, void *stream)
{
    OCN_REQUIRE(d, "ocn_model_driver_time_step: null driver");
    const double g1 = 8.0 / 15, g2 = 5.0 / 12, g3 = 3.0 / 4, z2 = -17.0 / 60, z3 = -5.0 / 12;  // runge_kutta_3.jl:53-62
    int st;
    if (!d->started) {  // iteration 0: update_state!(model) with the tendencies
        st = update_state(d, stream);
        if (st != OCN_SUCCESS) return st;
        st = compute_tendencies(d, stream);
        if (st != OCN_SUCCESS) return st;
        d->started = true;
    }
    const double first_stage_dt = g1 * dt, second_stage_dt = (g2 + z2) * dt, third_stage_dt = (g3 + z3) * dt;
    // ---- first stage
    if (d->pending)
        st = fused_launch(d, dt, g1, 0.0, 0, stream);  // the deferred compute_tendencies! of the last step + this step's first substep
    else
        st = ocn_rk3_substep(&d->grid, d->n, d->U, d->Gn, d->Gm, d->locs, dt, g1, 0.0, 0, stream);
    if (st != OCN_SUCCESS) return st;
    st = project_and_advance(d, dt, first_stage_dt, g2, z2, stream);   // ... ends with the second substep
    if (st != OCN_SUCCESS) return st;
    st = project_and_advance(d, dt, second_stage_dt, g3, z3, stream);  // ... ends with the third substep
    if (st != OCN_SUCCESS) return st;
    // ---- third stage: projection, update_state!; its compute_tendencies! is fused into the next step's first substep
    st = project(d, third_stage_dt, false, stream);
    if (st != OCN_SUCCESS) return st;
    st = update_state(d, stream);
    if (st != OCN_SUCCESS) return st;
    d->pending = true;
    d->iteration += 1;
    return OCN_SUCCESS;
}

extern "C" int ocn_model_driver_flush(ocn_model_driver_t d, void *stream)
{
    OCN_REQUIRE(d, "ocn_model_driver_flush: null driver");
    if (d->pending) {
        int st = compute_tendencies(d, stream);
        if (st != OCN_SUCCESS) return st;
    }
    if (d->U[0] != d->user[0]) {  // an odd number of fused launches since the last flush: bring the fields home
        for (int f = 0; f < d->n; ++f) {
            OCN_CHECK_HIP(hipMemcpyAsync(d->user[f], d->U[f], d->bytes[f], hipMemcpyDeviceToDevice, ocn::as_stream(stream)));
            std::swap(d->U[f], d->A[f]);
        }
        refresh_terms(d);
    }
    return OCN_SUCCESS;
}

extern "C" int ocn_model_driver_field(ocn_model_driver_t d, int32_t f, double **field, double **G)
{
    OCN_REQUIRE(d && f >= 0 && f < d->n, "ocn_model_driver_field: null driver or field %d outside 0..%d", f, d ? d->n - 1 : -1);
    if (field) *field = d->U[f];
    if (G) *G = d->Gn[f];
    return OCN_SUCCESS;
}
