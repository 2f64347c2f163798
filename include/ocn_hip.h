/*
 * ocn_hip.h -- C ABI of libocn_hip.so, the MI355X (gfx950) implementation of the
 * NonhydrostaticModel time-stepping hot path of Oceananigans.jl (v0.96.19).
 *
 * The reference has no C plugin API: a backend plugs in by Julia multiple dispatch on the
 * architecture / array type (pattern: ext/OceananigansMetalExt.jl:11-35).  Each entry point
 * below replaces the `launch!(arch, grid, workspec, kernel!, args...)` of one reference
 * kernel (src/Utils/kernel_launching.jl:258-302) or one solver call, and takes exactly what
 * that kernel takes: parent-array device pointers, the grid description, scalars.  The
 * Julia-side `ccall` bindings a maintainer would add are shown in INTEGRATION.md.
 *
 * Conventions
 *  - Every function returns OCN_SUCCESS (0) or a negative status; ocn_last_error() returns a
 *    thread-local message (the Julia wrapper rethrows it as ArgumentError / ErrorException).
 *    The library never aborts and never falls back to a CPU path.
 *  - All pointers are DEVICE pointers to float64 unless stated.  Arrays are the OffsetArray
 *    *parents* of the reference (src/Grids/new_data.jl:36-70): column-major, x fastest, halos
 *    included.  Parent extent per dimension: N+2H, or N+1+2H for a Face-located field in a
 *    Bounded dimension (src/Grids/grid_utils.jl:66-72).  Flat dimensions have N=1, H=0.
 *  - Ownership: the caller owns every array; the library keeps no pointer past a call except
 *    inside explicit handles (ocn_poisson_t, ocn_model_t), mirroring the reference's plans which
 *    capture `storage` (src/Solvers/fft_based_poisson_solver.jl:65-67).
 *  - Ordering: work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream) and is asynchronous w.r.t. the host, like KernelAbstractions launches
 *    (src/Utils/kernel_launching.jl:252-253).  One host thread per handle.
 *  - Supported scope: RectilinearGrid, x and y regular, z regular or stretched; every combination of
 *    Periodic / Bounded / Flat topologies (LDS-tiled kernels where x and y are Periodic or rank-local
 *    FullyConnected, direction-generic kernels otherwise: csrc/general.hip); slabs of an x-partitioned
 *    (Periodic, Periodic, *), (Periodic, Bounded, Bounded) or (Bounded, Bounded, Bounded) grid; Float64.
 */
#ifndef OCN_HIP_H
#define OCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCN_SUCCESS 0
#define OCN_ERR_INVALID_ARGUMENT (-1)
#define OCN_ERR_UNSUPPORTED (-2)
#define OCN_ERR_HIP (-3)
#define OCN_ERR_ROCFFT (-4)
#define OCN_ERR_ALLOC (-5)
#define OCN_ERR_COMM (-6) /* an RCCL call failed */
#define OCN_ERR_TIMEOUT (-7) /* ocn_sync_timeout / ocn_comm_wait: the device work did not finish within the deadline */

/* topology codes: src/Grids/Grids.jl Periodic / Bounded / Flat */
#define OCN_PERIODIC 0
#define OCN_BOUNDED 1
#define OCN_FLAT 2
/* x-partitioned local grid (src/DistributedComputations/distributed_grids.jl:339-346
 * turns Periodic into FullyConnected): halos come from neighbour ranks, interior
 * arithmetic identical to Periodic. */
#define OCN_FULLY_CONNECTED 3
/* the first and the last slab of a grid whose partitioned x is Bounded (distributed_grids.jl:339-346): RightConnected = wall on the
 * west side, neighbour on the east (rank 0); LeftConnected = neighbour on the west, wall on the east (the last rank; its x-Face fields
 * carry the wall face: Nx + 1 points, grid_utils.jl:43).  "Half Bounded" topologies (inactive_node.jl:48-52): the order reduction of
 * the reconstructions, the excluded periphery, the wall fills and the active-node tests apply on the walled side only. */
#define OCN_RIGHT_CONNECTED 4
#define OCN_LEFT_CONNECTED 5

/* field location bitmask: bit0 = Face in x, bit1 = Face in y, bit2 = Face in z */
#define OCN_LOC_CCC 0
#define OCN_LOC_FCC 1 /* u */
#define OCN_LOC_CFC 2 /* v */
#define OCN_LOC_CCF 4 /* w */

/* math modes for the WENO tendency kernels (see ocn_set_math_mode) */
#define OCN_MATH_STRICT 0 /* reference evaluation order, no FMA contraction, IEEE division */
#define OCN_MATH_FAST 1   /* algebraically identical, FMA + fused-division form */
/* per-grid selection (ocn_grid.math) */
#define OCN_GRID_MATH_DEFAULT 0
#define OCN_GRID_MATH_STRICT 1
#define OCN_GRID_MATH_FAST 2

/* RectilinearGrid (src/Grids/rectilinear_grid.jl:1-23) reduced to what kernels read. */
typedef struct ocn_grid {
    int32_t Nx, Ny, Nz;  /* interior size */
    int32_t Hx, Hy, Hz;  /* halo size */
    int32_t tx, ty, tz;  /* topology codes */
    int32_t math;        /* arithmetic variant of the kernels launched for THIS grid: OCN_GRID_MATH_DEFAULT (the process default,
                          * ocn_set_math_mode), OCN_GRID_MATH_STRICT or OCN_GRID_MATH_FAST: two models of one process may differ */
    double dx, dy, dz;   /* regular spacings (Flat: 1.0); dz ignored when dzc != NULL */
    double Lx, Ly, Lz;   /* domain extents grid.Lx/Ly/Lz (Flat: 1.0), used by poisson_eigenvalues */
    const double *dzc;   /* DEVICE: Δzᵃᵃᶜ[k], element 0 <-> k = 1-Hz, length Nz+2Hz; NULL if z regular */
    const double *dzf;   /* DEVICE: Δzᵃᵃᶠ[k], element 0 <-> k = 1-Hz, length Nz+2Hz; NULL if z regular */
} ocn_grid;

const char *ocn_last_error(void);
const char *ocn_version(void);

/* ---- Architectures (src/Architectures.jl:70-146; zeros(arch,...) src/Grids/zeros_and_ones.jl:9;
 *      sync_device! src/Utils/multi_region_transformation.jl:183-186) ---- */
int ocn_device_count(int *count);
int ocn_set_device(int device);
int ocn_malloc(void **ptr, size_t bytes);            /* zero-initialised, like the reference's zeros() */
int ocn_free(void *ptr);                             /* unsafe_free! */
int ocn_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int ocn_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int ocn_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream); /* device_copy_to! */
int ocn_memset(void *ptr, int value, size_t bytes, void *stream);
int ocn_sync(void *stream);                          /* sync_device! */
/* sync_device! with a host-side deadline: polls the stream (no blocking wait) and returns OCN_ERR_TIMEOUT after `seconds` -- a
 * stuck collective of a multi-GPU run must end in an error of THIS rank, not in a hang (the caller exits non-zero; nothing is
 * retried or re-executed). */
int ocn_sync_timeout(void *stream, double seconds);
/* Measurement aid: an empty one-thread kernel (`ocn::profile_marker_kernel`) whose dispatches delimit a region of interest in
 * rocprofv3's per-dispatch output (kernel trace and --pmc passes carry no other marker: marker trace domains cannot be combined with
 * counter collection on this pool).  bench.py brackets its timed steps with two of them; tools/summarize_profile.py sums what lies
 * between.  Replaces nothing in the reference. */
int ocn_profile_marker(void *stream);

/* Process default of the arithmetic variant (every kernel compiled in two variants: tendencies, extra terms, AMD, hydrostatic momentum),
 * used by grids whose `math` field is OCN_GRID_MATH_DEFAULT.  Atomic; handles created from a grid keep that grid's `math`. */
int ocn_set_math_mode(int mode);
int ocn_get_math_mode(void);

/* ---- Halo fills: fill_halo_regions! (src/BoundaryConditions/fill_halo_regions.jl:50-67) with the
 *      default boundary conditions of src/BoundaryConditions/field_boundary_conditions.jl:15-33.
 *      `fields`/`locs` are HOST arrays of n device pointers / location masks (the tupled fill of
 *      src/Fields/field_tuples.jl:56-101).  Per field and direction: Periodic -> periodic copy over the
 *      whole parent cross-section (fill_halo_regions_periodic.jl:40-71); Bounded & Center-located ->
 *      no-flux mirror of one cell (fill_halo_regions_flux.jl:14-33); Bounded & Face-located -> the
 *      wall-normal velocity on both boundary faces is set to 0 iff fill_boundary_normal_velocities
 *      (fill_halo_regions_open.jl:9-70).  Non-periodic fills run before periodic ones
 *      (fill_halo_regions.jl:148-196).  FullyConnected directions are skipped (communication). ---- */
int ocn_fill_halo_regions(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n,
                          int32_t fill_boundary_normal_velocities, void *stream);
/* single-direction periodic fill, K18 (dir 0/1/2) */
int ocn_fill_halo_periodic(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, int32_t dir,
                           void *stream);

/* ---- Tendencies: compute_Gu!/Gv!/Gw! fused (src/Models/NonhydrostaticModels/
 *      compute_nonhydrostatic_tendencies.jl:57-179, nonhydrostatic_tendency_kernel_functions.jl:47-200)
 *      for advection = WENO() (5th order), every other term `nothing`.
 *      range = NULL -> launch!(..., :xyz; exclude_periphery=true); otherwise range[6] =
 *      {i0,i1,j0,j1,k0,k1} (1-based, inclusive) = KernelParameters, periphery NOT excluded
 *      (kernel_launching.jl:236-240), used for the interior / buffer split of
 *      src/Models/interleave_communication_and_computation.jl:29-67. ---- */
int ocn_compute_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                    double *Gv, double *Gw, const int32_t *range, void *stream);
/* Fused stage boundary of RK3 (no reference counterpart; results identical to the two calls it replaces):
 * compute_Gu!/Gv!/Gw! over :xyz followed by the NEXT stage's rk3_substep_field! (runge_kutta_3.jl:194-200),
 *   G = tendencies(u, v, w);   U_out = U + dt*(gamma*G + zeta*Gm)   (has_zeta = 0: U + (dt*gamma)*G, Gm unused;
 *   wall faces copied unchanged),
 * written to SEPARATE output velocity arrays (other workgroups still read the stencil neighbourhood of U).
 * `range` as in ocn_compute_momentum_tendencies (NULL = :xyz); only cells inside the range are written.
 * p_correct != NULL additionally folds the PREVIOUS stage's _pressure_correct_velocities! (pressure_correction.jl:31-37)
 * into the loads: every velocity value is read as  u - ((p[i]-p[i-1])/dx)*dt_correct  (v, w alike) with periodically
 * wrapped indices, so (u, v, w) are the *uncorrected* fields with valid halos and p needs no halos.
 * (Periodic, Periodic, Periodic) grids with an interior of at least 16 x 8 x 4 and range = NULL only; otherwise
 * OCN_ERR_UNSUPPORTED.  On a (FullyConnected, Periodic, Periodic) local grid of a slab-x run the x indices are not wrapped: p must
 * hold the neighbours' planes in its x halos and u[1 - Hx] must already be corrected (ocn_halo_exchange_pressure).
 * Grids with a Bounded / Flat x or y (range = NULL, p_correct = NULL): the substep is the epilogue of the tiled kernel on the interior box
 * and one more per-cell kernel on the wall frames; the wall faces of u, v, w (first index and the face N + 1) are carried over. */
int ocn_compute_momentum_tendencies_rk3(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                        double *Gv, double *Gw, const double *Gmu, const double *Gmv, const double *Gmw,
                                        double *u_out, double *v_out, double *w_out, double dt, double gamma, double zeta,
                                        int32_t has_zeta, const double *p_correct, double dt_correct, const int32_t *range,
                                        void *stream);
/* The same launch for the correction-on-load stage of ONE RANK of a slab-x run (p_correct required, full range) that additionally writes
 * the stepped velocities of its Hx westmost / eastmost columns into the send buffers of the next x-halo exchange
 * (ocn_halo_exchange_buffers: field q at q * field_doubles, then h + Hx * parent row; interior rows only), so that
 * ocn_halo_exchange_begin_packed can post the exchange without a pack launch (Fields/field_boundary_buffers.jl:276-308 fills its
 * buffers with a separate broadcast per field and side).  No reference counterpart; results identical. */
int ocn_compute_momentum_tendencies_rk3_strips(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                               double *Gv, double *Gw, const double *Gmu, const double *Gmv, const double *Gmw,
                                               double *u_out, double *v_out, double *w_out, double dt, double gamma, double zeta,
                                               int32_t has_zeta, const double *p_correct, double dt_correct, double *strip_west,
                                               double *strip_east, int64_t field_doubles, void *stream);
/* compute_Gc! (compute_nonhydrostatic_tendencies.jl:186-195; tracer_tendency :228-259) */
int ocn_compute_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w,
                                const double *c, double *Gc, const int32_t *range, void *stream);

/* ---- the other terms of u/v/w_velocity_tendency and tracer_tendency ----
 * advection schemes */
#define OCN_ADVECTION_WENO5 0     /* WENO() = WENO(order=5)             src/Advection/weno_reconstruction.jl:98-123 */
#define OCN_ADVECTION_CENTERED2 1 /* Centered() = Centered(order=2), the reference default  centered_reconstruction.jl:39-60 */
#define OCN_ADVECTION_UPWIND5 2   /* UpwindBiased(order=5)            upwind_biased_reconstruction.jl:41-140 */
/* buoyancy formulations (gravity_unit_vector = NegativeZDirection()) */
#define OCN_BUOYANCY_NONE 0
#define OCN_BUOYANCY_TRACER 1      /* BuoyancyTracer(): b = tracer `T` slot               buoyancy_tracer.jl:12 */
#define OCN_BUOYANCY_SEAWATER_TS 2 /* SeawaterBuoyancy(LinearEquationOfState): g(αT - βS) linear_equation_of_state.jl:58-60 */
#define OCN_BUOYANCY_SEAWATER_T 3  /* ... constant_salinity:     g α T                    :62-63 */
#define OCN_BUOYANCY_SEAWATER_S 4  /* ... constant_temperature: -g β S                    :65-66 */

/* The keyword arguments of NonhydrostaticModel(; advection, coriolis, closure, buoyancy) (nonhydrostatic_model.jl:114-135)
 * that change the tendency kernels, plus the device fields those terms read. */
typedef struct ocn_model_terms {
    int32_t advection; /* OCN_ADVECTION_* */
    int32_t coriolis;  /* 0 nothing, 1 FPlane(f), 2 BetaPlane(f, coriolis_beta)      src/Coriolis/f_plane.jl:44-46, beta_plane.jl:43-57 */
    int32_t closure;   /* 0 nothing, 1 ScalarDiffusivity(ν, κ): ThreeDimensionalFormulation, ExplicitTimeDiscretization,
                          constant coefficients                    abstract_scalar_diffusivity_closure.jl:158-223;
                          2 AnisotropicMinimumDissipation: the same isotropic stress with the eddy viscosity field nu_e
                          (anisotropic_minimum_dissipation.jl:20-21) */
    int32_t buoyancy;  /* OCN_BUOYANCY_* */
    double f;          /* FPlane.f or BetaPlane.f₀ */
    double nu;         /* ScalarDiffusivity.ν */
    double g, alpha, beta; /* gravitational_acceleration, thermal_expansion, haline_contraction */
    const double *T;   /* DEVICE: temperature (or the buoyancy tracer b); NULL if unused */
    const double *S;   /* DEVICE: salinity; NULL if unused */
    const double *pHY; /* DEVICE: hydrostatic pressure anomaly pHY′ (model.pressures.pHY′) or NULL = `nothing`
                          (then w receives z_dot_g_b directly, nonhydrostatic_tendency_kernel_functions.jl:141-143) */
    const double *nu_e; /* DEVICE: closure == 2: diffusivity_fields.νₑ (Center field, halos filled); else NULL */
    /* coriolis == 2: BetaPlane(f₀ = f, β = coriolis_beta), f = f₀ + β ynode (src/Coriolis/beta_plane.jl:43-57): yc / yf are DEVICE vectors
     * of the grid's y nodes at Centers / Faces WITH halos (element 0 <-> j = 1 - Hy: yᵃᶜᵃ, yᵃᶠᵃ of the reference); x_f_cross_U reads
     * y at (Face, Center, Center), y_f_cross_U at (Center, Face, Center).  Unused otherwise. */
    double coriolis_beta;
    const double *yc, *yf;
} ocn_model_terms;

/* compute_Gu!/Gv!/Gw! with every supported term:
 *   G = ((((-div_𝐯u) + x_dot_g_b) - x_f_cross_U) - ∂x pHY′) - ∂ⱼτ₁ⱼ      (nonhydrostatic_tendency_kernel_functions.jl:66-75,
 *   128-137, 193-200), evaluated in that order.  terms->advection selects the flux scheme. */
int ocn_compute_momentum_tendencies_terms(const ocn_grid *grid, const ocn_model_terms *terms, const double *u, const double *v,
                                          const double *w, double *Gu, double *Gv, double *Gw, const int32_t *range,
                                          void *stream);
/* compute_Gc! with advection scheme terms->advection and, when terms->closure != 0, the diffusive flux divergence
 *   Gc = -div_Uc - ∇_dot_qᶜ,  q = -κ ∇c      (:250-256; closure_kernel_operators.jl:48-53)
 * kappa_e != NULL (closure == 2): the tracer's eddy diffusivity field κₑ (Center, halos filled) replaces the number kappa. */
int ocn_compute_tracer_tendency_terms(const ocn_grid *grid, const ocn_model_terms *terms, double kappa, const double *kappa_e,
                                      const double *u, const double *v, const double *w, const double *c, double *Gc,
                                      const int32_t *range, void *stream);
/* compute_diffusivities!(diffusivity_fields, ::AnisotropicMinimumDissipation, model) with Cb = nothing
 * (anisotropic_minimum_dissipation.jl:125-188): νₑ = max(0, -Cν δ² r / q) and, per tracer, κₑ = max(0, -Cκ δ² ϑ / σ) over the
 * interior; the caller fills their halos afterwards (update_nonhydrostatic_model_state.jl:48).  Velocity / tracer halos
 * must be filled; z must not be Flat. */
int ocn_compute_amd_viscosity(const ocn_grid *grid, double C_nu, const double *u, const double *v, const double *w, double *nu_e,
                              void *stream);
int ocn_compute_amd_diffusivity(const ocn_grid *grid, double C_kappa, const double *u, const double *v, const double *w,
                                const double *c, double *kappa_e, void *stream);
/* Both of the above for νₑ and n_tracers <= 4 tracers in ONE launch (the velocity gradients are evaluated once). */
int ocn_compute_amd_diffusivities(const ocn_grid *grid, double C_nu, const double *u, const double *v, const double *w,
                                  double *nu_e, int32_t n_tracers, const double *C_kappa, const double *const *tracers,
                                  double *const *kappa_e, void *stream);
/* The same over the columns i_first..i_last only, which may include the halo columns 0 and Nx+1: a distributed run computes
 * the interior while its halo exchange is in flight and the edge and halo columns afterwards, from the exchanged halos
 * (compute_nonhydrostatic_buffer_tendencies.jl:55-68) -- the halo values equal what the neighbour computes for its own edge. */
int ocn_compute_amd_diffusivities_range(const ocn_grid *grid, double C_nu, const double *u, const double *v, const double *w,
                                        double *nu_e, int32_t n_tracers, const double *C_kappa, const double *const *tracers,
                                        double *const *kappa_e, int32_t i_first, int32_t i_last, void *stream);
/* update_hydrostatic_pressure! (update_hydrostatic_pressure.jl:12-53): pHY′ by downward integration of the buoyancy
 * perturbation over i in 0:Nx+1, j in 0:Ny+1 (tracer halos must be filled).  No-op on a z-Flat grid. */
int ocn_update_hydrostatic_pressure(const ocn_grid *grid, const ocn_model_terms *terms, double *pHY, void *stream);
/* columns i_first..i_last of 0..Nx+1 only (see ocn_compute_amd_diffusivities_range) */
int ocn_update_hydrostatic_pressure_range(const ocn_grid *grid, const ocn_model_terms *terms, double *pHY, int32_t i_first,
                                          int32_t i_last, void *stream);

/* Boundary conditions (src/BoundaryConditions/boundary_condition.jl).  The condition evaluates to
 *   values[(i-1) + Nx*(j-1)]              if values != NULL  (array boundary condition, DEVICE pointer), else
 *   value + coeff * c[i, j, interior]     (coeff = 0: a plain number; coeff != 0 restates the ContinuousBoundaryFunction
 *                                          f(x, y, t, c, p) = p * c with field_dependencies = the field itself, whose
 *                                          argument is c[i, j, Nz] / c[i, j, 1]: continuous_boundary_function.jl:107-115). */
#define OCN_BC_DEFAULT 0  /* Periodic / no-flux / impenetrable by topology (field_boundary_conditions.jl:15-33) */
#define OCN_BC_FLUX 1
#define OCN_BC_VALUE 2
#define OCN_BC_GRADIENT 3
#define OCN_BC_OPEN 4      /* OpenBoundaryCondition(value): the wall-normal velocity on the boundary face (fill_halo_regions_open.jl:65-70); on the
                            * side of a field that is not Face-located along it the kind is ignored */
typedef struct ocn_bc {
    int32_t kind;
    int32_t _pad;
    double value, coeff;
    const double *values;
} ocn_bc;
/* FieldBoundaryConditions of one field.  Only bottom / top may differ from OCN_BC_DEFAULT (x, y are Periodic in the
 * supported scope); w (Face in z) keeps its impenetrable default. */
typedef struct ocn_field_bcs {
    ocn_bc west, east, south, north, bottom, top;
} ocn_field_bcs;
/* fill_halo_regions! with user boundary conditions: Value / Gradient sides are linearly extrapolated into the first halo
 * cell (fill_halo_regions_value_gradient.jl:5-103), Flux sides get the no-flux fill (fill_halo_regions_flux.jl:14-33).
 * bcs[f] == NULL keeps field f on its defaults. */
int ocn_fill_halo_regions_bcs(const ocn_grid *grid, double *const *fields, const int32_t *locs,
                              const ocn_field_bcs *const *bcs, int32_t n, int32_t fill_boundary_normal_velocities,
                              void *stream);
/* compute_boundary_tendency_contributions! (compute_nonhydrostatic_tendencies.jl:204-213) = apply_z_bcs! of every field:
 *   G[i,j,1] += flux*Az/V (bottom),  G[i,j,Nz] -= flux*Az/V (top)      (apply_flux_bcs.jl:107-160) */
int ocn_apply_flux_bcs(const ocn_grid *grid, double *const *G, const double *const *fields, const int32_t *locs,
                       const ocn_field_bcs *const *bcs, int32_t n, void *stream);

/* One RK3 stage boundary of a model with the §8(f) terms in as few passes as possible (no reference counterpart; results
 * identical to ocn_compute_*_tendencies_terms + ocn_apply_flux_bcs + ocn_rk3_substep, bit for bit in strict math):
 * the tendencies of the current state INCLUDING the bottom / top flux boundary contributions land in G, and the NEXT stage's
 *   U_out = U + Δt (γ G + ζ G⁻)   (has_zeta = 0:  U_out = U + (Δt γ) G)
 * is written to a second storage (must not alias the inputs; the wall faces of w are carried over).  bcs_* may be NULL.
 * Grids with a Bounded / Flat x or y (range = NULL): the finishing pass with fluxes and substep runs inside the tiled kernel on the interior
 * box and as per-cell kernels on the wall frames (wall faces of u, v carried over too); conditions on the x / y walls may be Value /
 * Gradient / Open (they live in the halo fills) but not Flux (OCN_ERR_INVALID_ARGUMENT: ocn_apply_flux_bcs adds those, unfused). */
int ocn_compute_momentum_tendencies_terms_rk3(const ocn_grid *grid, const ocn_model_terms *terms, const ocn_field_bcs *bcs_u,
                                              const ocn_field_bcs *bcs_v, const double *u, const double *v, const double *w,
                                              double *Gu, double *Gv, double *Gw, const double *Gmu, const double *Gmv,
                                              const double *Gmw, double *u_out, double *v_out, double *w_out, double dt,
                                              double gamma, double zeta, int32_t has_zeta, const int32_t *range, void *stream);
/* same for one tracer; advection must be OCN_ADVECTION_WENO5 or OCN_ADVECTION_UPWIND5 (advection, diffusion, boundary flux
 * and substep are ONE kernel -- on grids with walls in x / y: on the interior box; three per-cell kernels on the frames) */
int ocn_compute_tracer_tendency_terms_rk3(const ocn_grid *grid, const ocn_model_terms *terms, double kappa, const double *kappa_e,
                                          const ocn_field_bcs *bcs_c, const double *u, const double *v, const double *w,
                                          const double *c, double *Gc, const double *Gmc, double *c_out, double dt, double gamma,
                                          double zeta, int32_t has_zeta, const int32_t *range, void *stream);

/* ---- NaN check ----
 * hasnan(field) = any(isnan, parent(field)) (src/Models/nan_checker.jl:33).  Scans n_elements doubles (the whole parent array)
 * and sets *flag_device (a DEVICE int32 the caller zeroed) to 1 if any is NaN; asynchronous, the caller reads the flag when
 * it wants the answer.  field must be 16-byte aligned (any ocn_malloc / torch allocation is). */
int ocn_hasnan(const double *field, int64_t n_elements, int32_t *flag_device, void *stream);

/* cell_advection_timescale(grid, velocities) (src/Advection/cell_advection_timescale.jl:13-35), the quantity behind AdvectiveCFL
 * and TimeStepWizard: min over the interior of 1 / (|u|/Δx + |v|/Δy + |w|/Δzᶜᶜᶠ).  *result_device (DEVICE double) receives the
 * minimum (+inf for a fluid at rest); asynchronous. */
int ocn_cell_advection_timescale(const ocn_grid *grid, const double *u, const double *v, const double *w, double *result_device,
                                 void *stream);

/* ---- Time steppers ----
 * rk3_substep_field! for n fields in one launch (src/TimeSteppers/runge_kutta_3.jl:160-208).
 * has_zeta = 0 selects the first-stage method  U += (Δt*γ)*Gⁿ. */
int ocn_rk3_substep(const ocn_grid *grid, int32_t n, double *const *U, const double *const *Gn,
                    const double *const *Gm, const int32_t *locs, double dt, double gamma, double zeta, int32_t has_zeta,
                    void *stream);
/* split_rk3_substep_field! of the HydrostaticFreeSurfaceModel's SplitRungeKutta3 stepper, stages 2 and 3
 * (hydrostatic_free_surface_rk3_step.jl:30-60):  U = zeta * Psi + gamma * (U + dt * G)  over the interior of every field of the tuple,
 * Psi = the field cached at the start of the step (cache_previous_fields!); stage 1 is ocn_rk3_substep with gamma = 1, has_zeta = 0. */
int ocn_split_rk3_substep(const ocn_grid *grid, int32_t n, double *const *U, const double *const *G, const double *const *Psi,
                          const int32_t *locs, double dt, double gamma, double zeta, void *stream);
/* ab2_step_field! (src/TimeSteppers/quasi_adams_bashforth_2.jl:128-175) */
int ocn_ab2_step(const ocn_grid *grid, int32_t n, double *const *U, const double *const *Gn, const double *const *Gm,
                 const int32_t *locs, double dt, double chi, void *stream);
/* cache_previous_tendencies! (src/TimeSteppers/store_tendencies.jl:6-22) */
int ocn_cache_previous_tendencies(const ocn_grid *grid, int32_t n, double *const *Gm, const double *const *Gn,
                                  const int32_t *locs, void *stream);

/* ---- Pressure ----
 * _pressure_correct_velocities! (src/Models/NonhydrostaticModels/pressure_correction.jl:31-50) */
int ocn_pressure_correct_velocities(const ocn_grid *grid, double *u, double *v, double *w, const double *p, double dt,
                                    void *stream);
/* divᶜᶜᶜ into a halo-free Nx*Ny*Nz array (src/Operators/divergence_operators.jl:16-19) */
int ocn_divergence(const ocn_grid *grid, const double *u, const double *v, const double *w, double *div, void *stream);

/* Poisson solver handle = FFTBasedPoissonSolver (src/Solvers/fft_based_poisson_solver.jl:5-125) when z is
 * regular and Periodic/Flat, FourierTridiagonalPoissonSolver (fourier_tridiagonal_poisson_solver.jl:6-147)
 * when z is Bounded (regular or stretched) -- the dispatch of
 * src/Models/NonhydrostaticModels/NonhydrostaticModels.jl:25-62.  Grids with a Bounded / Flat x or y: the FFT-based
 * solver with cosine transforms when every direction is regular, the Fourier-tridiagonal solver (cosine / Fourier
 * transforms along x and y, Thomas sweep along z) when z is stretched -- or when the environment variable
 * OCN_POISSON_GENERAL_TRI=1 asks for it on a regular Bounded z.  The handle owns its storage and rocFFT plans. */
typedef struct ocn_poisson *ocn_poisson_t;
int ocn_poisson_create(ocn_poisson_t *solver, const ocn_grid *grid);
int ocn_poisson_destroy(ocn_poisson_t solver);
/* introspection: kind 0 = FFT-based, 1 = Fourier-tridiagonal, 2 = FFT-based with cosine transforms (a Bounded / Flat x or y),
 * 3 = Fourier-tridiagonal on such a grid; r2c = real-to-complex transforms in use;
 * direct_out = bit 0: inverse transform writes straight into the pressure interior (no copy_real_component! pass); bit 1: the fused
 * FFT_z / division / IFFT_z column pass; bit 2: the library's own row / column kernels for x and y; bit 3: kind 1 on a REGULAR z of a
 * column-kernel length -- the Thomas sweep replaced by its exact spectral twin, cosine transform / division / inverse cosine transform in
 * one column pass (what the reference's FFTBasedPoissonSolver does on such a grid; OCN_POISSON_DCT_Z=0 keeps the sweep) */
int ocn_poisson_info(ocn_poisson_t solver, int32_t *kind, int32_t *r2c, int32_t *direct_out);
/* compute_source_term! (src/Models/NonhydrostaticModels/solve_for_pressure.jl:12-17,33-38,57-76) */
int ocn_poisson_compute_source_term(ocn_poisson_t solver, const double *u, const double *v, const double *w, double dt,
                                    void *stream);
/* set the source term from a halo-free real Nx*Ny*Nz array R (tests: ∇²ϕ = R;
 * set_source_term! fourier_tridiagonal_poisson_solver.jl:155-161 applies Δzᶜ) */
int ocn_poisson_set_source_term(ocn_poisson_t solver, const double *R, void *stream);
/* solve!(ϕ, solver): writes the interior of the haloed pressure field p (fft_based_poisson_solver.jl:95-125) */
int ocn_poisson_solve(ocn_poisson_t solver, double *p, void *stream);
/* solve_for_pressure! = source term + solve (solve_for_pressure.jl:78-82) */
/* solve!(phi, solver, b, m) with m != 0: (laplacian + m) phi = b, no zero-mode gauge; FFT-based handles on the plain transform path */
int ocn_poisson_solve_shifted(ocn_poisson_t solver, double *phi, double m, void *stream);
int ocn_solve_for_pressure(ocn_poisson_t solver, double *p, const double *u, const double *v, const double *w, double dt,
                           void *stream);

/* ---- HydrostaticFreeSurfaceModel, first slice: explicit free surface, static (Periodic, Periodic,
 * Bounded) grid.  u, v tendencies = ocn_compute_momentum_tendencies (flux-form advection; Gw is not used), then
 * ocn_add_barotropic_pressure_gradient, then ocn_add_momentum_terms -- the order of the reference's sum
 * (hydrostatic_free_surface_tendency_kernel_functions.jl:45-52).  eta / G_eta are (Nx+2Hx) x (Ny+2Hy) planes, x fastest. */
int ocn_add_momentum_terms(const ocn_grid *grid, const ocn_model_terms *terms, const double *u, const double *v, const double *w,
                           double *Gu, double *Gv, double *Gw, const int32_t *range, void *stream);
/* Gu = -U_dot_grad(u), Gv = -U_dot_grad(v) with momentum_advection = VectorInvariant() (the model's default: EnstrophyConserving
 * vorticity flux, EnergyConserving vertical advection and kinetic-energy gradient; Advection/vector_invariant_advection.jl:269-361);
 * with eta, the barotropic pressure gradient of ocn_add_barotropic_pressure_gradient is subtracted in the same pass */
int ocn_compute_vector_invariant_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                                                     double *Gv, const double *eta /* or NULL: without - g grad(eta) */,
                                                     double gravitational_acceleration, void *stream);
/* SplitExplicitFreeSurface with the ForwardBackwardScheme (the SplitExplicitFreeSurfaces directory), (Periodic, Periodic) static grid of column
 * depth H; every 2-D quantity is an (Nx+2Hx) x (Ny+2Hy) plane like eta, of which only the interior is used.
 *   forcing:   compute_split_explicit_forcing! (compute_slow_tendencies.jl:12-32)
 *   substeps:  initialize_free_surface_state! + iterate_split_explicit! + _update_split_explicit_state!
 *              (step_split_explicit_free_surface.jl:3-108); `weights`: n averaging weights, HOST array; dtau = fractional step x dt
 *   mode:      integrate_barotropic_mode! (initialize_free_surface!)
 *   corrector: barotropic_split_explicit_corrector! (barotropic_split_explicit_corrector.jl:44-71) */
int ocn_split_explicit_forcing(const ocn_grid *grid, const double *Gu, const double *Gu_previous, const double *Gv, const double *Gv_previous,
                               double chi, double *GU, double *GV, void *stream);
int ocn_split_explicit_substeps(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                double column_depth, double *eta, double *U, double *V, double *eta_filtered, double *U_filtered,
                                double *V_filtered, const double *GU, const double *GV, void *stream);
int ocn_compute_barotropic_mode(const ocn_grid *grid, const double *u, const double *v, double *U, double *V, void *stream);
int ocn_barotropic_split_explicit_corrector(const ocn_grid *grid, double *u, double *v, const double *U, const double *V, double *U_filtered,
                                            double *V_filtered, double column_depth, void *stream);
/* The same substep loop, temporally blocked: ceil(n / 4) launches instead of 2 n (LDS patches with a 4-cell rim run 4 substeps each);
 * bit-identical results.  `work`: device scratch of 3 planes (3 (Nx+2Hx)(Ny+2Hy) doubles).
 * Replaces iterate_split_explicit! + _update_split_explicit_state! (step_split_explicit_free_surface.jl:62-108). */
int ocn_split_explicit_substeps_blocked(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                        double column_depth, double *eta, double *U, double *V, double *eta_filtered, double *U_filtered,
                                        double *V_filtered, const double *GU, const double *GV, double *work, void *stream);
/* ocn_compute_tracer_tendency_terms_rk3 for TWO tracers in one launch (T and S): u, v, w at the faces, the upwind directions, the metrics
 * and the tile staging are shared; per tracer the arithmetic is unchanged (bit-identical to two separate launches).  Every per-tracer
 * argument is a HOST array of 2.  *launched = 0 when the range is too small for the tiled kernel (nothing was done: call the
 * single-tracer entry twice). */
int ocn_compute_tracer_pair_tendency_terms_rk3(const ocn_grid *grid, const ocn_model_terms *terms, const double *kappa, const double *const *kappa_e,
                                               const ocn_field_bcs *const *bcs_c, const double *u, const double *v, const double *w,
                                               const double *const *c, double *const *Gc, const double *const *Gc_previous, double *const *c_out,
                                               double dt, double gamma, double zeta, int32_t has_zeta, const int32_t *range,
                                               int32_t *launched, void *stream);
/* ImplicitFreeSurface with the FFT solver (the reference's default free surface on an xy-regular RectilinearGrid;
 * implicit_free_surface.jl:112-145, fft_based_implicit_free_surface_solver.jl:76-115, barotropic_pressure_correction.jl:21-47):
 *   ocn_implicit_free_surface_rhs: compute_vertically_integrated_volume_flux! (Qu = sum_k Ax u, Qv = sum_k Ay v into (sx, sy) planes) and
 *     fft_implicit_free_surface_right_hand_side!: rhs = (dx Qu + dy Qv - Az eta / dt) / (g Lz dt Az) into a halo-free (Nx, Ny) array;
 *   the solve is ocn_poisson_set_source_term(rhs) + ocn_poisson_solve_shifted(handle of the (TX, TY, Flat) horizontal grid, eta, m)
 *     with m = -1 / (g Lz dt^2): solve!(phi, solver, b, m) of the screened equation (fft_based_poisson_solver.jl:95-125);
 *   ocn_barotropic_pressure_correction: u -= g dt dx(eta), v -= g dt dy(eta) at every level (eta halos filled). */
int ocn_implicit_free_surface_rhs(const ocn_grid *grid, const double *u, const double *v, const double *eta, double gravitational_acceleration,
                                  double dt, double *Qu, double *Qv, double *rhs, void *stream);
int ocn_barotropic_pressure_correction(const ocn_grid *grid, double *u, double *v, const double *eta, double gravitational_acceleration,
                                       double dt, void *stream);
/* The substep loop with timestepper = AdamsBashforth3Scheme() (split_explicit_timesteppers.jl:19-159; the reference's two launches per
 * substep).  coefficients: HOST array {alpha, theta, beta, delta, mu, gamma, epsilon}; work: 7 planes (the scheme's history fields,
 * re-initialised from the current state at every call as initialize_free_surface_timestepper! does). */
int ocn_split_explicit_substeps_ab3(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                    double column_depth, const double *coefficients, double *eta, double *U, double *V, double *eta_filtered,
                                    double *U_filtered, double *V_filtered, const double *GU, const double *GV, double *work, void *stream);
/* The substep loop on a slab-x rank: DistributedSplitExplicitFreeSurface (distributed_split_explicit_free_surface.jl: the x halos of
 * eta, U, V, GU, GV are extended to the number of substeps and filled ONCE per baroclinic step; the substeps then run without
 * communication over ranges that reach into the halos).  W = n (the number of substeps; needs W <= Nx of the slab).
 *   _begin: gathers the interiors into wide work planes and packs the W-wide west / east strips of the 5 planes
 *           (5 W Ny doubles per side) -> exchange them with the x neighbours (ocn_comm_exchange_strips, or any transport)
 *   _run:   received strips -> wide halos, temporally blocked substeps over shrinking ranges, interior averages -> eta, U, V.
 * work: 11 (Nx + 2 W) Ny doubles.  Interior results are bit-identical to the single-rank substepping. */
int ocn_split_explicit_dist_begin(const ocn_grid *grid, int32_t n, const double *eta, const double *U, const double *V, const double *GU,
                                  const double *GV, double *work, double *send_west, double *send_east, void *stream);
int ocn_split_explicit_dist_run(const ocn_grid *grid, int32_t n, const double *weights, double dtau, double gravitational_acceleration,
                                double column_depth, double *eta, double *U, double *V, double *work, const double *recv_west,
                                const double *recv_east, void *stream);
/* One pass over the columns for the whole horizontal-momentum part of a QuasiAdamsBashforth2 step of the HydrostaticFreeSurfaceModel
 * with momentum_advection = VectorInvariant():
 *   compute_hydrostatic_free_surface_Gu!/Gv! (hydrostatic_free_surface_tendency_kernel_functions.jl:29-97; all terms of `terms` and the
 *     flux conditions of bcs_u / bcs_v; - g grad(eta) when eta != NULL, i.e. with an ExplicitFreeSurface)        -> Gu, Gv
 *   ab2_step_velocities! (hydrostatic_free_surface_ab2_step.jl:41-63): u_out = u + dt ((3/2 + chi) G - (1/2 + chi) G_previous), Euler
 *     (chi = -1/2, G_previous not read) when euler != 0; u_out / v_out must not alias u / v
 *   GU, GV != NULL (SplitExplicitFreeSurface): compute_split_explicit_forcing! (compute_slow_tendencies.jl:12-46) -> GU, GV and
 *     compute_barotropic_mode! of the stepped velocities (barotropic_split_explicit_corrector.jl:13-32) -> U_star, V_star
 * Strict math: bit-identical to the separate entry points. */
int ocn_hydrostatic_momentum_ab2_step(const ocn_grid *grid, const ocn_model_terms *terms, const ocn_field_bcs *bcs_u,
                                      const ocn_field_bcs *bcs_v, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                                      const double *Gu_previous, const double *Gv_previous, double *u_out, double *v_out, double dt,
                                      double chi, int32_t euler, const double *eta, double gravitational_acceleration, double *GU,
                                      double *GV, double *U_star, double *V_star, void *stream);
/* barotropic_split_explicit_corrector! (barotropic_split_explicit_corrector.jl:44-71) + compute_w_from_continuity!
 * (compute_w_from_continuity.jl:31-40) in one pass: u = u_star + (U - U_star) / H, v likewise (written to u, v, which must not alias
 * u_star, v_star), w from the corrected velocities for the interior columns (the caller's halo fill of w supplies the halo columns).
 * U == NULL (ExplicitFreeSurface): u = u_star, no correction. */
int ocn_barotropic_corrector_and_w(const ocn_grid *grid, const double *u_star, const double *v_star, double *u, double *v, double *w,
                                   const double *U, const double *V, const double *U_star, const double *V_star, double column_depth,
                                   void *stream);
/* fill_halo_regions!(eta): periodic x, y halos of the free-surface plane */
int ocn_fill_free_surface_halos(const ocn_grid *grid, double *eta, void *stream);
/* _compute_w_from_continuity! (compute_w_from_continuity.jl:31-40) for every parent column with east / north neighbours */
int ocn_compute_w_from_continuity(const ocn_grid *grid, const double *u, const double *v, double *w, void *stream);
/* Gu -= g dx(eta), Gv -= g dy(eta) (explicit_free_surface.jl:36-40) */
int ocn_add_barotropic_pressure_gradient(const ocn_grid *grid, double gravitational_acceleration, const double *eta, double *Gu, double *Gv,
                                         void *stream);
/* G_eta = w[:, :, Nz+1] (explicit_free_surface.jl:98-140), then eta += dt ((1.5 + chi) G_eta - (0.5 + chi) G_eta_previous not_euler) (:84-96) */
int ocn_explicit_free_surface_ab2_step(const ocn_grid *grid, const double *w, double *eta, double *G_eta, const double *G_eta_previous,
                                       double dt, double chi, void *stream);

/* ---- time_step!(model, Δt) of the RungeKutta3 NonhydrostaticModel in ONE call (src/TimeSteppers/runge_kutta_3.jl:77-151):
 * WENO5 advection, no tracers / extra terms, one GPU: (Periodic, Periodic, Periodic | Bounded | Flat), and since round 4 grids with a
 * Bounded / Flat x or y (channels, closed boxes, slices: `solver` then is the general / Fourier-tridiagonal handle of that grid, the
 * fused launch runs the tiled epilogue on the interior box and a finishing kernel on the wall frames).  The handle owns a second
 * set of velocity arrays, G^n, G^- and the pressure solver, and alternates their roles so that the stage boundaries run fused
 * (DESIGN.md section 5); it issues the same entry points in the same order as the Python host and is bit-identical to it.
 *   create:    u, v, w, p = the caller's parent arrays (initial velocities in the interiors); fills their halos.  `solver`: an
 *              existing ocn_poisson_t of the same grid to borrow, or NULL.
 *   time_step: one RK3 step; afterwards the velocities live EITHER in the caller's arrays or in the second set, and the last
 *              compute_tendencies! is deferred into the next step.
 *   flush:     completes the deferred tendencies and brings the velocities into the caller's arrays (one copy, if needed);
 *              call it before reading u, v, w, G^n on the host side or handing them to other code.
 *   fields:    where the velocities and G^n are right now (device pointers, valid until the next call). */
typedef struct ocn_rk3_driver *ocn_rk3_driver_t;
int ocn_rk3_driver_create(ocn_rk3_driver_t *driver, const ocn_grid *grid, double *u, double *v, double *w, double *p,
                          ocn_poisson_t solver /* NULL: the handle creates its own */, void *stream);
int ocn_rk3_driver_destroy(ocn_rk3_driver_t driver);
int ocn_rk3_driver_time_step(ocn_rk3_driver_t driver, double dt, void *stream);
int ocn_rk3_driver_flush(ocn_rk3_driver_t driver, void *stream);
int ocn_rk3_driver_fields(ocn_rk3_driver_t driver, double **u, double **v, double **w, double **Gu, double **Gv, double **Gw);
/* defer_correction != 0 (the default on all-periodic grids; OCN_DRIVER_DEFER_CORRECTION=0 in the environment changes the default):
 * pressure_correct_velocities! of the THIRD stage is not launched either -- the next step's first fused launch applies it on load
 * like stages 1 and 2 (three identical stage boundaries per step, no pressure-correction pass, no halo fill after it).  Between
 * time_step and flush the velocity arrays then hold the uncorrected u*, v*, w*; ocn_rk3_driver_flush applies the correction, fills
 * the halos and completes the tendencies -- the state it leaves is the reference's, bit for bit in strict math. */
int ocn_rk3_driver_configure(ocn_rk3_driver_t driver, int32_t defer_correction);

/* ---- the same for a model with tracers and the SURVEY 8(f) terms (config 4's term set: WENO5 / UpwindBiased5 advection, FPlane,
 * ScalarDiffusivity or AnisotropicMinimumDissipation, BuoyancyTracer / SeawaterBuoyancy with or without a separate pHY', bottom / top
 * boundary conditions), (Periodic, Periodic, Bounded | Periodic | Flat), one GPU: time_step!(model, dt) of
 * runge_kutta_3.jl:77-151 with update_state! (update_nonhydrostatic_model_state.jl:20-70: halo fills, compute_auxiliaries!, the diffusivity
 * halo fill), compute_tendencies! (+ boundary contributions) and the pressure projection issued by the library -- the launch sequence of
 * the Python host (models.py::_time_step_rk3 on the general fused path), bit-identical to it.  The handle owns a second set of every
 * prognostic array, G^n and G^-; nu_e, kappa_e, pHY and p stay the caller's.  Number-valued and array-valued conditions are supported
 * (the arrays are the caller's device arrays, read at every stage); conditions that are functions of time must be refreshed by the
 * caller between steps (they are then piecewise constant over a step, unlike the reference, which evaluates them per stage).
 * One GPU: Periodic x and y, and since round 4 grids with a Bounded / Flat x or y (the fused stage boundaries of such grids; no flux
 * condition on an x / y wall, no array-valued bottom / top flux next to x walls: OCN_ERR_INVALID_ARGUMENT at creation). */
#define OCN_MODEL_MAX_TRACERS 4
typedef struct ocn_model_driver_desc {
    ocn_model_terms terms;    /* terms.T / terms.S are ignored: tracer_T / tracer_S below name the tracers the buoyancy reads */
    int32_t n_tracers;        /* 0 .. OCN_MODEL_MAX_TRACERS */
    int32_t tracer_T, tracer_S; /* index into tracers[] of the temperature (or buoyancy tracer) and of the salinity; -1 = unused */
    int32_t _pad;
    double kappa[OCN_MODEL_MAX_TRACERS];   /* ScalarDiffusivity kappa of every tracer (closure == 1) */
    double C_nu, C_kappa[OCN_MODEL_MAX_TRACERS]; /* AnisotropicMinimumDissipation Cnu and per-tracer Ckappa (closure == 2) */
    double *tracers[OCN_MODEL_MAX_TRACERS];      /* DEVICE: the caller's tracer parents */
    double *nu_e, *kappa_e[OCN_MODEL_MAX_TRACERS]; /* DEVICE: diffusivity fields (closure == 2), default boundary conditions */
    double *pHY;              /* DEVICE: hydrostatic pressure anomaly or NULL (must equal terms.pHY) */
    const ocn_field_bcs *bcs[3 + OCN_MODEL_MAX_TRACERS]; /* u, v, w, tracers...; NULL = defaults.  Copied at creation. */
} ocn_model_driver_desc;
typedef struct ocn_model_driver *ocn_model_driver_t;
int ocn_model_driver_create(ocn_model_driver_t *driver, const ocn_grid *grid, const ocn_model_driver_desc *desc, double *u, double *v,
                            double *w, double *p, ocn_poisson_t solver /* NULL: the handle creates its own */, void *stream);
int ocn_model_driver_destroy(ocn_model_driver_t driver);
int ocn_model_driver_time_step(ocn_model_driver_t driver, double dt, void *stream);
/* completes the deferred compute_tendencies! and brings every prognostic field into the caller's arrays */
int ocn_model_driver_flush(ocn_model_driver_t driver, void *stream);
/* where field f (0, 1, 2 = u, v, w; 3 + n = tracer n) and its G^n are right now */
int ocn_model_driver_field(ocn_model_driver_t driver, int32_t f, double **field, double **G);

/* solve!(ϕ, ::BatchedTridiagonalSolver, rhs), z direction (src/Solvers/batched_tridiagonal_solver.jl:100-123,
 * 203-235).  a, c: real Nz-1; b: real Nx*Ny*Nz; f, phi: complex interleaved Nx*Ny*Nz; t: real scratch. */
int ocn_batched_tridiagonal_solve_z(int32_t Nx, int32_t Ny, int32_t Nz, const double *a, const double *b, const double *c,
                                    const double *f, double *t, double *phi, void *stream);

/* ---- Distributed slab-x support (src/DistributedComputations/) ----
 * Halo staging buffers (src/Fields/field_boundary_buffers.jl:276-308): pack send-west = parent[1+Hx:2Hx,:,:],
 * send-east = parent[1+nx:nx+Hx,:,:] into dense (Hx, sy, sz) buffers; unpack recv-west -> parent[1:Hx,:,:],
 * recv-east -> parent[1+nx+Hx:nx+2Hx,:,:]. */
int ocn_halo_pack_x(const ocn_grid *grid, const double *field, int32_t loc, double *send_west, double *send_east,
                    void *stream);
int ocn_halo_unpack_x(const ocn_grid *grid, double *field, int32_t loc, const double *recv_west, const double *recv_east,
                      void *stream);
/* One x-plane with its full (y, z) cross-section.  which = 0 (west): pack parent[1+Hx,:,:] (first interior plane), unpack into
 * parent[Hx,:,:] (the halo plane next to it); which = 1 (east): pack parent[nx+Hx,:,:], unpack into parent[1+nx+Hx,:,:].
 * What divᶜᶜᶜ needs of u (its east neighbour plane) and ∂xᶠᶜᶜ of the pressure (its west one): the projection's two
 * synchronous halo fills (pressure_correction.jl:10-17) then move one plane instead of 2 Hx per field; the complete fill of
 * the same fields follows in update_state!. */
int ocn_halo_plane_x(const ocn_grid *grid, double *field, int32_t loc, int32_t which, double *buffer, int32_t unpack, void *stream);
/* x-halo payload of the correction-on-load stage of a slab-x rank (no reference counterpart; results identical to
 * fill_halo_regions!(pNHS) + pressure_correct_velocities! + fill_halo_regions!(velocities), pressure_correction.jl:8-50 and
 * update_nonhydrostatic_model_state.jl:33-37).  After solve_for_pressure! the rank sends, per x neighbour, ONE message of
 * (Hx + 1) values per row of the (y, z) cross-section: the Hx pressure planes next to the interface and -- eastwards only -- the u plane
 * i = nx - Hx + 1 with the pressure correction of this stage already applied (the receiver's westmost halo column of u, whose
 * correction would need p[-Hx]).  ocn_compute_momentum_tendencies_rk3 with p_correct on a FullyConnected x then corrects everything
 * else on load.  pack: send buffers of (Hx + 1) * sy * sz doubles each; unpack: the neighbour's send_east arrives as recv_west and
 * vice versa.  (Periodic y, z; nx >= Hx + 1.)  u is the UNCORRECTED u* of this stage with its x halos already exchanged. */
int ocn_halo_pack_pressure(const ocn_grid *grid, const double *p, const double *u, double dt_correct, double *send_west,
                           double *send_east, void *stream);
int ocn_halo_unpack_pressure(const ocn_grid *grid, double *p, double *u, const double *recv_west, const double *recv_east,
                             void *stream);
/* The same for a tuple of fields in one launch: the strips of the fields follow one another in the buffers (field q starts at
 * Hx * sum_{r<q} sy_r sz_r), so the exchange is one message per neighbour for the whole tuple
 * (fill_halo_regions! of a tuple, src/DistributedComputations/halo_communication.jl:95-128). */
int ocn_halo_pack_x_fields(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, double *send_west,
                           double *send_east, void *stream);
int ocn_halo_unpack_x_fields(const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, const double *recv_west,
                             const double *recv_east, void *stream);
/* Transposes between the y-local layout (nx, Ny, Nz) and the x-local layout (Nx, ny, Nz) of complex data
 * (src/DistributedComputations/distributed_transpose.jl:25-95), R = number of ranks:
 * pack_y_to_x fills the send buffer (chunk m = j in [m*ny, (m+1)*ny)), unpack_x_from_y reads the received one. */
int ocn_transpose_pack_y_to_x(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *yfield, double *send,
                              void *stream);
int ocn_transpose_unpack_x_from_y(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *recv, double *xfield,
                                  void *stream);
int ocn_transpose_pack_x_to_y(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *xfield, double *send,
                              void *stream);
int ocn_transpose_unpack_y_from_x(int32_t nx, int32_t Ny, int32_t Nz, int32_t R, const double *recv, double *yfield,
                                  void *stream);

/* Distributed FFT-based Poisson solver pieces (distributed_fft_based_poisson_solver.jl:141-178), slab-x:
 * the handle plans  FFT_z,FFT_y on (nx,Ny,Nz)  and  FFT_x on (Nx,ny,Nz); the host moves data between the
 * two layouts with the transposes above and an all-to-all. */
typedef struct ocn_dist_poisson *ocn_dist_poisson_t;
/* global_Lx: extent of the *global* domain in x (global_grid.Lx), used for the global eigenvalues λx (:104-106) */
int ocn_dist_poisson_create(ocn_dist_poisson_t *solver, const ocn_grid *local_grid, int32_t rank, int32_t nranks,
                            double global_Lx);
/* ... with the topology of the GLOBAL x direction said (the local grids of a Bounded x are RightConnected / FullyConnected /
 * LeftConnected slabs): global_tx = OCN_PERIODIC is ocn_dist_poisson_create; OCN_BOUNDED needs Bounded y and z
 * (distributed_fft_based_poisson_solver.jl:62-66) and takes cosine transforms along x after the transpose. */
int ocn_dist_poisson_create_global(ocn_dist_poisson_t *solver, const ocn_grid *local_grid, int32_t rank, int32_t nranks,
                                   double global_Lx, int32_t global_tx);
int ocn_dist_poisson_destroy(ocn_dist_poisson_t solver);
/* device pointers to the solver's y-local field, x-local field and the two transpose buffers */
int ocn_dist_poisson_buffers(ocn_dist_poisson_t solver, double **yfield, double **xfield, double **send, double **recv);
/* y extent of the transposed complex data (Ny, or Ny/2+1 padded to a multiple of nranks when real-to-complex transforms
 * are in use), the number of complex elements of each of the four buffers, and whether r2c is active */
int ocn_dist_poisson_layout(ocn_dist_poisson_t solver, int32_t *ny_transposed, int64_t *complex_elements, int32_t *r2c);
/* fast != 0: the handle runs the slab pipeline (no pack / unpack passes): forward_yz leaves the all-to-all payload in `send`
 * (chunk d goes to rank d); exchange send -> recv; then
 *   fast == 1 (periodic z): solve_x works in place on `recv`; exchange recv -> send; backward_yz reads `send`
 *             (payload partitioned by stored kz position, Nz / nranks per rank);
 *   fast == 2 (Bounded z, Fourier-tridiagonal): solve_x reads `recv` and leaves the solution in `send`; exchange send -> recv
 *             again; backward_yz reads `recv` (payload partitioned by ky, ceil((Ny/2+1) / nranks) per rank, zero padded);
 *   fast == 3 (periodic z, nranks > 1 by default; OCN_DIST_POISSON_XTRI=0/1): NO transposes.  forward_yz transforms y and z in
 *             place and solves the local x blocks of the cyclic tridiagonal systems p[i-1] - (2 + dx^2 (ly + lz)) p[i] + p[i+1] =
 *             dx^2 F[i] (the same operator the reference inverts with FFT_x and the division by lx + ly + lz,
 *             distributed_fft_based_poisson_solver.jl:141-178, poisson_eigenvalues.jl:8-31); ONE all-gather of the gather `send`
 *             buffer (2 complex numbers per (ky, kz) mode + the mean mode's line) into the gather `recv` buffer
 *             (ocn_dist_poisson_gather_buffers; ocn_dist_poisson_exchange direction 0, direction 1 is a no-op); solve_x solves the
 *             interface systems and corrects the local blocks; backward_yz inverts z and y in place.  csrc/xtri.hip. */
int ocn_dist_poisson_pipeline(ocn_dist_poisson_t solver, int32_t *fast);
/* fast == 3: the all-gather buffers (send: doubles_per_rank doubles; recv: nranks x doubles_per_rank, chunk s from rank s) */
int ocn_dist_poisson_gather_buffers(ocn_dist_poisson_t solver, double **send, double **recv, int64_t *doubles_per_rank);
int ocn_dist_poisson_source_term(ocn_dist_poisson_t solver, const double *u, const double *v, const double *w, double dt,
                                 void *stream);
int ocn_dist_poisson_forward_yz(ocn_dist_poisson_t solver, void *stream);
int ocn_dist_poisson_solve_x(ocn_dist_poisson_t solver, void *stream); /* FFT_x, divide by eigenvalues, IFFT_x */
int ocn_dist_poisson_backward_yz(ocn_dist_poisson_t solver, double *p, void *stream);

/* ---- Collectives of the slab-x decomposition: RCCL (librccl, linked directly) over xGMI, one communicator per process / GPU.
 *      Replaces the MPI.jl calls of src/DistributedComputations/: Distributed(...) set-up (distributed_architectures.jl:167-297),
 *      fill_halo_event! / synchronize_communication! west-east (halo_communication.jl:210-229, 267-366; distributed_fields.jl:58-75;
 *      staging buffers Fields/field_boundary_buffers.jl:276-308), Alltoallv! of the transposes (distributed_transpose.jl:185-191),
 *      Allreduce of scalars (Simulations/simulation.jl:128-134).  A Julia host therefore needs no MPI.jl in the time-stepping
 *      loop: only the 128-byte unique id has to reach every rank once (any channel: a file, a socket, MPI.bcast).
 *      Ordering: exchanges run on the communicator's own stream and are ordered against the caller's `stream` with events;
 *      no call blocks the host except ocn_comm_barrier. ---- */
typedef void *ocn_comm_t;
#define OCN_COMM_UNIQUE_ID_BYTES 128
#define OCN_COMM_MAX_RANKS 64
/* The point-to-point schedule a rank issues inside ONE RCCL group, as a pure host function (no device, no communicator): what
 * ocn_halo_exchange_begin / ocn_comm_exchange_strips (OCN_SCHED_STRIPS: slots 0 send_west, 1 send_east, 2 recv_west, 3 recv_east),
 * ocn_halo_exchange_plane (OCN_SCHED_PLANE_EAST / _WEST: slot 0 send, 1 recv), ocn_comm_all_to_all (OCN_SCHED_ALL_TO_ALL: slot =
 * chunk index) and ocn_comm_all_gather (OCN_SCHED_ALL_GATHER: send slot 0 to every peer, receive slot s from rank s; the default
 * on a node's fully connected xGMI links, OCN_COMM_ALL_GATHER=collective selects ncclAllGather) execute, in issue order.  RCCL pairs the k-th send of rank a to rank b with the k-th receive of b from a; the test
 * suite replays the schedules of all ranks for R = 1, 2, 3, 8 and checks every pairing (the MPI tags of
 * halo_communication.jl:100-150 do this job in the reference).  self_via_rccl: a rank's transfers to itself are sends too. */
#define OCN_SCHED_STRIPS 0
#define OCN_SCHED_PLANE_EAST 1
#define OCN_SCHED_PLANE_WEST 2
#define OCN_SCHED_ALL_TO_ALL 3
#define OCN_SCHED_ALL_GATHER 4
typedef struct ocn_comm_op {
    int32_t is_recv; /* 0 send, 1 receive */
    int32_t peer;    /* rank */
    int32_t slot;    /* which buffer (see above) */
} ocn_comm_op;
int ocn_comm_schedule(int32_t kind, int32_t rank, int32_t nranks, int32_t self_via_rccl, ocn_comm_op *ops, int32_t capacity,
                      int32_t *n_ops);
int ocn_comm_unique_id(void *id_out);                       /* rank 0: ncclGetUniqueId -> 128 bytes for every rank */
int ocn_comm_init(ocn_comm_t *comm, int32_t rank, int32_t nranks, const void *unique_id); /* on the current device; collective */
/* An in-process transport instead of RCCL, for ranks that are THREADS of one process sharing ONE GPU (RCCL refuses two ranks on one
 * device): every entry point below then runs unchanged with R = 2, 4, 8 on a one-GPU box -- the send / recv schedules, pack / unpack,
 * event ordering, the pressure-plane exchange, the distributed drivers.  Staging copies through per-pair mailboxes with RCCL's pairing
 * rule (the k-th send a -> b meets the k-th receive of b from a); host-blocking; not a product path.  All ranks pass the same group_key;
 * the call returns when all have joined.  ocn_comm_info reports rccl_version 0. */
int ocn_comm_init_local(ocn_comm_t *comm, int32_t rank, int32_t nranks, int64_t group_key);
/* Measurement transport: this process is rank 0 of `nranks` IDENTICAL ranks (an x-periodic flow of period Lx / nranks).  No peer
 * exists: what a peer would send is what this rank sends to the peer's mirror image, so every receive of a schedule is an asynchronous
 * device copy from one of the rank's own send buffers.  The schedules, pack / unpack launches, stream ordering, the nranks-rank
 * interface systems of the transpose-free pressure solve and the C drivers are those of a real nranks-rank run: the time of a step is
 * what ONE rank of nranks costs before any link time (tools/bench_dist_rank.py).  All-to-all exchanges are refused (OCN_ERR_UNSUPPORTED):
 * their chunks are addressed by absolute rank and have no mirror image.
 * Replaces nothing in the reference. */
int ocn_comm_init_replica(ocn_comm_t *comm, int32_t nranks);
int ocn_comm_destroy(ocn_comm_t comm);
/* rank, the number of ranks RCCL itself reports (ncclCommCount), RCCL version code */
int ocn_comm_info(ocn_comm_t comm, int32_t *rank, int32_t *nranks, int32_t *rccl_version);
/* fill_halo_event!(...; async = true) in x for a tuple of fields (local y / z fills first, fill_halo_regions.jl:148-196): packs the
 * 2 x Hx-wide full-cross-section strips of all fields (corners travel with the sides, OneDBuffers) on `stream` and starts one
 * grouped send / recv per neighbour (rank -+ 1, wrapping) on the communication stream.  `stream` stays free for the interior
 * tendency launch (interleave_communication_and_computation.jl:29-67). */
int ocn_halo_exchange_begin(ocn_comm_t comm, const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, void *stream);
/* The send buffers of the next ocn_halo_exchange_begin_packed for fields of these locations (one cross-section, Periodic y and z) and the
 * number of doubles per field in them; valid until the next exchange of a larger tuple. */
int ocn_halo_exchange_buffers(ocn_comm_t comm, const ocn_grid *grid, const int32_t *locs, int32_t n, double **send_west, double **send_east,
                              int64_t *field_doubles);
/* ocn_halo_exchange_begin without its pack launch: the send buffers were filled on `stream` (interior rows only) by
 * ocn_compute_momentum_tendencies_rk3_strips; ocn_halo_exchange_end unpacks with periodically wrapped (j, k), which gives every
 * halo cell -- corners included -- the value the local fills + full-cross-section strips of ocn_halo_exchange_begin give it. */
int ocn_halo_exchange_begin_packed(ocn_comm_t comm, const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, void *stream);
/* synchronize_communication!: `stream` waits for the exchange (event) and unpacks the received strips into the x halos. */
int ocn_halo_exchange_end(ocn_comm_t comm, const ocn_grid *grid, double *const *fields, const int32_t *locs, int32_t n, void *stream);
/* one x plane from a neighbour, in stream order (side 0: field[nx+1] <- east neighbour's field[1]; 1: field[0] <- west's field[nx]) */
int ocn_halo_exchange_plane(ocn_comm_t comm, const ocn_grid *grid, double *field, int32_t loc, int32_t side, void *stream);
/* ocn_halo_pack_pressure -> one grouped send / recv per neighbour -> ocn_halo_unpack_pressure, in stream order on `stream` (it sits
 * on the critical path between the pressure solve and the tendency launch that corrects on load) */
int ocn_halo_exchange_pressure(ocn_comm_t comm, const ocn_grid *grid, double *p, double *u, double dt_correct, void *stream);
/* one contiguous strip per x neighbour, in stream order: send_west -> rank - 1 (arrives as its recv_east), send_east -> rank + 1 */
int ocn_comm_exchange_strips(ocn_comm_t comm, const double *send_west, const double *send_east, double *recv_west, double *recv_east,
                             size_t count, void *stream);
/* Alltoallv! with equal counts: chunk d of `send` (count doubles) goes to rank d, chunk s of `recv` comes from rank s; stream order */
int ocn_comm_all_to_all(ocn_comm_t comm, const double *send, double *recv, size_t count, void *stream);
/* MPI.Allgather with equal counts: chunk s of `recv` is rank s's `send` (count doubles each); in stream order on `stream` */
int ocn_comm_all_gather(ocn_comm_t comm, const double *send, double *recv, size_t count, void *stream);
/* transpose_y_to_x! (direction 0) / transpose_x_to_y! (direction 1) of a distributed Poisson handle's exchange buffers */
int ocn_dist_poisson_exchange(ocn_dist_poisson_t solver, ocn_comm_t comm, int32_t direction, void *stream);
int ocn_comm_allreduce(ocn_comm_t comm, double *buffer, size_t count, int32_t op /* 0 sum, 1 max, 2 min */, void *stream);
int ocn_comm_barrier(ocn_comm_t comm);
/* Device-side timing of the exchanges, for diagnosing a multi-GPU run (off by default: timing events cost a little per call).  While
 * enabled, every exchange is bracketed by events on the stream it runs on; ocn_comm_stats synchronises the device, adds up the
 * elapsed milliseconds per category since the last call and clears them.  out_ms[8]:
 *   [0] strip exchange on the communication stream (grouped send / recv of ocn_halo_exchange_begin: link time + peer skew)
 *   [1] what the CALLER's stream waited at ocn_halo_exchange_end (0 when the exchange was hidden under the pressure solve)
 *   [2] all-gather / all-to-all of the pressure solve (in stream order: on the critical path)
 *   [3] pressure-plane exchange of the correction-on-load stage (in stream order)
 *   [4] single-plane exchanges (in stream order)
 *   [5] number of strip exchanges, [6] number of all-gathers / all-to-alls, [7] reserved */
int ocn_comm_enable_stats(ocn_comm_t comm, int32_t enable);
int ocn_comm_stats(ocn_comm_t comm, double *out_ms);
/* Host-side wait for the communication stream with a deadline: OCN_ERR_TIMEOUT if the exchanges posted so far have not completed
 * within `seconds` (a peer that never posted its half). */
int ocn_comm_wait(ocn_comm_t comm, double seconds);

/* The same one-call RK3 time_step! for ONE RANK of a slab-x run (Distributed(GPU(); partition = Partition(R)),
 * distributed_architectures.jl:167-297): local (FullyConnected, Periodic, Periodic) grid, the rank's ocn_dist_poisson_t (slab
 * pipelines: ocn_dist_poisson_pipeline 1 or 3) and its communicator, both borrowed.  Per stage: local y / z halo fills -> the u plane
 * the divergence reads -> the strips of u*, v*, w* posted on the communication stream (they fly under the pressure solve) ->
 * distributed solve -> strips unpacked -> ocn_halo_exchange_pressure -> ONE launch over the whole slab that corrects on load, computes
 * the tendencies and takes the next substep.  No host synchronisation, no Python between the launches; collectives included.
 * Bit-identical to the per-call sequence of oceananigans.jl_amd/distributed.py (tests/test_gpu_distributed.py).  The third stage's
 * correction is always deferred (see ocn_rk3_driver_configure); flush / fields / destroy as above. */
int ocn_rk3_driver_create_distributed(ocn_rk3_driver_t *driver, const ocn_grid *local_grid, double *u, double *v, double *w, double *p,
                                      ocn_dist_poisson_t solver, ocn_comm_t comm, void *stream);
/* ONE RANK of a slab-x run (local grid (FullyConnected, Periodic, *), RCCL communicator, the rank's distributed Poisson handle on one of
 * its slab pipelines): every x exchange -- the strips of the prognostic fields and of nu_e / kappa_e in update_state!, the single planes
 * of u and p inside the projection of stages 1 and 2, the solver's transposes or all-gather -- is issued by the library.  The exchange is
 * synchronous here (no interior / buffer split: distributed.py::update_state_general overlaps it); results are identical. */
int ocn_model_driver_create_distributed(ocn_model_driver_t *driver, const ocn_grid *local_grid, const ocn_model_driver_desc *desc, double *u,
                                        double *v, double *w, double *p, ocn_dist_poisson_t solver, ocn_comm_t comm, void *stream);                      /* MPI.Barrier; blocks the host */

#ifdef __cplusplus
}
#endif
#endif /* OCN_HIP_H */
