/*
 * oracle/ocn_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * A plain-C, strict-IEEE (compile with -ffp-contract=off) CPU restatement of the
 * NonhydrostaticModel hot path of Oceananigans.jl v0.96.19 on a RectilinearGrid
 * (x, y regular; z regular or stretched; each direction Periodic / Bounded / Flat).
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree).  Expression shapes (operand order, left-associated n-ary +,
 * "multiply by computed reciprocal", no FMA contraction) follow the Julia source so
 * that results are reproducible to the last bit on any IEEE-754 CPU.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (oceananigans.jl_amd/) never links, imports or calls it.
 *
 * Parity status: the reference is Julia and cannot be executed in this project
 * (no julia binary, no network).  This restatement is pinned by the reference's own
 * property / known-answer tests re-expressed in tests/ (see DESIGN.md "Oracle"),
 * and by the jldoctest coefficient vectors in src/Advection/reconstruction_coefficients.jl.
 * Bit-level parity of the WENO arithmetic with a live Julia run is UNPINNED.
 *
 * Memory layout (src/Grids/new_data.jl:36-70, src/Grids/grid_utils.jl:66-72):
 * every field is its OffsetArray *parent*: column-major, x fastest, with halos.
 * Interior index (i,j,k), 1-based, maps to parent offset
 *   (i+Hx-1) + sx*((j+Hy-1) + sy*(k+Hz-1)),  sx,sy,sz = parent extents.
 * Parent extent along a dimension: N+2H, or N+1+2H for a Face-located field in a
 * Bounded dimension.  Flat dimensions have N=1, H=0.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define OCN_PERIODIC 0
#define OCN_BOUNDED 1
#define OCN_FLAT 2

typedef struct {
    int32_t Nx, Ny, Nz;
    int32_t Hx, Hy, Hz;
    int32_t tx, ty, tz;
    double dx, dy, dz;  /* regular spacings; dz ignored when dzc != NULL */
    const double *dzc;  /* dz at centres, element 0 <-> k = 1-Hz, length Nz+2Hz, or NULL */
    const double *dzf;  /* dz at faces,   element 0 <-> k = 1-Hz, length Nz+2Hz (+1 if Bounded), or NULL */
} ocn_grid;

/* parent extents for a field at location (fx,fy,fz); f? = 1 for Face */
static inline int ext(int N, int H, int topo, int face) { return N + 2 * H + ((face && topo == OCN_BOUNDED) ? 1 : 0); }

typedef struct {
    ptrdiff_t sx, sy, sz; /* parent extents */
    ptrdiff_t s1, s2, s3; /* strides: 1, sx, sx*sy */
    ptrdiff_t o;          /* offset of interior (1,1,1) */
} lay;

static inline lay mklay(const ocn_grid *g, int fx, int fy, int fz)
{
    lay L;
    L.sx = ext(g->Nx, g->Hx, g->tx, fx);
    L.sy = ext(g->Ny, g->Hy, g->ty, fy);
    L.sz = ext(g->Nz, g->Hz, g->tz, fz);
    L.s1 = 1;
    L.s2 = L.sx;
    L.s3 = L.sx * L.sy;
    L.o = g->Hx + L.s2 * g->Hy + L.s3 * g->Hz;
    return L;
}
/* 1-based interior index -> parent offset */
#define AT(L, i, j, k) ((L).o + ((i)-1) + (L).s2 * ((j)-1) + (L).s3 * ((k)-1))

/* ---- grid metrics: src/Operators/spacings_and_areas_and_volumes.jl:106-140,263-345 ---- */
static inline double dzc_at(const ocn_grid *g, int k) { return g->dzc ? g->dzc[k + g->Hz - 1] : g->dz; }
static inline double dzf_at(const ocn_grid *g, int k) { return g->dzf ? g->dzf[k + g->Hz - 1] : g->dz; }
/* zf: 1 if the z-location is Face */
static inline double dz_at(const ocn_grid *g, int k, int zf) { return zf ? dzf_at(g, k) : dzc_at(g, k); }
static inline double Ax_at(const ocn_grid *g, int k, int zf) { return g->dy * dz_at(g, k, zf); } /* Ax = dy*dz */
static inline double Ay_at(const ocn_grid *g, int k, int zf) { return g->dx * dz_at(g, k, zf); } /* Ay = dx*dz */
static inline double Az_at(const ocn_grid *g) { return g->dx * g->dy; }                         /* Az = dx*dy */
static inline double V_at(const ocn_grid *g, int k, int zf) { return Az_at(g) * dz_at(g, k, zf); } /* V = Az*dz */

/* =====================================================================================
 * Reconstruction coefficients.
 * src/Advection/reconstruction_coefficients.jl:100-115 (stencil_coefficients): the
 * Int/Int quotient is a Float64, accumulated in BigFloat, rounded to FT, and the last
 * coefficient is 1 - sum(others).  Values below are produced by oracle/coefficients.py
 * (tests/test_oracle_coefficients.py checks them against that generator and against
 * the reference's jldoctest vectors).
 * ===================================================================================== */
/* Centered(order=4): applied to psi[n-2], psi[n-1], psi[n], psi[n+1] for face n
 * (calc_reconstruction_stencil, reconstruction_coefficients.jl:173-203: coeff[order-idx+1]) */
static const double C4[4] = {-0.08333333333333326, 0.5833333333333333, 0.5833333333333333, -0.08333333333333333};
/* WENO{3} per-stencil coefficients coeff_p (weno_interpolants.jl:118-119), stencil r = 0,1,2 */
static const double W5P[3][3] = {{0.33333333333333337, 0.8333333333333334, -0.16666666666666674},
                                 {-0.16666666666666669, 0.8333333333333333, 0.3333333333333335},
                                 {0.33333333333333326, -1.1666666666666667, 1.8333333333333335}};
/* WENO{2} per-stencil coefficients */
static const double W3P[2][2] = {{0.5, 0.5}, {-0.5, 1.5}};
/* smoothness coefficients, weno_interpolants.jl:178-183 */
static const double W5B[3][6] = {{10, -31, 11, 25, -19, 4}, {4, -13, 5, 13, -13, 4}, {4, -19, 11, 25, -31, 10}};
static const double W3B[2][3] = {{1, -2, 1}, {1, -2, 1}};
/* optimal weights C*, weno_interpolants.jl:77-82 */
static const double W5C[3] = {3.0 / 10.0, 3.0 / 5.0, 1.0 / 10.0};
static const double W3C[2] = {2.0 / 3.0, 1.0 / 3.0};
/* const eps = 1f-8 widened to Float64 on use (weno_interpolants.jl:70) */
static const double WENO_EPS = (double)1e-8f;

/* exported for tests */
void ocn_oracle_coefficients(double *c4, double *w5p, double *w3p, double *eps)
{
    memcpy(c4, C4, sizeof C4);
    memcpy(w5p, W5P, sizeof W5P);
    memcpy(w3p, W3P, sizeof W3P);
    *eps = WENO_EPS;
}

/* ---- WENO5: weno_interpolants.jl:341-348 (weights), 445-447 (stencils), 475-511 (reconstruction)
 * S = psi[n-3..n+2] for a reconstruction at face n; left != 0 selects LeftBias. */
static inline double beta3(const double *p, const double *C)
{
    /* smoothness_operation for buffer 3 (weno_interpolants.jl:213-225) */
    return p[0] * ((C[0] * p[0] + C[1] * p[1]) + C[2] * p[2]) + p[1] * (C[3] * p[1] + C[4] * p[2]) + (p[2] * p[2]) * C[5];
}
static inline double weno5(const double S[6], int left)
{
    double s0[3], s1[3], s2[3];
    if (left) {
        s0[0] = S[2]; s0[1] = S[3]; s0[2] = S[4];
        s1[0] = S[1]; s1[1] = S[2]; s1[2] = S[3];
        s2[0] = S[0]; s2[1] = S[1]; s2[2] = S[2];
    } else {
        s0[0] = S[3]; s0[1] = S[2]; s0[2] = S[1];
        s1[0] = S[4]; s1[1] = S[3]; s1[2] = S[2];
        s2[0] = S[5]; s2[1] = S[4]; s2[2] = S[3];
    }
    double b0 = beta3(s0, W5B[0]), b1 = beta3(s1, W5B[1]), b2 = beta3(s2, W5B[2]);
    double tau = fabs(b0 - b2); /* global_smoothness_indicator(Val(3)), :318 */
    double q0 = tau / (b0 + WENO_EPS), q1 = tau / (b1 + WENO_EPS), q2 = tau / (b2 + WENO_EPS);
    double a0 = W5C[0] * (1 + q0 * q0), a1 = W5C[1] * (1 + q1 * q1), a2 = W5C[2] * (1 + q2 * q2); /* :299-306 */
    double sa = (a0 + a1) + a2;
    double w0 = a0 / sa, w1 = a1 / sa, w2 = a2 / sa;
    double p0 = (W5P[0][0] * s0[0] + W5P[0][1] * s0[1]) + W5P[0][2] * s0[2]; /* biased_p :144-145 */
    double p1 = (W5P[1][0] * s1[0] + W5P[1][1] * s1[1]) + W5P[1][2] * s1[2];
    double p2 = (W5P[2][0] * s2[0] + W5P[2][1] * s2[1]) + W5P[2][2] * s2[2];
    return (w0 * p0 + w1 * p1) + w2 * p2;
}
/* ---- WENO3 (buffer_scheme of WENO5): S = psi[n-2..n+1] */
static inline double beta2(const double *p, const double *C) { return p[0] * (C[0] * p[0] + C[1] * p[1]) + (p[1] * p[1]) * C[2]; }
static inline double weno3(const double S[4], int left)
{
    double s0[2], s1[2];
    if (left) {
        s0[0] = S[1]; s0[1] = S[2];
        s1[0] = S[0]; s1[1] = S[1];
    } else {
        s0[0] = S[2]; s0[1] = S[1];
        s1[0] = S[3]; s1[1] = S[2];
    }
    double b0 = beta2(s0, W3B[0]), b1 = beta2(s1, W3B[1]);
    double tau = fabs(b0 - b1);
    double q0 = tau / (b0 + WENO_EPS), q1 = tau / (b1 + WENO_EPS);
    double a0 = W3C[0] * (1 + q0 * q0), a1 = W3C[1] * (1 + q1 * q1);
    double sa = a0 + a1;
    double w0 = a0 / sa, w1 = a1 / sa;
    double p0 = W3P[0][0] * s0[0] + W3P[0][1] * s0[1];
    double p1 = W3P[1][0] * s1[0] + W3P[1][1] * s1[1];
    return w0 * p0 + w1 * p1;
}

/* =====================================================================================
 * Topology-conditional interpolation (topologically_conditional_interpolation.jl:37-128)
 * A "line accessor" describes a 1-D line through a field (or through the function
 * metric*field): value(m) = scale(m) * p[m*stride] where m is the index offset from the
 * face index n along the interpolation direction.
 * ===================================================================================== */
typedef struct {
    const double *p;   /* points at the element with index n (the face index) along the line */
    ptrdiff_t s;       /* stride along the line */
    const ocn_grid *g; /* for z-varying metrics */
    int metric;        /* 0: none, 1: Ax (dy*dz), 2: Ay (dx*dz), 3: Az (dx*dy) */
    int zf;            /* z-location of the metric (1 Face) */
    int along_z;       /* 1 if the line runs along z (metric index varies with m) */
    int k0;            /* z index at m = 0 (1-based) */
} line;

static inline double lval(const line *L, int m)
{
    double v = L->p[m * L->s];
    if (!L->metric) return v;
    int k = L->along_z ? L->k0 + m : L->k0;
    double a = L->metric == 1 ? Ax_at(L->g, k, L->zf) : L->metric == 2 ? Ay_at(L->g, k, L->zf) : Az_at(L->g);
    return a * v; /* Ax_q(i,j,k,grid,u) = Ax(i,j,k) * u[i,j,k]; products_between_fields_and_grid_metrics.jl:5-14 */
}

/* symmetric (advecting velocity) interpolation to face n; N = grid size along the line, topo its topology.
 * WENO{3}.advecting_velocity_scheme = Centered(order=4), buffer chain -> Centered(order=2)
 * (weno_reconstruction.jl:117-120, upwind_biased_reconstruction.jl:101-107).
 * `center` selects the *ᶜ variant (interpolating to centre n == face n+1 with the ᶜ halo test):
 * the caller passes the line already shifted to face n+1 and idx = n. */
static inline double sym_interp(const line *L, int idx, int N, int topo, int center)
{
    if (topo == OCN_FLAT) return lval(L, center ? -1 : 0); /* flat_advective_fluxes.jl:26-44: psi[i,j,k] at the point itself */
    int hi_ok = 1;
    if (topo == OCN_BOUNDED) {
        /* outside_symmetric_halo, required_halo_size(WENO{3}) = 3 (:46-47) */
        hi_ok = center ? (idx >= 3 && idx <= N + 1 - 3) : (idx >= 3 + 1 && idx <= N + 1 - 3);
    }
    if (hi_ok) return ((C4[0] * lval(L, -2) + C4[1] * lval(L, -1)) + C4[2] * lval(L, 0)) + C4[3] * lval(L, 1);
    /* Centered(order=2): 0.5*psi[n-1] + 0.5*psi[n]  (both deeper fallbacks are Centered2) */
    return 0.5 * lval(L, -1) + 0.5 * lval(L, 0);
}

/* biased interpolation to face n (or centre idx = n-1 when center != 0) */
/* UpwindBiased(order=5) / (order=3) stencils (upwind_biased_reconstruction.jl:91-117 -> calc_reconstruction_stencil,
 * reconstruction_coefficients.jl:173-203), coefficients from oracle/coefficients.py; n-ary + is left-associated. */
static inline double upwind5(const double S[6], int left)
{
    if (left) return (((0.033333333333333326 * S[0] + -0.21666666666666667 * S[1]) + 0.7833333333333333 * S[2]) + 0.45 * S[3]) + -0.04999999999999998 * S[4];
    return (((-0.050000000000000044 * S[1] + 0.45 * S[2]) + 0.7833333333333333 * S[3]) + -0.21666666666666667 * S[4]) + 0.03333333333333331 * S[5];
}
static inline double upwind3(const double S[4], int left)
{
    if (left) return (-0.16666666666666674 * S[0] + 0.8333333333333334 * S[1]) + 0.33333333333333337 * S[2];
    return (0.3333333333333335 * S[1] + 0.8333333333333333 * S[2]) + -0.16666666666666669 * S[3];
}
void ocn_oracle_upwind_coefficients(double *u5l, double *u5r, double *u3l, double *u3r)
{
    const double a[5] = {0.033333333333333326, -0.21666666666666667, 0.7833333333333333, 0.45, -0.04999999999999998};
    const double b[5] = {-0.050000000000000044, 0.45, 0.7833333333333333, -0.21666666666666667, 0.03333333333333331};
    const double c[3] = {-0.16666666666666674, 0.8333333333333334, 0.33333333333333337};
    const double d[3] = {0.3333333333333335, 0.8333333333333333, -0.16666666666666669};
    memcpy(u5l, a, sizeof a); memcpy(u5r, b, sizeof b); memcpy(u3l, c, sizeof c); memcpy(u3r, d, sizeof d);
}

/* scheme: 0 WENO(order=5) (fallback WENO3 -> Upwind1), 2 UpwindBiased(order=5) (fallback Upwind3 -> Upwind1); the halo
 * conditions are the same (both have buffer 3) */
static inline double bias_interp_scheme(const line *L, int idx, int N, int topo, int center, int left, int scheme);
static inline double bias_interp(const line *L, int idx, int N, int topo, int center, int left)
{
    return bias_interp_scheme(L, idx, N, topo, center, left, 0);
}
static inline double bias_interp_scheme(const line *L, int idx, int N, int topo, int center, int left, int scheme)
{
    if (topo == OCN_FLAT) return lval(L, center ? -1 : 0);
    int ok5 = 1, ok3 = 1;
    if (topo == OCN_BOUNDED) {
        if (center) { /* outside_biased_haloᶜ :51-52 */
            ok5 = (idx >= 3) && (idx <= N + 1 - 2) && (idx >= 2) && (idx <= N + 1 - 3);
            ok3 = (idx >= 2) && (idx <= N + 1 - 1) && (idx >= 1) && (idx <= N + 1 - 2);
        } else { /* outside_biased_haloᶠ :49-50 */
            ok5 = (idx >= 4) && (idx <= N + 1 - 2) && (idx >= 3) && (idx <= N + 1 - 3);
            ok3 = (idx >= 3) && (idx <= N + 1 - 1) && (idx >= 2) && (idx <= N + 1 - 2);
        }
    }
    if (ok5) {
        double S[6];
        for (int m = 0; m < 6; ++m) S[m] = lval(L, m - 3);
        return scheme == 2 ? upwind5(S, left) : weno5(S, left);
    }
    if (ok3) {
        double S[4];
        for (int m = 0; m < 4; ++m) S[m] = lval(L, m - 2);
        return scheme == 2 ? upwind3(S, left) : weno3(S, left);
    }
    /* UpwindBiased(order=1): left -> psi[n-1], right -> psi[n] (reconstruction_coefficients.jl:153-157) */
    return left ? lval(L, -1) : lval(L, 0);
}

/* =====================================================================================
 * Momentum advective fluxes, upwind_biased_advective_fluxes.jl:23-93.
 * dir codes: 0 x, 1 y, 2 z.  Every flux is  U~ * psi^R  with
 *   U~   = symmetric interpolation, along direction `da`, of (area * advecting velocity)
 *   psiR = biased interpolation, along direction `db`, of the advected component.
 * ===================================================================================== */
typedef struct {
    const ocn_grid *g;
    const double *u, *v, *w;
    lay Lu, Lv, Lw;
    int scheme; /* 0: WENO(order=5) (the default of every entry point), 1: Centered(order=2), 2: UpwindBiased(order=5) */
} vel;
#define OCN_ADV_WENO5 0
#define OCN_ADV_CENTERED2 1
#define OCN_ADV_UPWIND5 2

/* Centered(order=2) interpolation: FT(0.5)*psi[n-1] + FT(0.5)*psi[n]; Centered{1} is a "low order" scheme and has no
 * topology conditions (topologically_conditional_interpolation.jl:24-27 LOADV); Flat: the value itself
 * (flat_advective_fluxes.jl:26-44). */
static inline double c2_interp(const line *L, int topo, int center)
{
    if (topo == OCN_FLAT) return lval(L, center ? -1 : 0);
    return 0.5 * lval(L, -1) + 0.5 * lval(L, 0);
}

static inline int gridN(const ocn_grid *g, int d) { return d == 0 ? g->Nx : d == 1 ? g->Ny : g->Nz; }
static inline int gridT(const ocn_grid *g, int d) { return d == 0 ? g->tx : d == 1 ? g->ty : g->tz; }
static inline ptrdiff_t strd(const lay *L, int d) { return d == 0 ? L->s1 : d == 1 ? L->s2 : L->s3; }

/* generic flux: advecting component `ca` (0 u,1 v,2 w) interpolated along `da` (to face or centre),
 * advected component `cb` interpolated along `db`. (i,j,k) is the flux location index triple as used
 * by the reference's advective_momentum_flux_* functions. */
static double mom_flux(const vel *V, int ca, int da, int a_center, int cb, int db, int b_center, int i, int j, int k)
{
    const ocn_grid *g = V->g;
    /* flat_advective_fluxes.jl:8-22: flux *through* a Flat direction is zero */
    if (gridT(g, ca) == OCN_FLAT) return 0.0;
    const double *fa = ca == 0 ? V->u : ca == 1 ? V->v : V->w;
    const lay *La = ca == 0 ? &V->Lu : ca == 1 ? &V->Lv : &V->Lw;
    const double *fb = cb == 0 ? V->u : cb == 1 ? V->v : V->w;
    const lay *Lb = cb == 0 ? &V->Lu : cb == 1 ? &V->Lv : &V->Lw;
    int ijk[3] = {i, j, k};

    line A;
    A.g = g;
    A.metric = ca + 1;          /* Ax_q(fcc) for u, Ay_q(cfc) for v, Az_q(ccf) for w */
    A.zf = (ca == 2);           /* z-location of that metric: w lives on z faces */
    A.along_z = (da == 2);
    A.s = strd(La, da);
    {
        int q[3] = {i, j, k};
        if (a_center) q[da] += 1; /* symmetric_interpolate_*ᶜ: inner(..., i+1, ...) reconstruction_coefficients.jl:30-34 */
        A.p = fa + AT(*La, q[0], q[1], q[2]);
        A.k0 = q[2];
    }
    if (V->scheme == OCN_ADV_CENTERED2) {
        /* centered_advective_fluxes.jl:7-17:  A(flux location) * sym(U) * sym(u), left-associated; the area is NOT
         * inside the interpolation.  z-location of the flux: Face for Uw, Vw; Center otherwise (Az has no z metric). */
        A.metric = 0;
        double ua = c2_interp(&A, gridT(g, da), a_center);
        line Bc;
        Bc.g = g;
        Bc.metric = 0;
        Bc.zf = 0;
        Bc.along_z = (db == 2);
        Bc.s = strd(Lb, db);
        int q[3] = {i, j, k};
        if (b_center) q[db] += 1;
        Bc.p = fb + AT(*Lb, q[0], q[1], q[2]);
        Bc.k0 = q[2];
        double ub = c2_interp(&Bc, gridT(g, db), b_center);
        int zf = (cb == 2 && ca != 2);
        double area = ca == 0 ? Ax_at(g, k, zf) : ca == 1 ? Ay_at(g, k, zf) : Az_at(g);
        return (area * ua) * ub;
    }
    double ut = sym_interp(&A, ijk[da], gridN(g, da), gridT(g, da), a_center);

    line B;
    B.g = g;
    B.metric = 0;
    B.zf = 0;
    B.along_z = (db == 2);
    B.s = strd(Lb, db);
    {
        int q[3] = {i, j, k};
        if (b_center) q[db] += 1;
        B.p = fb + AT(*Lb, q[0], q[1], q[2]);
        B.k0 = q[2];
    }
    int left = ut > 0; /* bias(u) = ifelse(u > 0, LeftBias(), RightBias()) :21 */
    double pr = bias_interp_scheme(&B, ijk[db], gridN(g, db), gridT(g, db), b_center, left, V->scheme);
    return ut * pr;
}

/* named wrappers matching the reference's function names */
#define F_Uu(V, i, j, k) mom_flux(V, 0, 0, 1, 0, 0, 1, i, j, k) /* :23-29  sym xᶜ of Ax*u ; biased xᶜ of u */
#define F_Vu(V, i, j, k) mom_flux(V, 1, 0, 0, 0, 1, 0, i, j, k) /* :31-37  sym xᶠ of Ay*v ; biased yᶠ of u */
#define F_Wu(V, i, j, k) mom_flux(V, 2, 0, 0, 0, 2, 0, i, j, k) /* :39-45  sym xᶠ of Az*w ; biased zᶠ of u */
#define F_Uv(V, i, j, k) mom_flux(V, 0, 1, 0, 1, 0, 0, i, j, k) /* :47-53  sym yᶠ of Ax*u ; biased xᶠ of v */
#define F_Vv(V, i, j, k) mom_flux(V, 1, 1, 1, 1, 1, 1, i, j, k) /* :55-61 */
#define F_Wv(V, i, j, k) mom_flux(V, 2, 1, 0, 1, 2, 0, i, j, k) /* :63-69 */
#define F_Uw(V, i, j, k) mom_flux(V, 0, 2, 0, 2, 0, 0, i, j, k) /* :71-77 */
#define F_Vw(V, i, j, k) mom_flux(V, 1, 2, 0, 2, 1, 0, i, j, k) /* :79-85 */
#define F_Ww(V, i, j, k) mom_flux(V, 2, 2, 1, 2, 2, 1, i, j, k) /* :87-93 */

/* delta operators return zero(FT) along Flat dims (difference_operators.jl:33-49) */
#define DFLAT(g, d) (gridT(g, d) == OCN_FLAT)

/* K1-K3: compute_Gu!/Gv!/Gw! (compute_nonhydrostatic_tendencies.jl:146-179) with every optional
 * term `nothing`: G = -div_Uu (momentum_advection_operators.jl:46-83).  Work range follows
 * launch!(..., :xyz; exclude_periphery=true) (kernel_launching.jl:113-161): Face-located in a
 * Bounded dim starts at 2. */
void ocn_oracle_momentum_tendencies_scheme(const ocn_grid *g, int scheme, const double *u, const double *v, const double *w,
                                           double *Gu, double *Gv, double *Gw)
{
    vel V;
    V.scheme = scheme;
    V.g = g;
    V.u = u;
    V.v = v;
    V.w = w;
    V.Lu = mklay(g, 1, 0, 0);
    V.Lv = mklay(g, 0, 1, 0);
    V.Lw = mklay(g, 0, 0, 1);
    const int Nx = g->Nx, Ny = g->Ny, Nz = g->Nz;
    const int ox = (g->tx == OCN_BOUNDED && Nx > 1), oy = (g->ty == OCN_BOUNDED && Ny > 1), oz = (g->tz == OCN_BOUNDED && Nz > 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= Nz; ++k)
        for (int j = 1; j <= Ny; ++j)
            for (int i = 1; i <= Nx; ++i) {
                if (i >= 1 + ox) { /* Gu at (f,c,c) */
                    double dxF = DFLAT(g, 0) ? 0.0 : F_Uu(&V, i, j, k) - F_Uu(&V, i - 1, j, k);     /* δxᶠᵃᵃ */
                    double dyF = DFLAT(g, 1) ? 0.0 : F_Vu(&V, i, j + 1, k) - F_Vu(&V, i, j, k);     /* δyᵃᶜᵃ */
                    double dzF = DFLAT(g, 2) ? 0.0 : F_Wu(&V, i, j, k + 1) - F_Wu(&V, i, j, k);     /* δzᵃᵃᶜ */
                    double rV = 1 / V_at(g, k, 0);
                    Gu[AT(V.Lu, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
                }
                if (j >= 1 + oy) { /* Gv at (c,f,c) */
                    double dxF = DFLAT(g, 0) ? 0.0 : F_Uv(&V, i + 1, j, k) - F_Uv(&V, i, j, k);     /* δxᶜᵃᵃ */
                    double dyF = DFLAT(g, 1) ? 0.0 : F_Vv(&V, i, j, k) - F_Vv(&V, i, j - 1, k);     /* δyᵃᶠᵃ */
                    double dzF = DFLAT(g, 2) ? 0.0 : F_Wv(&V, i, j, k + 1) - F_Wv(&V, i, j, k);     /* δzᵃᵃᶜ */
                    double rV = 1 / V_at(g, k, 0);
                    Gv[AT(V.Lv, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
                }
                if (k >= 1 + oz) { /* Gw at (c,c,f) */
                    double dxF = DFLAT(g, 0) ? 0.0 : F_Uw(&V, i + 1, j, k) - F_Uw(&V, i, j, k);     /* δxᶜᵃᵃ */
                    double dyF = DFLAT(g, 1) ? 0.0 : F_Vw(&V, i, j + 1, k) - F_Vw(&V, i, j, k);     /* δyᵃᶜᵃ */
                    double dzF = DFLAT(g, 2) ? 0.0 : F_Ww(&V, i, j, k) - F_Ww(&V, i, j, k - 1);     /* δzᵃᵃᶠ */
                    double rV = 1 / V_at(g, k, 1);
                    Gw[AT(V.Lw, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
                }
            }
}

void ocn_oracle_momentum_tendencies(const ocn_grid *g, const double *u, const double *v, const double *w, double *Gu,
                                    double *Gv, double *Gw)
{
    ocn_oracle_momentum_tendencies_scheme(g, OCN_ADV_WENO5, u, v, w, Gu, Gv, Gw);
}

/* K4: tracer tendency  Gc = -div_Uc (tracer_advection_operators.jl:30-34), fluxes
 * upwind_biased_advective_fluxes.jl:99-121:  Ax * u[i,j,k] * cR  (left-assoc). */
static double tracer_flux(const vel *V, const double *c, const lay *Lc, int d, int i, int j, int k)
{
    const ocn_grid *g = V->g;
    if (gridT(g, d) == OCN_FLAT) return 0.0;
    const double *fa = d == 0 ? V->u : d == 1 ? V->v : V->w;
    const lay *La = d == 0 ? &V->Lu : d == 1 ? &V->Lv : &V->Lw;
    double ut = fa[AT(*La, i, j, k)];
    line B;
    B.g = g;
    B.metric = 0;
    B.zf = 0;
    B.along_z = (d == 2);
    B.s = strd(Lc, d);
    B.p = c + AT(*Lc, i, j, k);
    B.k0 = k;
    int ijk[3] = {i, j, k};
    double area = d == 0 ? Ax_at(g, k, 0) : d == 1 ? Ay_at(g, k, 0) : Az_at(g);
    if (V->scheme == OCN_ADV_CENTERED2) /* centered_advective_fluxes.jl:23-25: Ax_q(U) * sym(c) */
        return (area * ut) * c2_interp(&B, gridT(g, d), 0);
    double cr = bias_interp_scheme(&B, ijk[d], gridN(g, d), gridT(g, d), 0, ut > 0, V->scheme);
    return (area * ut) * cr;
}
void ocn_oracle_tracer_tendency_scheme(const ocn_grid *g, int scheme, const double *u, const double *v, const double *w,
                                       const double *c, double *Gc)
{
    vel V;
    V.scheme = scheme;
    V.g = g;
    V.u = u;
    V.v = v;
    V.w = w;
    V.Lu = mklay(g, 1, 0, 0);
    V.Lv = mklay(g, 0, 1, 0);
    V.Lw = mklay(g, 0, 0, 1);
    lay Lc = mklay(g, 0, 0, 0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                double dxF = DFLAT(g, 0) ? 0.0 : tracer_flux(&V, c, &Lc, 0, i + 1, j, k) - tracer_flux(&V, c, &Lc, 0, i, j, k);
                double dyF = DFLAT(g, 1) ? 0.0 : tracer_flux(&V, c, &Lc, 1, i, j + 1, k) - tracer_flux(&V, c, &Lc, 1, i, j, k);
                double dzF = DFLAT(g, 2) ? 0.0 : tracer_flux(&V, c, &Lc, 2, i, j, k + 1) - tracer_flux(&V, c, &Lc, 2, i, j, k);
                double rV = 1 / V_at(g, k, 0);
                Gc[AT(Lc, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
            }
}

void ocn_oracle_tracer_tendency(const ocn_grid *g, const double *u, const double *v, const double *w, const double *c,
                                double *Gc)
{
    ocn_oracle_tracer_tendency_scheme(g, OCN_ADV_WENO5, u, v, w, c, Gc);
}

/* =====================================================================================
 * the non-advective terms of u/v/w_velocity_tendency and tracer_tendency
 * (nonhydrostatic_tendency_kernel_functions.jl:47-259), hydrostatic pressure anomaly, flux /
 * value / gradient boundary conditions.
 * ===================================================================================== */
typedef struct {
    int32_t coriolis; /* 0 nothing, 1 FPlane (Coriolis/f_plane.jl:44-46), 2 BetaPlane (beta_plane.jl:43-57) */
    int32_t closure;  /* 0 nothing, 1 ScalarDiffusivity(ν, κ): ThreeDimensionalFormulation, ExplicitTimeDiscretization, constants */
    int32_t buoyancy; /* 0 nothing, 1 BuoyancyTracer, 2 SeawaterBuoyancy(LinearEquationOfState) with T and S,
                         3 ... with T only (constant_salinity), 4 ... with S only (constant_temperature) */
    int32_t _pad;
    double f;            /* FPlane.f, or BetaPlane.f₀ */
    double nu;           /* ScalarDiffusivity.ν */
    double g, alpha, beta; /* gravitational_acceleration, thermal_expansion, haline_contraction */
    /* coriolis == 2: BetaPlane(f₀, β) (Coriolis/beta_plane.jl:43-57): f = f₀ + β y with y = ynode at the velocity point, i.e.
     * yᵃᶜᵃ[j] for x_f_cross_U and yᵃᶠᵃ[j] for y_f_cross_U; the node vectors include the halos (element 0 <-> j = 1 - Hy) */
    double coriolis_beta;
    const double *yc, *yf;
} ocn_physics;
static inline double coriolis_f_at(const ocn_grid *g, const ocn_physics *ph, int j, int face)
{
    if (ph->coriolis != 2) return ph->f;
    const double y = (face ? ph->yf : ph->yc)[j + g->Hy - 1];
    return ph->f + ph->coriolis_beta * y;
}

/* inactive_cell (Grids/inactive_node.jl:35-95): outside the domain in a Bounded direction */
static inline int inactive_cell(const ocn_grid *g, int i, int j, int k)
{
    int r = 0;
    if (g->tx == OCN_BOUNDED) r |= (i < 1) | (i > g->Nx);
    if (g->ty == OCN_BOUNDED) r |= (j < 1) | (j > g->Ny);
    if (g->tz == OCN_BOUNDED) r |= (k < 1) | (k > g->Nz);
    return r;
}
/* not_peripheral_node at (c,f,c) and (f,c,c) (inactive_node.jl:145-149) as 0/1 */
static inline double active_cfc(const ocn_grid *g, int i, int j, int k) { return !(inactive_cell(g, i, j, k) | inactive_cell(g, i, j - 1, k)); }
static inline double active_fcc(const ocn_grid *g, int i, int j, int k) { return !(inactive_cell(g, i, j, k) | inactive_cell(g, i - 1, j, k)); }

/* ℑxyᶠᶜᵃ(q) = ℑyᵃᶜᵃ(ℑxᶠᵃᵃ q) (interpolation_operators.jl:8-26,46); along a Flat dim the interpolation is the identity (:103-110) */
static inline double ixy_fc(const ocn_grid *g, const double *q, const lay *L, int i, int j, int k)
{
#define IXF(jj) (g->tx == OCN_FLAT ? q[AT(*L, i, jj, k)] : 0.5 * (q[AT(*L, i - 1, jj, k)] + q[AT(*L, i, jj, k)]))
    if (g->ty == OCN_FLAT) return IXF(j);
    return 0.5 * (IXF(j) + IXF(j + 1));
#undef IXF
}
static inline double ixy_fc_active(const ocn_grid *g, int i, int j, int k)
{
#define IXF(jj) (g->tx == OCN_FLAT ? active_cfc(g, i, jj, k) : 0.5 * (active_cfc(g, i - 1, jj, k) + active_cfc(g, i, jj, k)))
    if (g->ty == OCN_FLAT) return IXF(j);
    return 0.5 * (IXF(j) + IXF(j + 1));
#undef IXF
}
/* ℑxyᶜᶠᵃ(q) = ℑyᵃᶠᵃ(ℑxᶜᵃᵃ q) */
static inline double ixy_cf(const ocn_grid *g, const double *q, const lay *L, int i, int j, int k)
{
#define IXC(jj) (g->tx == OCN_FLAT ? q[AT(*L, i, jj, k)] : 0.5 * (q[AT(*L, i, jj, k)] + q[AT(*L, i + 1, jj, k)]))
    if (g->ty == OCN_FLAT) return IXC(j);
    return 0.5 * (IXC(j - 1) + IXC(j));
#undef IXC
}
static inline double ixy_cf_active(const ocn_grid *g, int i, int j, int k)
{
#define IXC(jj) (g->tx == OCN_FLAT ? active_fcc(g, i, jj, k) : 0.5 * (active_fcc(g, i, jj, k) + active_fcc(g, i + 1, jj, k)))
    if (g->ty == OCN_FLAT) return IXC(j);
    return 0.5 * (IXC(j - 1) + IXC(j));
#undef IXC
}

/* buoyancy_perturbationᶜᶜᶜ (buoyancy_tracer.jl:12, linear_equation_of_state.jl:58-66) */
static inline double buoyancy_ccc(const ocn_physics *ph, const double *T, const double *S, ptrdiff_t a)
{
    switch (ph->buoyancy) {
        case 1: return T[a];
        case 2: return ph->g * (ph->alpha * T[a] - ph->beta * S[a]);
        case 3: return ph->g * ph->alpha * T[a];
        case 4: return -ph->g * ph->beta * S[a];
        default: return 0.0;
    }
}

/* _update_hydrostatic_pressure! (update_hydrostatic_pressure.jl:12-20) over p_kernel_parameters (:45-53):
 * i in 0:Nx+1, j in 0:Ny+1 (1:N along Flat).  z_dot_g_bᶜᶜᶠ = ĝ_z * ℑzᵃᵃᶠ(b), ĝ_z = 1 (g_dot_b.jl:3, buoyancy_force.jl:39).
 * Nothing happens on a z-Flat grid (:26). */
void ocn_oracle_update_hydrostatic_pressure(const ocn_grid *g, const ocn_physics *ph, const double *T, const double *S,
                                            double *pHY)
{
    if (g->tz == OCN_FLAT || ph->buoyancy == 0) return;
    lay L = mklay(g, 0, 0, 0);
    const int i0 = g->tx == OCN_FLAT ? 1 : 0, i1 = g->tx == OCN_FLAT ? g->Nx : g->Nx + 1;
    const int j0 = g->ty == OCN_FLAT ? 1 : 0, j1 = g->ty == OCN_FLAT ? g->Ny : g->Ny + 1;
    const int Nz = g->Nz;
#pragma omp parallel for schedule(static)
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
#define ZB(k) (1 * (0.5 * (buoyancy_ccc(ph, T, S, AT(L, i, j, (k)-1)) + buoyancy_ccc(ph, T, S, AT(L, i, j, (k))))))
            pHY[AT(L, i, j, Nz)] = -ZB(Nz + 1) * dzf_at(g, Nz + 1);
            for (int k = Nz - 1; k >= 1; --k) pHY[AT(L, i, j, k)] = pHY[AT(L, i, j, k + 1)] - ZB(k + 1) * dzf_at(g, k + 1);
#undef ZB
        }
}

/* Adds, to Gu/Gv/Gw that already hold the advective tendency, the remaining terms in the reference's order
 *   G = ((((-div_𝐯u - 0) + x_dot_g_b) - x_f_cross_U) - ∂x pHY′) - ∂ⱼτ₁ⱼ      (…kernel_functions.jl:66-75, 128-137, 193-200)
 * x_dot_g_b = y_dot_g_b = 0 for the default NegativeZDirection gravity (g_dot_b.jl:7-8); in Gw the buoyancy term
 * is z_dot_g_b only when there is no separate hydrostatic pressure (pHY == NULL) (…kernel_functions.jl:141-143).
 * Same index ranges as the advective kernels (Face-located in Bounded starts at 2). */
void ocn_oracle_momentum_extra_tendencies_nu(const ocn_grid *g, const ocn_physics *ph, const double *u, const double *v,
                                             const double *w, const double *T, const double *S, const double *pHY,
                                             const double *nu_e, double *Gu, double *Gv, double *Gw)
{
    const lay Lu = mklay(g, 1, 0, 0), Lv = mklay(g, 0, 1, 0), Lw = mklay(g, 0, 0, 1), Lc = mklay(g, 0, 0, 0);
    const int Nx = g->Nx, Ny = g->Ny, Nz = g->Nz;
    const int ox = (g->tx == OCN_BOUNDED && Nx > 1), oy = (g->ty == OCN_BOUNDED && Ny > 1), oz = (g->tz == OCN_BOUNDED && Nz > 1);
    const int fx = DFLAT(g, 0), fy = DFLAT(g, 1), fz = DFLAT(g, 2);
    const double dx = g->dx, dy = g->dy, nu = ph->nu;
#define U_(i, j, k) u[AT(Lu, i, j, k)]
#define V_(i, j, k) v[AT(Lv, i, j, k)]
#define W_(i, j, k) w[AT(Lw, i, j, k)]
    /* derivative operators (derivative_operators.jl:20-30); δ along Flat = 0 */
#define DXU_C(i, j, k) (fx ? 0.0 : (U_((i) + 1, j, k) - U_(i, j, k)) / dx)                /* ∂xᶜᶜᶜ u */
#define DYV_C(i, j, k) (fy ? 0.0 : (V_(i, (j) + 1, k) - V_(i, j, k)) / dy)                /* ∂yᶜᶜᶜ v */
#define DZW_C(i, j, k) (fz ? 0.0 : (W_(i, j, (k) + 1) - W_(i, j, k)) / dzc_at(g, k))      /* ∂zᶜᶜᶜ w */
#define DYU_FF(i, j, k) (fy ? 0.0 : (U_(i, j, k) - U_(i, (j)-1, k)) / dy)                 /* ∂yᶠᶠᶜ u */
#define DXV_FF(i, j, k) (fx ? 0.0 : (V_(i, j, k) - V_((i)-1, j, k)) / dx)                 /* ∂xᶠᶠᶜ v */
#define DZU_FF(i, j, k) (fz ? 0.0 : (U_(i, j, k) - U_(i, j, (k)-1)) / dzf_at(g, k))       /* ∂zᶠᶜᶠ u */
#define DXW_FF(i, j, k) (fx ? 0.0 : (W_(i, j, k) - W_((i)-1, j, k)) / dx)                 /* ∂xᶠᶜᶠ w */
#define DZV_FF(i, j, k) (fz ? 0.0 : (V_(i, j, k) - V_(i, j, (k)-1)) / dzf_at(g, k))       /* ∂zᶜᶠᶠ v */
#define DYW_FF(i, j, k) (fy ? 0.0 : (W_(i, j, k) - W_(i, (j)-1, k)) / dy)                 /* ∂yᶜᶠᶠ w */
    /* viscous fluxes, isotropic (abstract_scalar_diffusivity_closure.jl:163-175):  -2 * (ν * Σᵢⱼ), Σ₁₂ = 0.5*(∂y u + ∂x v)
     * (velocity_tracer_gradients.jl:15-27) */
    /* viscosity at the stress locations: a number, or (nu_e != NULL) the ccc array νₑ of an eddy-viscosity closure
     * interpolated with ℑxyᶠᶠᵃ / ℑxzᶠᵃᶠ / ℑyzᵃᶠᶠ (abstract_scalar_diffusivity_closure.jl:291-296) */
#define NE(i, j, k) nu_e[AT(Lc, i, j, k)]
#define NU_C(i, j, k) (nu_e ? NE(i, j, k) : nu)
#define NU_FFC(i, j, k) (nu_e ? 0.5 * (0.5 * (NE((i)-1, (j)-1, k) + NE(i, (j)-1, k)) + 0.5 * (NE((i)-1, j, k) + NE(i, j, k))) : nu)
#define NU_FCF(i, j, k) (nu_e ? 0.5 * (0.5 * (NE((i)-1, j, (k)-1) + NE(i, j, (k)-1)) + 0.5 * (NE((i)-1, j, k) + NE(i, j, k))) : nu)
#define NU_CFF(i, j, k) (nu_e ? 0.5 * (0.5 * (NE(i, (j)-1, (k)-1) + NE(i, j, (k)-1)) + 0.5 * (NE(i, (j)-1, k) + NE(i, j, k))) : nu)
#define T11(i, j, k) (-2 * (NU_C(i, j, k) * DXU_C(i, j, k)))
#define T22(i, j, k) (-2 * (NU_C(i, j, k) * DYV_C(i, j, k)))
#define T33(i, j, k) (-2 * (NU_C(i, j, k) * DZW_C(i, j, k)))
#define T12(i, j, k) (-2 * (NU_FFC(i, j, k) * (0.5 * (DYU_FF(i, j, k) + DXV_FF(i, j, k)))))
#define T13(i, j, k) (-2 * (NU_FCF(i, j, k) * (0.5 * (DZU_FF(i, j, k) + DXW_FF(i, j, k)))))
#define T23(i, j, k) (-2 * (NU_CFF(i, j, k) * (0.5 * (DZV_FF(i, j, k) + DYW_FF(i, j, k)))))
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= Nz; ++k)
        for (int j = 1; j <= Ny; ++j)
            for (int i = 1; i <= Nx; ++i) {
                const double Axc = Ax_at(g, k, 0), Ayc = Ay_at(g, k, 0), Az = Az_at(g);
                if (i >= 1 + ox) {
                    double G = Gu[AT(Lu, i, j, k)];
                    if (ph->buoyancy) G = G + 0.0; /* x_dot_g_b = 0 */
                    if (ph->coriolis) { /* x_f_cross_U = -f * active_weighted_ℑxyᶠᶜᶜ(v) (interpolation_operators.jl:127-131) */
                        double an = ixy_fc_active(g, i, j, k);
                        double vi = (an == 0) ? 0.0 : ixy_fc(g, v, &Lv, i, j, k) / an;
                        G = G - (-coriolis_f_at(g, ph, j, 0) * vi);
                    }
                    if (pHY) G = G - (fx ? 0.0 : (pHY[AT(Lc, i, j, k)] - pHY[AT(Lc, i - 1, j, k)]) / dx); /* ∂xᶠᶜᶜ pHY′ */
                    if (ph->closure) { /* ∂ⱼ_τ₁ⱼ (closure_kernel_operators.jl:27-32) */
                        double dxF = fx ? 0.0 : Axc * T11(i, j, k) - Axc * T11(i - 1, j, k);          /* δxᶠᵃᵃ Ax_qᶜᶜᶜ */
                        double dyF = fy ? 0.0 : Ayc * T12(i, j + 1, k) - Ayc * T12(i, j, k);          /* δyᵃᶜᵃ Ay_qᶠᶠᶜ */
                        double dzF = fz ? 0.0 : Az * T13(i, j, k + 1) - Az * T13(i, j, k);            /* δzᵃᵃᶜ Az_qᶠᶜᶠ */
                        G = G - 1 / V_at(g, k, 0) * ((dxF + dyF) + dzF);
                    }
                    Gu[AT(Lu, i, j, k)] = G;
                }
                if (j >= 1 + oy) {
                    double G = Gv[AT(Lv, i, j, k)];
                    if (ph->buoyancy) G = G + 0.0;
                    if (ph->coriolis) { /* y_f_cross_U = f * active_weighted_ℑxyᶜᶠᶜ(u) */
                        double an = ixy_cf_active(g, i, j, k);
                        double ui = (an == 0) ? 0.0 : ixy_cf(g, u, &Lu, i, j, k) / an;
                        G = G - coriolis_f_at(g, ph, j, 1) * ui;
                    }
                    if (pHY) G = G - (fy ? 0.0 : (pHY[AT(Lc, i, j, k)] - pHY[AT(Lc, i, j - 1, k)]) / dy);
                    if (ph->closure) { /* ∂ⱼ_τ₂ⱼ (:34-39) */
                        double dxF = fx ? 0.0 : Axc * T12(i + 1, j, k) - Axc * T12(i, j, k);          /* δxᶜᵃᵃ Ax_qᶠᶠᶜ */
                        double dyF = fy ? 0.0 : Ayc * T22(i, j, k) - Ayc * T22(i, j - 1, k);          /* δyᵃᶠᵃ Ay_qᶜᶜᶜ */
                        double dzF = fz ? 0.0 : Az * T23(i, j, k + 1) - Az * T23(i, j, k);            /* δzᵃᵃᶜ Az_qᶜᶠᶠ */
                        G = G - 1 / V_at(g, k, 0) * ((dxF + dyF) + dzF);
                    }
                    Gv[AT(Lv, i, j, k)] = G;
                }
                if (k >= 1 + oz) {
                    double G = Gw[AT(Lw, i, j, k)];
                    if (ph->buoyancy) { /* maybe_z_dot_g_bᶜᶜᶠ */
                        double zb = 0.0;
                        if (!pHY) zb = fz ? buoyancy_ccc(ph, T, S, AT(Lc, i, j, k))
                                          : 1 * (0.5 * (buoyancy_ccc(ph, T, S, AT(Lc, i, j, k - 1)) + buoyancy_ccc(ph, T, S, AT(Lc, i, j, k))));
                        G = G + zb;
                    }
                    if (ph->coriolis) G = G - 0.0; /* z_f_cross_U = 0 (f_plane.jl:46) */
                    if (ph->closure) { /* ∂ⱼ_τ₃ⱼ (:41-46): areas and volume at (c,c,f) */
                        const double Axf = Ax_at(g, k, 1), Ayf = Ay_at(g, k, 1);
                        double dxF = fx ? 0.0 : Axf * T13(i + 1, j, k) - Axf * T13(i, j, k);          /* δxᶜᵃᵃ Ax_qᶠᶜᶠ */
                        double dyF = fy ? 0.0 : Ayf * T23(i, j + 1, k) - Ayf * T23(i, j, k);          /* δyᵃᶜᵃ Ay_qᶜᶠᶠ */
                        double dzF = fz ? 0.0 : Az * T33(i, j, k) - Az * T33(i, j, k - 1);            /* δzᵃᵃᶠ Az_qᶜᶜᶜ */
                        G = G - 1 / V_at(g, k, 1) * ((dxF + dyF) + dzF);
                    }
                    Gw[AT(Lw, i, j, k)] = G;
                }
            }
#undef T11
#undef T22
#undef T33
#undef T12
#undef T13
#undef T23
#undef NE
#undef NU_C
#undef NU_FFC
#undef NU_FCF
#undef NU_CFF
}
void ocn_oracle_momentum_extra_tendencies(const ocn_grid *g, const ocn_physics *ph, const double *u, const double *v,
                                          const double *w, const double *T, const double *S, const double *pHY, double *Gu,
                                          double *Gv, double *Gw)
{
    ocn_oracle_momentum_extra_tendencies_nu(g, ph, u, v, w, T, S, pHY, NULL, Gu, Gv, Gw);
}

/* Gc <- Gc - ∇_dot_qᶜ (closure_kernel_operators.jl:48-53) with diffusive_flux_x = -(κ * ∂xᶠᶜᶜ c)
 * (abstract_scalar_diffusivity_closure.jl:221-223) */
/* kappa_e != NULL: the ccc array κₑ of an eddy-diffusivity closure, interpolated to the flux faces with ℑxᶠᵃᵃ / ℑyᵃᶠᵃ / ℑzᵃᵃᶠ
 * (abstract_scalar_diffusivity_closure.jl:298-300) */
void ocn_oracle_tracer_diffusion_kappa(const ocn_grid *g, double kappa, const double *kappa_e, const double *c, double *Gc)
{
    const lay L = mklay(g, 0, 0, 0);
    const int fx = DFLAT(g, 0), fy = DFLAT(g, 1), fz = DFLAT(g, 2);
#define C_(i, j, k) c[AT(L, i, j, k)]
#define KE(i, j, k) kappa_e[AT(L, i, j, k)]
#define QX(i, j, k) (-((kappa_e ? 0.5 * (KE((i)-1, j, k) + KE(i, j, k)) : kappa) * ((C_(i, j, k) - C_((i)-1, j, k)) / g->dx)))
#define QY(i, j, k) (-((kappa_e ? 0.5 * (KE(i, (j)-1, k) + KE(i, j, k)) : kappa) * ((C_(i, j, k) - C_(i, (j)-1, k)) / g->dy)))
#define QZ(i, j, k) (-((kappa_e ? 0.5 * (KE(i, j, (k)-1) + KE(i, j, k)) : kappa) * ((C_(i, j, k) - C_(i, j, (k)-1)) / dzf_at(g, k))))
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                const double Ax = Ax_at(g, k, 0), Ay = Ay_at(g, k, 0), Az = Az_at(g);
                double dxF = fx ? 0.0 : Ax * QX(i + 1, j, k) - Ax * QX(i, j, k);
                double dyF = fy ? 0.0 : Ay * QY(i, j + 1, k) - Ay * QY(i, j, k);
                double dzF = fz ? 0.0 : Az * QZ(i, j, k + 1) - Az * QZ(i, j, k);
                Gc[AT(L, i, j, k)] = Gc[AT(L, i, j, k)] - 1 / V_at(g, k, 0) * ((dxF + dyF) + dzF);
            }
#undef C_
#undef KE
#undef QX
#undef QY
#undef QZ
}
void ocn_oracle_tracer_diffusion(const ocn_grid *g, double kappa, const double *c, double *Gc)
{
    ocn_oracle_tracer_diffusion_kappa(g, kappa, NULL, c, Gc);
}

/* Boundary conditions on one side.  kind: 0 default (Periodic / no-flux / impenetrable), 1 Flux, 2 Value, 3 Gradient, 4 Open (a
 * prescribed wall-normal velocity).
 * The condition is  value + coeff * c[interior cell next to the boundary]  (coeff = 0 for a plain number; the
 * coeff form restates a ContinuousBoundaryFunction  f(x, y, t, c, p) = p * c  with field_dependencies = the field itself,
 * whose argument is c[i, j, Nz] (continuous_boundary_function.jl:107-115, 73-74)), or values[(i-1) + n1*(j-1)]
 * when `values` != NULL (an array boundary condition, boundary_condition.jl getbc for AbstractArray). */
typedef struct {
    int32_t kind, _pad;
    double value, coeff;
    const double *values;
} ocn_bc;

static inline double getbc(const ocn_bc *bc, int a, int b, int n1, double c_int)
{
    if (bc->values) return bc->values[(a - 1) + (ptrdiff_t)n1 * (b - 1)];
    if (bc->coeff != 0.0) return bc->value + bc->coeff * c_int;
    return bc->value;
}

/* apply_x/y/z_bcs! (apply_flux_bcs.jl:38-160) along `dir` for a field at `loc`:
 *   left:   G[1] += flux * A(1, flipped loc) / V(1)          right:  G[N] -= flux * A(N+1, flipped loc) / V(N)  */
void ocn_oracle_apply_flux_bcs(const ocn_grid *g, int loc, int dir, const ocn_bc *left, const ocn_bc *right, const double *c,
                               double *G)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
    int N[3] = {g->Nx, g->Ny, g->Nz};
    int d1 = dir == 0 ? 1 : 0, d2 = dir == 2 ? 1 : 2;
    const int zf = (loc >> 2) & 1;
    for (int b = 1; b <= N[d2]; ++b)
        for (int a = 1; a <= N[d1]; ++a) {
            int q[3];
            q[d1] = a;
            q[d2] = b;
            for (int side = 0; side < 2; ++side) {
                const ocn_bc *bc = side ? right : left;
                if (!bc || bc->kind != 1) continue;
                q[dir] = side ? N[dir] : 1;
                const int kk = q[2];
                ptrdiff_t o = AT(L, q[0], q[1], q[2]);
                double area; /* area of the boundary face, with the location flipped along dir */
                if (dir == 0) area = g->dy * dz_at(g, kk, zf);
                else if (dir == 1) area = g->dx * dz_at(g, kk, zf);
                else area = Az_at(g);
                double vol = V_at(g, kk, zf);
                double flux = getbc(bc, a, b, N[d1], c[o]);
                if (side) G[o] -= flux * area / vol;
                else G[o] += flux * area / vol;
            }
        }
}

/* Value / Gradient halo fill on one pair of sides (fill_halo_regions_value_gradient.jl:5-103): the first halo cell is the
 * linear extrapolation  c[0] = c[1] + ∇c * (-Δ),  c[N+1] = c[N] + ∇c * Δ,  Δ = spacing at the boundary face (location flipped
 * along dir); Value: ∇c = (c¹ - v)/(Δ/2) (left), (v - cᴺ)/(Δ/2) (right).  Sides with kind 0/1 get the no-flux fill. */
void ocn_oracle_fill_value_gradient(const ocn_grid *g, int loc, int dir, const ocn_bc *left, const ocn_bc *right, double *c)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
    int N[3] = {g->Nx, g->Ny, g->Nz};
    int d1 = dir == 0 ? 1 : 0, d2 = dir == 2 ? 1 : 2;
    const int face = (loc >> dir) & 1;
    for (int b = 1; b <= N[d2]; ++b)
        for (int a = 1; a <= N[d1]; ++a) {
            int q[3], h[3];
            q[d1] = h[d1] = a;
            q[d2] = h[d2] = b;
            for (int side = 0; side < 2; ++side) {
                const ocn_bc *bc = side ? right : left;
                q[dir] = side ? N[dir] : 1;
                h[dir] = side ? N[dir] + 1 : 0;
                ptrdiff_t oi = AT(L, q[0], q[1], q[2]), oh = AT(L, h[0], h[1], h[2]);
                if (!bc || bc->kind < 2) {
                    c[oh] = c[oi];
                    continue;
                }
                const int ib = side ? N[dir] + 1 : 1; /* boundary index iᴮ */
                double D;
                if (dir == 0) D = g->dx;
                else if (dir == 1) D = g->dy;
                else D = face ? dzc_at(g, ib) : dzf_at(g, ib); /* flipped location */
                double bv = getbc(bc, a, b, N[d1], c[oi]);
                double grad;
                if (bc->kind == 3) grad = bv;
                else grad = side ? (bv - c[oi]) / (D / 2) : (c[oi] - bv) / (D / 2);
                c[oh] = side ? c[oi] + grad * D : c[oi] + grad * (-D);
            }
        }
}

/* =====================================================================================
 * AnisotropicMinimumDissipation (Cb = nothing)
 * turbulence_closure_implementations/anisotropic_minimum_dissipation.jl:125-341, velocity_tracer_gradients.jl:66-140.
 * Filter widths Δᶠx = 2Δx etc. are the ccc values AT THE GIVEN INDEX for every location (:192-203).
 * ===================================================================================== */
typedef struct {
    const ocn_grid *g;
    const double *u, *v, *w, *c;
    lay Lu, Lv, Lw, Lc;
    double Fx, Fy; /* Δᶠx, Δᶠy */
} amd_ctx;
typedef double (*amd_fn)(const amd_ctx *, int, int, int);
/* a Flat direction: its derivatives vanish and its interpolations return the value itself (derivative_operators.jl, interpolation_operators.jl
 * :103-110 on Flat grids; Δ = 1) -- every index along it is the one cell there is */
#define AFI(i) (A->g->tx == OCN_FLAT ? 1 : (i))
#define AFJ(j) (A->g->ty == OCN_FLAT ? 1 : (j))
#define AU(i, j, k) A->u[AT(A->Lu, AFI(i), AFJ(j), k)]
#define AV(i, j, k) A->v[AT(A->Lv, AFI(i), AFJ(j), k)]
#define AW(i, j, k) A->w[AT(A->Lw, AFI(i), AFJ(j), k)]
#define AC(i, j, k) A->c[AT(A->Lc, AFI(i), AFJ(j), k)]
static inline double amd_Fz(const amd_ctx *A, int k) { return 2 * dzc_at(A->g, k); }
static double n_dx_u(const amd_ctx *A, int i, int j, int k) { return (AU(i + 1, j, k) - AU(i, j, k)) / A->g->dx; }
static double n_dy_v(const amd_ctx *A, int i, int j, int k) { return (AV(i, j + 1, k) - AV(i, j, k)) / A->g->dy; }
static double n_dz_w(const amd_ctx *A, int i, int j, int k) { return (AW(i, j, k + 1) - AW(i, j, k)) / dzc_at(A->g, k); }
static double n_dx_v(const amd_ctx *A, int i, int j, int k) { return A->Fx / A->Fy * ((AV(i, j, k) - AV(i - 1, j, k)) / A->g->dx); }
static double n_dy_u(const amd_ctx *A, int i, int j, int k) { return A->Fy / A->Fx * ((AU(i, j, k) - AU(i, j - 1, k)) / A->g->dy); }
static double n_dx_w(const amd_ctx *A, int i, int j, int k) { return A->Fx / amd_Fz(A, k) * ((AW(i, j, k) - AW(i - 1, j, k)) / A->g->dx); }
static double n_dz_u(const amd_ctx *A, int i, int j, int k) { return amd_Fz(A, k) / A->Fx * ((AU(i, j, k) - AU(i, j, k - 1)) / dzf_at(A->g, k)); }
static double n_dy_w(const amd_ctx *A, int i, int j, int k) { return A->Fy / amd_Fz(A, k) * ((AW(i, j, k) - AW(i, j - 1, k)) / A->g->dy); }
static double n_dz_v(const amd_ctx *A, int i, int j, int k) { return amd_Fz(A, k) / A->Fy * ((AV(i, j, k) - AV(i, j, k - 1)) / dzf_at(A->g, k)); }
static double n_dx_c(const amd_ctx *A, int i, int j, int k) { return A->Fx * ((AC(i, j, k) - AC(i - 1, j, k)) / A->g->dx); }
static double n_dy_c(const amd_ctx *A, int i, int j, int k) { return A->Fy * ((AC(i, j, k) - AC(i, j - 1, k)) / A->g->dy); }
static double n_dz_c(const amd_ctx *A, int i, int j, int k) { return amd_Fz(A, k) * ((AC(i, j, k) - AC(i, j, k - 1)) / dzf_at(A->g, k)); }
static double n_S12(const amd_ctx *A, int i, int j, int k) { return 0.5 * (n_dy_u(A, i, j, k) + n_dx_v(A, i, j, k)); }
static double n_S13(const amd_ctx *A, int i, int j, int k) { return 0.5 * (n_dz_u(A, i, j, k) + n_dx_w(A, i, j, k)); }
static double n_S23(const amd_ctx *A, int i, int j, int k) { return 0.5 * (n_dz_v(A, i, j, k) + n_dy_w(A, i, j, k)); }
#define AMD_SQ(name, f) static double name(const amd_ctx *A, int i, int j, int k) { double t = f(A, i, j, k); return t * t; }
#define AMD_PR(name, f, g2) static double name(const amd_ctx *A, int i, int j, int k) { return f(A, i, j, k) * g2(A, i, j, k); }
AMD_SQ(n_dx_v2, n_dx_v) AMD_SQ(n_dy_u2, n_dy_u) AMD_SQ(n_dx_w2, n_dx_w) AMD_SQ(n_dz_u2, n_dz_u) AMD_SQ(n_dy_w2, n_dy_w) AMD_SQ(n_dz_v2, n_dz_v)
AMD_SQ(n_dx_c2, n_dx_c) AMD_SQ(n_dy_c2, n_dy_c) AMD_SQ(n_dz_c2, n_dz_c)
AMD_PR(n_dx_v_S12, n_dx_v, n_S12) AMD_PR(n_dy_u_S12, n_dy_u, n_S12) AMD_PR(n_dx_w_S13, n_dx_w, n_S13) AMD_PR(n_dz_u_S13, n_dz_u, n_S13)
AMD_PR(n_dz_v_S23, n_dz_v, n_S23) AMD_PR(n_dy_w_S23, n_dy_w, n_S23)
/* interpolations of functions (interpolation_operators.jl:20-26, 44-57) */
static inline double Ix_c(const amd_ctx *A, amd_fn f, int i, int j, int k) { return 0.5 * (f(A, i, j, k) + f(A, i + 1, j, k)); }
static inline double Iy_c(const amd_ctx *A, amd_fn f, int i, int j, int k) { return 0.5 * (f(A, i, j, k) + f(A, i, j + 1, k)); }
static inline double Iz_c(const amd_ctx *A, amd_fn f, int i, int j, int k) { return 0.5 * (f(A, i, j, k) + f(A, i, j, k + 1)); }
static inline double Ixy_cc(const amd_ctx *A, amd_fn f, int i, int j, int k) { return 0.5 * (Ix_c(A, f, i, j, k) + Ix_c(A, f, i, j + 1, k)); }
static inline double Ixz_cc(const amd_ctx *A, amd_fn f, int i, int j, int k) { return 0.5 * (Ix_c(A, f, i, j, k) + Ix_c(A, f, i, j, k + 1)); }
static inline double Iyz_cc(const amd_ctx *A, amd_fn f, int i, int j, int k) { return 0.5 * (Iy_c(A, f, i, j, k) + Iy_c(A, f, i, j, k + 1)); }

static amd_ctx amd_make(const ocn_grid *g, const double *u, const double *v, const double *w, const double *c)
{
    amd_ctx A;
    A.g = g; A.u = u; A.v = v; A.w = w; A.c = c;
    A.Lu = mklay(g, 1, 0, 0); A.Lv = mklay(g, 0, 1, 0); A.Lw = mklay(g, 0, 0, 1); A.Lc = mklay(g, 0, 0, 0);
    A.Fx = 2 * g->dx; A.Fy = 2 * g->dy;
    return A;
}
static inline double amd_delta2(const amd_ctx *A, int k)
{
    double Fz = amd_Fz(A, k);
    return 3 / ((1 / (A->Fx * A->Fx) + 1 / (A->Fy * A->Fy)) + 1 / (Fz * Fz));
}
static inline double julia_max0(double x) { return (x > 0 || x != x) ? x : 0.0; }

/* _compute_AMD_viscosity! (:125-147) over :xyz; q = norm_tr_∇uᶜᶜᶜ (:263-281), r = norm_uᵢₐ_uⱼₐ_Σᵢⱼᶜᶜᶜ (:208-257) */
void ocn_oracle_amd_viscosity(const ocn_grid *g, double Cnu, const double *u, const double *v, const double *w, double *nu_e)
{
    const amd_ctx ctx = amd_make(g, u, v, w, NULL);
    const amd_ctx *A = &ctx;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                const double dxu = n_dx_u(A, i, j, k), dyv = n_dy_v(A, i, j, k), dzw = n_dz_w(A, i, j, k);
                const double q = (((((((dxu * dxu + dyv * dyv) + dzw * dzw) + Ixy_cc(A, n_dx_v2, i, j, k)) + Ixy_cc(A, n_dy_u2, i, j, k)) +
                                    Ixz_cc(A, n_dx_w2, i, j, k)) + Ixz_cc(A, n_dz_u2, i, j, k)) + Iyz_cc(A, n_dy_w2, i, j, k)) + Iyz_cc(A, n_dz_v2, i, j, k);
                double nu = 0.0;
                if (q != 0) {
                    const double r1 = ((((dxu * (dxu * dxu) + dyv * Ixy_cc(A, n_dx_v2, i, j, k)) + dzw * Ixz_cc(A, n_dx_w2, i, j, k)) +
                                        2 * dxu * Ixy_cc(A, n_dx_v_S12, i, j, k)) + 2 * dxu * Ixz_cc(A, n_dx_w_S13, i, j, k)) +
                                      2 * Ixy_cc(A, n_dx_v, i, j, k) * Ixz_cc(A, n_dx_w, i, j, k) * Iyz_cc(A, n_S23, i, j, k);
                    const double r2 = ((((dxu * Ixy_cc(A, n_dy_u2, i, j, k) + dyv * (dyv * dyv)) + dzw * Iyz_cc(A, n_dy_w2, i, j, k)) +
                                        2 * dyv * Ixy_cc(A, n_dy_u_S12, i, j, k)) +
                                       2 * Ixy_cc(A, n_dy_u, i, j, k) * Iyz_cc(A, n_dy_w, i, j, k) * Ixz_cc(A, n_S13, i, j, k)) +
                                      2 * dyv * Iyz_cc(A, n_dy_w_S23, i, j, k);
                    const double r3 = ((((dxu * Ixz_cc(A, n_dz_u2, i, j, k) + dyv * Iyz_cc(A, n_dz_v2, i, j, k)) + dzw * (dzw * dzw)) +
                                        2 * Ixz_cc(A, n_dz_u, i, j, k) * Iyz_cc(A, n_dz_v, i, j, k) * Ixy_cc(A, n_S12, i, j, k)) +
                                       2 * dzw * Ixz_cc(A, n_dz_u_S13, i, j, k)) + 2 * dzw * Iyz_cc(A, n_dz_v_S23, i, j, k);
                    const double r = (r1 + r2) + r3;
                    const double Cb_zeta = 0.0 / amd_Fz(A, k); /* Cb = nothing */
                    nu = -Cnu * amd_delta2(A, k) * (r - Cb_zeta) / q;
                }
                nu_e[AT(A->Lc, i, j, k)] = julia_max0(nu);
            }
}

/* _compute_AMD_diffusivity! (:149-169); σ = norm_θᵢ²ᶜᶜᶜ (:325-327), ϑ = norm_uᵢⱼ_cⱼ_cᵢᶜᶜᶜ (:297-323), including the
 * ℑxzᶜᵃᶜ (not ℑyz) interpolation of norm_∂y_w exactly as the reference writes it (:313) */
void ocn_oracle_amd_diffusivity(const ocn_grid *g, double Ck, const double *u, const double *v, const double *w, const double *c,
                                double *kappa_e)
{
    const amd_ctx ctx = amd_make(g, u, v, w, c);
    const amd_ctx *A = &ctx;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                const double sigma = (Ix_c(A, n_dx_c2, i, j, k) + Iy_c(A, n_dy_c2, i, j, k)) + Iz_c(A, n_dz_c2, i, j, k);
                double kap = 0.0;
                if (sigma != 0) {
                    const double cx = Ix_c(A, n_dx_c, i, j, k), cy = Iy_c(A, n_dy_c, i, j, k), cz = Iz_c(A, n_dz_c, i, j, k);
                    const double cx_ux = (n_dx_u(A, i, j, k) * Ix_c(A, n_dx_c2, i, j, k) + Ixy_cc(A, n_dx_v, i, j, k) * cx * cy) +
                                         Ixz_cc(A, n_dx_w, i, j, k) * cx * cz;
                    const double cy_uy = (Ixy_cc(A, n_dy_u, i, j, k) * cy * cx + n_dy_v(A, i, j, k) * Iy_c(A, n_dy_c2, i, j, k)) +
                                         Ixz_cc(A, n_dy_w, i, j, k) * cy * cz;
                    const double cz_uz = (Ixz_cc(A, n_dz_u, i, j, k) * cz * cx + Iyz_cc(A, n_dz_v, i, j, k) * cz * cy) +
                                         n_dz_w(A, i, j, k) * Iz_c(A, n_dz_c2, i, j, k);
                    const double theta = (cx_ux + cy_uy) + cz_uz;
                    kap = -Ck * amd_delta2(A, k) * theta / sigma;
                }
                kappa_e[AT(A->Lc, i, j, k)] = julia_max0(kap);
            }
}
#undef AU
#undef AV
#undef AW
#undef AC
#undef AFI
#undef AFJ

/* =====================================================================================
 * Time-stepper kernels
 * ===================================================================================== */
/* loc: bit0 x-face, bit1 y-face, bit2 z-face.  Range = :xyz with exclude_periphery=true. */
static void field_range(const ocn_grid *g, int loc, int *i0, int *j0, int *k0)
{
    *i0 = 1 + ((loc & 1) && g->tx == OCN_BOUNDED && g->Nx > 1);
    *j0 = 1 + ((loc & 2) && g->ty == OCN_BOUNDED && g->Ny > 1);
    *k0 = 1 + ((loc & 4) && g->tz == OCN_BOUNDED && g->Nz > 1);
}
/* K5 rk3_substep_field! (runge_kutta_3.jl:194-208) */
void ocn_oracle_rk3_substep(const ocn_grid *g, int loc, double *U, const double *Gn, const double *Gm, double dt,
                            double gamma, double zeta, int has_zeta)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
    int i0, j0, k0;
    field_range(g, loc, &i0, &j0, &k0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = k0; k <= g->Nz; ++k)
        for (int j = j0; j <= g->Ny; ++j)
            for (int i = i0; i <= g->Nx; ++i) {
                ptrdiff_t a = AT(L, i, j, k);
                if (has_zeta)
                    U[a] += dt * (gamma * Gn[a] + zeta * Gm[a]);
                else
                    U[a] += (dt * gamma) * Gn[a];
            }
}
/* K6 ab2_step_field! (quasi_adams_bashforth_2.jl:162-175) */
void ocn_oracle_ab2_step(const ocn_grid *g, int loc, double *U, const double *Gn, const double *Gm, double dt, double chi)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
    int i0, j0, k0;
    field_range(g, loc, &i0, &j0, &k0);
    /* not_euler is a Julia Bool: x * false is a "strong zero" (0.0 even for NaN x), protecting against
     * leftover NaNs in G⁻ (quasi_adams_bashforth_2.jl:169) */
    int not_euler = (chi != -0.5);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = k0; k <= g->Nz; ++k)
        for (int j = j0; j <= g->Ny; ++j)
            for (int i = i0; i <= g->Nx; ++i) {
                ptrdiff_t a = AT(L, i, j, k);
                double G = (1.5 + chi) * Gn[a] - (not_euler ? (0.5 + chi) * Gm[a] : 0.0);
                U[a] += dt * G;
            }
}
/* K7 _cache_field_tendencies! (store_tendencies.jl:6-9): G⁻ <- Gⁿ over :xyz (interior 1:N) */
void ocn_oracle_cache_tendency(const ocn_grid *g, int loc, double *Gm, const double *Gn)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) Gm[AT(L, i, j, k)] = Gn[AT(L, i, j, k)];
}

/* =====================================================================================
 * Pressure: source term, correction
 * ===================================================================================== */
/* divᶜᶜᶜ (divergence_operators.jl:16-19) */
static inline double div_ccc(const ocn_grid *g, const double *u, const double *v, const double *w, const lay *Lu,
                             const lay *Lv, const lay *Lw, int i, int j, int k)
{
    double dxu = DFLAT(g, 0) ? 0.0 : Ax_at(g, k, 0) * u[AT(*Lu, i + 1, j, k)] - Ax_at(g, k, 0) * u[AT(*Lu, i, j, k)];
    double dyv = DFLAT(g, 1) ? 0.0 : Ay_at(g, k, 0) * v[AT(*Lv, i, j + 1, k)] - Ay_at(g, k, 0) * v[AT(*Lv, i, j, k)];
    double dzw = DFLAT(g, 2) ? 0.0 : Az_at(g) * w[AT(*Lw, i, j, k + 1)] - Az_at(g) * w[AT(*Lw, i, j, k)];
    return (1 / V_at(g, k, 0)) * ((dxu + dyv) + dzw);
}
void ocn_oracle_divergence(const ocn_grid *g, const double *u, const double *v, const double *w, double *div /* Nx*Ny*Nz, no halo */)
{
    lay Lu = mklay(g, 1, 0, 0), Lv = mklay(g, 0, 1, 0), Lw = mklay(g, 0, 0, 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i)
                div[(i - 1) + (ptrdiff_t)g->Nx * ((j - 1) + (ptrdiff_t)g->Ny * (k - 1))] = div_ccc(g, u, v, w, &Lu, &Lv, &Lw, i, j, k);
}
/* K8 _compute_source_term! (solve_for_pressure.jl:12-17) and K9 _fourier_tridiagonal_source_term!
 * ZDirection (:33-38).  rhs is an interleaved complex array Nx*Ny*Nz (no halo). */
void ocn_oracle_source_term(const ocn_grid *g, const double *u, const double *v, const double *w, double dt, int times_dz,
                            double *rhs_complex)
{
    lay Lu = mklay(g, 1, 0, 0), Lv = mklay(g, 0, 1, 0), Lw = mklay(g, 0, 0, 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                double d = div_ccc(g, u, v, w, &Lu, &Lv, &Lw, i, j, k);
                double r = times_dz ? (dzc_at(g, k) * d) / dt : d / dt; /* active * Δz * δ / Δt, left-assoc */
                ptrdiff_t a = (i - 1) + (ptrdiff_t)g->Nx * ((j - 1) + (ptrdiff_t)g->Ny * (k - 1));
                rhs_complex[2 * a] = r;
                rhs_complex[2 * a + 1] = 0.0;
            }
}
/* K13 copy_real_component! (fft_based_poisson_solver.jl:129-137) */
void ocn_oracle_copy_real(const ocn_grid *g, const double *phi_complex, double *p)
{
    lay L = mklay(g, 0, 0, 0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i)
                p[AT(L, i, j, k)] = phi_complex[2 * ((i - 1) + (ptrdiff_t)g->Nx * ((j - 1) + (ptrdiff_t)g->Ny * (k - 1)))];
}
/* K17 _pressure_correct_velocities! (pressure_correction.jl:31-37); :xyz over 1:N in all dims.
 * ∂xᶠᶜᶜ = δxᶠᵃᵃ(p)/Δxᶠᶜᶜ (derivative_operators.jl:20-30); δ along a Flat dim is zero. */
void ocn_oracle_pressure_correct(const ocn_grid *g, double *u, double *v, double *w, const double *p, double dt)
{
    lay Lu = mklay(g, 1, 0, 0), Lv = mklay(g, 0, 1, 0), Lw = mklay(g, 0, 0, 1), Lp = mklay(g, 0, 0, 0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                double pc = p[AT(Lp, i, j, k)];
                double px = DFLAT(g, 0) ? 0.0 : (pc - p[AT(Lp, i - 1, j, k)]) / g->dx;
                double py = DFLAT(g, 1) ? 0.0 : (pc - p[AT(Lp, i, j - 1, k)]) / g->dy;
                double pz = DFLAT(g, 2) ? 0.0 : (pc - p[AT(Lp, i, j, k - 1)]) / dzf_at(g, k);
                u[AT(Lu, i, j, k)] -= px * dt;
                v[AT(Lv, i, j, k)] -= py * dt;
                w[AT(Lw, i, j, k)] -= pz * dt;
            }
}
/* ∇²ᶜᶜᶜ (laplacian_operators.jl:36-40) for the Poisson property tests:
 * 1/V * (δx(Ax ∂x c) + δy(Ay ∂y c) + δz(Az ∂z c)) */
void ocn_oracle_laplacian(const ocn_grid *g, const double *p, double *lap /* Nx*Ny*Nz */)
{
    lay L = mklay(g, 0, 0, 0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->Nz; ++k)
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
#define P(a, b, c) p[AT(L, a, b, c)]
                double fx1 = Ax_at(g, k, 0) * ((P(i + 1, j, k) - P(i, j, k)) / g->dx), fx0 = Ax_at(g, k, 0) * ((P(i, j, k) - P(i - 1, j, k)) / g->dx);
                double fy1 = Ay_at(g, k, 0) * ((P(i, j + 1, k) - P(i, j, k)) / g->dy), fy0 = Ay_at(g, k, 0) * ((P(i, j, k) - P(i, j - 1, k)) / g->dy);
                double fz1 = Az_at(g) * ((P(i, j, k + 1) - P(i, j, k)) / dzf_at(g, k + 1)), fz0 = Az_at(g) * ((P(i, j, k) - P(i, j, k - 1)) / dzf_at(g, k));
#undef P
                double sx = DFLAT(g, 0) ? 0.0 : fx1 - fx0, sy = DFLAT(g, 1) ? 0.0 : fy1 - fy0, sz = DFLAT(g, 2) ? 0.0 : fz1 - fz0;
                lap[(i - 1) + (ptrdiff_t)g->Nx * ((j - 1) + (ptrdiff_t)g->Ny * (k - 1))] = (1 / V_at(g, k, 0)) * ((sx + sy) + sz);
            }
}

/* =====================================================================================
 * Halo fills
 * ===================================================================================== */
/* K18 fill_periodic_*_halo! (fill_halo_regions_periodic.jl:40-71): on the parent array, over the
 * full parent cross-section (corners included).  dir 0/1/2.  (sx,sy,sz) parent extents. */
void ocn_oracle_fill_periodic(double *c, int sx, int sy, int sz, int dir, int N, int H)
{
    ptrdiff_t s[3] = {1, sx, (ptrdiff_t)sx * sy};
    int e[3] = {sx, sy, sz};
    int d1 = dir == 0 ? 1 : 0, d2 = dir == 2 ? 1 : 2;
    for (int b = 0; b < e[d2]; ++b)
        for (int a = 0; a < e[d1]; ++a) {
            double *base = c + a * s[d1] + b * s[d2];
            for (int h = 0; h < H; ++h) {
                base[h * s[dir]] = base[(N + h) * s[dir]];         /* c[i]     = c[N+i]  (1-based parent) */
                base[(N + H + h) * s[dir]] = base[(H + h) * s[dir]]; /* c[N+H+i] = c[H+i] */
            }
        }
}
/* K19 flux (no-flux) fill along z: c[.,.,0] = c[.,.,1], c[.,.,Nz+1] = c[.,.,Nz]
 * (fill_halo_regions_flux.jl:14-33), over the interior (i,j) range (fill_halo_size for a 2-D :xy side).
 * Works for any direction `dir`; N = interior size of the field along dir. */
void ocn_oracle_fill_flux(const ocn_grid *g, int loc, double *c, int dir)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
    int N[3] = {g->Nx, g->Ny, g->Nz};
    int d1 = dir == 0 ? 1 : 0, d2 = dir == 2 ? 1 : 2;
    for (int b = 1; b <= N[d2]; ++b)
        for (int a = 1; a <= N[d1]; ++a) {
            int lo[3], hi[3], lo_src[3], hi_src[3];
            lo[d1] = hi[d1] = lo_src[d1] = hi_src[d1] = a;
            lo[d2] = hi[d2] = lo_src[d2] = hi_src[d2] = b;
            lo[dir] = 0;
            lo_src[dir] = 1;
            hi[dir] = N[dir] + 1;
            hi_src[dir] = N[dir];
            c[AT(L, lo[0], lo[1], lo[2])] = c[AT(L, lo_src[0], lo_src[1], lo_src[2])];
            c[AT(L, hi[0], hi[1], hi[2])] = c[AT(L, hi_src[0], hi_src[1], hi_src[2])];
        }
}
/* Open fill (fill_halo_regions_open.jl:65-70): the wall-normal velocity on the two boundary faces is set to getbc(bc, ...) -- 0 for the
 * default Impenetrable condition Open(nothing) (boundary_condition.jl:90,113), the number / array of an OpenBoundaryCondition(value)
 * (kind 4) otherwise; left / right may be NULL (default). */
void ocn_oracle_fill_open_bcs(const ocn_grid *g, int loc, double *c, int dir, const ocn_bc *left, const ocn_bc *right)
{
    lay L = mklay(g, loc & 1, (loc >> 1) & 1, (loc >> 2) & 1);
    int N[3] = {g->Nx, g->Ny, g->Nz};
    int d1 = dir == 0 ? 1 : 0, d2 = dir == 2 ? 1 : 2;
    for (int b = 1; b <= N[d2]; ++b)
        for (int a = 1; a <= N[d1]; ++a) {
            int lo[3], hi[3];
            lo[d1] = hi[d1] = a;
            lo[d2] = hi[d2] = b;
            lo[dir] = 1;
            hi[dir] = N[dir] + 1;
            c[AT(L, lo[0], lo[1], lo[2])] = (left && left->kind == 4) ? getbc(left, a, b, N[d1], 0.0) : 0.0;
            c[AT(L, hi[0], hi[1], hi[2])] = (right && right->kind == 4) ? getbc(right, a, b, N[d1], 0.0) : 0.0;
        }
}
void ocn_oracle_fill_open(const ocn_grid *g, int loc, double *c, int dir) { ocn_oracle_fill_open_bcs(g, loc, c, dir, NULL, NULL); }

/* =====================================================================================
 * Fourier-tridiagonal solver pieces
 * ===================================================================================== */
/* K15 compute_main_diagonal! ZDirection (fourier_tridiagonal_poisson_solver.jl:41-51) */
void ocn_oracle_main_diagonal_z(const ocn_grid *g, const double *lx, const double *ly, double *D /* Nx*Ny*Nz */)
{
    const int Nx = g->Nx, Ny = g->Ny, Nz = g->Nz;
    for (int j = 1; j <= Ny; ++j)
        for (int i = 1; i <= Nx; ++i) {
            double lam = lx[i - 1] + ly[j - 1];
#define DD(k) D[(i - 1) + (ptrdiff_t)Nx * ((j - 1) + (ptrdiff_t)Ny * ((k)-1))]
            DD(1) = -1 / dzf_at(g, 2) - dzc_at(g, 1) * lam;
            for (int k = 2; k <= Nz - 1; ++k) DD(k) = -(1 / dzf_at(g, k + 1) + 1 / dzf_at(g, k)) - dzc_at(g, k) * lam;
            DD(Nz) = -1 / dzf_at(g, Nz) - dzc_at(g, Nz) * lam;
#undef DD
        }
}
/* complex helpers with Julia's semantics for Complex/Real and Real*Complex (componentwise) */
/* K14 solve_batched_tridiagonal_system_z! (batched_tridiagonal_solver.jl:209-235).
 * a,c: real 1-D (Nz-1), b: real 3-D, f/phi: complex 3-D interleaved, t: real 3-D scratch.
 * phi holds its previous contents on entry (needed for the "not diagonally dominant" guard). */
void ocn_oracle_tridiag_solve_z(int Nx, int Ny, int Nz, const double *a, const double *b, const double *c, const double *f,
                                double *t, double *phi)
{
    const double tiny = 10 * 2.220446049250313e-16; /* 10*eps(Float64) */
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
#define IX(k) ((ptrdiff_t)i + (ptrdiff_t)Nx * ((ptrdiff_t)j + (ptrdiff_t)Ny * (k)))
            double beta = b[IX(0)];
            phi[2 * IX(0)] = f[2 * IX(0)] / beta;
            phi[2 * IX(0) + 1] = f[2 * IX(0) + 1] / beta;
            for (int k = 1; k < Nz; ++k) {
                double ck = c[k - 1], bk = b[IX(k)], ak = a[k - 1];
                t[IX(k)] = ck / beta;
                beta = bk - ak * t[IX(k)];
                int dd = fabs(beta) > tiny;
                double sr = (f[2 * IX(k)] - ak * phi[2 * IX(k - 1)]) / beta;
                double si = (f[2 * IX(k) + 1] - ak * phi[2 * IX(k - 1) + 1]) / beta;
                if (dd) {
                    phi[2 * IX(k)] = sr;
                    phi[2 * IX(k) + 1] = si;
                }
            }
            for (int k = Nz - 2; k >= 0; --k) {
                phi[2 * IX(k)] -= t[IX(k + 1)] * phi[2 * IX(k + 1)];
                phi[2 * IX(k) + 1] -= t[IX(k + 1)] * phi[2 * IX(k + 1) + 1];
            }
#undef IX
        }
}

/* 1-D reconstruction probes for unit tests (order-of-accuracy, known answers) */
double ocn_oracle_weno5(const double *S6, int left) { return weno5(S6, left); }
double ocn_oracle_upwind5(const double *S6, int left) { return upwind5(S6, left); }
double ocn_oracle_upwind3(const double *S4, int left) { return upwind3(S4, left); }
double ocn_oracle_weno3(const double *S4, int left) { return weno3(S4, left); }
double ocn_oracle_centered4(const double *S4) { return ((C4[0] * S4[0] + C4[1] * S4[1]) + C4[2] * S4[2]) + C4[3] * S4[3]; }
