// physics.hip -- SURVEY §8(f) rank 1: everything in u/v/w_velocity_tendency and tracer_tendency besides WENO advection.
//
//   * advection = Centered(order=2)            src/Advection/centered_advective_fluxes.jl:7-25 (the reference's default scheme)
//   * coriolis  = FPlane(f)                    src/Coriolis/f_plane.jl:44-46, Operators/interpolation_operators.jl:121-131
//   * closure   = ScalarDiffusivity(ν, κ)      src/TurbulenceClosures/abstract_scalar_diffusivity_closure.jl:158-223,
//                                              closure_kernel_operators.jl:27-53, velocity_tracer_gradients.jl:15-27
//   * buoyancy  = BuoyancyTracer / SeawaterBuoyancy(LinearEquationOfState) through the hydrostatic pressure anomaly
//                                              src/Models/NonhydrostaticModels/update_hydrostatic_pressure.jl:12-53,
//                                              BuoyancyFormulations/{buoyancy_tracer.jl:12, linear_equation_of_state.jl:58-66, g_dot_b.jl:1-8}
//   * flux boundary conditions (top / bottom)  src/BoundaryConditions/apply_flux_bcs.jl:38-160
//
// The extra momentum terms are ADDED to a G that already holds the advective tendency, in the reference's order
//   G = ((((-div_𝐯u - 0) + x_dot_g_b) - x_f_cross_U) - ∂x pHY′) - ∂ⱼτ₁ⱼ      (nonhydrostatic_tendency_kernel_functions.jl:66-75)
// so the strict build (-ffp-contract=off, true divisions) is bit-identical to the CPU oracle.  The fast build multiplies
// by reciprocal spacings and lets the compiler contract to FMA.  All kernels are one thread per cell, x fastest across
// the 64 lanes (512-B coalesced rows); x and y are Periodic in the supported scope, so every field shares one set of
// strides and active_weighted_ℑxy divides by exactly 1.
#include <cstdlib>
#include <cstring>

#include "ocn_weno.h"

namespace OCN_NS {

using ocn::GridDev;
using ocn::Lay;
using ocn::TermsDev;

struct PRange {
    int i0, i1, j0, j1, k0, k1;
    int ow;  // first k written for Gw (periphery exclusion of Face-in-Bounded)
};

// XCD-aware workgroup -> tile mapping (see block_coords in tendencies.hip): each of the 8 XCDs walks a contiguous range of tiles
__device__ __forceinline__ void xcd_block_coords(int on, int &bx, int &by, int &bz)
{
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (!on) return;
    const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
    const unsigned b = bx + nx * (by + ny * bz);
    const unsigned q = b & 7u, chunk = n >> 3, rem = n & 7u;
    const unsigned logical = q * chunk + (q < rem ? q : rem) + (b >> 3);
    bx = logical % nx;
    by = (logical / nx) % ny;
    bz = logical / (nx * ny);
}
static int xcd_remap_on()
{
    static const int v = getenv("OCN_XCD_REMAP") ? atoi(getenv("OCN_XCD_REMAP")) : 1;
    return v;
}

static int make_prange(const ocn_grid *grid, const int32_t *range, PRange &r)
{
    if (range) {
        r.i0 = range[0]; r.i1 = range[1]; r.j0 = range[2]; r.j1 = range[3]; r.k0 = range[4]; r.k1 = range[5];
        if (r.i0 < 1 || r.i1 > grid->Nx || r.j0 < 1 || r.j1 > grid->Ny || r.k0 < 1 || r.k1 > grid->Nz) {
            ocn::set_error("tendency range {%d:%d,%d:%d,%d:%d} outside the interior %dx%dx%d", r.i0, r.i1, r.j0, r.j1, r.k0,
                           r.k1, grid->Nx, grid->Ny, grid->Nz);
            return OCN_ERR_INVALID_ARGUMENT;
        }
        r.ow = 1;
    } else {
        r.i0 = 1; r.i1 = grid->Nx; r.j0 = 1; r.j1 = grid->Ny; r.k0 = 1; r.k1 = grid->Nz;
        r.ow = (grid->tz == OCN_BOUNDED && grid->Nz > 1) ? 2 : 1;
    }
    return OCN_SUCCESS;
}

// a / d  in the strict build;  a * (1/d) with the reciprocal hoisted in the fast build
#if OCN_STRICT
#define OCN_DIV(a, d, rd) ((a) / (d))
#else
#define OCN_DIV(a, d, rd) ((a) * (rd))
#endif

// ---------------------------------------------------------------------------------------------------
// Centered(order=2) momentum advection: flux = A(flux location) * sym(U) * sym(u), left-associated
// ---------------------------------------------------------------------------------------------------
template <int TZ>
__global__ __launch_bounds__(256) void momentum_tendencies_centered2(GridDev g, const double *__restrict__ u,
                                                                     const double *__restrict__ v,
                                                                     const double *__restrict__ w, double *__restrict__ Gu,
                                                                     double *__restrict__ Gv, double *__restrict__ Gw, PRange r)
{
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1) return;
    constexpr bool ZF = (TZ == OCN_FLAT);
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);  // strides and offset shared by all locations (x, y Periodic)
    const long long s2 = L.s2, s3 = ZF ? 0 : L.s3, o = ocn::at(L, i, j, k);
    const double *pu = u + o, *pv = v + o, *pw = w + o;
#define U_(a, b, c) pu[(a) + (b)*s2 + (c)*s3]
#define V_(a, b, c) pv[(a) + (b)*s2 + (c)*s3]
#define W_(a, b, c) pw[(a) + (b)*s2 + (c)*s3]
#define AVG(x, y) (0.5 * (x) + 0.5 * (y))
    const double Axc = M.Ax(k), Ayc = M.Ay(k), Az = M.Az;
    {   // Gu
        const double ue = AVG(U_(0, 0, 0), U_(1, 0, 0)), uw = AVG(U_(-1, 0, 0), U_(0, 0, 0));
        const double fx = (Axc * ue) * ue - (Axc * uw) * uw;                                             // δxᶠᵃᵃ F_Uu
        const double fyn = (Ayc * AVG(V_(-1, 1, 0), V_(0, 1, 0))) * AVG(U_(0, 0, 0), U_(0, 1, 0));       // F_Vu(j+1)
        const double fys = (Ayc * AVG(V_(-1, 0, 0), V_(0, 0, 0))) * AVG(U_(0, -1, 0), U_(0, 0, 0));      // F_Vu(j)
        double dzF = 0.0;
        if (!ZF) {
            const double ft = (Az * AVG(W_(-1, 0, 1), W_(0, 0, 1))) * AVG(U_(0, 0, 0), U_(0, 0, 1));     // F_Wu(k+1)
            const double fb = (Az * AVG(W_(-1, 0, 0), W_(0, 0, 0))) * AVG(U_(0, 0, -1), U_(0, 0, 0));    // F_Wu(k)
            dzF = ft - fb;
        }
        const double rV = 1 / (Az * M.dzC(k));
        Gu[o] = -(rV * ((fx + (fyn - fys)) + dzF));
    }
    {   // Gv
        const double fxe = (Axc * AVG(U_(1, -1, 0), U_(1, 0, 0))) * AVG(V_(0, 0, 0), V_(1, 0, 0));       // F_Uv(i+1)
        const double fxw = (Axc * AVG(U_(0, -1, 0), U_(0, 0, 0))) * AVG(V_(-1, 0, 0), V_(0, 0, 0));      // F_Uv(i)
        const double vn = AVG(V_(0, 0, 0), V_(0, 1, 0)), vs = AVG(V_(0, -1, 0), V_(0, 0, 0));
        const double fy = (Ayc * vn) * vn - (Ayc * vs) * vs;                                             // δyᵃᶠᵃ F_Vv
        double dzF = 0.0;
        if (!ZF) {
            const double ft = (Az * AVG(W_(0, -1, 1), W_(0, 0, 1))) * AVG(V_(0, 0, 0), V_(0, 0, 1));     // F_Wv(k+1)
            const double fb = (Az * AVG(W_(0, -1, 0), W_(0, 0, 0))) * AVG(V_(0, 0, -1), V_(0, 0, 0));    // F_Wv(k)
            dzF = ft - fb;
        }
        const double rV = 1 / (Az * M.dzC(k));
        Gv[o] = -(rV * (((fxe - fxw) + fy) + dzF));
    }
    if (k >= r.ow) {  // Gw: areas and volume at (c,c,f)
        const double Axf = M.dy * M.dzF(k), Ayf = M.dx * M.dzF(k);
        // sym_z of u, v to the z face; along a Flat z the value itself (flat_advective_fluxes.jl:26-44)
        const double uze = ZF ? U_(1, 0, 0) : AVG(U_(1, 0, -1), U_(1, 0, 0)), uzw = ZF ? U_(0, 0, 0) : AVG(U_(0, 0, -1), U_(0, 0, 0));
        const double vzn = ZF ? V_(0, 1, 0) : AVG(V_(0, 1, -1), V_(0, 1, 0)), vzs = ZF ? V_(0, 0, 0) : AVG(V_(0, 0, -1), V_(0, 0, 0));
        const double fxe = (Axf * uze) * AVG(W_(0, 0, 0), W_(1, 0, 0));                                  // F_Uw(i+1)
        const double fxw = (Axf * uzw) * AVG(W_(-1, 0, 0), W_(0, 0, 0));                                 // F_Uw(i)
        const double fyn = (Ayf * vzn) * AVG(W_(0, 0, 0), W_(0, 1, 0));                                  // F_Vw(j+1)
        const double fys = (Ayf * vzs) * AVG(W_(0, -1, 0), W_(0, 0, 0));                                 // F_Vw(j)
        double dzF = 0.0;
        if (!ZF) {
            const double wt = AVG(W_(0, 0, 0), W_(0, 0, 1)), wb = AVG(W_(0, 0, -1), W_(0, 0, 0));
            dzF = (Az * wt) * wt - (Az * wb) * wb;                                                       // δzᵃᵃᶠ F_Ww
        }
        const double rV = 1 / (Az * M.dzF(k));
        Gw[o] = -(rV * (((fxe - fxw) + (fyn - fys)) + dzF));
    }
}

// Centered(order=2) tracer advection (+ optional diffusion, see tracer_diffusion below): Ax_q(U) * sym(c)
template <int TZ>
__global__ __launch_bounds__(256) void tracer_tendency_centered2(GridDev g, const double *__restrict__ u,
                                                                 const double *__restrict__ v, const double *__restrict__ w,
                                                                 const double *__restrict__ c, double *__restrict__ Gc, PRange r)
{
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1) return;
    constexpr bool ZF = (TZ == OCN_FLAT);
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long s2 = L.s2, s3 = ZF ? 0 : L.s3, o = ocn::at(L, i, j, k);
    const double *pu = u + o, *pv = v + o, *pw = w + o, *pc = c + o;
#define C_(a, b, cc) pc[(a) + (b)*s2 + (cc)*s3]
    const double Ax = M.Ax(k), Ay = M.Ay(k), Az = M.Az;
    const double fx = (Ax * U_(1, 0, 0)) * AVG(C_(0, 0, 0), C_(1, 0, 0)) - (Ax * U_(0, 0, 0)) * AVG(C_(-1, 0, 0), C_(0, 0, 0));
    const double fy = (Ay * V_(0, 1, 0)) * AVG(C_(0, 0, 0), C_(0, 1, 0)) - (Ay * V_(0, 0, 0)) * AVG(C_(0, -1, 0), C_(0, 0, 0));
    double fzz = 0.0;
    if (!ZF) fzz = (Az * W_(0, 0, 1)) * AVG(C_(0, 0, 0), C_(0, 0, 1)) - (Az * W_(0, 0, 0)) * AVG(C_(0, 0, -1), C_(0, 0, 0));
    const double rV = 1 / (Az * M.dzC(k));
    Gc[o] = -(rV * ((fx + fy) + fzz));
}
#undef AVG

// ---------------------------------------------------------------------------------------------------
// Extra momentum terms: Coriolis, hydrostatic pressure gradient, buoyancy (no pHY′), isotropic viscous stress divergence
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double buoyancy_ccc(const TermsDev &t, long long a)
{
    switch (t.buoyancy) {
        case OCN_BUOYANCY_TRACER: return t.T[a];
        case OCN_BUOYANCY_SEAWATER_TS: return t.g * (t.alpha * t.T[a] - t.beta * t.S[a]);
        case OCN_BUOYANCY_SEAWATER_T: return t.g * t.alpha * t.T[a];
        case OCN_BUOYANCY_SEAWATER_S: return -t.g * t.beta * t.S[a];
        default: return 0.0;
    }
}

// One cell of the momentum finishing pass.  Uf/Vf/Wf/NEf(a, b, c) return u, v, w, νₑ at (i+a, j+b, k+c): global memory in the
// direct kernel, LDS planes in the tiled one -- the arithmetic is the same text, so both are bit-identical to the oracle.
// HYD (HydrostaticFreeSurfaceModel, hydrostatic_momentum_tiled below): the advective G of u, v arrives in G0u / G0v instead of
// Gu[o] / Gv[o], there is no w tendency, and res[] returns {Gu, u_out, Gv, v_out} for the column sums of the split-explicit
// free surface.
// The global values one cell of the finishing pass reads besides the LDS-staged velocities: G⁻ of the substep, pHY′ and its west / south
// neighbours, the incoming advective G, the buoyancy at the two cells around the w face.  Requested by momentum_extra_loads, consumed
// by momentum_extra_cell: the tiled kernels issue them BEFORE the plane-ahead prefetch of the next staging step -- vector-memory
// returns are in order, so a wait for these values would otherwise drain the whole prefetch group in the same iteration
// (profiles/r03b_config4.md: 52 % of the wave cycles parked).
struct ExtraLoads {
    double gm_u, gm_v, gm_w, ph_c, ph_w, ph_s, Gu_in, Gv_in, Gw_in, zb_w;
    bool w_cell;
};
// offsets of the cell (i, j, k) in the parent arrays of the Face fields where they differ from the centre fields' `o` (grids with a Bounded
// x / y: one more point along it, grid_utils.jl:66-72; the interior box of general.hip) and the plane stride of w; NULL: all of them are `o`
struct FieldOffs {
    long long u, v, w, w3;
};
template <int TZ, bool HYD = false>
__device__ __forceinline__ ExtraLoads momentum_extra_loads(const TermsDev &t, const ocn::MomentumFinal &mf, const PRange &r, int k, long long o,
                                                           long long s2, long long s3, const double *__restrict__ Gu,
                                                           const double *__restrict__ Gv, const double *__restrict__ Gw,
                                                           const FieldOffs *fo = nullptr)
{
    constexpr bool ZF = (TZ == OCN_FLAT);
    ExtraLoads ld;
    const long long ou = fo ? fo->u : o, ov = fo ? fo->v : o, ow = fo ? fo->w : o;
    const bool use_gm = mf.sc.has_zeta && (HYD || mf.sc.on);
    ld.gm_u = use_gm ? mf.sub[0].Gm[ou] : 0.0;
    ld.gm_v = use_gm ? mf.sub[1].Gm[ov] : 0.0;
    ld.w_cell = !HYD && k >= r.ow;
    ld.gm_w = (use_gm && ld.w_cell) ? mf.sub[2].Gm[ow] : 0.0;
    ld.ph_c = t.pHY ? t.pHY[o] : 0.0;
    ld.ph_w = t.pHY ? t.pHY[o - 1] : 0.0;
    ld.ph_s = t.pHY ? t.pHY[o - s2] : 0.0;
    ld.Gu_in = (HYD || mf.pre) ? 0.0 : Gu[ou];
    ld.Gv_in = (HYD || mf.pre) ? 0.0 : Gv[ov];
    ld.Gw_in = (ld.w_cell && !mf.pre) ? Gw[ow] : 0.0;
    ld.zb_w = 0.0;  // maybe_z_dot_g_bᶜᶜᶠ: only without a separate hydrostatic pressure anomaly
    if (ld.w_cell && t.buoyancy && !t.pHY) ld.zb_w = ZF ? buoyancy_ccc(t, o) : 1 * (0.5 * (buoyancy_ccc(t, o - s3) + buoyancy_ccc(t, o)));
    return ld;
}

// The 15 stress values the three components of a cell read, when the caller has evaluated every stress ONCE per face / centre and shares
// them between the cells that read them (momentum_extra_tiled): T11, T22, T33 at centres, T12 at (f,f,c), T13 at (f,c,f), T23 at (c,f,f).
struct Stresses {
    double t11e, t11w, t12n, t12c, t12e, t13t, t13c, t13e, t22n, t22s, t23t, t23c, t23n, t33t, t33b;
};

template <int TZ, bool HYD = false, class FU, class FV, class FW, class FN>
__device__ __forceinline__ void momentum_extra_cell(const GridDev &g, const TermsDev &t, const Metrics &M, int i, int j, int k,
                                                    long long o, long long s2, long long s3, bool has_nu, FU Uf, FV Vf, FW Wf,
                                                    FN NEf, double *__restrict__ Gu, double *__restrict__ Gv,
                                                    double *__restrict__ Gw, const PRange &r, const ocn::MomentumFinal &mf,
                                                    const ExtraLoads &ld, double G0u = 0.0, double G0v = 0.0, double *res = nullptr,
                                                    const Stresses *sh = nullptr, const FieldOffs *fo = nullptr)
{
    constexpr bool ZF = (TZ == OCN_FLAT);
    const long long o_u = fo ? fo->u : o, o_v = fo ? fo->v : o, o_w = fo ? fo->w : o, w3 = fo ? fo->w3 : s3;
    const double dx = M.dx, dy = M.dy, nu = t.nu;
    const double dzc = M.dzC(k), dzf = M.dzF(k), dzf1 = ZF ? dzf : M.dzF(k + 1), dzcm = ZF ? dzc : M.dzC(k - 1);
#if !OCN_STRICT
    const double rdx = fast_rcp(dx), rdy = fast_rcp(dy), rdzc = fast_rcp(dzc), rdzf = fast_rcp(dzf), rdzf1 = fast_rcp(dzf1), rdzcm = fast_rcp(dzcm);
#endif
    // strain-rate pieces (velocity_tracer_gradients.jl); δ along a Flat z is 0
#define DX(a, b) OCN_DIV((a) - (b), dx, rdx)
#define DY(a, b) OCN_DIV((a) - (b), dy, rdy)
#define DZF(a, b) (ZF ? 0.0 : OCN_DIV((a) - (b), dzf, rdzf))      /* derivative at z-face k   */
#define DZF1(a, b) (ZF ? 0.0 : OCN_DIV((a) - (b), dzf1, rdzf1))   /* derivative at z-face k+1 */
#define TAU(nuv, s) (-2 * ((nuv) * (s)))
    const double Axc = M.Ax(k), Ayc = M.Ay(k), Az = M.Az;
    // viscosity at the stress locations: the number ν, or the ccc array νₑ of an eddy-viscosity closure interpolated with
    // ℑxyᶠᶠᵃ / ℑxzᶠᵃᶠ / ℑyzᵃᶠᶠ (abstract_scalar_diffusivity_closure.jl:291-296)
    auto nuC = [&](int a, int b, int c) { return has_nu ? NEf(a, b, c) : nu; };
    auto nuFFC = [&](int a, int b, int c) {
        return has_nu ? 0.5 * (0.5 * (NEf(a - 1, b - 1, c) + NEf(a, b - 1, c)) + 0.5 * (NEf(a - 1, b, c) + NEf(a, b, c))) : nu;
    };
    auto nuFCF = [&](int a, int b, int c) {
        return has_nu ? 0.5 * (0.5 * (NEf(a - 1, b, c - 1) + NEf(a, b, c - 1)) + 0.5 * (NEf(a - 1, b, c) + NEf(a, b, c))) : nu;
    };
    auto nuCFF = [&](int a, int b, int c) {
        return has_nu ? 0.5 * (0.5 * (NEf(a, b - 1, c - 1) + NEf(a, b, c - 1)) + 0.5 * (NEf(a, b - 1, c) + NEf(a, b, c))) : nu;
    };

    const double gm_u = ld.gm_u, gm_v = ld.gm_v, gm_w = ld.gm_w, ph_c = ld.ph_c, ph_w = ld.ph_w, ph_s = ld.ph_s;
    const double Gu_in = HYD ? G0u : ld.Gu_in, Gv_in = HYD ? G0v : ld.Gv_in, Gw_in = ld.Gw_in, zb_w = ld.zb_w;

    {   // ---------------- Gu at (f,c,c)
        double G = Gu_in;
        if (t.buoyancy) G = G + 0.0;  // x_dot_g_b = 0 (NegativeZDirection)
        if (t.coriolis) {             // - x_f_cross_U,  x_f_cross_U = -f * ℑxyᶠᶜᵃ(v) / 1
            const double vi = 0.5 * (0.5 * (Vf(-1, 0, 0) + Vf(0, 0, 0)) + 0.5 * (Vf(-1, 1, 0) + Vf(0, 1, 0)));
            G = G - (-ocn::coriolis_f_at(t, g.Hy, j, 0) * vi);
        }
        if (t.pHY) G = G - DX(ph_c, ph_w);  // ∂xᶠᶜᶜ pHY′
        if (t.closure) {
            double t11e, t11w, t12n, t12s, dzF = 0.0;
            if (sh) {  // evaluated once by the owner of each face / centre: the same expressions, hence the same bits
                t11e = sh->t11e; t11w = sh->t11w; t12n = sh->t12n; t12s = sh->t12c;
                if (!ZF) dzF = Az * sh->t13t - Az * sh->t13c;
            } else {
                t11e = TAU(nuC(0, 0, 0), DX(Uf(1, 0, 0), Uf(0, 0, 0)));
                t11w = TAU(nuC(-1, 0, 0), DX(Uf(0, 0, 0), Uf(-1, 0, 0)));
                t12n = TAU(nuFFC(0, 1, 0), 0.5 * (DY(Uf(0, 1, 0), Uf(0, 0, 0)) + DX(Vf(0, 1, 0), Vf(-1, 1, 0))));
                t12s = TAU(nuFFC(0, 0, 0), 0.5 * (DY(Uf(0, 0, 0), Uf(0, -1, 0)) + DX(Vf(0, 0, 0), Vf(-1, 0, 0))));
                if (!ZF) {
                    const double t13t = TAU(nuFCF(0, 0, 1), 0.5 * (DZF1(Uf(0, 0, 1), Uf(0, 0, 0)) + DX(Wf(0, 0, 1), Wf(-1, 0, 1))));
                    const double t13b = TAU(nuFCF(0, 0, 0), 0.5 * (DZF(Uf(0, 0, 0), Uf(0, 0, -1)) + DX(Wf(0, 0, 0), Wf(-1, 0, 0))));
                    dzF = Az * t13t - Az * t13b;
                }
            }
            G = G - recip_volume(Az * dzc) * (((Axc * t11e - Axc * t11w) + (Ayc * t12n - Ayc * t12s)) + dzF);
        }
        if (TZ == OCN_BOUNDED) {
            if (k == 1 && mf.bottom[0].kind == OCN_BC_FLUX) G += ocn::bc_condition(mf.bottom[0], i, j, g.Nx, Uf(0, 0, 0)) * Az / (Az * M.dzC(1));
            if (k == g.Nz && mf.top[0].kind == OCN_BC_FLUX) G -= ocn::bc_condition(mf.top[0], i, j, g.Nx, Uf(0, 0, 0)) * Az / (Az * M.dzC(g.Nz));
        }
        Gu[o_u] = G;
        if (HYD) {
            const double gm = gm_u;
            const double un = Uf(0, 0, 0) + (mf.sc.has_zeta ? mf.sc.dt * (mf.sc.gamma * G + mf.sc.zeta * gm) : (mf.sc.dt * mf.sc.gamma) * G);
            mf.sub[0].out[o] = un;
            res[0] = mf.sc.has_zeta ? mf.sc.gamma * G - (-mf.sc.zeta) * gm : mf.sc.gamma * G;  // ab2_step_G (compute_slow_tendencies.jl:34-46)
            res[1] = un;
        } else if (mf.sc.on)
            mf.sub[0].out[o_u] = Uf(0, 0, 0) + (mf.sc.has_zeta ? mf.sc.dt * (mf.sc.gamma * G + mf.sc.zeta * gm_u) : (mf.sc.dt * mf.sc.gamma) * G);
    }
    {   // ---------------- Gv at (c,f,c)
        double G = Gv_in;
        if (t.buoyancy) G = G + 0.0;
        if (t.coriolis) {  // - y_f_cross_U,  y_f_cross_U = f * ℑxyᶜᶠᵃ(u) / 1
            const double ui = 0.5 * (0.5 * (Uf(0, -1, 0) + Uf(1, -1, 0)) + 0.5 * (Uf(0, 0, 0) + Uf(1, 0, 0)));
            G = G - ocn::coriolis_f_at(t, g.Hy, j, 1) * ui;
        }
        if (t.pHY) G = G - DY(ph_c, ph_s);
        if (t.closure) {
            double t12e, t12w, t22n, t22s, dzF = 0.0;
            if (sh) {
                t12e = sh->t12e; t12w = sh->t12c; t22n = sh->t22n; t22s = sh->t22s;
                if (!ZF) dzF = Az * sh->t23t - Az * sh->t23c;
            } else {
                t12e = TAU(nuFFC(1, 0, 0), 0.5 * (DY(Uf(1, 0, 0), Uf(1, -1, 0)) + DX(Vf(1, 0, 0), Vf(0, 0, 0))));
                t12w = TAU(nuFFC(0, 0, 0), 0.5 * (DY(Uf(0, 0, 0), Uf(0, -1, 0)) + DX(Vf(0, 0, 0), Vf(-1, 0, 0))));
                t22n = TAU(nuC(0, 0, 0), DY(Vf(0, 1, 0), Vf(0, 0, 0)));
                t22s = TAU(nuC(0, -1, 0), DY(Vf(0, 0, 0), Vf(0, -1, 0)));
                if (!ZF) {
                    const double t23t = TAU(nuCFF(0, 0, 1), 0.5 * (DZF1(Vf(0, 0, 1), Vf(0, 0, 0)) + DY(Wf(0, 0, 1), Wf(0, -1, 1))));
                    const double t23b = TAU(nuCFF(0, 0, 0), 0.5 * (DZF(Vf(0, 0, 0), Vf(0, 0, -1)) + DY(Wf(0, 0, 0), Wf(0, -1, 0))));
                    dzF = Az * t23t - Az * t23b;
                }
            }
            G = G - recip_volume(Az * dzc) * (((Axc * t12e - Axc * t12w) + (Ayc * t22n - Ayc * t22s)) + dzF);
        }
        if (TZ == OCN_BOUNDED) {
            if (k == 1 && mf.bottom[1].kind == OCN_BC_FLUX) G += ocn::bc_condition(mf.bottom[1], i, j, g.Nx, Vf(0, 0, 0)) * Az / (Az * M.dzC(1));
            if (k == g.Nz && mf.top[1].kind == OCN_BC_FLUX) G -= ocn::bc_condition(mf.top[1], i, j, g.Nx, Vf(0, 0, 0)) * Az / (Az * M.dzC(g.Nz));
        }
        Gv[o_v] = G;
        if (HYD) {
            const double gm = gm_v;
            const double vn = Vf(0, 0, 0) + (mf.sc.has_zeta ? mf.sc.dt * (mf.sc.gamma * G + mf.sc.zeta * gm) : (mf.sc.dt * mf.sc.gamma) * G);
            mf.sub[1].out[o] = vn;
            res[2] = mf.sc.has_zeta ? mf.sc.gamma * G - (-mf.sc.zeta) * gm : mf.sc.gamma * G;
            res[3] = vn;
        } else if (mf.sc.on)
            mf.sub[1].out[o_v] = Vf(0, 0, 0) + (mf.sc.has_zeta ? mf.sc.dt * (mf.sc.gamma * G + mf.sc.zeta * gm_v) : (mf.sc.dt * mf.sc.gamma) * G);
    }
    if (HYD) return;  // w is diagnostic in the hydrostatic model
    if (k >= r.ow) {  // ---------------- Gw at (c,c,f)
        double G = Gw_in;
        if (t.buoyancy) G = G + zb_w;
        if (t.coriolis) G = G - 0.0;  // z_f_cross_U = 0
        if (t.closure) {
            const double Axf = dy * dzf, Ayf = dx * dzf;
            double t13e, t13w, t23n, t23s, dzF = 0.0;
            if (sh) {
                t13e = sh->t13e; t13w = sh->t13c; t23n = sh->t23n; t23s = sh->t23c;
                if (!ZF) dzF = Az * sh->t33t - Az * sh->t33b;
            } else {
                t13e = TAU(nuFCF(1, 0, 0), 0.5 * (DZF(Uf(1, 0, 0), Uf(1, 0, -1)) + DX(Wf(1, 0, 0), Wf(0, 0, 0))));
                t13w = TAU(nuFCF(0, 0, 0), 0.5 * (DZF(Uf(0, 0, 0), Uf(0, 0, -1)) + DX(Wf(0, 0, 0), Wf(-1, 0, 0))));
                t23n = TAU(nuCFF(0, 1, 0), 0.5 * (DZF(Vf(0, 1, 0), Vf(0, 1, -1)) + DY(Wf(0, 1, 0), Wf(0, 0, 0))));
                t23s = TAU(nuCFF(0, 0, 0), 0.5 * (DZF(Vf(0, 0, 0), Vf(0, 0, -1)) + DY(Wf(0, 0, 0), Wf(0, -1, 0))));
                if (!ZF) {
                    const double t33t = TAU(nuC(0, 0, 0), OCN_DIV(Wf(0, 0, 1) - Wf(0, 0, 0), dzc, rdzc));    // Σ₃₃ at centre k
                    const double t33b = TAU(nuC(0, 0, -1), OCN_DIV(Wf(0, 0, 0) - Wf(0, 0, -1), dzcm, rdzcm));  // Σ₃₃ at centre k-1
                    dzF = Az * t33t - Az * t33b;
                }
            }
            G = G - recip_volume(Az * dzf) * (((Axf * t13e - Axf * t13w) + (Ayf * t23n - Ayf * t23s)) + dzF);
        }
        Gw[o_w] = G;
        const bool wall = (TZ == OCN_BOUNDED) && k == 1 && g.Nz > 1;  // rk3_substep! never steps the wall face
        if (mf.sc.on) mf.sub[2].out[o_w] = wall ? Wf(0, 0, 0) : Wf(0, 0, 0) + (mf.sc.has_zeta ? mf.sc.dt * (mf.sc.gamma * G + mf.sc.zeta * gm_w) : (mf.sc.dt * mf.sc.gamma) * G);
    } else if (mf.sc.on) {
        mf.sub[2].out[o_w] = Wf(0, 0, 0);  // wall face (exclude_periphery): carried over unchanged
    }
    if (mf.sc.on && TZ == OCN_BOUNDED && k == g.Nz) mf.sub[2].out[o_w + w3] = Wf(0, 0, 1);  // top wall face k = Nz+1
}

// `mf`: the flux boundary contributions of u, v (apply_flux_bcs.jl:107-160) and the NEXT stage's rk3 substep of u, v, w into a
// second storage, folded into this last pass over G (same operations as apply_flux_bcs_kernel / stepper_kernel).
template <int TZ>
__global__ __launch_bounds__(256) void momentum_extra_kernel(GridDev g, TermsDev t, const double *__restrict__ u,
                                                             const double *__restrict__ v, const double *__restrict__ w,
                                                             double *__restrict__ Gu, double *__restrict__ Gv,
                                                             double *__restrict__ Gw, PRange r, ocn::MomentumFinal mf)
{
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1) return;
    constexpr bool ZF = (TZ == OCN_FLAT);
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long s2 = L.s2, s3 = ZF ? 0 : L.s3, o = ocn::at(L, i, j, k);
    const double *pu = u + o, *pv = v + o, *pw = w + o, *pn = t.nu_e ? t.nu_e + o : nullptr;
    const ExtraLoads ld = momentum_extra_loads<TZ>(t, mf, r, k, o, s2, s3, Gu, Gv, Gw);
    momentum_extra_cell<TZ>(
        g, t, M, i, j, k, o, s2, s3, pn != nullptr, [&](int a, int b, int c) { return pu[a + b * s2 + c * s3]; },
        [&](int a, int b, int c) { return pv[a + b * s2 + c * s3]; }, [&](int a, int b, int c) { return pw[a + b * s2 + c * s3]; },
        [&](int a, int b, int c) { return pn[a + b * s2 + c * s3]; }, Gu, Gv, Gw, r, mf, ld);
}

// Tiled variant of the finishing pass: a workgroup owns a 32 x 8 patch of columns and marches KZ planes upward; planes
// k-1, k, k+1 of u, v, w (and νₑ) live in a 3-slot LDS ring with a one-cell rim, so every value enters the workgroup once per
// plane (1.33x with the rim) instead of once per stencil tap (~60 taps per cell hit L2 in the direct kernel: the 3 planes x
// 4 fields of a workgroup do not fit the 32 KB L1).  pHY′, G, G⁻ are touched once per cell and stay in global memory.
// GL ("general layouts"): the interior box of a grid with a Bounded x / y (general.hip) -- u, v, w (and their G, G⁻, stepped copies) have
// their own parent layouts; every cell of the box is a full stencil away from the walls, where the expressions are the Periodic ones.
template <int TZ, bool SH, bool GL = false>
__global__ __launch_bounds__(256, SH ? 3 : 4) void momentum_extra_tiled(GridDev g, TermsDev t, const double *__restrict__ u,
                                                            const double *__restrict__ v, const double *__restrict__ w,
                                                            double *__restrict__ Gu, double *__restrict__ Gv,
                                                            double *__restrict__ Gw, PRange r, ocn::MomentumFinal mf, int KZ)
{
    constexpr int TX = 32, TY = 8, SX = TX + 2, SY = TY + 2, PL = SX * SY;
    __shared__ double Lu[3][PL], Lv[3][PL], Lw[3][PL], Ln[3][PL];
    // Stresses shared between the cells that read them (closure != 0): every cell evaluates the six stresses it OWNS -- T11, T22, T33 at
    // its centre, T12 at its south-west edge, T13, T23 at its lower west / south edges -- once per plane instead of the 18 values its three
    // components read (each stress is read by 2 to 4 cells); T33 stays in registers (same column), T13 / T23 of plane k + 1 become plane k
    // of the next iteration.  Same expressions, same operands: bit-identical to the unshared evaluation.
    constexpr int SPL = SH ? PL : 1;
    __shared__ double S11[SPL], S22[SPL], S12[SPL], S13[2][SPL], S23[2][SPL];
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    int bx, by, bz;
    xcd_block_coords(mf.xcd, bx, by, bz);
    const int i0 = r.i0 + bx * TX, j0 = r.j0 + by * TY;
    const int kb = r.k0 + bz * KZ, ke = min(kb + KZ - 1, r.k1);
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i <= r.i1) && (j <= r.j1);
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long s2 = L.s2, s3 = L.s3;
    const Lay LFu = GL ? ocn::make_lay(g, OCN_LOC_FCC) : L, LFv = GL ? ocn::make_lay(g, OCN_LOC_CFC) : L, LFw = GL ? ocn::make_lay(g, OCN_LOC_CCF) : L;
    const bool has_nu = t.nu_e != nullptr;
    // Staging of plane kk (tile + rim, indices clamped to the first halo cell) into ring slot kk % 3 is split in two so that the
    // global loads of plane k+2 are in flight while plane k is being computed: fetch() -> registers, commit() -> LDS.
    constexpr int NS = (PL + TX * TY - 1) / (TX * TY);  // cells staged per thread (2)
    long long soff[NS], soffu[GL ? NS : 1], soffv[GL ? NS : 1], soffw[GL ? NS : 1];
    bool son[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int idx = tid + q * TX * TY;
        son[q] = idx < PL;
        const int li = son[q] ? idx % SX : 0, lj = son[q] ? idx / SX : 0;
        const int si = min(i0 - 1 + li, g.Nx + 1), sj = min(j0 - 1 + lj, g.Ny + 1);
        soff[q] = ocn::at(L, si, sj, 0);  // plane 0: add kk * s3
        if (GL) {
            soffu[q] = ocn::at(LFu, si, sj, 0);
            soffv[q] = ocn::at(LFv, si, sj, 0);
            soffw[q] = ocn::at(LFw, si, sj, 0);
        }
    }
    double fu[NS], fv[NS], fw[NS], fn[NS];
    auto fetch = [&](int kk) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const long long oo = soff[q] + (long long)kk * s3;
            fu[q] = son[q] ? u[GL ? soffu[q] + (long long)kk * LFu.s3 : oo] : 0.0;
            fv[q] = son[q] ? v[GL ? soffv[q] + (long long)kk * LFv.s3 : oo] : 0.0;
            fw[q] = son[q] ? w[GL ? soffw[q] + (long long)kk * LFw.s3 : oo] : 0.0;
            fn[q] = (son[q] && has_nu) ? t.nu_e[oo] : 0.0;
        }
    };
    auto commit = [&](int kk) {
        const int slot = kk % 3;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (!son[q]) continue;
            const int idx = tid + q * TX * TY;
            Lu[slot][idx] = fu[q];
            Lv[slot][idx] = fv[q];
            Lw[slot][idx] = fw[q];
            if (has_nu) Ln[slot][idx] = fn[q];
        }
    };
    fetch(kb - 1);
    commit(kb - 1);
    fetch(kb);
    commit(kb);
    fetch(kb + 1);
    const int c0 = (ty + 1) * SX + (tx + 1);
    // rim positions whose stresses the tile's cells read: one per lane of the first 32 / 32 / 8 / 8 lanes of waves 0 .. 3
    //   wave 0: south row (T22 of row j0 - 1), wave 1: north row (T12, T23 of row j0 + TY), wave 2: west column (T11 of column i0 - 1),
    //   wave 3: east column (T12, T13 of column i0 + TX)
    const int wv = tid >> 6, ln = tid & 63;
    const int rim = (wv == 0 && ln < TX) ? 1 : (wv == 1 && ln < TX) ? 2 : (wv == 2 && ln < TY) ? 3 : (wv == 3 && ln < TY) ? 4 : 0;
    const int cr = rim == 1 ? (ln + 1) : rim == 2 ? (TY + 1) * SX + (ln + 1) : rim == 3 ? (ln + 1) * SX : rim == 4 ? (ln + 1) * SX + (TX + 1) : c0;
    constexpr bool shared = SH;
    const double dx = M.dx, dy = M.dy, nu = t.nu;
#if !OCN_STRICT
    const double rdx = fast_rcp(dx), rdy = fast_rcp(dy);
#endif
    double t33_prev = 0.0;
    for (int k = kb; k <= ke; ++k) {
        commit(k + 1);
        __syncthreads();
        const int ia = active ? i : r.i1, ja = active ? j : r.j1;
        const long long o = ocn::at(L, ia, ja, k);
        const FieldOffs fov{GL ? ocn::at(LFu, ia, ja, k) : o, GL ? ocn::at(LFv, ia, ja, k) : o, GL ? ocn::at(LFw, ia, ja, k) : o, GL ? LFw.s3 : s3};
        const FieldOffs *fo = GL ? &fov : nullptr;
        ExtraLoads ld{};
        if (active) ld = momentum_extra_loads<TZ>(t, mf, r, k, o, s2, s3, Gu, Gv, Gw, fo);  // this plane's own values first ...
        OCN_ISSUE_LOADS_HERE();
        if (k < ke) fetch(k + 2);  // ... then the staging values of plane k + 2, consumed by the next iteration's commit
        OCN_ISSUE_LOADS_HERE();
        const int base = k + 3;  // (k + c) % 3 for c in {-1, 0, 1} without negative operands
        Stresses sh{};
        if (shared) {
            const double dzc = M.dzC(k), dzf = M.dzF(k), dzf1 = M.dzF(k + 1), dzcm = M.dzC(k - 1);
#if !OCN_STRICT
            const double rdzc = fast_rcp(dzc), rdzf = fast_rcp(dzf), rdzf1 = fast_rcp(dzf1), rdzcm = fast_rcp(dzcm);
#endif
            auto sU = [&](int c, int a, int b, int d) { return Lu[(base + d) % 3][c + a + b * SX]; };
            auto sV = [&](int c, int a, int b, int d) { return Lv[(base + d) % 3][c + a + b * SX]; };
            auto sW = [&](int c, int a, int b, int d) { return Lw[(base + d) % 3][c + a + b * SX]; };
            auto sN = [&](int c, int a, int b, int d) { return Ln[(base + d) % 3][c + a + b * SX]; };
            auto nuC = [&](int c, int d) { return has_nu ? sN(c, 0, 0, d) : nu; };
            auto nuFFC = [&](int c) {
                return has_nu ? 0.5 * (0.5 * (sN(c, -1, -1, 0) + sN(c, 0, -1, 0)) + 0.5 * (sN(c, -1, 0, 0) + sN(c, 0, 0, 0))) : nu;
            };
            auto nuFCF = [&](int c, int d) {
                return has_nu ? 0.5 * (0.5 * (sN(c, -1, 0, d - 1) + sN(c, 0, 0, d - 1)) + 0.5 * (sN(c, -1, 0, d) + sN(c, 0, 0, d))) : nu;
            };
            auto nuCFF = [&](int c, int d) {
                return has_nu ? 0.5 * (0.5 * (sN(c, 0, -1, d - 1) + sN(c, 0, 0, d - 1)) + 0.5 * (sN(c, 0, -1, d) + sN(c, 0, 0, d))) : nu;
            };
            // the expressions of momentum_extra_cell, with the centre of evaluation as an argument (d: z-face k + d)
            auto T11 = [&](int c) { return TAU(nuC(c, 0), DX(sU(c, 1, 0, 0), sU(c, 0, 0, 0))); };
            auto T22 = [&](int c) { return TAU(nuC(c, 0), DY(sV(c, 0, 1, 0), sV(c, 0, 0, 0))); };
            auto T12 = [&](int c) { return TAU(nuFFC(c), 0.5 * (DY(sU(c, 0, 0, 0), sU(c, 0, -1, 0)) + DX(sV(c, 0, 0, 0), sV(c, -1, 0, 0)))); };
            auto T13 = [&](int c, int d) {
                const double dzu = d ? OCN_DIV(sU(c, 0, 0, 1) - sU(c, 0, 0, 0), dzf1, rdzf1) : OCN_DIV(sU(c, 0, 0, 0) - sU(c, 0, 0, -1), dzf, rdzf);
                return TAU(nuFCF(c, d), 0.5 * (dzu + DX(sW(c, 0, 0, d), sW(c, -1, 0, d))));
            };
            auto T23 = [&](int c, int d) {
                const double dzv = d ? OCN_DIV(sV(c, 0, 0, 1) - sV(c, 0, 0, 0), dzf1, rdzf1) : OCN_DIV(sV(c, 0, 0, 0) - sV(c, 0, 0, -1), dzf, rdzf);
                return TAU(nuCFF(c, d), 0.5 * (dzv + DY(sW(c, 0, 0, d), sW(c, 0, -1, d))));
            };
            const int cur = k & 1, nxt = cur ^ 1;
            const bool first = (k == kb);
            // own position (every thread, active or not: the neighbours of the last active column / row read these)
            const double o11 = T11(c0), o22 = T22(c0), o12 = T12(c0), o13n = T13(c0, 1), o23n = T23(c0, 1);
            double o13c, o23c;
            if (first) {
                o13c = T13(c0, 0);
                o23c = T23(c0, 0);
                S13[cur][c0] = o13c;
                S23[cur][c0] = o23c;
                t33_prev = TAU(nuC(c0, -1), OCN_DIV(sW(c0, 0, 0, 0) - sW(c0, 0, 0, -1), dzcm, rdzcm));
            } else {
                o13c = S13[cur][c0];  // (written by this thread in the previous iteration)
                o23c = S23[cur][c0];
            }
            S11[c0] = o11; S22[c0] = o22; S12[c0] = o12; S13[nxt][c0] = o13n; S23[nxt][c0] = o23n;
            const double o33 = TAU(nuC(c0, 0), OCN_DIV(sW(c0, 0, 0, 1) - sW(c0, 0, 0, 0), dzc, rdzc));
            // rim positions
            if (rim == 1) {
                S22[cr] = T22(cr);
            } else if (rim == 2) {
                S12[cr] = T12(cr);
                S23[nxt][cr] = T23(cr, 1);
                if (first) S23[cur][cr] = T23(cr, 0);
            } else if (rim == 3) {
                S11[cr] = T11(cr);
            } else if (rim == 4) {
                S12[cr] = T12(cr);
                S13[nxt][cr] = T13(cr, 1);
                if (first) S13[cur][cr] = T13(cr, 0);
            }
            __syncthreads();
            sh.t11e = o11; sh.t11w = S11[c0 - 1];
            sh.t12c = o12; sh.t12n = S12[c0 + SX]; sh.t12e = S12[c0 + 1];
            sh.t13t = o13n; sh.t13c = o13c; sh.t13e = S13[cur][c0 + 1];
            sh.t22n = o22; sh.t22s = S22[c0 - SX];
            sh.t23t = o23n; sh.t23c = o23c; sh.t23n = S23[cur][c0 + SX];
            sh.t33t = o33; sh.t33b = t33_prev;
            t33_prev = o33;
        }
        if (active) {
            momentum_extra_cell<TZ>(
                g, t, M, i, j, k, o, s2, s3, has_nu, [&](int a, int b, int c) { return Lu[(base + c) % 3][c0 + a + b * SX]; },
                [&](int a, int b, int c) { return Lv[(base + c) % 3][c0 + a + b * SX]; },
                [&](int a, int b, int c) { return Lw[(base + c) % 3][c0 + a + b * SX]; },
                [&](int a, int b, int c) { return Ln[(base + c) % 3][c0 + a + b * SX]; }, Gu, Gv, Gw, r, mf, ld, 0.0, 0.0, nullptr,
                SH ? &sh : nullptr, fo);
        }
        // Unshared: everyone must be done with slot (k - 1) % 3 before the next iteration's commit overwrites it.  Shared: nothing reads plane
        // k - 1 after the first iteration's stress phase (T13, T23 of plane k and T33 of k - 1 are carried), which the barrier above already
        // closed; the plane-k stress arrays are next written after the NEXT iteration's first barrier, which every wave reaches only after its
        // cell phase -- so two barriers per plane suffice.
        if (!SH) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// HydrostaticFreeSurfaceModel: the whole horizontal-momentum part of one QuasiAdamsBashforth2 step in ONE pass over the columns.
//
//   compute_hydrostatic_free_surface_Gu!/Gv!   (hydrostatic_free_surface_tendency_kernel_functions.jl:29-97):
//       G = - U_dot_∇u [VectorInvariant(): vector_invariant_advection.jl:269-275, 304-319, 360-361] - g ∂x η (ExplicitFreeSurface
//           only) - f x U - ∂x pHY′ - ∂ⱼτ₁ⱼ, + the flux boundary contributions (apply_flux_bcs.jl:107-160)
//   ab2_step_velocities!                       (hydrostatic_free_surface_ab2_step.jl:41-63; TimeSteppers ab2_step_field!):
//       u* = u + Δt ((3/2 + χ) Gⁿ - (1/2 + χ) G⁻ not_euler)            -> second storage (the neighbours still read u)
//   compute_split_explicit_forcing!            (compute_slow_tendencies.jl:12-46):   Gᵁ = Σₖ Δz ((3/2 + χ) Gⁿ - (1/2 + χ) G⁻ not_euler)
//   compute_barotropic_mode! of u*             (barotropic_split_explicit_corrector.jl:13-32), which the barotropic corrector needs
//       after the substepping: U̅* = Σₖ Δz u* σ, σ = 1 on a static grid
//
// A workgroup owns a 32 x 8 patch of columns and marches k = 1 .. Nz; planes k-1, k, k+1 of u, v, w live in a 3-slot LDS ring with a
// one-cell rim (every stencil of the vector-invariant form and of the stress divergence fits), pHY′, G⁻ are touched once per cell.
// The column sums ride along in registers in the reference's order (k ascending, first term assigned), so the strict build is
// bit-identical to the unfused kernels (vector_invariant_kernel, momentum_extra_kernel, stepper_kernel<2>,
// barotropic_forcing_kernel, barotropic_mode_kernel).  HBM per cell: read u, v, w, pHY′, G⁻u, G⁻v; write Gu, Gv, u*, v* = 80 B
// (the five launches it replaces: 264 B).
// ---------------------------------------------------------------------------------------------------
template <class FU, class FV, class FW>
__device__ __forceinline__ void vector_invariant_cell(const Metrics &M, int k, FU uu, FV vv, FW ww, double &Gu0, double &Gv0)
{
    const double dx = M.dx, dy = M.dy, Az = dx * dy;
    const double dzf0 = M.dzF(k), dzf1 = M.dzF(k + 1);  // Δzᶠ at faces k, k+1
    auto zeta = [&](int a, int b) {  // ζ₃ᶠᶠᶜ at (i + a, j + b)  (Operators/vorticity_operators.jl:4-11)
        const double gam = (dy * vv(a, b, 0) - dy * vv(a - 1, b, 0)) - (dx * uu(a, b, 0) - dx * uu(a, b - 1, 0));
        return gam / Az;
    };
    auto Kh = [&](int a, int b) {    // Khᶜᶜᶜ at (i + a, j + b)
        return (0.5 * (uu(a, b, 0) * uu(a, b, 0) + uu(a + 1, b, 0) * uu(a + 1, b, 0)) +
                0.5 * (vv(a, b, 0) * vv(a, b, 0) + vv(a, b + 1, 0) * vv(a, b + 1, 0))) / 2;
    };
    {
        auto m = [&](int a) { return 0.5 * (dx * vv(a, 0, 0) + dx * vv(a, 1, 0)); };  // ℑyᵃᶜᵃ(Δx_qᶜᶠᶜ v) at (i + a, j)
        const double hadv = -(0.5 * (zeta(0, 0) + zeta(0, 1))) * (0.5 * (m(-1) + m(0))) / dx;
        auto Z = [&](int c, double dzf) { return (0.5 * (Az * ww(-1, 0, c) + Az * ww(0, 0, c))) * ((uu(0, 0, c) - uu(0, 0, c - 1)) / dzf); };
        const double vadv = (0.5 * (Z(0, dzf0) + Z(1, dzf1))) / Az;
        const double bern = (Kh(0, 0) - Kh(-1, 0)) / dx;
        Gu0 = -((hadv + vadv) + bern);
    }
    {
        auto n = [&](int b) { return 0.5 * (dy * uu(0, b, 0) + dy * uu(1, b, 0)); };  // ℑxᶜᵃᵃ(Δy_qᶠᶜᶜ u) at (i, j + b)
        const double hadv = (0.5 * (zeta(0, 0) + zeta(1, 0))) * (0.5 * (n(-1) + n(0))) / dy;
        auto Z = [&](int c, double dzf) { return (0.5 * (Az * ww(0, -1, c) + Az * ww(0, 0, c))) * ((vv(0, 0, c) - vv(0, 0, c - 1)) / dzf); };
        const double vadv = (0.5 * (Z(0, dzf0) + Z(1, dzf1))) / Az;
        const double bern = (Kh(0, 0) - Kh(0, -1)) / dy;
        Gv0 = -((hadv + vadv) + bern);
    }
}

template <int W>
__global__ __launch_bounds__(256, W) void hydrostatic_momentum_tiled(GridDev g, TermsDev t, const double *__restrict__ u,
                                                                  const double *__restrict__ v, const double *__restrict__ w,
                                                                  double *__restrict__ Gu, double *__restrict__ Gv,
                                                                  ocn::MomentumFinal mf, ocn::HydroFuse hf)
{
    constexpr int TZ = OCN_BOUNDED;
    constexpr int TX = 32, TY = 8, SX = TX + 2, SY = TY + 2, PL = SX * SY;
    __shared__ double Lu[3][PL], Lv[3][PL], Lw[3][PL];
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    int bx, by, bz;
    xcd_block_coords(mf.xcd, bx, by, bz);
    const int i0 = 1 + bx * TX, j0 = 1 + by * TY;
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i <= g.Nx) && (j <= g.Ny);
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long s2 = L.s2, s3 = L.s3;
    PRange r{1, g.Nx, 1, g.Ny, 1, g.Nz, 1};
    constexpr int NS = (PL + TX * TY - 1) / (TX * TY);  // cells staged per thread (2)
    long long soff[NS];
    bool son[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int idx = tid + q * TX * TY;
        son[q] = idx < PL;
        const int li = son[q] ? idx % SX : 0, lj = son[q] ? idx / SX : 0;
        soff[q] = ocn::at(L, min(i0 - 1 + li, g.Nx + 1), min(j0 - 1 + lj, g.Ny + 1), 0);  // plane 0: add kk * s3
    }
    double fu[NS], fv[NS], fw[NS];
    auto fetch = [&](int kk) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const long long oo = soff[q] + (long long)kk * s3;
            fu[q] = son[q] ? u[oo] : 0.0;
            fv[q] = son[q] ? v[oo] : 0.0;
            fw[q] = son[q] ? w[oo] : 0.0;
        }
    };
    auto commit = [&](int kk) {
        const int slot = (kk + 3) % 3;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (!son[q]) continue;
            const int idx = tid + q * TX * TY;
            Lu[slot][idx] = fu[q];
            Lv[slot][idx] = fv[q];
            Lw[slot][idx] = fw[q];
        }
    };
    fetch(0);
    commit(0);
    fetch(1);
    commit(1);
    fetch(2);
    const int c0 = (ty + 1) * SX + (tx + 1);
    const long long e = (i - 1 + g.Hx) + (long long)L.sx * (j - 1 + g.Hy);  // this column in the (sx, sy) planes of η, Gᵁ, U̅
    double gxe = 0.0, gye = 0.0;
    if (active && hf.eta) {  // explicit_barotropic_pressure_x/y_gradient (explicit_free_surface.jl:36-40), the same at every k
        gxe = hf.grav * ((hf.eta[e] - hf.eta[e - 1]) / g.dx);
        gye = hf.grav * ((hf.eta[e] - hf.eta[e - L.sx]) / g.dy);
    }
    double aGU = 0.0, aGV = 0.0, aU = 0.0, aV = 0.0;
    for (int k = 1; k <= g.Nz; ++k) {
        commit(k + 1);
        __syncthreads();
        const long long o = ocn::at(L, active ? i : g.Nx, active ? j : g.Ny, k);
        ExtraLoads ld{};
        if (active) ld = momentum_extra_loads<TZ, true>(t, mf, r, k, o, s2, s3, Gu, Gv, nullptr);  // this plane's own values first ...
        OCN_ISSUE_LOADS_HERE();
        if (k < g.Nz) fetch(k + 2);  // ... then the staging values of plane k + 2, consumed by the next iteration's commit
        OCN_ISSUE_LOADS_HERE();
        if (active) {
            const int base = k + 3;  // (k + c) % 3 for c in {-1, 0, 1} without negative operands
            auto Uf = [&](int a, int b, int c) { return Lu[(base + c) % 3][c0 + a + b * SX]; };
            auto Vf = [&](int a, int b, int c) { return Lv[(base + c) % 3][c0 + a + b * SX]; };
            auto Wf = [&](int a, int b, int c) { return Lw[(base + c) % 3][c0 + a + b * SX]; };
            double G0u, G0v, res[4];
            vector_invariant_cell(M, k, Uf, Vf, Wf, G0u, G0v);
            if (hf.eta) {
                G0u -= gxe;
                G0v -= gye;
            }
            momentum_extra_cell<TZ, true>(g, t, M, i, j, k, o, s2, s3, false, Uf, Vf, Wf, [&](int, int, int) { return 0.0; }, Gu, Gv,
                                          nullptr, r, mf, ld, G0u, G0v, res);
            const double dz = M.dzC(k);
            if (k == 1) {
                aGU = dz * res[0];
                aGV = dz * res[2];
                aU = dz * res[1] * 1.0;
                aV = dz * res[3] * 1.0;
            } else {
                aGU = aGU + dz * res[0];
                aGV = aGV + dz * res[2];
                aU = aU + dz * res[1] * 1.0;
                aV = aV + dz * res[3] * 1.0;
            }
        }
        __syncthreads();  // everyone is done with slot (k - 1) % 3 before the next iteration overwrites it
    }
    if (active && hf.GU) {
        hf.GU[e] = aGU;
        hf.GV[e] = aGV;
        hf.Ub[e] = aU;
        hf.Vb[e] = aV;
    }
}

// Gc <- Gc - ∇_dot_qᶜ,  q = -(κ ∂c)  (closure_kernel_operators.jl:48-53)
template <int TZ>
__global__ __launch_bounds__(256) void tracer_diffusion_kernel(GridDev g, double kappa, const double *__restrict__ kappa_e,
                                                               const double *__restrict__ c, double *__restrict__ Gc, PRange r)
{
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1) return;
    constexpr bool ZF = (TZ == OCN_FLAT);
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long s2 = L.s2, s3 = ZF ? 0 : L.s3, o = ocn::at(L, i, j, k);
    const double *pc = c + o;
    const double dx = M.dx, dy = M.dy, dzc = M.dzC(k), dzf = M.dzF(k), dzf1 = ZF ? dzf : M.dzF(k + 1);
#if !OCN_STRICT
    const double rdx = 1 / dx, rdy = 1 / dy, rdzf = 1 / dzf, rdzf1 = 1 / dzf1;
#endif
    const double Ax = M.Ax(k), Ay = M.Ay(k), Az = M.Az;
    const double c0 = C_(0, 0, 0);
    // κ at the flux faces: the number κ, or κₑ interpolated with ℑxᶠᵃᵃ / ℑyᵃᶠᵃ / ℑzᵃᵃᶠ (abstract_scalar_diffusivity_closure.jl:298-300)
    const double *pk = kappa_e ? kappa_e + o : nullptr;
#define KE(a, b, cc) pk[(a) + (b)*s2 + (cc)*s3]
    const double k0 = pk ? KE(0, 0, 0) : kappa;
    const double kxe = pk ? 0.5 * (k0 + KE(1, 0, 0)) : kappa, kxw = pk ? 0.5 * (KE(-1, 0, 0) + k0) : kappa;
    const double kyn = pk ? 0.5 * (k0 + KE(0, 1, 0)) : kappa, kys = pk ? 0.5 * (KE(0, -1, 0) + k0) : kappa;
    const double qxe = -(kxe * DX(C_(1, 0, 0), c0)), qxw = -(kxw * DX(c0, C_(-1, 0, 0)));
    const double qyn = -(kyn * DY(C_(0, 1, 0), c0)), qys = -(kys * DY(c0, C_(0, -1, 0)));
    double dzF = 0.0;
    if (!ZF) {
        const double kzt = pk ? 0.5 * (k0 + KE(0, 0, 1)) : kappa, kzb = pk ? 0.5 * (KE(0, 0, -1) + k0) : kappa;
        const double qzt = -(kzt * DZF1(C_(0, 0, 1), c0)), qzb = -(kzb * DZF(c0, C_(0, 0, -1)));
        dzF = Az * qzt - Az * qzb;
    }
#undef KE
    Gc[o] = Gc[o] - 1 / (Az * dzc) * (((Ax * qxe - Ax * qxw) + (Ay * qyn - Ay * qys)) + dzF);
}
#undef U_
#undef V_
#undef W_
#undef C_
#undef DX
#undef DY
#undef DZF
#undef DZF1
#undef TAU
#undef NE

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
#define OCN_LAUNCH_TZ(KERNEL, ...)                                                                              \
    do {                                                                                                        \
        switch (grid->tz) {                                                                                     \
            case OCN_PERIODIC: hipLaunchKernelGGL(KERNEL<OCN_PERIODIC>, nb, block, 0, stream, __VA_ARGS__); break; \
            case OCN_BOUNDED: hipLaunchKernelGGL(KERNEL<OCN_BOUNDED>, nb, block, 0, stream, __VA_ARGS__); break;   \
            case OCN_FLAT: hipLaunchKernelGGL(KERNEL<OCN_FLAT>, nb, block, 0, stream, __VA_ARGS__); break;         \
            default: ocn::set_error("unsupported z topology %d", grid->tz); return OCN_ERR_UNSUPPORTED;         \
        }                                                                                                       \
        OCN_CHECK_HIP(hipGetLastError());                                                                       \
    } while (0)

int launch_momentum_centered2(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                              double *Gw, const int32_t *range, hipStream_t stream)
{
    PRange r;
    int st = make_prange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const dim3 block = ocn::range_block(r.i1 - r.i0 + 1), nb = ocn::range_grid(block, r.i1 - r.i0 + 1, r.j1 - r.j0 + 1, r.k1 - r.k0 + 1);
    OCN_LAUNCH_TZ(momentum_tendencies_centered2, g, u, v, w, Gu, Gv, Gw, r);
    return OCN_SUCCESS;
}

int launch_tracer_centered2(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                            double *Gc, const int32_t *range, hipStream_t stream)
{
    PRange r;
    int st = make_prange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const dim3 block = ocn::range_block(r.i1 - r.i0 + 1), nb = ocn::range_grid(block, r.i1 - r.i0 + 1, r.j1 - r.j0 + 1, r.k1 - r.k0 + 1);
    OCN_LAUNCH_TZ(tracer_tendency_centered2, g, u, v, w, c, Gc, r);
    return OCN_SUCCESS;
}

int launch_momentum_extra(const ocn_grid *grid, const TermsDev &t, const double *u, const double *v, const double *w,
                          double *Gu, double *Gv, double *Gw, const int32_t *range, hipStream_t stream,
                          const ocn::MomentumFinal *fin)
{
    ocn::MomentumFinal mf{};
    if (fin) mf = *fin;
    mf.xcd = xcd_remap_on();
    static const bool share_env = !(getenv("OCN_SHARE_STRESSES") && getenv("OCN_SHARE_STRESSES")[0] == '0');
    const bool share = share_env && t.closure != 0;  // every stress evaluated once per face / centre and shared through LDS
    PRange r;
    int st = make_prange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = r.k1 - r.k0 + 1;
    static const int force_direct = (getenv("OCN_EXTRA_KERNEL") && !strcmp(getenv("OCN_EXTRA_KERNEL"), "direct"));
    if (!force_direct && grid->tz != OCN_FLAT && wx >= 16 && wy >= 8 && wz >= 4 && grid->Hz >= 1) {
        const int tiles = ((wx + 31) / 32) * ((wy + 7) / 8);
        int KZ = wz;  // z-chunk: enough workgroups to fill the chip, long enough to amortise the two-plane prologue
        while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < 4096) KZ = (KZ + 1) / 2;
        dim3 nbt((wx + 31) / 32, (wy + 7) / 8, (wz + KZ - 1) / KZ);
        if (grid->tz == OCN_PERIODIC) {
            if (share)
                hipLaunchKernelGGL((momentum_extra_tiled<OCN_PERIODIC, true>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
            else
                hipLaunchKernelGGL((momentum_extra_tiled<OCN_PERIODIC, false>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
        } else {
            if (share)
                hipLaunchKernelGGL((momentum_extra_tiled<OCN_BOUNDED, true>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
            else
                hipLaunchKernelGGL((momentum_extra_tiled<OCN_BOUNDED, false>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
        }
        OCN_CHECK_HIP(hipGetLastError());
        return OCN_SUCCESS;
    }
    const dim3 block = ocn::range_block(r.i1 - r.i0 + 1), nb = ocn::range_grid(block, r.i1 - r.i0 + 1, r.j1 - r.j0 + 1, r.k1 - r.k0 + 1);
    OCN_LAUNCH_TZ(momentum_extra_kernel, g, t, u, v, w, Gu, Gv, Gw, r, mf);
    return OCN_SUCCESS;
}

// The finishing pass on the INTERIOR BOX {i0, i1, j0, j1} of a grid with walls in x / y (general.hip: every cell at least a full stencil
// away from the walls, where the per-cell kernel with its run-time topology evaluates exactly these expressions): the tiled kernel with
// per-field parent layouts.  *launched = 0 when the box is too small for the tiles (the caller then runs the per-cell kernel everywhere).
int launch_momentum_extra_box(const ocn_grid *grid, const TermsDev &t, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                              double *Gw, const int32_t box[4], int *launched, hipStream_t stream, const ocn::MomentumFinal *fin, int ranged)
{
    *launched = 0;
    ocn::MomentumFinal mf{};
    if (fin) mf = *fin;
    mf.xcd = xcd_remap_on();
    static const bool share_env = !(getenv("OCN_SHARE_STRESSES") && getenv("OCN_SHARE_STRESSES")[0] == '0');
    const bool share = share_env && t.closure != 0;
    PRange r;
    r.i0 = box[0]; r.i1 = box[1]; r.j0 = box[2]; r.j1 = box[3]; r.k0 = 1; r.k1 = grid->Nz;
    r.ow = (!ranged && grid->tz == OCN_BOUNDED && grid->Nz > 1) ? 2 : 1;  // (KernelParameters: periphery not excluded, as make_prange)
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = grid->Nz;
    if (grid->tz == OCN_FLAT || wx < 16 || wy < 8 || wz < 4 || grid->Hz < 1) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const int tiles = ((wx + 31) / 32) * ((wy + 7) / 8);
    int KZ = wz;
    while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < 4096) KZ = (KZ + 1) / 2;
    dim3 nbt((wx + 31) / 32, (wy + 7) / 8, (wz + KZ - 1) / KZ);
    if (grid->tz == OCN_PERIODIC) {
        if (share)
            hipLaunchKernelGGL((momentum_extra_tiled<OCN_PERIODIC, true, true>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
        else
            hipLaunchKernelGGL((momentum_extra_tiled<OCN_PERIODIC, false, true>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
    } else {
        if (share)
            hipLaunchKernelGGL((momentum_extra_tiled<OCN_BOUNDED, true, true>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
        else
            hipLaunchKernelGGL((momentum_extra_tiled<OCN_BOUNDED, false, true>), nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, Gw, r, mf, KZ);
    }
    OCN_CHECK_HIP(hipGetLastError());
    *launched = 1;
    return OCN_SUCCESS;
}

int launch_hydrostatic_momentum(const ocn_grid *grid, const TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                double *Gv, const ocn::MomentumFinal &mf, const ocn::HydroFuse &hf, hipStream_t stream)
{
    GridDev g = ocn::to_dev(*grid);
    dim3 nbt((g.Nx + 31) / 32, (g.Ny + 7) / 8, 1);
    ocn::MomentumFinal mfx = mf;
    mfx.xcd = xcd_remap_on();
    static const int waves = getenv("OCN_HYDRO_WAVES") ? atoi(getenv("OCN_HYDRO_WAVES")) : 3;  // min waves / SIMD the build targets
    if (waves >= 4)
        hipLaunchKernelGGL(hydrostatic_momentum_tiled<4>, nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, mfx, hf);
    else
        hipLaunchKernelGGL(hydrostatic_momentum_tiled<1>, nbt, dim3(256), 0, stream, g, t, u, v, w, Gu, Gv, mfx, hf);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_tracer_diffusion(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                            const int32_t *range, hipStream_t stream)
{
    PRange r;
    int st = make_prange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const dim3 block = ocn::range_block(r.i1 - r.i0 + 1), nb = ocn::range_grid(block, r.i1 - r.i0 + 1, r.j1 - r.j0 + 1, r.k1 - r.k0 + 1);
    OCN_LAUNCH_TZ(tracer_diffusion_kernel, g, kappa, kappa_e, c, Gc, r);
    return OCN_SUCCESS;
}

}  // namespace OCN_NS
