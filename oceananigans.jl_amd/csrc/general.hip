// general.hip -- the tendency kernels on grids with a Bounded or Flat x / y direction.
//
// The production kernels (tendencies.hip, physics.hip) assume Periodic x, y: every staggered location then shares ONE parent layout and
// no x / y stencil ever meets a wall.  On (Periodic, Bounded, Bounded), (Bounded, Bounded, Bounded), (Periodic, Flat, Bounded) ... grids
//   * Face-located fields have N + 1 + 2H points along a Bounded direction (grid_utils.jl:66-72): u, v, w, c have different strides;
//   * reconstructions lose order near the walls (topologically_conditional_interpolation.jl:37-128) in x and y too;
//   * interpolation along a Flat direction is the identity and differences / fluxes through it vanish
//     (flat_advective_fluxes.jl:8-44, difference_operators.jl:33-49, interpolation_operators.jl:103-110);
//   * Face-located fields skip their first index in a Bounded direction (launch!(...; exclude_periphery = true),
//     kernel_launching.jl:113-161);
//   * the Coriolis term averages over ACTIVE nodes only (active_weighted_ℑxy, interpolation_operators.jl:121-131).
// These kernels are direction-generic, one thread per cell, topologies read at run time (they are uniform over a launch): correctness and
// coverage first (every reference example on a closed box or an x-z slice runs), not the roofline -- the BASELINE configurations are
// all (Periodic, Periodic, .) and keep the tiled kernels.  Strict build: the operand order of the reference's kernel functions
// (nonhydrostatic_tendency_kernel_functions.jl:47-259, momentum_advection_operators.jl:46-83, closure_kernel_operators.jl:27-53),
// bit-identical to the CPU oracle.
//
// Compiled twice like tendencies.hip (namespaces via ocn_weno.h): strict / fast, WENO5 or UpwindBiased(order = 5).
#include <algorithm>
#include <cstring>

#include "ocn_weno.h"

namespace OCN_NS {

using ocn::GridDev;
using ocn::Lay;

namespace gen {

// topology of one direction as the kernels see it: Periodic / Bounded / Flat, and for Bounded which sides have a wall -- both, or only
// one on the first / last slab of a grid whose partitioned x is Bounded (RightConnected / LeftConnected: the "half Bounded" topologies
// of topologically_conditional_interpolation.jl:55-65, whose order reduction applies on the walled side only)
struct Topo {
    int t;
    bool wl, wh;  // wall at the low-index / high-index end
};

// topologically conditional interpolation with the topology as a run-time value (ocn_weno.h: sym_interp / bias_interp)
template <bool CENTER, class V>
__device__ __forceinline__ double sym_rt(Topo tp, V val, int idx, int N)
{
    if (tp.t == OCN_FLAT) return val(CENTER ? -1 : 0);
    if (tp.t == OCN_BOUNDED) {
        const bool hi = (!tp.wl || idx >= (CENTER ? 3 : 4)) && (!tp.wh || idx <= N + 1 - 3);
        if (!hi) return 0.5 * val(-1) + 0.5 * val(0);
    }
    return centered4(val(-2), val(-1), val(0), val(1));
}

template <bool CENTER, class V>
__device__ __forceinline__ double bias_rt(Topo tp, V val, int idx, int N, bool left)
{
    if (tp.t == OCN_FLAT) return val(CENTER ? -1 : 0);
    if (tp.t == OCN_BOUNDED) {
        const bool ok5 = (!tp.wl || idx >= (CENTER ? 3 : 4)) && (!tp.wh || idx <= N + 1 - 3);
        const bool ok3 = (!tp.wl || idx >= (CENTER ? 2 : 3)) && (!tp.wh || idx <= N + 1 - 2);
        if (!ok5) {
            if (ok3) return weno3(val(-2), val(-1), val(0), val(1), left);
            return left ? val(-1) : val(0);
        }
    }
    return weno5(val(-3), val(-2), val(-1), val(0), val(1), val(2), left);
}

// Centered(order = 2): no topology conditions (a "low order" scheme); the value itself along a Flat direction
template <bool CENTER, class V>
__device__ __forceinline__ double c2_rt(int topo, V val)
{
    if (topo == OCN_FLAT) return val(CENTER ? -1 : 0);
    return 0.5 * val(-1) + 0.5 * val(0);
}

struct Fields {
    GridDev g;
    const double *u, *v, *w;
    Lay Lu, Lv, Lw, Lc;
    int centered2;  // advection = Centered(order = 2) instead of the build's upwind scheme
};

__device__ __forceinline__ Topo topo_of(const GridDev &g, int d)
{
    if (d == 0) return Topo{g.tx, g.xw != 0, g.xe != 0};
    const int t = d == 1 ? g.ty : g.tz;
    return Topo{t, t == OCN_BOUNDED, t == OCN_BOUNDED};
}
__device__ __forceinline__ int size_of(const GridDev &g, int d) { return d == 0 ? g.Nx : d == 1 ? g.Ny : g.Nz; }
__device__ __forceinline__ long long stride_of(const Lay &L, int d) { return d == 0 ? 1 : d == 1 ? L.s2 : L.s3; }

// One momentum flux  U~ * psi^R  (upwind_biased_advective_fluxes.jl:23-93; centered_advective_fluxes.jl:7-17):
//   advecting component CA interpolated along DA (to the centre when ACEN), advected component CB interpolated along DB.
template <int CA, int DA, bool ACEN, int CB, int DB, bool BCEN>
__device__ __forceinline__ double mom_flux(const Fields &F, const Metrics &M, int i, int j, int k)
{
    const GridDev &g = F.g;
    if (topo_of(g, CA).t == OCN_FLAT) return 0.0;  // the flux THROUGH a Flat direction is zero (flat_advective_fluxes.jl:8-22)
    const double *fa = CA == 0 ? F.u : CA == 1 ? F.v : F.w;
    const double *fb = CB == 0 ? F.u : CB == 1 ? F.v : F.w;
    const Lay &La = CA == 0 ? F.Lu : CA == 1 ? F.Lv : F.Lw;
    const Lay &Lb = CB == 0 ? F.Lu : CB == 1 ? F.Lv : F.Lw;
    const int ijk[3] = {i, j, k};
    int qa[3] = {i, j, k}, qb[3] = {i, j, k};
    if (ACEN) qa[DA] += 1;  // symmetric_interpolate_*ᶜ: the line shifted to face idx + 1
    if (BCEN) qb[DB] += 1;
    const double *pa = fa + ocn::at(La, qa[0], qa[1], qa[2]);
    const double *pb = fb + ocn::at(Lb, qb[0], qb[1], qb[2]);
    const long long sa = stride_of(La, DA), sb = stride_of(Lb, DB);
    const int ka = qa[2];
    if (F.centered2) {
        // A(flux location) * sym(U) * sym(u), left-associated; the area is NOT inside the interpolation
        const double ua = c2_rt<ACEN>(topo_of(g, DA).t, [&](int m) { return pa[m * sa]; });
        const double ub = c2_rt<BCEN>(topo_of(g, DB).t, [&](int m) { return pb[m * sb]; });
        const bool zf = (CB == 2 && CA != 2);  // z-location of the flux: Face for Uw, Vw
        const double dzk = zf ? M.dzF(k) : M.dzC(k);
        const double area = CA == 0 ? M.dy * dzk : CA == 1 ? M.dx * dzk : M.Az;
        return (area * ua) * ub;
    }
    double ut;
    if (CA == 2) {
        const double a = M.Az;
        ut = sym_rt<ACEN>(topo_of(g, DA), [&](int m) { return a * pa[m * sa]; }, ijk[DA], size_of(g, DA));
    } else if (DA == 2) {  // the line runs along z: the area changes along it
        ut = sym_rt<ACEN>(topo_of(g, DA), [&](int m) { return (CA == 0 ? M.Ax(ka + m) : M.Ay(ka + m)) * pa[m * sa]; }, ijk[DA], size_of(g, DA));
    } else {
        const double a = CA == 0 ? M.Ax(ka) : M.Ay(ka);
#if OCN_STRICT
        ut = sym_rt<ACEN>(topo_of(g, DA), [&](int m) { return a * pa[m * sa]; }, ijk[DA], size_of(g, DA));
#else
        ut = a * sym_rt<ACEN>(topo_of(g, DA), [&](int m) { return pa[m * sa]; }, ijk[DA], size_of(g, DA));
#endif
    }
    const double pr = bias_rt<BCEN>(topo_of(g, DB), [&](int m) { return pb[m * sb]; }, ijk[DB], size_of(g, DB), ut > 0);
    return ut * pr;
}

// tracer flux through face (i, j, k) of direction D:  (A * U) * cR  (upwind_biased_advective_fluxes.jl:99-121)
template <int D>
__device__ __forceinline__ double tracer_flux(const Fields &F, const Metrics &M, const double *__restrict__ c, int i, int j, int k)
{
    const GridDev &g = F.g;
    if (topo_of(g, D).t == OCN_FLAT) return 0.0;
    const double *fa = D == 0 ? F.u : D == 1 ? F.v : F.w;
    const Lay &La = D == 0 ? F.Lu : D == 1 ? F.Lv : F.Lw;
    const double ut = fa[ocn::at(La, i, j, k)];
    const double *pc = c + ocn::at(F.Lc, i, j, k);
    const long long sc = stride_of(F.Lc, D);
    const int ijk[3] = {i, j, k};
    const double area = D == 0 ? M.Ax(k) : D == 1 ? M.Ay(k) : M.Az;
    if (F.centered2) return (area * ut) * c2_rt<false>(topo_of(g, D).t, [&](int m) { return pc[m * sc]; });
    const double cr = bias_rt<false>(topo_of(g, D), [&](int m) { return pc[m * sc]; }, ijk[D], size_of(g, D), ut > 0);
    return (area * ut) * cr;
}

struct GRange {
    int i0, i1, j0, j1, k0, k1;
    int ou, ov, ow;  // first index written for Gu (in i), Gv (in j), Gw (in k)
};

// The ranges of ONE launch: a whole range, or the (up to four) wall frames around the interior box -- four thin launches in a row are
// four times the latency of one thread's work (a frame fills a fraction of the chip), one launch over all of them is once.  Blocks are
// 256 threads, bdx x (256 / bdx) cells of range f (ocn::range_block: thin x strips fold towards y); linear block b belongs to the range
// whose [first[f], first[f + 1]) holds it; all ranges share k0 .. k1 (blockIdx.y).
struct GFrames {
    int n;
    GRange r[4];
    int bdx[4], nbx[4], first[5];
};

__device__ __forceinline__ bool frame_cell(const GFrames &F, GRange &r, int &i, int &j, int &k)
{
    const int b = blockIdx.x;
    int f = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < F.n && b >= F.first[q]) f = q;
    r = F.r[f];
    const int local = b - F.first[f], bdx = F.bdx[f], nbx = F.nbx[f];
    const int tx = threadIdx.x % bdx, ty = threadIdx.x / bdx;
    i = r.i0 + (local % nbx) * bdx + tx;
    j = r.j0 + (local / nbx) * (256 / bdx) + ty;
    k = r.k0 + blockIdx.y;
    return i <= r.i1 && j <= r.j1;
}

}  // namespace gen

// compute_Gu! / Gv! / Gw!: G = -div_𝐯u etc. (momentum_advection_operators.jl:46-83)
__global__ __launch_bounds__(256) void momentum_tendencies_general(gen::Fields F, double *__restrict__ Gu, double *__restrict__ Gv,
                                                                   double *__restrict__ Gw, gen::GFrames fr)
{
    using namespace gen;
    gen::GRange r;
    int i, j, k;
    if (!gen::frame_cell(fr, r, i, j, k)) return;
    const GridDev &g = F.g;
    const Metrics M = make_metrics(g);
    const bool fx = g.tx == OCN_FLAT, fy = g.ty == OCN_FLAT, fz = g.tz == OCN_FLAT;
    if (i >= r.ou) {  // Gu at (f,c,c)
        const double dxF = fx ? 0.0 : mom_flux<0, 0, true, 0, 0, true>(F, M, i, j, k) - mom_flux<0, 0, true, 0, 0, true>(F, M, i - 1, j, k);
        const double dyF = fy ? 0.0 : mom_flux<1, 0, false, 0, 1, false>(F, M, i, j + 1, k) - mom_flux<1, 0, false, 0, 1, false>(F, M, i, j, k);
        const double dzF = fz ? 0.0 : mom_flux<2, 0, false, 0, 2, false>(F, M, i, j, k + 1) - mom_flux<2, 0, false, 0, 2, false>(F, M, i, j, k);
        const double rV = 1 / (M.Az * M.dzC(k));
        Gu[ocn::at(F.Lu, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
    }
    if (j >= r.ov) {  // Gv at (c,f,c)
        const double dxF = fx ? 0.0 : mom_flux<0, 1, false, 1, 0, false>(F, M, i + 1, j, k) - mom_flux<0, 1, false, 1, 0, false>(F, M, i, j, k);
        const double dyF = fy ? 0.0 : mom_flux<1, 1, true, 1, 1, true>(F, M, i, j, k) - mom_flux<1, 1, true, 1, 1, true>(F, M, i, j - 1, k);
        const double dzF = fz ? 0.0 : mom_flux<2, 1, false, 1, 2, false>(F, M, i, j, k + 1) - mom_flux<2, 1, false, 1, 2, false>(F, M, i, j, k);
        const double rV = 1 / (M.Az * M.dzC(k));
        Gv[ocn::at(F.Lv, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
    }
    if (k >= r.ow) {  // Gw at (c,c,f)
        const double dxF = fx ? 0.0 : mom_flux<0, 2, false, 2, 0, false>(F, M, i + 1, j, k) - mom_flux<0, 2, false, 2, 0, false>(F, M, i, j, k);
        const double dyF = fy ? 0.0 : mom_flux<1, 2, false, 2, 1, false>(F, M, i, j + 1, k) - mom_flux<1, 2, false, 2, 1, false>(F, M, i, j, k);
        const double dzF = fz ? 0.0 : mom_flux<2, 2, true, 2, 2, true>(F, M, i, j, k) - mom_flux<2, 2, true, 2, 2, true>(F, M, i, j, k - 1);
        const double rV = 1 / (M.Az * M.dzF(k));
        Gw[ocn::at(F.Lw, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
    }
}

// compute_Gc!: Gc = -div_Uc (tracer_advection_operators.jl:30-34)
__global__ __launch_bounds__(256) void tracer_tendency_general(gen::Fields F, const double *__restrict__ c, double *__restrict__ Gc, gen::GFrames fr)
{
    using namespace gen;
    gen::GRange r;
    int i, j, k;
    if (!gen::frame_cell(fr, r, i, j, k)) return;
    const GridDev &g = F.g;
    const Metrics M = make_metrics(g);
    const double dxF = g.tx == OCN_FLAT ? 0.0 : tracer_flux<0>(F, M, c, i + 1, j, k) - tracer_flux<0>(F, M, c, i, j, k);
    const double dyF = g.ty == OCN_FLAT ? 0.0 : tracer_flux<1>(F, M, c, i, j + 1, k) - tracer_flux<1>(F, M, c, i, j, k);
    const double dzF = g.tz == OCN_FLAT ? 0.0 : tracer_flux<2>(F, M, c, i, j, k + 1) - tracer_flux<2>(F, M, c, i, j, k);
    const double rV = 1 / (M.Az * M.dzC(k));
    Gc[ocn::at(F.Lc, i, j, k)] = -(rV * ((dxF + dyF) + dzF));
}

// ---------------------------------------------------------------------------------------------------
// The non-advective terms, ADDED to G in the reference's order  ((((-div𝐯u) + gb) - f×U) - ∇pHY′) - ∂ⱼτᵢⱼ
// (nonhydrostatic_tendency_kernel_functions.jl:47-200): FPlane Coriolis with active-node weighting, the hydrostatic pressure gradient,
// z_dot_g_b without a separate pHY′, the isotropic viscous stress divergence with ν a number or the eddy viscosity νₑ.
// ---------------------------------------------------------------------------------------------------
struct ExtraArgs {
    ocn::TermsDev t;
    const double *nu_e;
};

__device__ __forceinline__ double gen_buoyancy(const ocn::TermsDev &t, long long a)
{
    switch (t.buoyancy) {
        case OCN_BUOYANCY_TRACER: return t.T[a];
        case OCN_BUOYANCY_SEAWATER_TS: return t.g * (t.alpha * t.T[a] - t.beta * t.S[a]);
        case OCN_BUOYANCY_SEAWATER_T: return t.g * t.alpha * t.T[a];
        case OCN_BUOYANCY_SEAWATER_S: return -t.g * t.beta * t.S[a];
        default: return 0.0;
    }
}

__global__ __launch_bounds__(256) void momentum_extra_general(gen::Fields F, ocn::TermsDev t, double *__restrict__ Gu, double *__restrict__ Gv,
                                                              double *__restrict__ Gw, gen::GFrames fr)
{
    using namespace gen;
    gen::GRange r;
    int i, j, k;
    if (!gen::frame_cell(fr, r, i, j, k)) return;
    const GridDev &g = F.g;
    const Metrics M = make_metrics(g);
    const Lay &Lu = F.Lu, &Lv = F.Lv, &Lw = F.Lw, &Lc = F.Lc;
    const double *u = F.u, *v = F.v, *w = F.w, *nu_e = t.nu_e;
    const bool fx = g.tx == OCN_FLAT, fy = g.ty == OCN_FLAT, fz = g.tz == OCN_FLAT;
    const double dx = M.dx, dy = M.dy, nu = t.nu;
#define U_(a, b, c) u[ocn::at(Lu, a, b, c)]
#define V_(a, b, c) v[ocn::at(Lv, a, b, c)]
#define W_(a, b, c) w[ocn::at(Lw, a, b, c)]
#define NE(a, b, c) nu_e[ocn::at(Lc, a, b, c)]
    // derivative operators (derivative_operators.jl:20-30); a difference along a Flat direction is 0
    auto DXU_C = [&](int a, int b, int c) { return fx ? 0.0 : (U_(a + 1, b, c) - U_(a, b, c)) / dx; };
    auto DYV_C = [&](int a, int b, int c) { return fy ? 0.0 : (V_(a, b + 1, c) - V_(a, b, c)) / dy; };
    auto DZW_C = [&](int a, int b, int c) { return fz ? 0.0 : (W_(a, b, c + 1) - W_(a, b, c)) / M.dzC(c); };
    auto DYU_FF = [&](int a, int b, int c) { return fy ? 0.0 : (U_(a, b, c) - U_(a, b - 1, c)) / dy; };
    auto DXV_FF = [&](int a, int b, int c) { return fx ? 0.0 : (V_(a, b, c) - V_(a - 1, b, c)) / dx; };
    auto DZU_FF = [&](int a, int b, int c) { return fz ? 0.0 : (U_(a, b, c) - U_(a, b, c - 1)) / M.dzF(c); };
    auto DXW_FF = [&](int a, int b, int c) { return fx ? 0.0 : (W_(a, b, c) - W_(a - 1, b, c)) / dx; };
    auto DZV_FF = [&](int a, int b, int c) { return fz ? 0.0 : (V_(a, b, c) - V_(a, b, c - 1)) / M.dzF(c); };
    auto DYW_FF = [&](int a, int b, int c) { return fy ? 0.0 : (W_(a, b, c) - W_(a, b - 1, c)) / dy; };
    // viscosity at the stress locations (abstract_scalar_diffusivity_closure.jl:291-296)
    auto NU_C = [&](int a, int b, int c) { return nu_e ? NE(a, b, c) : nu; };
    auto NU_FFC = [&](int a, int b, int c) { return nu_e ? 0.5 * (0.5 * (NE(a - 1, b - 1, c) + NE(a, b - 1, c)) + 0.5 * (NE(a - 1, b, c) + NE(a, b, c))) : nu; };
    auto NU_FCF = [&](int a, int b, int c) { return nu_e ? 0.5 * (0.5 * (NE(a - 1, b, c - 1) + NE(a, b, c - 1)) + 0.5 * (NE(a - 1, b, c) + NE(a, b, c))) : nu; };
    auto NU_CFF = [&](int a, int b, int c) { return nu_e ? 0.5 * (0.5 * (NE(a, b - 1, c - 1) + NE(a, b, c - 1)) + 0.5 * (NE(a, b - 1, c) + NE(a, b, c))) : nu; };
    auto T11 = [&](int a, int b, int c) { return -2 * (NU_C(a, b, c) * DXU_C(a, b, c)); };
    auto T22 = [&](int a, int b, int c) { return -2 * (NU_C(a, b, c) * DYV_C(a, b, c)); };
    auto T33 = [&](int a, int b, int c) { return -2 * (NU_C(a, b, c) * DZW_C(a, b, c)); };
    auto T12 = [&](int a, int b, int c) { return -2 * (NU_FFC(a, b, c) * (0.5 * (DYU_FF(a, b, c) + DXV_FF(a, b, c)))); };
    auto T13 = [&](int a, int b, int c) { return -2 * (NU_FCF(a, b, c) * (0.5 * (DZU_FF(a, b, c) + DXW_FF(a, b, c)))); };
    auto T23 = [&](int a, int b, int c) { return -2 * (NU_CFF(a, b, c) * (0.5 * (DZV_FF(a, b, c) + DYW_FF(a, b, c)))); };
    // inactive_cell (Grids/inactive_node.jl:35-95) and the peripheral-node tests of the Coriolis average
    auto inactive = [&](int a, int b, int c) {
        bool q = false;
        if (g.tx == OCN_BOUNDED) q |= (g.xw && a < 1) | (g.xe && a > g.Nx);  // (inactive_node.jl:5-25: by side on the half-Bounded slabs)
        if (g.ty == OCN_BOUNDED) q |= (b < 1) | (b > g.Ny);
        if (g.tz == OCN_BOUNDED) q |= (c < 1) | (c > g.Nz);
        return q;
    };
    auto act_cfc = [&](int a, int b, int c) { return (inactive(a, b, c) || inactive(a, b - 1, c)) ? 0.0 : 1.0; };
    auto act_fcc = [&](int a, int b, int c) { return (inactive(a, b, c) || inactive(a - 1, b, c)) ? 0.0 : 1.0; };
    const double Axc = M.Ax(k), Ayc = M.Ay(k), Az = M.Az;
    if (i >= r.ou) {
        const long long o = ocn::at(Lu, i, j, k);
        double G = Gu[o];
        if (t.buoyancy) G = G + 0.0;
        if (t.coriolis) {  // x_f_cross_U = -f * active_weighted_ℑxyᶠᶜᶜ(v)
            auto IXF = [&](int jj) { return fx ? V_(i, jj, k) : 0.5 * (V_(i - 1, jj, k) + V_(i, jj, k)); };
            auto IXFa = [&](int jj) { return fx ? act_cfc(i, jj, k) : 0.5 * (act_cfc(i - 1, jj, k) + act_cfc(i, jj, k)); };
            const double an = fy ? IXFa(j) : 0.5 * (IXFa(j) + IXFa(j + 1));
            const double vi = (an == 0) ? 0.0 : (fy ? IXF(j) : 0.5 * (IXF(j) + IXF(j + 1))) / an;
            G = G - (-ocn::coriolis_f_at(t, g.Hy, j, 0) * vi);
        }
        if (t.pHY) G = G - (fx ? 0.0 : (t.pHY[ocn::at(Lc, i, j, k)] - t.pHY[ocn::at(Lc, i - 1, j, k)]) / dx);
        if (t.closure) {
            const double dxF = fx ? 0.0 : Axc * T11(i, j, k) - Axc * T11(i - 1, j, k);
            const double dyF = fy ? 0.0 : Ayc * T12(i, j + 1, k) - Ayc * T12(i, j, k);
            const double dzF = fz ? 0.0 : Az * T13(i, j, k + 1) - Az * T13(i, j, k);
            G = G - 1 / (Az * M.dzC(k)) * ((dxF + dyF) + dzF);
        }
        Gu[o] = G;
    }
    if (j >= r.ov) {
        const long long o = ocn::at(Lv, i, j, k);
        double G = Gv[o];
        if (t.buoyancy) G = G + 0.0;
        if (t.coriolis) {  // y_f_cross_U = f * active_weighted_ℑxyᶜᶠᶜ(u)
            auto IXC = [&](int jj) { return fx ? U_(i, jj, k) : 0.5 * (U_(i, jj, k) + U_(i + 1, jj, k)); };
            auto IXCa = [&](int jj) { return fx ? act_fcc(i, jj, k) : 0.5 * (act_fcc(i, jj, k) + act_fcc(i + 1, jj, k)); };
            const double an = fy ? IXCa(j) : 0.5 * (IXCa(j - 1) + IXCa(j));
            const double ui = (an == 0) ? 0.0 : (fy ? IXC(j) : 0.5 * (IXC(j - 1) + IXC(j))) / an;
            G = G - ocn::coriolis_f_at(t, g.Hy, j, 1) * ui;
        }
        if (t.pHY) G = G - (fy ? 0.0 : (t.pHY[ocn::at(Lc, i, j, k)] - t.pHY[ocn::at(Lc, i, j - 1, k)]) / dy);
        if (t.closure) {
            const double dxF = fx ? 0.0 : Axc * T12(i + 1, j, k) - Axc * T12(i, j, k);
            const double dyF = fy ? 0.0 : Ayc * T22(i, j, k) - Ayc * T22(i, j - 1, k);
            const double dzF = fz ? 0.0 : Az * T23(i, j, k + 1) - Az * T23(i, j, k);
            G = G - 1 / (Az * M.dzC(k)) * ((dxF + dyF) + dzF);
        }
        Gv[o] = G;
    }
    if (k >= r.ow) {
        const long long o = ocn::at(Lw, i, j, k);
        double G = Gw[o];
        if (t.buoyancy) {
            double zb = 0.0;
            if (!t.pHY) zb = fz ? gen_buoyancy(t, ocn::at(Lc, i, j, k))
                                : 1 * (0.5 * (gen_buoyancy(t, ocn::at(Lc, i, j, k - 1)) + gen_buoyancy(t, ocn::at(Lc, i, j, k))));
            G = G + zb;
        }
        if (t.coriolis) G = G - 0.0;
        if (t.closure) {
            const double Axf = dy * M.dzF(k), Ayf = dx * M.dzF(k);
            const double dxF = fx ? 0.0 : Axf * T13(i + 1, j, k) - Axf * T13(i, j, k);
            const double dyF = fy ? 0.0 : Ayf * T23(i, j + 1, k) - Ayf * T23(i, j, k);
            const double dzF = fz ? 0.0 : Az * T33(i, j, k) - Az * T33(i, j, k - 1);
            G = G - 1 / (Az * M.dzF(k)) * ((dxF + dyF) + dzF);
        }
        Gw[o] = G;
    }
#undef U_
#undef V_
#undef W_
#undef NE
}

// Gc <- Gc - ∇_dot_qᶜ (closure_kernel_operators.jl:48-53), κ a number or the eddy diffusivity field interpolated to the faces
__global__ __launch_bounds__(256) void tracer_diffusion_general(GridDev g, double kappa, const double *__restrict__ kappa_e,
                                                                const double *__restrict__ c, double *__restrict__ Gc, gen::GFrames fr)
{
    gen::GRange r;
    int i, j, k;
    if (!gen::frame_cell(fr, r, i, j, k)) return;
    const Metrics M = make_metrics(g);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const bool fx = g.tx == OCN_FLAT, fy = g.ty == OCN_FLAT, fz = g.tz == OCN_FLAT;
#define C_(a, b, cc) c[ocn::at(L, a, b, cc)]
#define KE(a, b, cc) kappa_e[ocn::at(L, a, b, cc)]
    auto QX = [&](int a, int b, int cc) { return -((kappa_e ? 0.5 * (KE(a - 1, b, cc) + KE(a, b, cc)) : kappa) * ((C_(a, b, cc) - C_(a - 1, b, cc)) / M.dx)); };
    auto QY = [&](int a, int b, int cc) { return -((kappa_e ? 0.5 * (KE(a, b - 1, cc) + KE(a, b, cc)) : kappa) * ((C_(a, b, cc) - C_(a, b - 1, cc)) / M.dy)); };
    auto QZ = [&](int a, int b, int cc) { return -((kappa_e ? 0.5 * (KE(a, b, cc - 1) + KE(a, b, cc)) : kappa) * ((C_(a, b, cc) - C_(a, b, cc - 1)) / M.dzF(cc))); };
    const double Ax = M.Ax(k), Ay = M.Ay(k), Az = M.Az;
    const double dxF = fx ? 0.0 : Ax * QX(i + 1, j, k) - Ax * QX(i, j, k);
    const double dyF = fy ? 0.0 : Ay * QY(i, j + 1, k) - Ay * QY(i, j, k);
    const double dzF = fz ? 0.0 : Az * QZ(i, j, k + 1) - Az * QZ(i, j, k);
    const long long o = ocn::at(L, i, j, k);
    Gc[o] = Gc[o] - 1 / (Az * M.dzC(k)) * ((dxF + dyF) + dzF);
#undef C_
#undef KE
}

// ---------------------------------------------------------------------------------------------------
// The fused stage boundary on the FRAMES of a grid with walls (the interior box takes the epilogues of the tiled kernels): the bottom / top
// flux contributions (apply_z_bcs!, apply_flux_bcs.jl:107-160) added to a complete G, then the NEXT stage's rk3_substep! into the second
// storage (runge_kutta_3.jl:150-175) -- the expressions of momentum_extra_cell / tracer_finish.  Wall faces (the excluded periphery of a
// Face field along a Bounded direction: first index, and the face N + 1 behind the last cell) are carried over unchanged.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double substep_value(const ocn::SubstepCoef &sc, double U, double G, const double *Gm, long long o)
{
    return U + (sc.has_zeta ? sc.dt * (sc.gamma * G + sc.zeta * Gm[o]) : (sc.dt * sc.gamma) * G);
}

__global__ __launch_bounds__(256) void momentum_finish_general(gen::Fields F, double *__restrict__ Gu, double *__restrict__ Gv, double *__restrict__ Gw,
                                                               ocn::MomentumFinal mf, gen::GFrames fr)
{
    using namespace gen;
    gen::GRange r;
    int i, j, k;
    if (!gen::frame_cell(fr, r, i, j, k)) return;
    const GridDev &g = F.g;
    const Metrics M = make_metrics(g);
    const double Az = M.Az;
    const bool zb = g.tz == OCN_BOUNDED;
    for (int f = 0; f < 2; ++f) {  // u, v
        const Lay &L = f ? F.Lv : F.Lu;
        const double *U = f ? F.v : F.u;
        double *G_ = f ? Gv : Gu;
        const long long o = ocn::at(L, i, j, k);
        const double val = U[o];
        const bool stepped = f ? (j >= r.ov) : (i >= r.ou);
        double G = G_[o];
        if (zb) {  // (apply_z_bcs! runs over every column of the grid, the excluded wall faces included: their G is never read)
            bool touched = false;
            if (k == 1 && mf.bottom[f].kind == OCN_BC_FLUX) { G += ocn::bc_condition(mf.bottom[f], i, j, g.Nx, val) * Az / (Az * M.dzC(1)); touched = true; }
            if (k == g.Nz && mf.top[f].kind == OCN_BC_FLUX) { G -= ocn::bc_condition(mf.top[f], i, j, g.Nx, val) * Az / (Az * M.dzC(g.Nz)); touched = true; }
            if (touched) G_[o] = G;
        }
        if (mf.sc.on) mf.sub[f].out[o] = stepped ? substep_value(mf.sc, val, G, mf.sub[f].Gm, o) : val;
        if (mf.sc.on) {  // the wall face behind the last cell
            if (!f && i == g.Nx && g.xe) { const long long o2 = ocn::at(L, g.Nx + 1, j, k); mf.sub[0].out[o2] = U[o2]; }
            if (f && j == g.Ny && g.ty == OCN_BOUNDED) { const long long o2 = ocn::at(L, i, g.Ny + 1, k); mf.sub[1].out[o2] = U[o2]; }
        }
    }
    if (mf.sc.on) {
        const long long o = ocn::at(F.Lw, i, j, k);
        const double val = F.w[o];
        const bool wall = zb && k == 1 && g.Nz > 1;  // rk3_substep! never steps the wall face
        mf.sub[2].out[o] = (k >= r.ow && !wall) ? substep_value(mf.sc, val, Gw[o], mf.sub[2].Gm, o) : val;
        if (zb && k == g.Nz) mf.sub[2].out[o + F.Lw.s3] = F.w[o + F.Lw.s3];
    }
}

__global__ __launch_bounds__(256) void tracer_finish_general(GridDev g, const double *__restrict__ c, double *__restrict__ Gc, ocn::TracerFuse tf,
                                                             gen::GFrames fr)
{
    gen::GRange r;
    int i, j, k;
    if (!gen::frame_cell(fr, r, i, j, k)) return;
    const Metrics M = make_metrics(g);
    const double az = M.Az;
    const long long o = ocn::at(ocn::make_lay(g, OCN_LOC_CCC), i, j, k);
    const double c0 = c[o];
    double G = Gc[o];
    if (g.tz == OCN_BOUNDED) {
        bool touched = false;
        if (k == 1 && tf.bottom.kind == OCN_BC_FLUX) { G += ocn::bc_condition(tf.bottom, i, j, g.Nx, c0) * az / (az * M.dzC(1)); touched = true; }
        if (k == g.Nz && tf.top.kind == OCN_BC_FLUX) { G -= ocn::bc_condition(tf.top, i, j, g.Nx, c0) * az / (az * M.dzC(g.Nz)); touched = true; }
        if (touched) Gc[o] = G;
    }
    if (tf.sc.on) tf.sub.out[o] = substep_value(tf.sc, c0, G, tf.sub.Gm, o);
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static int make_grange(const ocn_grid *grid, const int32_t *range, gen::GRange &r)
{
    if (range) {
        r.i0 = range[0]; r.i1 = range[1]; r.j0 = range[2]; r.j1 = range[3]; r.k0 = range[4]; r.k1 = range[5];
        if (r.i0 < 1 || r.i1 > grid->Nx || r.j0 < 1 || r.j1 > grid->Ny || r.k0 < 1 || r.k1 > grid->Nz) {
            ocn::set_error("tendency range {%d:%d,%d:%d,%d:%d} outside the interior %dx%dx%d", r.i0, r.i1, r.j0, r.j1, r.k0, r.k1, grid->Nx,
                           grid->Ny, grid->Nz);
            return OCN_ERR_INVALID_ARGUMENT;
        }
        r.ou = r.ov = r.ow = 1;  // KernelParameters: periphery not excluded
    } else {
        r.i0 = 1; r.i1 = grid->Nx; r.j0 = 1; r.j1 = grid->Ny; r.k0 = 1; r.k1 = grid->Nz;
        r.ou = (ocn::x_wall_west(*grid) && grid->Nx > 1) ? 2 : 1;  // periphery_offset(Face, Bounded, N) (kernel_launching.jl:113-114)
        r.ov = (grid->ty == OCN_BOUNDED && grid->Ny > 1) ? 2 : 1;
        r.ow = (grid->tz == OCN_BOUNDED && grid->Nz > 1) ? 2 : 1;
    }
    return OCN_SUCCESS;
}

static gen::Fields make_fields(const ocn_grid *grid, const double *u, const double *v, const double *w, int centered2)
{
    gen::Fields F;
    F.g = ocn::to_dev(*grid);
    F.u = u; F.v = v; F.w = w;
    F.Lu = ocn::make_lay(F.g, OCN_LOC_FCC);
    F.Lv = ocn::make_lay(F.g, OCN_LOC_CFC);
    F.Lw = ocn::make_lay(F.g, OCN_LOC_CCF);
    F.Lc = ocn::make_lay(F.g, OCN_LOC_CCC);
    F.centered2 = centered2;
    return F;
}

// tendencies.hip (same namespace): the LDS-tiled kernels over the interior box of a grid with walls in x / y
int launch_momentum_tendencies_box(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu, double *Gv, double *Gw,
                                   const int32_t box[4], int *launched, hipStream_t stream, const ocn::FuseArgs *fuse = nullptr, int ranged = 0);
int launch_tracer_tendency_box(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c, double *Gc,
                               const int32_t box[4], int *launched, hipStream_t stream, const ocn::TracerFuse *fuse = nullptr, int ranged = 0);
#if !OCN_UPWIND
// physics.hip (compiled for the WENO namespaces only; the extra terms do not depend on the advection scheme): the tiled finishing pass
int launch_momentum_extra_box(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu, double *Gv,
                              double *Gw, const int32_t box[4], int *launched, hipStream_t stream, const ocn::MomentumFinal *fin, int ranged = 0);
#endif

// Interior box + wall frames.  The topology-conditional reconstructions of a Bounded direction differ from the Periodic ones only within
// a stencil of the walls (topologically_conditional_interpolation.jl:46-52: full order for faces 4 .. N-2 and centres 3 .. N-2), so every
// cell with 4 <= i <= Nx-3 (and the same in y) gets the LDS-tiled shared-flux kernel of tendencies.hip with per-field layouts -- the same
// expressions on the same operands as the per-cell kernel, bit for bit -- and only the frames next to the walls run the per-cell kernel
// with its run-time topology.  Returns the box {i0, i1, j0, j1}, or false when the grid has no wall in x / y, a Flat x / y / z, a scheme
// the tiles do not carry, or a box too small for them (OCN_GENERAL_TILED=0 switches the decomposition off).
// A range (KernelParameters: the interior / buffer split of a slab, interleave_communication_and_computation.jl:29-67) with the full
// k extent is intersected with the box: the interior range of a channel's slab keeps the tiled kernels.
static bool interior_box(const ocn_grid *grid, int centered2, const int32_t *range, int32_t box[4])
{
    static const bool off = [] { const char *e = getenv("OCN_GENERAL_TILED"); return e && e[0] == '0'; }();
    if (off || centered2) return false;
    if (range && (range[4] != 1 || range[5] != grid->Nz)) return false;
    if (grid->tx == OCN_FLAT || grid->ty == OCN_FLAT || grid->tz == OCN_FLAT) return false;
    const bool xw = ocn::x_wall_west(*grid), xe = ocn::x_wall_east(*grid);
    if (!xw && !xe && grid->ty != OCN_BOUNDED) return false;
    box[0] = xw ? 4 : 1; box[1] = xe ? grid->Nx - 3 : grid->Nx;
    box[2] = grid->ty == OCN_BOUNDED ? 4 : 1; box[3] = grid->ty == OCN_BOUNDED ? grid->Ny - 3 : grid->Ny;
    if (range) {
        box[0] = std::max(box[0], range[0]); box[1] = std::min(box[1], range[1]);
        box[2] = std::max(box[2], range[2]); box[3] = std::min(box[3], range[3]);
    }
    return box[1] - box[0] + 1 >= 16 && box[3] - box[2] + 1 >= 8 && grid->Nz >= 4 && grid->Hx >= 3 && grid->Hy >= 3 && grid->Hz >= 3;
}

// one more range of a launch (empty ranges are skipped)
static void add_range(gen::GFrames &F, const gen::GRange &r)
{
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = r.k1 - r.k0 + 1;
    if (wx < 1 || wy < 1 || wz < 1 || F.n >= 4) return;
    const dim3 block = ocn::range_block(wx);
    const int f = F.n++;
    F.r[f] = r;
    F.bdx[f] = (int)block.x;
    F.nbx[f] = (wx + (int)block.x - 1) / (int)block.x;
    F.first[f + 1] = F.first[f] + F.nbx[f] * ((wy + (int)block.y - 1) / (int)block.y);
}
static gen::GFrames whole_range(const gen::GRange &r)
{
    gen::GFrames F{};
    add_range(F, r);
    return F;
}
// the (up to four) frames around the box as the ranges of ONE launch; `whole` carries the periphery offsets of the whole grid
static gen::GFrames frames_around(const ocn_grid *grid, const int32_t box[4], const gen::GRange &whole)
{
    (void)grid;
    const int spans[4][4] = {{whole.i0, box[0] - 1, whole.j0, whole.j1},          // west
                             {box[1] + 1, whole.i1, whole.j0, whole.j1},          // east
                             {box[0], box[1], whole.j0, box[2] - 1},              // south (between the x frames)
                             {box[0], box[1], box[3] + 1, whole.j1}};             // north
    gen::GFrames F{};
    for (const auto &sp : spans) {
        gen::GRange r = whole;
        r.i0 = sp[0]; r.i1 = sp[1]; r.j0 = sp[2]; r.j1 = sp[3];
        add_range(F, r);
    }
    return F;
}
// launch `kernel(args..., F)` over the ranges of F -- or, OCN_GENERAL_FRAMES=separate, one launch per range as before round 4's last change
#define OCN_GEN_LAUNCH(kernel, F_, ...)                                                                                              \
    do {                                                                                                                             \
        static const bool sep_ = [] { const char *e = getenv("OCN_GENERAL_FRAMES"); return e && !strcmp(e, "separate"); }();         \
        if ((F_).n > 0 && !sep_) {                                                                                                   \
            hipLaunchKernelGGL(kernel, dim3((F_).first[(F_).n], (F_).r[0].k1 - (F_).r[0].k0 + 1), dim3(256), 0, stream, __VA_ARGS__, (F_)); \
        } else {                                                                                                                     \
            for (int f_ = 0; f_ < (F_).n; ++f_) {                                                                                    \
                const gen::GFrames one_ = whole_range((F_).r[f_]);                                                                   \
                hipLaunchKernelGGL(kernel, dim3(one_.first[1], one_.r[0].k1 - one_.r[0].k0 + 1), dim3(256), 0, stream, __VA_ARGS__, one_); \
            }                                                                                                                        \
        }                                                                                                                            \
    } while (0)

// fin != NULL (a model without extra terms): the next substep rides on this launch -- the epilogue of the tiled kernel on the box, the
// finishing kernel on the frames
int launch_momentum_tendencies_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, double *Gu,
                                       double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin)
{
    gen::GRange r;
    int st = make_grange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    const bool fluxes = fin && (fin->bottom[0].kind || fin->bottom[1].kind || fin->top[0].kind || fin->top[1].kind);
    const bool finish = fin && (fin->sc.on || fluxes);
    const gen::Fields F = make_fields(grid, u, v, w, centered2);
    gen::GFrames cells = whole_range(r);
    int32_t box[4];
    if (interior_box(grid, centered2, range, box) && !fluxes) {
        int launched = 0;
        ocn::FuseArgs fz{};
        if (fin && fin->sc.on) {
            for (int f = 0; f < 3; ++f) { fz.Gm[f] = fin->sub[f].Gm; fz.Uo[f] = fin->sub[f].out; }
            fz.dt = fin->sc.dt; fz.gamma = fin->sc.gamma; fz.zeta = fin->sc.zeta; fz.on = 1; fz.has_zeta = fin->sc.has_zeta;
        }
        st = launch_momentum_tendencies_box(grid, u, v, w, Gu, Gv, Gw, box, &launched, stream, (fin && fin->sc.on) ? &fz : nullptr, range != nullptr);
        if (st != OCN_SUCCESS) return st;
        if (launched) cells = frames_around(grid, box, r);
    }
    OCN_GEN_LAUNCH(momentum_tendencies_general, cells, F, Gu, Gv, Gw);
    if (finish) OCN_GEN_LAUNCH(momentum_finish_general, cells, F, Gu, Gv, Gw, *fin);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

// fuse != NULL: what the Periodic grids' tiled kernel folds in besides advection -- Gc = -div_Uc - ∇_dot_qᶜ (κ a number or the field κₑ), the
// bottom / top flux contributions, the next substep -- in the same call: the box kernel takes the whole TracerFuse (centre fields: their
// layout has no walls in it), the frames run the per-cell kernels one after the other (same sums in the same order)
int launch_tracer_tendency_general(const ocn_grid *grid, int centered2, const double *u, const double *v, const double *w, const double *c,
                                   double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse)
{
    gen::GRange r;
    int st = make_grange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    const GridDev gd = ocn::to_dev(*grid);
    ocn::TracerFuse tf{};
    if (fuse) tf = *fuse;
    const bool diffusion = tf.diffusion != 0, finish = tf.bottom.kind || tf.top.kind || tf.sc.on;
    const gen::Fields F = make_fields(grid, u, v, w, centered2);
    gen::GFrames cells = whole_range(r);
    int32_t box[4];
    if (interior_box(grid, centered2, range, box)) {
        int launched = 0;
        st = launch_tracer_tendency_box(grid, u, v, w, c, Gc, box, &launched, stream, fuse ? &tf : nullptr, range != nullptr);
        if (st != OCN_SUCCESS) return st;
        if (launched) cells = frames_around(grid, box, r);
    }
    OCN_GEN_LAUNCH(tracer_tendency_general, cells, F, c, Gc);
    if (diffusion) OCN_GEN_LAUNCH(tracer_diffusion_general, cells, gd, tf.kappa, tf.kappa_e, c, Gc);
    if (finish) OCN_GEN_LAUNCH(tracer_finish_general, cells, gd, c, Gc, tf);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

// fin != NULL: the u / v bottom / top flux contributions and the next substep of u, v, w ride on this last pass over G (the finishing pass of
// the Periodic grids): inside the tiled kernel on the box, as one more per-cell kernel on the frames
int launch_momentum_extra_general(const ocn_grid *grid, const ocn::TermsDev &t, const double *u, const double *v, const double *w, double *Gu,
                                  double *Gv, double *Gw, const int32_t *range, hipStream_t stream, const ocn::MomentumFinal *fin)
{
    const bool finish = fin && (fin->sc.on || fin->bottom[0].kind || fin->bottom[1].kind || fin->top[0].kind || fin->top[1].kind);
    gen::GRange r;
    int st = make_grange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    const gen::Fields F = make_fields(grid, u, v, w, 0);
    gen::GFrames cells = whole_range(r);
#if !OCN_UPWIND
    // interior box (the tiled finishing pass with per-field layouts: every stress once per face, velocities through LDS) + wall frames
    int32_t box[4];
    if (interior_box(grid, 0, range, box)) {
        int launched = 0;
        st = launch_momentum_extra_box(grid, t, u, v, w, Gu, Gv, Gw, box, &launched, stream, fin, range != nullptr);
        if (st != OCN_SUCCESS) return st;
        if (launched) cells = frames_around(grid, box, r);
    }
#endif
    OCN_GEN_LAUNCH(momentum_extra_general, cells, F, t, Gu, Gv, Gw);
    if (finish) OCN_GEN_LAUNCH(momentum_finish_general, cells, F, Gu, Gv, Gw, *fin);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_tracer_diffusion_general(const ocn_grid *grid, double kappa, const double *kappa_e, const double *c, double *Gc,
                                    const int32_t *range, hipStream_t stream)
{
    gen::GRange r;
    int st = make_grange(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    const gen::GFrames cells = whole_range(r);
    OCN_GEN_LAUNCH(tracer_diffusion_general, cells, ocn::to_dev(*grid), kappa, kappa_e, c, Gc);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

}  // namespace OCN_NS
