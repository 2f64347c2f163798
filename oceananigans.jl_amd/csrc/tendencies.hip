// tendencies.hip -- K1-K4: fused compute_Gu!/Gv!/Gw! and compute_Gc! for advection = WENO() (5th order).
// Reference: src/Models/NonhydrostaticModels/compute_nonhydrostatic_tendencies.jl:57-195,
// nonhydrostatic_tendency_kernel_functions.jl:47-259, src/Advection/momentum_advection_operators.jl:46-83,
// upwind_biased_advective_fluxes.jl:23-121, tracer_advection_operators.jl:30-34.
//
// Compiled twice (see ocn_weno.h): namespace ocn_strict / ocn_fast.
//
// Kernel "direct": one thread per cell, x fastest across the 64 lanes of a wave so every stencil row is a
// coalesced 512-B read; the 6 (u) + 6 (v) + 6 (w) face fluxes a cell needs are evaluated in registers.
// k is blockIdx.z, so all Bounded-z order-reduction tests are wave-uniform (no divergence).
#include <cstdlib>
#include <cstring>

#include "ocn_weno.h"


#ifndef OCN_NARROW_PAD
#define OCN_NARROW_PAD 0
#endif
#ifndef OCN_TRACER_LDS_SELECT
#define OCN_TRACER_LDS_SELECT 0
#endif

namespace OCN_NS {

using ocn::GridDev;
using ocn::Lay;

struct Range {
    int i0, i1, j0, j1, k0, k1;  // 1-based inclusive
    int ou, ov, ow;              // first index written for Gu (in i), Gv (in j), Gw (in k): periphery exclusion
    int xcd;                     // remap workgroups so that each XCD owns a contiguous band of tiles (block_coords)
};

// MI355X dispatches consecutive workgroup ids round-robin over its 8 XCDs, each with its own L2.  Neighbouring tiles share their
// stencil rings (a 32 x 8 patch reads 37 x 13 cells), so with the plain mapping every ring is fetched by up to 8 different L2s.
// With r.xcd set, hardware workgroup b (XCD b % 8) takes the logical tile start_{b % 8} + b / 8: every XCD walks its own contiguous
// range of tiles in x-fastest order and finds its neighbours' rings in its own L2.  A bijection for any grid size.
__device__ __forceinline__ void block_coords(const Range &r, int &bx, int &by, int &bz)
{
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (!r.xcd) return;
    const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
    const unsigned b = bx + nx * (by + ny * bz);
    const unsigned q = b & 7u, chunk = n >> 3, rem = n & 7u;
    const unsigned logical = q * chunk + (q < rem ? q : rem) + (b >> 3);
    bx = logical % nx;
    by = (logical / nx) % ny;
    bz = logical / (nx * ny);
}

// One momentum flux  U~ * psi^R.
//   advecting line: pointer pa at the face element, stride sa, metric MET (1 Ax, 2 Ay, 3 Az), AZ: line runs along z
//   advected  line: pointer pb at the face element, stride sb
template <int TA, bool ACEN, int MET, bool AZ, int TB, bool BCEN>
__device__ __forceinline__ double mom_flux(const Metrics &M, const double *__restrict__ pa, long long sa, int idxa, int Na,
                                           int ka, const double *__restrict__ pb, long long sb, int idxb, int Nb)
{
    double ut;
    if (MET == 3) {
        const double a = M.Az;
        ut = sym_interp<TA, ACEN>([&](int m) { return a * pa[m * sa]; }, idxa, Na);
    } else if (AZ) {
        ut = sym_interp<TA, ACEN>([&](int m) { return (MET == 1 ? M.Ax(ka + m) : M.Ay(ka + m)) * pa[m * sa]; }, idxa, Na);
    } else {
        const double a = (MET == 1) ? M.Ax(ka) : M.Ay(ka);
        ut = sym_interp<TA, ACEN>([&](int m) { return a * pa[m * sa]; }, idxa, Na);
    }
    const bool left = ut > 0;  // bias(u) = ifelse(u > 0, LeftBias(), RightBias())
    const double pr = bias_interp<TB, BCEN>([&](int m) { return pb[m * sb]; }, idxb, Nb, left);
    return ut * pr;
}

template <int TZ>
__global__ __launch_bounds__(256) void momentum_tendencies_direct(GridDev g, const double *__restrict__ u,
                                                                  const double *__restrict__ v,
                                                                  const double *__restrict__ w, double *__restrict__ Gu,
                                                                  double *__restrict__ Gv, double *__restrict__ Gw,
                                                                  Range r, ocn::FuseArgs fz)
{
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1) return;

    constexpr int P = OCN_PERIODIC;
    const Metrics M = make_metrics(g);
    const Lay Lu = ocn::make_lay(g, OCN_LOC_FCC), Lv = ocn::make_lay(g, OCN_LOC_CFC), Lw = ocn::make_lay(g, OCN_LOC_CCF);
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
    const long long su2 = Lu.s2, su3 = Lu.s3, sv2 = Lv.s2, sv3 = Lv.s3, sw2 = Lw.s2, sw3 = Lw.s3;
    const double *pu = u + ocn::at(Lu, i, j, k);
    const double *pv = v + ocn::at(Lv, i, j, k);
    const double *pw = w + ocn::at(Lw, i, j, k);
    constexpr bool ZF = (TZ == OCN_FLAT);

    // ---- Gu at (f,c,c): -(1/V) [ dx(F_Uu) + dy(F_Vu) + dz(F_Wu) ]  (momentum_advection_operators.jl:46-50)
    if (i >= r.ou) {
        // F_Uu(i) - F_Uu(i-1): sym x-centre of Ax*u, biased x-centre of u (lines shifted to face i+1 / i)
        const double fx1 = mom_flux<P, true, 1, false, P, true>(M, pu + 1, 1, i, Nx, k, pu + 1, 1, i, Nx);
        const double fx0 = mom_flux<P, true, 1, false, P, true>(M, pu, 1, i - 1, Nx, k, pu, 1, i - 1, Nx);
        // F_Vu(j+1) - F_Vu(j): sym x-face of Ay*v, biased y-face of u
        const double fy1 = mom_flux<P, false, 2, false, P, false>(M, pv + sv2, 1, i, Nx, k, pu + su2, su2, j + 1, Ny);
        const double fy0 = mom_flux<P, false, 2, false, P, false>(M, pv, 1, i, Nx, k, pu, su2, j, Ny);
        double dzF = 0.0;
        if (!ZF) {
            // F_Wu(k+1) - F_Wu(k): sym x-face of Az*w, biased z-face of u
            const double fz1 = mom_flux<P, false, 3, false, TZ, false>(M, pw + sw3, 1, i, Nx, k + 1, pu + su3, su3, k + 1, Nz);
            const double fz0 = mom_flux<P, false, 3, false, TZ, false>(M, pw, 1, i, Nx, k, pu, su3, k, Nz);
            dzF = fz1 - fz0;
        }
        const double rV = 1 / (M.Az * M.dzC(k));
        const long long o = ocn::at(Lu, i, j, k);
        double G = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
        if (fz.acc) G = G + Gu[o];
        Gu[o] = G;
        if (fz.on) fz.Uo[0][o] = pu[0] + (fz.has_zeta ? fz.dt * (fz.gamma * G + fz.zeta * fz.Gm[0][o]) : (fz.dt * fz.gamma) * G);
    }
    // ---- Gv at (c,f,c)  (:63-67)
    if (j >= r.ov) {
        // F_Uv(i+1) - F_Uv(i): sym y-face of Ax*u, biased x-face of v
        const double fx1 = mom_flux<P, false, 1, false, P, false>(M, pu + 1, su2, j, Ny, k, pv + 1, 1, i + 1, Nx);
        const double fx0 = mom_flux<P, false, 1, false, P, false>(M, pu, su2, j, Ny, k, pv, 1, i, Nx);
        // F_Vv(j) - F_Vv(j-1): sym y-centre of Ay*v, biased y-centre of v
        const double fy1 = mom_flux<P, true, 2, false, P, true>(M, pv + sv2, sv2, j, Ny, k, pv + sv2, sv2, j, Ny);
        const double fy0 = mom_flux<P, true, 2, false, P, true>(M, pv, sv2, j - 1, Ny, k, pv, sv2, j - 1, Ny);
        double dzF = 0.0;
        if (!ZF) {
            // F_Wv(k+1) - F_Wv(k): sym y-face of Az*w, biased z-face of v
            const double fz1 = mom_flux<P, false, 3, false, TZ, false>(M, pw + sw3, sw2, j, Ny, k + 1, pv + sv3, sv3, k + 1, Nz);
            const double fz0 = mom_flux<P, false, 3, false, TZ, false>(M, pw, sw2, j, Ny, k, pv, sv3, k, Nz);
            dzF = fz1 - fz0;
        }
        const double rV = 1 / (M.Az * M.dzC(k));
        const long long o = ocn::at(Lv, i, j, k);
        double G = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
        if (fz.acc) G = G + Gv[o];
        Gv[o] = G;
        if (fz.on) fz.Uo[1][o] = pv[0] + (fz.has_zeta ? fz.dt * (fz.gamma * G + fz.zeta * fz.Gm[1][o]) : (fz.dt * fz.gamma) * G);
    }
    // ---- Gw at (c,c,f)  (:79-83)
    if (k >= r.ow) {
        // F_Uw(i+1) - F_Uw(i): sym z-face of Ax*u, biased x-face of w
        const double fx1 = mom_flux<TZ, false, 1, true, P, false>(M, pu + 1, su3, k, Nz, k, pw + 1, 1, i + 1, Nx);
        const double fx0 = mom_flux<TZ, false, 1, true, P, false>(M, pu, su3, k, Nz, k, pw, 1, i, Nx);
        // F_Vw(j+1) - F_Vw(j): sym z-face of Ay*v, biased y-face of w
        const double fy1 = mom_flux<TZ, false, 2, true, P, false>(M, pv + sv2, sv3, k, Nz, k, pw + sw2, sw2, j + 1, Ny);
        const double fy0 = mom_flux<TZ, false, 2, true, P, false>(M, pv, sv3, k, Nz, k, pw, sw2, j, Ny);
        double dzF = 0.0;
        if (!ZF) {
            // F_Ww(k) - F_Ww(k-1): sym z-centre of Az*w, biased z-centre of w
            const double fz1 = mom_flux<TZ, true, 3, true, TZ, true>(M, pw + sw3, sw3, k, Nz, k + 1, pw + sw3, sw3, k, Nz);
            const double fz0 = mom_flux<TZ, true, 3, true, TZ, true>(M, pw, sw3, k - 1, Nz, k, pw, sw3, k - 1, Nz);
            dzF = fz1 - fz0;
        }
        const double rV = 1 / (M.Az * M.dzF(k));
        const long long o = ocn::at(Lw, i, j, k);
        double G = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
        if (fz.acc) G = G + Gw[o];
        Gw[o] = G;
        const bool wall = (TZ == OCN_BOUNDED) && k == 1 && Nz > 1;  // rk3_substep! never steps the wall face
        if (fz.on) fz.Uo[2][o] = wall ? pw[0] : pw[0] + (fz.has_zeta ? fz.dt * (fz.gamma * G + fz.zeta * fz.Gm[2][o]) : (fz.dt * fz.gamma) * G);
    } else if (fz.on) {
        fz.Uo[2][ocn::at(Lw, i, j, k)] = pw[0];  // wall face (exclude_periphery): carried over unchanged
    }
    if (fz.on && TZ == OCN_BOUNDED && k == Nz) fz.Uo[2][ocn::at(Lw, i, j, Nz + 1)] = pw[sw3];  // top wall face
}

// ---------------------------------------------------------------------------------------------------
// Kernel "tiled": each of the 9 momentum fluxes is evaluated ONCE per cell (the reference evaluates each twice,
// SURVEY.md Appendix B) and shared with the neighbouring cell through LDS.
//
//  * a workgroup owns a TX x TY patch of (i,j) columns and marches KZ planes upward in k;
//  * every thread keeps the z-stencils of its own column (u, v, w at k-2..k+3) in registers, so each global value
//    enters the workgroup once per (TX+5)x(TY+5) tile instead of once per stencil tap;
//  * plane k of u, v and planes k, k+1 of w live in LDS with a 3/2-cell ring for the x and y stencils
//    (own cells come from the register windows, only the ring is re-read from L2);
//  * fluxes are indexed so that a cell only needs its EAST and NORTH neighbours' values:
//      x-fluxes owned by thread (i,j): Fuu(i-1) [centre], Fuv(i) [face], Fuw(i) [face]
//      y-fluxes owned:                 Fvv(j-1) [centre], Fvu(j) [face], Fvw(j) [face]
//      z-fluxes: top-face Fwu(k+1), Fwv(k+1) and centre Fww(k); the bottom ones are last iteration's registers.
//    Patches therefore overlap by one column/row: (TX-1) x (TY-1) outputs per TX x TY threads.
// All flux expressions are the same as in the direct kernel, so strict mode stays bit-identical to the oracle.
// ---------------------------------------------------------------------------------------------------
// GL ("general layouts"): the fields have their own parent layouts -- a Bounded x (y) gives u (v) one more point along it
// (grid_utils.jl:66-72).  Used for the INTERIOR BOX of grids with walls in x / y (launch_momentum_tendencies_box): every cell of that
// box is at least a full stencil away from the walls, where the topology-conditional reconstructions are the Periodic ones.
// ST ("strips"): the epilogue also writes the stepped velocities of the Hx westmost / eastmost columns into the send buffers of the next
// x-halo exchange of a slab-x rank (FuseArgs::strip_w / strip_e): the exchange then needs no pack launch.
template <int TZ, int TX, int TY, int W, bool PC, bool OB = false, bool GL = false, bool ST = false>
__global__ __launch_bounds__(TX *TY, W) void momentum_tendencies_tiled(GridDev g, const double *__restrict__ u,
                                                                       const double *__restrict__ v,
                                                                       const double *__restrict__ w, double *__restrict__ Gu,
                                                                       double *__restrict__ Gv, double *__restrict__ Gw, Range r,
                                                                       int KZ, ocn::FuseArgs fz)
{
    constexpr int P = OCN_PERIODIC;
    constexpr int LX = TX + 5, LY = TY + 5, NT = TX * TY;
    constexpr int LXP = LX + (TX == 17 ? OCN_NARROW_PAD : 0);  // row stride of the LDS planes (padding experiment for the 17-wide patches)
    constexpr int NRING = LX * LY - NT;         // ring cells of one tile
    constexpr int RPT = (NRING + NT - 1) / NT;  // ring cells per thread (1 or 2)
    static_assert(RPT <= 2, "tile too small for its ring");
    // OB ("one barrier"): planes k and k+1 of u, v and k, k+1, k+2 of w are resident and the flux exchange is double-buffered, so the
    // staging of the NEXT plane moves behind this plane's flux evaluation and shares its barrier with the flux exchange.
    constexpr int NUV = OB ? 2 : 1, NW_ = OB ? 3 : 2, NEX = OB ? 2 : 1;
    __shared__ double su_[NUV][LY][LXP], sv_[NUV][LY][LXP], sw[NW_][LY][LXP];
    __shared__ double ex_[NEX][6][NT];  // Fuu_w, Fuv, Fuw (read by the west neighbour), Fvv_s, Fvu, Fvw (by the south one)
    auto uvslot = [](int kk) { return OB ? (kk & 1) : 0; };
    auto wslot = [](int kk) { return OB ? (kk % 3) : (kk & 1); };

    Metrics M = make_metrics(g);
    if (TZ == OCN_PERIODIC) M.dzc = M.dzf = nullptr;  // a Periodic z is never stretched: lets the compiler fold the metric loads
    // x and y are Periodic for every supported grid, so the row/plane strides and the interior offset are the same for
    // all staggered locations (only the number of z planes differs): one layout serves u, v, w, G and p.
    static_assert(!(GL && PC), "the correction on load shares one layout between p, u, v, w");
    static_assert(!ST || (PC && TZ == OCN_PERIODIC && !GL), "strips are written by the correction-on-load stage of a periodic-z slab");
    const Lay L0 = ocn::make_lay(g, OCN_LOC_CCC);
    const Lay Lu = GL ? ocn::make_lay(g, OCN_LOC_FCC) : L0, Lv = GL ? ocn::make_lay(g, OCN_LOC_CFC) : L0, Lw = GL ? ocn::make_lay(g, OCN_LOC_CCF) : L0;
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    int bx, by, bz;
    block_coords(r, bx, by, bz);
    const int ti0 = r.i0 + bx * (TX - 1), tj0 = r.j0 + by * (TY - 1);
    const int k_start = r.k0 + bz * KZ;
    const int k_end = min(k_start + KZ - 1, r.k1);
    // own column (clamped into the parent array; clamped duplicates are never written)
    const int imax = Nx + g.Hx, jmax = Ny + g.Hy;
    const int i = min(ti0 + tx, imax), j = min(tj0 + ty, jmax);
    const bool writes = (tx < TX - 1) && (ty < TY - 1) && (ti0 + tx <= r.i1) && (tj0 + ty <= r.j1);
    const int lx = tx + 3, ly = ty + 3;  // own cell inside the LDS tile

    // one element offset addresses u, v, w (and G) of the own column: same layout for every field (see above)
    const long long own0 = ocn::at(Lu, i, j, 1);
    const long long ownv = GL ? ocn::at(Lv, i, j, 1) : own0, ownw = GL ? ocn::at(Lw, i, j, 1) : own0;
    const double *pu = u + own0, *pv = v + ownv, *pw = w + ownw;
    const long long su3 = Lu.s3, sv3 = GL ? Lv.s3 : su3, sw3 = GL ? Lw.s3 : su3;
    // PC: the previous stage's pressure correction is applied on load (indices wrap periodically, so neither the
    // pressure halos nor re-filled velocity halos are needed): same expression as pressure_correct_kernel.
    auto wrp = [](int q, int N) { return q < 1 ? q + N : (q > N ? q - N : q); };
    // x of a rank-local slab (fz.pc_xhalo): p has its neighbours' planes in its x halos, nothing wraps
    const bool xhalo = PC && fz.pc_xhalo;
    auto wrx = [&](int q) { return xhalo ? q : (q < 1 ? q + Nx : (q > Nx ? q - Nx : q)); };
    const double *pC = nullptr, *pWn = nullptr, *pSn = nullptr;  // p columns at (i,j), (i-1,j), (i,j-1)
    if (PC) {
        const Lay &Lp = L0;
        pC = fz.pc_p + ocn::at(Lp, wrx(i), wrp(j, Ny), 1);
        pWn = fz.pc_p + ocn::at(Lp, wrx(i - 1), wrp(j, Ny), 1);
        pSn = fz.pc_p + ocn::at(Lp, wrx(i), wrp(j - 1, Ny), 1);
    }
    const double pcdt = fz.pc_dt;
#if OCN_STRICT
#define OCN_PC_APPLY(raw, d, h) ((raw) - ((d) / (h)) * pcdt)  // reference expression: difference / spacing, times the stage's dt
    const double hx = M.dx, hy = M.dy, hz = M.dz;
#else
    // fast math: one FMA with dt / spacing, which sits in scalar registers (in vector registers the three loop invariants push the
    // kernel over its 168-register budget: spills inside the plane loop)
#define OCN_PC_APPLY(raw, d, h) __builtin_fma(-(d), h, raw)
    const double hx = ocn::to_sgpr(pcdt * (1.0 / M.dx)), hy = ocn::to_sgpr(pcdt * (1.0 / M.dy)), hz = ocn::to_sgpr(pcdt * (1.0 / M.dz));
#endif
    auto zz = [&](int kk) { return (long long)(wrp(kk, Nz) - 1) * su3; };  // plane offset of p (same strides when periodic)
    auto own_u = [&](int kk) {
        const double raw = pu[(kk - 1) * su3];
        if (!PC) return raw;
        return OCN_PC_APPLY(raw, pC[zz(kk)] - pWn[zz(kk)], hx);
    };
    auto own_v = [&](int kk) {
        const double raw = pv[(kk - 1) * sv3];
        if (!PC) return raw;
        return OCN_PC_APPLY(raw, pC[zz(kk)] - pSn[zz(kk)], hy);
    };
    auto own_w = [&](int kk) {
        const double raw = pw[(kk - 1) * sw3];
        if (!PC) return raw;
        return OCN_PC_APPLY(raw, pC[zz(kk)] - pC[zz(kk - 1)], hz);
    };
#define ZU(k) own_u(k)
#define ZV(k) own_v(k)
#define ZW(k) own_w(k)

    // Static ring assignment: ring cell q (0 <= q < NRING) <-> tile cell (cx, cy) outside the TX x TY core.
    int rcx[RPT], rcy[RPT];
    bool ron[RPT];
    long long roff[RPT], roffv[RPT], roffw[RPT];  // ring cell offset (plane 1): one value shared by u, v, w unless GL
    const double *rpc[RPT] = {}, *rpw[RPT] = {}, *rps[RPT] = {};
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
        const int q = tid + s * NT;
        ron[s] = q < NRING;
        int cx, cy;
        if (q < 3 * LX) {  // 3 south rows
            cx = q % LX; cy = q / LX;
        } else if (q < 5 * LX) {  // 2 north rows
            cx = (q - 3 * LX) % LX; cy = 3 + TY + (q - 3 * LX) / LX;
        } else {  // 3 west + 2 east columns of the core rows
            const int t = q - 5 * LX, c = t % 5;
            cy = 3 + t / 5;
            cx = c < 3 ? c : TX + c;
        }
        if (!ron[s]) { cx = 0; cy = 0; }
        rcx[s] = cx; rcy[s] = cy;
        const int gi = min(ti0 - 3 + cx, imax), gj = min(tj0 - 3 + cy, jmax);
        roff[s] = ocn::at(Lu, gi, gj, 1);
        roffv[s] = GL ? ocn::at(Lv, gi, gj, 1) : roff[s];
        roffw[s] = GL ? ocn::at(Lw, gi, gj, 1) : roff[s];
        if (PC) {
            const Lay &Lp = L0;
            rpc[s] = fz.pc_p + ocn::at(Lp, wrx(gi), wrp(gj, Ny), 1);
            // the westmost halo column of u on a slab (gi = 1 - Hx) was corrected by its owner (p[-Hx] is not here): zero gradient
            rpw[s] = (xhalo && gi - 1 < 1 - g.Hx) ? rpc[s] : fz.pc_p + ocn::at(Lp, wrx(gi - 1), wrp(gj, Ny), 1);
            rps[s] = fz.pc_p + ocn::at(Lp, wrx(gi), wrp(gj - 1, Ny), 1);
        }
    }
    auto ring_u = [&](int s, int kk) {
        const double raw = u[roff[s] + (kk - 1) * su3];
        if (!PC) return raw;
        return OCN_PC_APPLY(raw, rpc[s][zz(kk)] - rpw[s][zz(kk)], hx);
    };
    auto ring_v = [&](int s, int kk) {
        const double raw = v[roffv[s] + (kk - 1) * sv3];
        if (!PC) return raw;
        return OCN_PC_APPLY(raw, rpc[s][zz(kk)] - rps[s][zz(kk)], hy);
    };
    auto ring_w = [&](int s, int kk) {
        const double raw = w[roffw[s] + (kk - 1) * sw3];
        if (!PC) return raw;
        return OCN_PC_APPLY(raw, rpc[s][zz(kk)] - rpc[s][zz(kk - 1)], hz);
    };

    // z-windows: index m <-> k-2+m
    double zu[6], zv[6], zw[6];
    int k = k_start;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        zu[m] = ZU(k - 2 + m);
        zv[m] = ZV(k - 2 + m);
        zw[m] = ZW(k - 2 + m);
    }
    // ---- prologue: bottom-face fluxes Fwu(k), Fwv(k) need w-tile(k) and u/v[k-3..k+2]; Fww(k-1) needs w[k-3..k+2]
    double fwu_bot, fwv_bot, fww_prev;
    double nu[RPT], nv[RPT], nw[RPT];  // prefetched ring values: u(k), v(k), w(k+1)
    double pc_prev = 0.0, rp_prev[RPT] = {};  // PC: pressure of the previous plane at the own column / ring cells
    {
        sw[wslot(k)][ly][lx] = zw[2];
#pragma unroll
        for (int s = 0; s < RPT; ++s)
            if (ron[s]) sw[wslot(k)][rcy[s]][rcx[s]] = ring_w(s, k);
        if (OB) {  // plane k of u, v and plane k+1 of w as well: the loop only ever stages the NEXT plane
            su_[uvslot(k)][ly][lx] = zu[2];
            sv_[uvslot(k)][ly][lx] = zv[2];
            sw[wslot(k + 1)][ly][lx] = zw[3];
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) {
                    su_[uvslot(k)][rcy[s]][rcx[s]] = ring_u(s, k);
                    sv_[uvslot(k)][rcy[s]][rcx[s]] = ring_v(s, k);
                    sw[wslot(k + 1)][rcy[s]][rcx[s]] = ring_w(s, k + 1);
                }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RPT; ++s) {
            nu[s] = (!OB && ron[s]) ? ring_u(s, k) : 0.0;
            nv[s] = (!OB && ron[s]) ? ring_v(s, k) : 0.0;
            nw[s] = (!OB && ron[s]) ? ring_w(s, k + 1) : 0.0;
        }
        if (PC) {
            pc_prev = pC[zz(k + 3)];
#pragma unroll
            for (int s = 0; s < RPT; ++s) rp_prev[s] = ron[s] ? rpc[s][zz(k + 1)] : 0.0;
        }
        const double um3 = ZU(k - 3), vm3 = ZV(k - 3), wm3 = ZW(k - 3);
        const double(*swk)[LXP] = sw[wslot(k)];
        {   // Fwu(k): sym x-face of Az*w at plane k ; biased z-face of u
            const double a = M.Az;
            const double wt = sym_interp_scaled<P, false>([&](int m) { return swk[ly][lx + m]; }, a, i, Nx);
            const double S[6] = {um3, zu[0], zu[1], zu[2], zu[3], zu[4]};
            fwu_bot = wt * bias_interp<TZ, false>([&](int m) { return S[m + 3]; }, k, Nz, wt > 0);
        }
        {   // Fwv(k): sym y-face of Az*w ; biased z-face of v
            const double a = M.Az;
            const double wt = sym_interp_scaled<P, false>([&](int m) { return swk[ly + m][lx]; }, a, j, Ny);
            const double S[6] = {vm3, zv[0], zv[1], zv[2], zv[3], zv[4]};
            fwv_bot = wt * bias_interp<TZ, false>([&](int m) { return S[m + 3]; }, k, Nz, wt > 0);
        }
        {   // Fww(k-1): sym/biased z-centre of w at centre k-1 (line shifted to face k): w[k-3..k+2]
            const double a = M.Az;
            const double S[6] = {wm3, zw[0], zw[1], zw[2], zw[3], zw[4]};
            const double wt = sym_interp_scaled<TZ, true>([&](int m) { return S[m + 3]; }, a, k - 1, Nz);
            fww_prev = wt * bias_interp<TZ, true>([&](int m) { return S[m + 3]; }, k - 1, Nz, wt > 0);
        }
    }

    // (two planes per iteration -- the body as a forced-inline lambda called twice, since the compiler never unrolls a loop with barriers --
    // were measured in round 4: more VALU instructions and scratch accesses than two single planes)
    for (; k <= k_end; ++k) {
        double(*su)[LXP] = su_[uvslot(k)];
        double(*sv)[LXP] = sv_[uvslot(k)];
        double(*ex)[NT] = ex_[uvslot(k)];
        if (!OB) {
            // ---- stage plane k of u, v and plane k+1 of w (plane k of w is already resident) from registers
            su[ly][lx] = zu[2];
            sv[ly][lx] = zv[2];
            sw[(k + 1) & 1][ly][lx] = zw[3];
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) {
                    su[rcy[s]][rcx[s]] = nu[s];
                    sv[rcy[s]][rcx[s]] = nv[s];
                    sw[(k + 1) & 1][rcy[s]][rcx[s]] = nw[s];
                }
            __syncthreads();
        }
        // ---- prefetch what the NEXT plane needs; the loads fly under this plane's arithmetic.  (Values that are loaded under a
        // condition and used under the same one are left uninitialised on purpose: a default costs a 64-bit move per plane each.)
        double zu_n, zv_n, zw_n;
        if (k < k_end) {
            if (PC) {
                // corrected values of the next plane, re-using last plane's pressure values (6 p loads instead of 8)
                const long long o4 = zz(k + 4), o1 = zz(k + 1), o2 = zz(k + 2);
                const double pc4 = pC[o4], pw4 = pWn[o4], ps4 = pSn[o4];
                zu_n = OCN_PC_APPLY(pu[(k + 3) * su3], pc4 - pw4, hx);
                zv_n = OCN_PC_APPLY(pv[(k + 3) * sv3], pc4 - ps4, hy);
                zw_n = OCN_PC_APPLY(pw[(k + 3) * sw3], pc4 - pc_prev, hz);
                pc_prev = pc4;
#pragma unroll
                for (int s = 0; s < RPT; ++s)
                    if (ron[s]) {
                        const double pr1 = rp_prev[s], pr2 = rpc[s][o2];
                        nu[s] = OCN_PC_APPLY(u[roff[s] + k * su3], pr1 - rpw[s][o1], hx);
                        nv[s] = OCN_PC_APPLY(v[roffv[s] + k * sv3], pr1 - rps[s][o1], hy);
                        nw[s] = OCN_PC_APPLY(w[roffw[s] + (k + 1) * sw3], pr2 - pr1, hz);
                        rp_prev[s] = pr2;
                    }
            } else {
                zu_n = ZU(k + 4);
                zv_n = ZV(k + 4);
                zw_n = ZW(k + 4);
#pragma unroll
                for (int s = 0; s < RPT; ++s)
                    if (ron[s]) {
                        nu[s] = ring_u(s, k + 1);
                        nv[s] = ring_v(s, k + 1);
                        nw[s] = ring_w(s, k + 2);
                    }
            }
        }
        // the previous stage's tendencies for the substep epilogue: loaded here so that they arrive under the arithmetic
        double gmu, gmv, gmw;
        if (fz.on && fz.has_zeta && writes) {
            const long long o = own0 + (long long)(k - 1) * su3;
            const long long o_v = GL ? ownv + (long long)(k - 1) * sv3 : o, o_w = GL ? ownw + (long long)(k - 1) * sw3 : o;
            gmu = fz.Gm[0][o]; gmv = fz.Gm[1][o_v]; gmw = fz.Gm[2][o_w];
        }
        // ... and the terms the finishing pass has already left in G (fz.acc)
        double eu, ev, ew;
        if (fz.acc && writes) {
            const long long o = own0 + (long long)(k - 1) * su3;
            const long long o_v = GL ? ownv + (long long)(k - 1) * sv3 : o, o_w = GL ? ownw + (long long)(k - 1) * sw3 : o;
            if (i >= r.ou) eu = Gu[o];
            if (j >= r.ov) ev = Gv[o_v];
            if (k >= r.ow) ew = Gw[o_w];
        }
        const double(*swk)[LXP] = sw[wslot(k)];
        const double(*swt)[LXP] = sw[wslot(k + 1)];
        const double ax = M.Ax(k), ay = M.Ay(k), az = M.Az;

        // this thread's six shared fluxes stay in registers where there is room, and only the neighbours' come back from LDS (the
        // correction-on-load variants sit at their register budget: one more spill per plane costs more than the six LDS reads)
        double myf0, myf1, myf2, myf3, myf4, myf5;
#define OCN_MYF(q) (PC ? ex[q][tid] : myf##q)
        // ---- x-fluxes (consumed by this cell and its WEST neighbour)
        {   // Fuu(i-1): centre i-1 == face i of the shifted line: u[i-3..i+2]
            const double ut = sym_interp_scaled<P, true>([&](int m) { return su[ly][lx + m]; }, ax, i - 1, Nx);
#if OCN_LDS_SELECT == 2  // same field and line as the symmetric interpolation above: the six loads serve both (value selects)
            myf0 = ut * bias_interp<P, true>([&](int m) { return su[ly][lx + m]; }, i - 1, Nx, ut > 0);
#else
            myf0 = ut * bias_interp_lds(&su[ly][lx], 1, ut > 0);
#endif
            ex[0][tid] = myf0;
        }
        {   // Fuv(i): sym y-face of Ax*u ; biased x-face of v
            const double ut = sym_interp_scaled<P, false>([&](int m) { return su[ly + m][lx]; }, ax, j, Ny);
            myf1 = ut * bias_interp_lds(&sv[ly][lx], 1, ut > 0);
            ex[1][tid] = myf1;
        }
        {   // Fuw(i): sym z-face of Ax*u (own column) ; biased x-face of w
            const double ut = TZ == OCN_PERIODIC ? sym_interp_scaled<TZ, false>([&](int m) { return zu[2 + m]; }, ax, k, Nz)  // never stretched
                                                 : sym_interp<TZ, false>([&](int m) { return M.Ax(k + m) * zu[2 + m]; }, k, Nz);
            myf2 = ut * bias_interp_lds(&swk[ly][lx], 1, ut > 0);
            ex[2][tid] = myf2;
        }
        // ---- y-fluxes (consumed by this cell and its SOUTH neighbour)
        {   // Fvv(j-1)
            const double vt = sym_interp_scaled<P, true>([&](int m) { return sv[ly + m][lx]; }, ay, j - 1, Ny);
#if OCN_LDS_SELECT == 2
            myf3 = vt * bias_interp<P, true>([&](int m) { return sv[ly + m][lx]; }, j - 1, Ny, vt > 0);
#else
            myf3 = vt * bias_interp_lds(&sv[ly][lx], LXP, vt > 0);
#endif
            ex[3][tid] = myf3;
        }
        {   // Fvu(j): sym x-face of Ay*v ; biased y-face of u
            const double vt = sym_interp_scaled<P, false>([&](int m) { return sv[ly][lx + m]; }, ay, i, Nx);
            myf4 = vt * bias_interp_lds(&su[ly][lx], LXP, vt > 0);
            ex[4][tid] = myf4;
        }
        {   // Fvw(j): sym z-face of Ay*v (own column) ; biased y-face of w
            const double vt = TZ == OCN_PERIODIC ? sym_interp_scaled<TZ, false>([&](int m) { return zv[2 + m]; }, ay, k, Nz)
                                                 : sym_interp<TZ, false>([&](int m) { return M.Ay(k + m) * zv[2 + m]; }, k, Nz);
            myf5 = vt * bias_interp_lds(&swk[ly][lx], LXP, vt > 0);
            ex[5][tid] = myf5;
        }
        // ---- z-fluxes on the top face k+1 and at centre k
        double fwu_top, fwv_top, fww;
        {
            const double wt = sym_interp_scaled<P, false>([&](int m) { return swt[ly][lx + m]; }, az, i, Nx);
            fwu_top = wt * bias_interp<TZ, false>([&](int m) { return zu[3 + m]; }, k + 1, Nz, wt > 0);
        }
        {
            const double wt = sym_interp_scaled<P, false>([&](int m) { return swt[ly + m][lx]; }, az, j, Ny);
            fwv_top = wt * bias_interp<TZ, false>([&](int m) { return zv[3 + m]; }, k + 1, Nz, wt > 0);
        }
        {   // Fww(k): line shifted to face k+1: w[k-2..k+3]
            const double wt = sym_interp_scaled<TZ, true>([&](int m) { return zw[3 + m]; }, az, k, Nz);
            fww = wt * bias_interp<TZ, true>([&](int m) { return zw[3 + m]; }, k, Nz, wt > 0);
        }
        if (OB && k < k_end) {  // stage plane k+1 of u, v and plane k+2 of w: their slots were last read before the previous barrier
            su_[uvslot(k + 1)][ly][lx] = zu[3];
            sv_[uvslot(k + 1)][ly][lx] = zv[3];
            sw[wslot(k + 2)][ly][lx] = zw[4];
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) {
                    su_[uvslot(k + 1)][rcy[s]][rcx[s]] = nu[s];
                    sv_[uvslot(k + 1)][rcy[s]][rcx[s]] = nv[s];
                    sw[wslot(k + 2)][rcy[s]][rcx[s]] = nw[s];
                }
        }
        __syncthreads();
        if (writes) {
            const int e = tid + 1, n = tid + TX;
            const double rVc = recip_volume(M.Az * M.dzC(k));
            const long long ou_ = ocn::at(Lu, i, j, k), ov_ = ocn::at(Lv, i, j, k), ow_ = ocn::at(Lw, i, j, k);
            // ST: this cell's slot in the west / east strip (or -1): h + Hx * (parent row), the layout of halo_pack_x_fields_kernel
            long long sw_ = -1, se_ = -1;
            if (ST) {
                const long long row = (j + g.Hy - 1) + (long long)L0.sy * (k + g.Hz - 1);
                if (i <= g.Hx) sw_ = (i - 1) + g.Hx * row;
                if (i > Nx - g.Hx) se_ = (i - (Nx - g.Hx + 1)) + g.Hx * row;
            }
#define OCN_STRIP(f, val)                                                   \
    do {                                                                    \
        if (ST) {                                                           \
            if (sw_ >= 0) fz.strip_w[(f) * fz.strip_field + sw_] = (val);   \
            if (se_ >= 0) fz.strip_e[(f) * fz.strip_field + se_] = (val);   \
        }                                                                   \
    } while (0)
            if (i >= r.ou) {
                double G = -(rVc * (((ex[0][e] - OCN_MYF(0)) + (ex[4][n] - OCN_MYF(4))) + (fwu_top - fwu_bot)));
                if (fz.acc) G = G + eu;
                Gu[ou_] = G;
                if (fz.on) {
                    const double un = zu[2] + (fz.has_zeta ? fz.dt * (fz.gamma * G + fz.zeta * gmu) : (fz.dt * fz.gamma) * G);
                    fz.Uo[0][ou_] = un;
                    OCN_STRIP(0, un);
                }
            }
            if (j >= r.ov) {
                double G = -(rVc * (((ex[1][e] - OCN_MYF(1)) + (ex[3][n] - OCN_MYF(3))) + (fwv_top - fwv_bot)));
                if (fz.acc) G = G + ev;
                Gv[ov_] = G;
                if (fz.on) {
                    const double vn = zv[2] + (fz.has_zeta ? fz.dt * (fz.gamma * G + fz.zeta * gmv) : (fz.dt * fz.gamma) * G);
                    fz.Uo[1][ov_] = vn;
                    OCN_STRIP(1, vn);
                }
            }
            if (k >= r.ow) {
                const double rVf = recip_volume(M.Az * M.dzF(k));
                double G = -(rVf * (((ex[2][e] - OCN_MYF(2)) + (ex[5][n] - OCN_MYF(5))) + (fww - fww_prev)));
                if (fz.acc) G = G + ew;
                Gw[ow_] = G;
                // rk3_substep! always excludes the wall face (runge_kutta_3.jl:171-174), even when a KernelParameters range
                // made the tendency kernel write Gw there
                const bool wall = (TZ == OCN_BOUNDED) && k == 1 && Nz > 1;
                if (fz.on) {
                    const double wn = wall ? zw[2] : zw[2] + (fz.has_zeta ? fz.dt * (fz.gamma * G + fz.zeta * gmw) : (fz.dt * fz.gamma) * G);
                    fz.Uo[2][ow_] = wn;
                    OCN_STRIP(2, wn);
                }
            } else if (fz.on) {
                fz.Uo[2][ow_] = zw[2];  // wall face: neither the tendency nor the substep touch it (exclude_periphery)
            }
            if (fz.on && TZ == OCN_BOUNDED && k == Nz) fz.Uo[2][ow_ + sw3] = zw[3];  // top wall face k = Nz+1
        }
        fwu_bot = fwu_top; fwv_bot = fwv_top; fww_prev = fww;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            zu[m] = zu[m + 1];
            zv[m] = zv[m + 1];
            zw[m] = zw[m + 1];
        }
        if (k < k_end) { zu[5] = zu_n; zv[5] = zv_n; zw[5] = zw_n; }
    }
#undef ZU
#undef ZV
#undef ZW
#undef OCN_PC_APPLY
#undef OCN_MYF
#undef OCN_STRIP
}

// K4 tracer: flux = (A * U[i,j,k]) * cR   (upwind_biased_advective_fluxes.jl:99-121)
template <int TB>
__device__ __forceinline__ double tracer_flux(double area, double ut, const double *__restrict__ pc, long long sc, int idx,
                                              int N)
{
    const double cr = bias_interp<TB, false>([&](int m) { return pc[m * sc]; }, idx, N, ut > 0);
    return (area * ut) * cr;
}

// The rest of tracer_tendency and of the stage boundary, folded into the tracer kernels: -∇_dot_qᶜ
// (closure_kernel_operators.jl:48-53, κ a number or the field κₑ interpolated to the faces) and the bottom / top flux boundary
// contributions (apply_flux_bcs.jl:107-160).  Same operations in the same order as the separate kernels (physics.hip
// tracer_diffusion_kernel, kernels.hip apply_flux_bcs_kernel): bit-identical in the strict build.
// c0 = c[i,j,k]; cxm/cxp, cym/cyp, czm/czp its six neighbours; K[7] = κ at the cell and at the same six neighbours
// (all equal to the number κ when there is no eddy-diffusivity field).
struct Kappa7 {
    double c, xm, xp, ym, yp, zm, zp;
};
__device__ __forceinline__ Kappa7 kappa_from_global(const ocn::TracerFuse &tf, long long o, long long s2, long long s3)
{
    if (!tf.kappa_e) return Kappa7{tf.kappa, tf.kappa, tf.kappa, tf.kappa, tf.kappa, tf.kappa, tf.kappa};
    const double *pk = tf.kappa_e + o;
    return Kappa7{pk[0], pk[-1], pk[1], pk[-s2], pk[s2], pk[-s3], pk[s3]};
}
template <int TZ>
__device__ __forceinline__ double tracer_finish(double G, const Metrics &M, const GridDev &g, const ocn::TracerFuse &tf, int i, int j,
                                                int k, const Kappa7 &K, double c0, double cxm, double cxp, double cym, double cyp,
                                                double czm, double czp, double ax, double ay, double az)
{
    if (tf.diffusion) {
        const double dx = M.dx, dy = M.dy, dzc = M.dzC(k);
        const bool fld = tf.kappa_e != nullptr;
        const double k0 = K.c;
        // κ at the faces: ℑxᶠᵃᵃ / ℑyᵃᶠᵃ / ℑzᵃᵃᶠ of κₑ (abstract_scalar_diffusivity_closure.jl:298-300), or the number itself
        const double kxe = fld ? 0.5 * (k0 + K.xp) : tf.kappa, kxw = fld ? 0.5 * (K.xm + k0) : tf.kappa;
        const double kyn = fld ? 0.5 * (k0 + K.yp) : tf.kappa, kys = fld ? 0.5 * (K.ym + k0) : tf.kappa;
#if OCN_STRICT
#define OCN_TD(a, d) ((a) / (d))
#else
#define OCN_TD(a, d) ((a) * fast_rcp(d))
#endif
        const double qxe = -(kxe * OCN_TD(cxp - c0, dx)), qxw = -(kxw * OCN_TD(c0 - cxm, dx));
        const double qyn = -(kyn * OCN_TD(cyp - c0, dy)), qys = -(kys * OCN_TD(c0 - cym, dy));
        double dzq = 0.0;
        if (TZ != OCN_FLAT) {
            const double kzt = fld ? 0.5 * (k0 + K.zp) : tf.kappa, kzb = fld ? 0.5 * (K.zm + k0) : tf.kappa;
            const double qzt = -(kzt * OCN_TD(czp - c0, M.dzF(k + 1))), qzb = -(kzb * OCN_TD(c0 - czm, M.dzF(k)));
            dzq = az * qzt - az * qzb;
        }
#undef OCN_TD
        G = G - recip_volume(az * dzc) * (((ax * qxe - ax * qxw) + (ay * qyn - ay * qys)) + dzq);
    }
    if (TZ == OCN_BOUNDED) {  // apply_z_bcs!: k is uniform across the workgroup
        if (k == 1 && tf.bottom.kind == OCN_BC_FLUX) G += ocn::bc_condition(tf.bottom, i, j, g.Nx, c0) * az / (az * M.dzC(1));
        if (k == g.Nz && tf.top.kind == OCN_BC_FLUX) G -= ocn::bc_condition(tf.top, i, j, g.Nx, c0) * az / (az * M.dzC(g.Nz));
    }
    return G;
}

// Direct kernel: one thread per cell, all six face fluxes evaluated in place.  `tf` (see tracer_finish) additionally carries
// the NEXT stage's rk3 substep into a second storage.
template <int TZ>
__global__ __launch_bounds__(256) void tracer_tendency_direct(GridDev g, const double *__restrict__ u,
                                                              const double *__restrict__ v, const double *__restrict__ w,
                                                              const double *__restrict__ c, double *__restrict__ Gc, Range r,
                                                              ocn::TracerFuse tf)
{
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1) return;
    constexpr int P = OCN_PERIODIC;
    const Metrics M = make_metrics(g);
    const Lay Lu = ocn::make_lay(g, OCN_LOC_FCC), Lv = ocn::make_lay(g, OCN_LOC_CFC), Lw = ocn::make_lay(g, OCN_LOC_CCF),
              Lc = ocn::make_lay(g, OCN_LOC_CCC);
    const double *pu = u + ocn::at(Lu, i, j, k);
    const double *pv = v + ocn::at(Lv, i, j, k);
    const double *pw = w + ocn::at(Lw, i, j, k);
    const double *pc = c + ocn::at(Lc, i, j, k);
    const double ax = M.Ax(k), ay = M.Ay(k), az = M.Az;
    const double fx1 = tracer_flux<P>(ax, pu[1], pc + 1, 1, i + 1, g.Nx), fx0 = tracer_flux<P>(ax, pu[0], pc, 1, i, g.Nx);
    const double fy1 = tracer_flux<P>(ay, pv[Lv.s2], pc + Lc.s2, Lc.s2, j + 1, g.Ny), fy0 = tracer_flux<P>(ay, pv[0], pc, Lc.s2, j, g.Ny);
    double dzF = 0.0;
    if (TZ != OCN_FLAT) {
        const double fz1 = tracer_flux<TZ>(az, pw[Lw.s3], pc + Lc.s3, Lc.s3, k + 1, g.Nz), fz0 = tracer_flux<TZ>(az, pw[0], pc, Lc.s3, k, g.Nz);
        dzF = fz1 - fz0;
    }
    const double rV = recip_volume(M.Az * M.dzC(k));
    double G = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
    const long long o = ocn::at(Lc, i, j, k), s2 = Lc.s2, s3 = (TZ == OCN_FLAT) ? 0 : Lc.s3;
    if (tf.diffusion || tf.bottom.kind || tf.top.kind)
        G = tracer_finish<TZ>(G, M, g, tf, i, j, k, kappa_from_global(tf, o, s2, s3), pc[0], pc[-1], pc[1], pc[-s2], pc[s2], pc[-s3],
                              pc[s3], ax, ay, az);
    Gc[o] = G;
    if (tf.sc.on) tf.sub.out[o] = pc[0] + (tf.sc.has_zeta ? tf.sc.dt * (tf.sc.gamma * G + tf.sc.zeta * tf.sub.Gm[o]) : (tf.sc.dt * tf.sc.gamma) * G);
}

// Tiled tracer kernel: every face flux is evaluated ONCE and shared through LDS (the direct kernel evaluates each twice), with
// the structure of momentum_tendencies_tiled: a workgroup owns a TX x TY patch of columns, overlapping its east / north
// neighbours by one so that (TX-1) x (TY-1) cells are written, and marches KZ planes upward;
//  * plane k of c lives in LDS with a 3 / 2-cell ring for the x and y stencils; the z stencil c[k-2..k+3] of the own column
//    is a register window, so each value enters the workgroup once per plane;
//  * thread (i, j) evaluates the west-face and south-face fluxes of its cell and the top-face flux; the east / north ones come
//    from the neighbouring threads through LDS, the bottom one is last iteration's top flux.
// 3 flux evaluations per thread and plane instead of 6; the flux expressions are those of the direct kernel (bit-identical).
template <int TZ, int TX, int TY, int W = 1, bool GL = false>
__global__ __launch_bounds__(TX *TY, W) void tracer_tendency_tiled(GridDev g, const double *__restrict__ u, const double *__restrict__ v,
                                                               const double *__restrict__ w, const double *__restrict__ c,
                                                               double *__restrict__ Gc, Range r, int KZ, ocn::TracerFuse tf)
{
    constexpr int P = OCN_PERIODIC;
    constexpr int LX = TX + 5, LY = TY + 5, NT = TX * TY;
    constexpr int NRING = LX * LY - NT;
    constexpr int RPT = (NRING + NT - 1) / NT;
    static_assert(RPT <= 2, "tile too small for its ring");
    __shared__ double sc[LY][LX];
    __shared__ double sk[LY][LX];  // plane k of the eddy diffusivity κₑ (same ring geometry), when there is one
    __shared__ double ex[2][NT];  // west-face flux (read by the west neighbour as its east flux), south-face flux

    Metrics M = make_metrics(g);
    if (TZ == OCN_PERIODIC) M.dzc = M.dzf = nullptr;
    const Lay L0 = ocn::make_lay(g, OCN_LOC_CCC);  // x, y Periodic: one layout for u, v, w, c, G
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    int bx, by, bz;
    block_coords(r, bx, by, bz);
    const int ti0 = r.i0 + bx * (TX - 1), tj0 = r.j0 + by * (TY - 1);
    const int k_start = r.k0 + bz * KZ, k_end = min(k_start + KZ - 1, r.k1);
    const int imax = Nx + g.Hx, jmax = Ny + g.Hy;
    const int i = min(ti0 + tx, imax), j = min(tj0 + ty, jmax);
    const bool writes = (tx < TX - 1) && (ty < TY - 1) && (ti0 + tx <= r.i1) && (tj0 + ty <= r.j1);
    const int lx = tx + 3, ly = ty + 3;
    const long long s3 = L0.s3;
    const long long own0 = ocn::at(L0, i, j, 1);
    // GL (see momentum_tendencies_tiled): the velocities have their own parent layouts on grids with a Bounded x / y
    const Lay Lu = GL ? ocn::make_lay(g, OCN_LOC_FCC) : L0, Lv = GL ? ocn::make_lay(g, OCN_LOC_CFC) : L0, Lw = GL ? ocn::make_lay(g, OCN_LOC_CCF) : L0;
    const long long su3 = GL ? Lu.s3 : s3, sv3 = GL ? Lv.s3 : s3, sw3 = GL ? Lw.s3 : s3;
    const double *pc = c + own0, *pu = u + (GL ? ocn::at(Lu, i, j, 1) : own0), *pv = v + (GL ? ocn::at(Lv, i, j, 1) : own0),
                 *pw = w + (GL ? ocn::at(Lw, i, j, 1) : own0);

    // static ring assignment (as in momentum_tendencies_tiled)
    int rcx[RPT], rcy[RPT];
    bool ron[RPT];
    long long roff[RPT];
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
        const int q = tid + s * NT;
        ron[s] = q < NRING;
        int cx, cy;
        if (q < 3 * LX) {
            cx = q % LX; cy = q / LX;
        } else if (q < 5 * LX) {
            cx = (q - 3 * LX) % LX; cy = 3 + TY + (q - 3 * LX) / LX;
        } else {
            const int t = q - 5 * LX, cc = t % 5;
            cy = 3 + t / 5;
            cx = cc < 3 ? cc : TX + cc;
        }
        if (!ron[s]) { cx = 0; cy = 0; }
        rcx[s] = cx; rcy[s] = cy;
        roff[s] = ocn::at(L0, min(ti0 - 3 + cx, imax), min(tj0 - 3 + cy, jmax), 1);
    }

    // z window: zc[m] <-> c[k-2+m]
    double zc[6];
    int k = k_start;
#pragma unroll
    for (int m = 0; m < 6; ++m) zc[m] = pc[(long long)(k - 3 + m) * s3];
    const double az = M.Az;
    double fzb;
    {   // bottom-face flux of plane k_start: stencil c[k-3 .. k+2]
        const double cm3 = pc[(long long)(k - 4) * s3];
        const double wf = pw[(long long)(k - 1) * sw3];
        const double S[6] = {cm3, zc[0], zc[1], zc[2], zc[3], zc[4]};
        fzb = (az * wf) * bias_interp<TZ, false>([&](int m) { return S[m + 3]; }, k, Nz, wf > 0);
    }
    // software pipeline: the global values of plane k+1 (ring cells of c, u, v at the own faces, w at the top face, the new
    // window entry) are requested one iteration ahead so that their latency hides behind the flux arithmetic of plane k
    const bool kfld = tf.diffusion && tf.kappa_e != nullptr;
    const double *pke = kfld ? tf.kappa_e + own0 : nullptr;
    double zk[3] = {0.0, 0.0, 0.0}, rk[RPT] = {};  // κₑ[k-1..k+1] of the own column; prefetched ring values of κₑ
    if (kfld) {
        zk[0] = pke[(long long)(k - 2) * s3];
        zk[1] = pke[(long long)(k - 1) * s3];
        zk[2] = pke[(long long)k * s3];
#pragma unroll
        for (int s = 0; s < RPT; ++s) rk[s] = ron[s] ? tf.kappa_e[roff[s] + (long long)(k - 1) * s3] : 0.0;
    }
    double rv[RPT], uf, vf, wf, znew;
#pragma unroll
    for (int s = 0; s < RPT; ++s) rv[s] = ron[s] ? c[roff[s] + (long long)(k - 1) * s3] : 0.0;
    uf = pu[(long long)(k - 1) * su3];
    vf = pv[(long long)(k - 1) * sv3];
    wf = pw[(long long)k * sw3];
    znew = (k < k_end) ? pc[(long long)(k + 3) * s3] : 0.0;
    for (; k <= k_end; ++k) {
        // stage plane k (the previous iteration's readers of sc are past its second barrier)
        sc[ly][lx] = zc[2];
#pragma unroll
        for (int s = 0; s < RPT; ++s)
            if (ron[s]) sc[rcy[s]][rcx[s]] = rv[s];
        if (kfld) {
            sk[ly][lx] = zk[1];
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) sk[rcy[s]][rcx[s]] = rk[s];
        }
        __syncthreads();
        const bool more = k < k_end;
        // (values loaded under a condition and used under the same one carry no default: a default is a 64-bit move or select per plane)
        double rk_n[RPT], zk_n, rv_n[RPT], uf_n, vf_n, wf_n, znew_n, gm;
        const long long o = own0 + (long long)(k - 1) * s3;
        if (more) {
            if (kfld) {
#pragma unroll
                for (int s = 0; s < RPT; ++s)
                    if (ron[s]) rk_n[s] = tf.kappa_e[roff[s] + (long long)k * s3];
                zk_n = pke[(long long)(k + 1) * s3];  // κₑ[k+2]
            }
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) rv_n[s] = c[roff[s] + (long long)k * s3];
            uf_n = pu[(long long)k * su3];
            vf_n = pv[(long long)k * sv3];
            wf_n = pw[(long long)(k + 1) * sw3];
            if (k + 1 < k_end) znew_n = pc[(long long)(k + 4) * s3];
        }
        if (writes && tf.sc.on && tf.sc.has_zeta) gm = tf.sub.Gm[o];
        // Everything above must be REQUESTED here: left alone, the scheduler sinks these loads to their first use -- behind the second
        // barrier, a few hundred cycles before the next iteration waits for them -- and, vector-memory returns being in order, a late G⁻
        // load in the epilogue then waits for the whole prefetch group (profiles/r03a_config4.md: 45 % of the wave cycles parked).
        OCN_ISSUE_LOADS_HERE();
        const double ax = M.Ax(k), ay = M.Ay(k);
#if OCN_TRACER_LDS_SELECT  // address-selected stencils as in the momentum kernel (A/B switch)
        const double fxw = (ax * uf) * bias_interp_lds(&sc[ly][lx], 1, uf > 0);
        const double fys = (ay * vf) * bias_interp_lds(&sc[ly][lx], LX, vf > 0);
#else
        const double fxw = (ax * uf) * bias_interp<P, false>([&](int m) { return sc[ly][lx + m]; }, i, Nx, uf > 0);
        const double fys = (ay * vf) * bias_interp<P, false>([&](int m) { return sc[ly + m][lx]; }, j, Ny, vf > 0);
#endif
        const double fzt = (az * wf) * bias_interp<TZ, false>([&](int m) { return zc[m + 3]; }, k + 1, Nz, wf > 0);
        const double cxm = sc[ly][lx - 1], cxp = sc[ly][lx + 1], cym = sc[ly - 1][lx], cyp = sc[ly + 1][lx];
        Kappa7 K{tf.kappa, tf.kappa, tf.kappa, tf.kappa, tf.kappa, tf.kappa, tf.kappa};
        if (kfld) K = Kappa7{zk[1], sk[ly][lx - 1], sk[ly][lx + 1], sk[ly - 1][lx], sk[ly + 1][lx], zk[0], zk[2]};
        ex[0][tid] = fxw;
        ex[1][tid] = fys;
        __syncthreads();
        if (writes) {
            const double fxe = ex[0][tid + 1], fyn = ex[1][tid + TX];
            const double rV = recip_volume(M.Az * M.dzC(k));
            double G = -(rV * (((fxe - fxw) + (fyn - fys)) + (fzt - fzb)));
            if (tf.diffusion || tf.bottom.kind || tf.top.kind)
                G = tracer_finish<TZ>(G, M, g, tf, i, j, k, K, zc[2], cxm, cxp, cym, cyp, zc[1], zc[3], ax, ay, az);
            Gc[o] = G;
            if (tf.sc.on) tf.sub.out[o] = zc[2] + (tf.sc.has_zeta ? tf.sc.dt * (tf.sc.gamma * G + tf.sc.zeta * gm) : (tf.sc.dt * tf.sc.gamma) * G);
        }
        fzb = fzt;
#pragma unroll
        for (int m = 0; m < 5; ++m) zc[m] = zc[m + 1];
        if (more) {
            zc[5] = znew;
            if (k + 1 < k_end) znew = znew_n;
            uf = uf_n; vf = vf_n; wf = wf_n;
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) rv[s] = rv_n[s];
            if (kfld) {
                zk[0] = zk[1]; zk[1] = zk[2]; zk[2] = zk_n;
#pragma unroll
                for (int s = 0; s < RPT; ++s)
                    if (ron[s]) rk[s] = rk_n[s];
            }
        }
    }
}

// Two tracers in ONE launch (T and S of the ocean configurations): the tiled kernel above with every tracer-specific quantity doubled --
// c plane + ring in LDS, z window, shared-flux exchange, κₑ plane, G⁻, substep output -- while u, v, w at the faces, the upwind
// directions, the metrics, the barriers and all index arithmetic are shared.  The two single-tracer launches read u, v, w twice
// (24 of their 56 algorithmic bytes per cell each) and are bound by HBM traffic + latency rather than VALU issue
// (profiles/r02a_config4.md: 4.6 TB/s, VALU busy 0.55); per tracer the arithmetic is the text of tracer_tendency_tiled, so results
// are bit-identical to two separate launches.
struct TracerPair {
    const double *c[2];
    double *G[2];
    ocn::TracerFuse tf[2];
};
template <int TZ, int TX, int TY>
__global__ __launch_bounds__(TX *TY) void tracer_pair_tendency_tiled(GridDev g, const double *__restrict__ u, const double *__restrict__ v,
                                                                    const double *__restrict__ w, TracerPair tp, Range r, int KZ)
{
    constexpr int P = OCN_PERIODIC, NTR = 2;
    constexpr int LX = TX + 5, LY = TY + 5, NT = TX * TY;
    constexpr int NRING = LX * LY - NT;
    constexpr int RPT = (NRING + NT - 1) / NT;
    static_assert(RPT <= 2, "tile too small for its ring");
    __shared__ double sc[NTR][LY][LX];
    __shared__ double sk[NTR][LY][LX];
    __shared__ double ex[2 * NTR][NT];

    Metrics M = make_metrics(g);
    if (TZ == OCN_PERIODIC) M.dzc = M.dzf = nullptr;
    const Lay L0 = ocn::make_lay(g, OCN_LOC_CCC);
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    int bx, by, bz;
    block_coords(r, bx, by, bz);
    const int ti0 = r.i0 + bx * (TX - 1), tj0 = r.j0 + by * (TY - 1);
    const int k_start = r.k0 + bz * KZ, k_end = min(k_start + KZ - 1, r.k1);
    const int imax = Nx + g.Hx, jmax = Ny + g.Hy;
    const int i = min(ti0 + tx, imax), j = min(tj0 + ty, jmax);
    const bool writes = (tx < TX - 1) && (ty < TY - 1) && (ti0 + tx <= r.i1) && (tj0 + ty <= r.j1);
    const int lx = tx + 3, ly = ty + 3;
    const long long s3 = L0.s3;
    const long long own0 = ocn::at(L0, i, j, 1);
    const double *pu = u + own0, *pv = v + own0, *pw = w + own0;
    const double *pc[NTR] = {tp.c[0] + own0, tp.c[1] + own0};

    int rcx[RPT], rcy[RPT];
    bool ron[RPT];
    long long roff[RPT];
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
        const int q = tid + s * NT;
        ron[s] = q < NRING;
        int cx, cy;
        if (q < 3 * LX) {
            cx = q % LX; cy = q / LX;
        } else if (q < 5 * LX) {
            cx = (q - 3 * LX) % LX; cy = 3 + TY + (q - 3 * LX) / LX;
        } else {
            const int t = q - 5 * LX, cc = t % 5;
            cy = 3 + t / 5;
            cx = cc < 3 ? cc : TX + cc;
        }
        if (!ron[s]) { cx = 0; cy = 0; }
        rcx[s] = cx; rcy[s] = cy;
        roff[s] = ocn::at(L0, min(ti0 - 3 + cx, imax), min(tj0 - 3 + cy, jmax), 1);
    }

    double zc[NTR][6], fzb[NTR];
    int k = k_start;
    const double az = M.Az;
    double uf, vf, wf;
    uf = pu[(long long)(k - 1) * s3];
    vf = pv[(long long)(k - 1) * s3];
    {
        const double wb = pw[(long long)(k - 1) * s3];  // bottom-face flux of plane k_start: stencil c[k-3 .. k+2]
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
#pragma unroll
            for (int m = 0; m < 6; ++m) zc[t][m] = pc[t][(long long)(k - 3 + m) * s3];
            const double cm3 = pc[t][(long long)(k - 4) * s3];
            const double S[6] = {cm3, zc[t][0], zc[t][1], zc[t][2], zc[t][3], zc[t][4]};
            fzb[t] = (az * wb) * bias_interp<TZ, false>([&](int m) { return S[m + 3]; }, k, Nz, wb > 0);
        }
    }
    wf = pw[(long long)k * s3];
    bool kfld[NTR];
    const double *pke[NTR];
    double zk[NTR][3], rk[NTR][RPT], rv[NTR][RPT], znew[NTR];
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
        kfld[t] = tp.tf[t].diffusion && tp.tf[t].kappa_e != nullptr;
        pke[t] = kfld[t] ? tp.tf[t].kappa_e + own0 : nullptr;
#pragma unroll
        for (int m = 0; m < 3; ++m) zk[t][m] = kfld[t] ? pke[t][(long long)(k - 2 + m) * s3] : 0.0;
#pragma unroll
        for (int s = 0; s < RPT; ++s) {
            rk[t][s] = (kfld[t] && ron[s]) ? tp.tf[t].kappa_e[roff[s] + (long long)(k - 1) * s3] : 0.0;
            rv[t][s] = ron[s] ? tp.c[t][roff[s] + (long long)(k - 1) * s3] : 0.0;
        }
        znew[t] = (k < k_end) ? pc[t][(long long)(k + 3) * s3] : 0.0;
    }
    for (; k <= k_end; ++k) {
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            sc[t][ly][lx] = zc[t][2];
#pragma unroll
            for (int s = 0; s < RPT; ++s)
                if (ron[s]) sc[t][rcy[s]][rcx[s]] = rv[t][s];
            if (kfld[t]) {
                sk[t][ly][lx] = zk[t][1];
#pragma unroll
                for (int s = 0; s < RPT; ++s)
                    if (ron[s]) sk[t][rcy[s]][rcx[s]] = rk[t][s];
            }
        }
        __syncthreads();
        const bool more = k < k_end;
        const long long o = own0 + (long long)(k - 1) * s3;
        double rk_n[NTR][RPT], zk_n[NTR], rv_n[NTR][RPT], znew_n[NTR], gm[NTR];
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                rk_n[t][s] = (kfld[t] && more && ron[s]) ? tp.tf[t].kappa_e[roff[s] + (long long)k * s3] : 0.0;
                rv_n[t][s] = (more && ron[s]) ? tp.c[t][roff[s] + (long long)k * s3] : 0.0;
            }
            zk_n[t] = (kfld[t] && more) ? pke[t][(long long)(k + 1) * s3] : 0.0;
            znew_n[t] = (k + 1 < k_end) ? pc[t][(long long)(k + 4) * s3] : 0.0;
            gm[t] = (writes && tp.tf[t].sc.on && tp.tf[t].sc.has_zeta) ? tp.tf[t].sub.Gm[o] : 0.0;
        }
        const double uf_n = more ? pu[(long long)k * s3] : 0.0, vf_n = more ? pv[(long long)k * s3] : 0.0;
        const double wf_n = more ? pw[(long long)(k + 1) * s3] : 0.0;
        const double ax = M.Ax(k), ay = M.Ay(k);
        const bool lu = uf > 0, lv = vf > 0, lw = wf > 0;
        double fxw[NTR], fys[NTR], fzt[NTR];
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            fxw[t] = (ax * uf) * bias_interp<P, false>([&](int m) { return sc[t][ly][lx + m]; }, i, Nx, lu);
            fys[t] = (ay * vf) * bias_interp<P, false>([&](int m) { return sc[t][ly + m][lx]; }, j, Ny, lv);
            fzt[t] = (az * wf) * bias_interp<TZ, false>([&](int m) { return zc[t][m + 3]; }, k + 1, Nz, lw);
            ex[2 * t][tid] = fxw[t];
            ex[2 * t + 1][tid] = fys[t];
        }
        double cxm[NTR], cxp[NTR], cym[NTR], cyp[NTR];
        Kappa7 K[NTR];
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            cxm[t] = sc[t][ly][lx - 1]; cxp[t] = sc[t][ly][lx + 1]; cym[t] = sc[t][ly - 1][lx]; cyp[t] = sc[t][ly + 1][lx];
            const double kp = tp.tf[t].kappa;
            K[t] = Kappa7{kp, kp, kp, kp, kp, kp, kp};
            if (kfld[t]) K[t] = Kappa7{zk[t][1], sk[t][ly][lx - 1], sk[t][ly][lx + 1], sk[t][ly - 1][lx], sk[t][ly + 1][lx], zk[t][0], zk[t][2]};
        }
        __syncthreads();
        if (writes) {
            const double rV = recip_volume(M.Az * M.dzC(k));
#pragma unroll
            for (int t = 0; t < NTR; ++t) {
                const ocn::TracerFuse &tf = tp.tf[t];
                const double fxe = ex[2 * t][tid + 1], fyn = ex[2 * t + 1][tid + TX];
                double G = -(rV * (((fxe - fxw[t]) + (fyn - fys[t])) + (fzt[t] - fzb[t])));
                if (tf.diffusion || tf.bottom.kind || tf.top.kind)
                    G = tracer_finish<TZ>(G, M, g, tf, i, j, k, K[t], zc[t][2], cxm[t], cxp[t], cym[t], cyp[t], zc[t][1], zc[t][3], ax, ay, az);
                tp.G[t][o] = G;
                if (tf.sc.on) tf.sub.out[o] = zc[t][2] + (tf.sc.has_zeta ? tf.sc.dt * (tf.sc.gamma * G + tf.sc.zeta * gm[t]) : (tf.sc.dt * tf.sc.gamma) * G);
            }
        }
        uf = uf_n; vf = vf_n; wf = wf_n;
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            fzb[t] = fzt[t];
#pragma unroll
            for (int m = 0; m < 5; ++m) zc[t][m] = zc[t][m + 1];
            zc[t][5] = znew[t];
            znew[t] = znew_n[t];
#pragma unroll
            for (int s = 0; s < RPT; ++s) {
                rv[t][s] = rv_n[t][s];
                rk[t][s] = rk_n[t][s];
            }
            zk[t][0] = zk[t][1]; zk[t][1] = zk[t][2]; zk[t][2] = zk_n[t];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// x-halo payload of the correction-on-load stage of a slab-x rank (distributed runs; momentum_tendencies_tiled with fz.pc_xhalo).
// After the pressure solve a rank needs, besides its own p[1..nx], the neighbours' pressure planes in p's x halos, and -- because
// the correction of u at the westmost halo column i = 1 - Hx would read p[-Hx], which no array holds -- that ONE u plane already
// corrected by its owner:   u[1-Hx] <- west neighbour's  u[nx-Hx+1] - ((p[nx-Hx+1] - p[nx-Hx]) / dx) dt,
// the expression of _pressure_correct_velocities! (pressure_correction.jl:31-37) in this build's arithmetic (same macro as the
// on-load correction), evaluated at every row of the cross-section with periodically wrapped (j, k): a halo row of u* is the copy of
// an interior row, so its corrected value is that row's.
//   buffers: (Hx + 1) values per row;  pack:  west[h] = p[1 + h],  east[h] = p[nx - Hx + 1 + h]  (h < Hx),  east[Hx] = corrected u
//   unpack:  p[1 - Hx + h] = west[h], u[1 - Hx] = west[Hx]  (from the west neighbour's `east`);  p[nx + 1 + h] = east[h]
__global__ void pressure_planes_kernel(GridDev g, double *__restrict__ p, double *__restrict__ u, double pcdt, double *__restrict__ west,
                                       double *__restrict__ east, int unpack)
{
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const int Hx = g.Hx, nx = g.Nx, W = Hx + 1;
    const long long rows = (long long)L.sy * L.sz;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * W) return;
    const int h = t % W;
    const long long row = t / W;
    double *prow = p + row * L.sx, *urow = u + row * L.sx;
    if (unpack) {
        if (h < Hx) {
            prow[h] = west[t];
            prow[nx + Hx + h] = east[t];
        } else {
            urow[0] = west[t];
        }
        return;
    }
    if (h < Hx) {
        west[t] = prow[Hx + h];
        east[t] = prow[nx + h];
        return;
    }
    // the corrected u plane, p taken at the periodically wrapped interior row
    const int jp = row % L.sy, kp = row / L.sy;
    int j = jp - g.Hy + 1, k = kp - g.Hz + 1;
    j = j < 1 ? j + g.Ny : (j > g.Ny ? j - g.Ny : j);
    k = k < 1 ? k + g.Nz : (k > g.Nz ? k - g.Nz : k);
    const double *pw = p + ocn::at(L, nx - Hx + 1, j, k);
    west[t] = 0.0;
#if OCN_STRICT
    east[t] = urow[nx] - ((pw[0] - pw[-1]) / g.dx) * pcdt;
#else
    east[t] = __builtin_fma(-(pw[0] - pw[-1]), pcdt * (1.0 / g.dx), urow[nx]);  // the expression of the correction on load
#endif
}
int launch_pressure_planes(const ocn_grid *grid, double *p, double *u, double dt, double *west, double *east, int unpack, hipStream_t stream)
{
    GridDev g = ocn::to_dev(*grid);
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long n = (long long)L.sy * L.sz * (g.Hx + 1);
    hipLaunchKernelGGL(pressure_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, p, u, dt, west, east, unpack);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

static int tile_variant()
{
    const char *e = getenv("OCN_TILE");
    return e ? atoi(e) : 0;
}

// z-chunking target: the launch should have at least this many workgroups (768 run concurrently: 256 CUs x 3)
static int min_blocks()
{
    static const int v = getenv("OCN_TEND_MIN_BLOCKS") ? atoi(getenv("OCN_TEND_MIN_BLOCKS")) : 8192;
    return v;
}

// the 3-wide buffer strips of a distributed run have few columns: shorter z-chunks put more workgroups in flight (they are latency
// bound at one wave per SIMD otherwise)
static int strip_min_kz()
{
    static const int v = getenv("OCN_STRIP_MIN_KZ") ? atoi(getenv("OCN_STRIP_MIN_KZ")) : 8;
    return v;
}

// One barrier per plane (template parameter OB of momentum_tendencies_tiled): bit 0 = the correction-on-load variant, bit 1 = the plain
// Periodic-z one, bit 2 = Bounded z.  Measured on one box, 512^3 step: 27.21 ms without, 27.06 with bit 0, 27.33 with bit 1, 27.09 with
// both; config 4 (bit 2) +0.45 ms.  The variant that waits for the most loads per plane is the one that gains from the missing barrier.
static int one_barrier()
{
    static const int v = getenv("OCN_TEND_ONE_BARRIER") ? atoi(getenv("OCN_TEND_ONE_BARRIER")) : 1;
    return v;
}

static int xcd_remap()
{
    static const int v = getenv("OCN_XCD_REMAP") ? atoi(getenv("OCN_XCD_REMAP")) : 1;  // measured: 4.72 -> 4.60 ms per 512^3 launch
    return v;
}

// lanes a tiling spends per owned column: (TX x TY threads per patch) x patches / columns
static double lanes_per_column(int TX, int TY, int wx, int wy)
{
    const double patches = (double)((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
    return patches * (double)(((TX * TY + 63) / 64) * 64) / ((double)wx * wy);
}
// 17 x 15 instead of 32 x 8 patches when they waste clearly fewer lanes (their 136-byte rows coalesce worse: measured 2.5 % slower
// than 32 x 8 at equal useful-lane count with 16 x 16 patches).  OCN_NARROW_TILE = 0 / 1 forces the choice.
static bool narrow_tile(int wx, int wy)
{
    static const int force = getenv("OCN_NARROW_TILE") ? atoi(getenv("OCN_NARROW_TILE")) : -1;
    if (wx < 16 || wy < 14) return false;
    if (force >= 0) return force != 0;
    return lanes_per_column(17, 15, wx, wy) * 1.08 < lanes_per_column(32, 8, wx, wy);
}

template <int TZ, int TX, int TY, bool PC, bool OB, bool ST = false>
static void launch_tiled(const GridDev &g, const double *u, const double *v, const double *w, double *Gu, double *Gv, double *Gw, const Range &r,
                         const ocn::FuseArgs &fz, int wx, int wy, int wz, hipStream_t stream)
{
    const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
    int KZ = wz;  // z-chunk: enough workgroups to fill the chip, long enough to amortise the 3-flux prologue
    while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < min_blocks()) KZ = (KZ + 1) / 2;
    dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);
    hipLaunchKernelGGL((momentum_tendencies_tiled<TZ, TX, TY, 3, PC, OB, false, ST>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, Gu, Gv, Gw, r, KZ, fz);
}

static int make_range(const ocn_grid *grid, const int32_t *range, Range &r)
{
    r.xcd = xcd_remap();
    if (range) {
        r.i0 = range[0]; r.i1 = range[1]; r.j0 = range[2]; r.j1 = range[3]; r.k0 = range[4]; r.k1 = range[5];
        if (r.i0 < 1 || r.i1 > grid->Nx || r.j0 < 1 || r.j1 > grid->Ny || r.k0 < 1 || r.k1 > grid->Nz) {
            ocn::set_error("tendency range {%d:%d,%d:%d,%d:%d} outside the interior %dx%dx%d", r.i0, r.i1, r.j0, r.j1, r.k0,
                           r.k1, grid->Nx, grid->Ny, grid->Nz);
            return OCN_ERR_INVALID_ARGUMENT;
        }
        r.ou = r.ov = r.ow = 1;  // KernelParameters: periphery not excluded (kernel_launching.jl:236-240)
    } else {
        r.i0 = 1; r.i1 = grid->Nx; r.j0 = 1; r.j1 = grid->Ny; r.k0 = 1; r.k1 = grid->Nz;
        r.ou = r.ov = 1;  // x, y are never Bounded in the supported scope
        r.ow = (grid->tz == OCN_BOUNDED && grid->Nz > 1) ? 2 : 1;  // periphery_offset(Face, Bounded, N) (:113-114)
    }
    return OCN_SUCCESS;
}

// The interior box of a grid with walls in x / y (csrc/general.hip launch_*_general): cells i0..i1 x j0..j1 whose x and y reconstructions
// are all the full-order ones (topologically_conditional_interpolation.jl:46-52: faces 4 .. N-2 and centres 3 .. N-2 of a Bounded
// direction), so the LDS-tiled kernels apply with per-field parent layouts (GL); z stays topology-conditional inside the kernel.
// Returns 0 in *launched when the box is too small for the tiles (the caller then covers everything with the per-cell kernel).
int launch_momentum_tendencies_box(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu, double *Gv, double *Gw,
                                   const int32_t box[4], int *launched, hipStream_t stream, const ocn::FuseArgs *fuse, int ranged)
{
    *launched = 0;
    Range r;
    r.xcd = xcd_remap();
    r.i0 = box[0]; r.i1 = box[1]; r.j0 = box[2]; r.j1 = box[3]; r.k0 = 1; r.k1 = grid->Nz;
    r.ou = r.ov = 1;
    r.ow = (!ranged && grid->tz == OCN_BOUNDED && grid->Nz > 1) ? 2 : 1;  // (KernelParameters: periphery not excluded)
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = grid->Nz;
    if (grid->tz == OCN_FLAT || wx < 16 || wy < 8 || wz < 4) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    ocn::FuseArgs fz{};
    if (fuse) fz = *fuse;  // (the next substep as the epilogue: per-field offsets like the G stores)
    constexpr int TX = 32, TY = 8;
    const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
    int KZ = wz;
    while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < min_blocks()) KZ = (KZ + 1) / 2;
    dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);
    if (grid->tz == OCN_PERIODIC)
        hipLaunchKernelGGL((momentum_tendencies_tiled<OCN_PERIODIC, TX, TY, 3, false, false, true>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, Gu, Gv, Gw, r, KZ, fz);
    else
        hipLaunchKernelGGL((momentum_tendencies_tiled<OCN_BOUNDED, TX, TY, 3, false, false, true>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, Gu, Gv, Gw, r, KZ, fz);
    OCN_CHECK_HIP(hipGetLastError());
    *launched = 1;
    return OCN_SUCCESS;
}

int launch_tracer_tendency_box(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c, double *Gc,
                               const int32_t box[4], int *launched, hipStream_t stream, const ocn::TracerFuse *fuse, int ranged)
{
    (void)ranged;  // (a centre field has no excluded periphery)
    *launched = 0;
    Range r;
    r.xcd = xcd_remap();
    r.i0 = box[0]; r.i1 = box[1]; r.j0 = box[2]; r.j1 = box[3]; r.k0 = 1; r.k1 = grid->Nz;
    r.ou = r.ov = r.ow = 1;
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = grid->Nz;
    if (grid->tz == OCN_FLAT || wx < 16 || wy < 8 || wz < 4) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    ocn::TracerFuse tf{};
    if (fuse) tf = *fuse;  // (diffusion, bottom / top fluxes, the next substep: all on centre fields, whose layout has no walls in it)
    constexpr int TX = 32, TY = 8;
    const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
    int KZ = wz;
    while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < min_blocks()) KZ = (KZ + 1) / 2;
    dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);
    if (grid->tz == OCN_PERIODIC)
        hipLaunchKernelGGL((tracer_tendency_tiled<OCN_PERIODIC, TX, TY, 1, true>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
    else
        hipLaunchKernelGGL((tracer_tendency_tiled<OCN_BOUNDED, TX, TY, 1, true>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
    OCN_CHECK_HIP(hipGetLastError());
    *launched = 1;
    return OCN_SUCCESS;
}

int launch_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                               double *Gv, double *Gw, const int32_t *range, const ocn::FuseArgs *fuse, hipStream_t stream)
{
    ocn::FuseArgs fz{};
    if (fuse) fz = *fuse;
    Range r;
    int st = make_range(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = r.k1 - r.k0 + 1;
    static const int force_direct = (getenv("OCN_TENDENCY_KERNEL") && !strcmp(getenv("OCN_TENDENCY_KERNEL"), "direct"));
    if (!force_direct && grid->tz != OCN_FLAT && wx >= 16 && wy >= 8 && wz >= 4) {
        // tile variants (TX, TY, min waves/SIMD); OCN_TILE selects one at run time for tuning
        static const int variant = tile_variant();
#define OCN_LAUNCH_TILED(TX, TY, W)                                                                                        \
    do {                                                                                                                   \
        const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));                                           \
        int KZ = wz; /* z-chunk: enough workgroups to fill the chip, long enough to amortise the 3-flux prologue */        \
        while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < min_blocks()) KZ = (KZ + 1) / 2;                                    \
        dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);                                   \
        if (grid->tz == OCN_PERIODIC)                                                                                      \
            hipLaunchKernelGGL((momentum_tendencies_tiled<OCN_PERIODIC, TX, TY, W, false>), nbt, dim3(TX * TY), 0, stream, g, \
                               u, v, w, Gu, Gv, Gw, r, KZ, fz);                                                             \
        else                                                                                                               \
            hipLaunchKernelGGL((momentum_tendencies_tiled<OCN_BOUNDED, TX, TY, W, false>), nbt, dim3(TX * TY), 0, stream, g,  \
                               u, v, w, Gu, Gv, Gw, r, KZ, fz);                                                             \
    } while (0)
        // narrow ranges (the 64-wide slab of one rank of eight): 32 x 8 patches own 31 columns each, so 64 columns take 3 patches (69 %
        // of the lanes useful in x); 17 x 15 patches own 16 x 14 and fit 64 = 4 x 16 exactly
        const bool narrow = variant == 0 && narrow_tile(wx, wy);
        if (fz.pc_on) {  // pressure correction on load: periodic z; x Periodic (wrapped) or FullyConnected (p halos exchanged), full range
            if ((grid->tx != OCN_PERIODIC && grid->tx != OCN_FULLY_CONNECTED) || grid->tz != OCN_PERIODIC || range != nullptr) {
                ocn::set_error("pressure correction on load needs a (Periodic | FullyConnected, Periodic, Periodic) grid and the full range");
                return OCN_ERR_UNSUPPORTED;
            }
            fz.pc_xhalo = grid->tx == OCN_FULLY_CONNECTED;
            if (fz.strip_w) {  // slab-x rank whose next exchange takes its strips from this launch
                if (!fz.pc_xhalo || !fz.strip_e || !fz.on || wx < 2 * g.Hx) {
                    ocn::set_error("strips are written by the correction-on-load stage of a slab at least 2 Hx wide");
                    return OCN_ERR_INVALID_ARGUMENT;
                }
                if (narrow) launch_tiled<OCN_PERIODIC, 17, 15, true, true, true>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
                else launch_tiled<OCN_PERIODIC, 32, 8, true, true, true>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
                OCN_CHECK_HIP(hipGetLastError());
                return OCN_SUCCESS;
            }
            // 17 x 15 patches (16 x 14 cells owned: 87.8 % of the lanes useful against 84.8 % of 32 x 8) also on full boxes: 24.42 against
            // 24.56 ms per 512^3 step, six same-box pairs (round 4; 16 x 16 patches: 24.9 - 25.4); OCN_PC_TILE=32 selects the 32 x 8 patches
            static const int pc_tile = getenv("OCN_PC_TILE") ? atoi(getenv("OCN_PC_TILE")) : 17;
            if (narrow || pc_tile == 17)
                launch_tiled<OCN_PERIODIC, 17, 15, true, true>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            else if (one_barrier() & 1)
                launch_tiled<OCN_PERIODIC, 32, 8, true, true>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            else
                launch_tiled<OCN_PERIODIC, 32, 8, true, false>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            OCN_CHECK_HIP(hipGetLastError());
            return OCN_SUCCESS;
        }
        if (narrow) {
            if (grid->tz == OCN_PERIODIC)
                launch_tiled<OCN_PERIODIC, 17, 15, false, false>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            else
                launch_tiled<OCN_BOUNDED, 17, 15, false, false>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            OCN_CHECK_HIP(hipGetLastError());
            return OCN_SUCCESS;
        }
        if (variant == 0 && (one_barrier() & (grid->tz == OCN_PERIODIC ? 2 : 4))) {  // the default 32 x 8 tile with one barrier per plane (see the kernel)
            if (grid->tz == OCN_PERIODIC)
                launch_tiled<OCN_PERIODIC, 32, 8, false, true>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            else
                launch_tiled<OCN_BOUNDED, 32, 8, false, true>(g, u, v, w, Gu, Gv, Gw, r, fz, wx, wy, wz, stream);
            OCN_CHECK_HIP(hipGetLastError());
            return OCN_SUCCESS;
        }
        switch (variant) {
            case 1: OCN_LAUNCH_TILED(32, 16, 4); break;
            case 7: OCN_LAUNCH_TILED(32, 16, 2); break;
            case 3: OCN_LAUNCH_TILED(64, 4, 3); break;
            case 4: OCN_LAUNCH_TILED(16, 16, 3); break;
            case 5: OCN_LAUNCH_TILED(64, 8, 2); break;
            case 6: OCN_LAUNCH_TILED(32, 8, 4); break;
            default: OCN_LAUNCH_TILED(32, 8, 3); break;  // fastest measured at 512^3 (6.9 ms vs 9.7 ms direct)
        }
#undef OCN_LAUNCH_TILED
        OCN_CHECK_HIP(hipGetLastError());
        return OCN_SUCCESS;
    }
    static const int strip_tiles = !(getenv("OCN_STRIP_TILES") && !strcmp(getenv("OCN_STRIP_TILES"), "0"));
    if (strip_tiles && !force_direct && !fz.pc_on && grid->tz != OCN_FLAT && wx <= 3 && wy >= 64 && wz >= 4) {
        // the halo-wide x-strips of a distributed run (buffer tendencies): the same shared-flux kernel with a 4 x 64 patch of
        // columns (3 x 63 owned), i.e. tiled in y instead of x
#define OCN_LAUNCH_STRIP(TZV)                                                                                                    \
    do {                                                                                                                         \
        constexpr int TX = 4, TY = 64;                                                                                           \
        const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));                                                \
        int KZ = wz;                                                                                                             \
        while (KZ > strip_min_kz() && tiles * ((wz + KZ - 1) / KZ) < 2048) KZ = (KZ + 1) / 2;                                     \
        dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);                                        \
        hipLaunchKernelGGL((momentum_tendencies_tiled<TZV, TX, TY, 3, false>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, Gu, Gv, \
                           Gw, r, KZ, fz);                                                                                        \
    } while (0)
        if (grid->tz == OCN_PERIODIC) OCN_LAUNCH_STRIP(OCN_PERIODIC); else OCN_LAUNCH_STRIP(OCN_BOUNDED);
#undef OCN_LAUNCH_STRIP
        OCN_CHECK_HIP(hipGetLastError());
        return OCN_SUCCESS;
    }
    if (fz.pc_on) {
        ocn::set_error("pressure correction on load needs the tiled kernel (non-Flat z, range at least 16 x 8 x 4)");
        return OCN_ERR_UNSUPPORTED;
    }
    const dim3 block = ocn::range_block(wx), nb = ocn::range_grid(block, wx, wy, wz);
    switch (grid->tz) {
        case OCN_PERIODIC: hipLaunchKernelGGL(momentum_tendencies_direct<OCN_PERIODIC>, nb, block, 0, stream, g, u, v, w, Gu, Gv, Gw, r, fz); break;
        case OCN_BOUNDED: hipLaunchKernelGGL(momentum_tendencies_direct<OCN_BOUNDED>, nb, block, 0, stream, g, u, v, w, Gu, Gv, Gw, r, fz); break;
        case OCN_FLAT: hipLaunchKernelGGL(momentum_tendencies_direct<OCN_FLAT>, nb, block, 0, stream, g, u, v, w, Gu, Gv, Gw, r, fz); break;
        default: ocn::set_error("unsupported z topology %d", grid->tz); return OCN_ERR_UNSUPPORTED;
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                           double *Gc, const int32_t *range, hipStream_t stream, const ocn::TracerFuse *fuse)
{
    ocn::TracerFuse tf{};
    if (fuse) tf = *fuse;
    Range r;
    int st = make_range(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = r.k1 - r.k0 + 1;
    static const int tracer_direct = (getenv("OCN_TRACER_KERNEL") && !strcmp(getenv("OCN_TRACER_KERNEL"), "direct"));
    if (!tracer_direct && grid->tz != OCN_FLAT && wx >= 16 && wy >= 8 && wz >= 4) {
        constexpr int TX = 32, TY = 8;
        const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
        int KZ = wz;  // z-chunk: enough workgroups to fill the chip, long enough to amortise the bottom-flux prologue
        while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < min_blocks()) KZ = (KZ + 1) / 2;
        dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);
        static const int waves = getenv("OCN_TRACER_WAVES") ? atoi(getenv("OCN_TRACER_WAVES")) : 3;  // min waves / SIMD the build targets
        if (waves >= 4) {
            if (grid->tz == OCN_PERIODIC)
                hipLaunchKernelGGL((tracer_tendency_tiled<OCN_PERIODIC, TX, TY, 4>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
            else
                hipLaunchKernelGGL((tracer_tendency_tiled<OCN_BOUNDED, TX, TY, 4>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
        } else if (grid->tz == OCN_PERIODIC)
            hipLaunchKernelGGL((tracer_tendency_tiled<OCN_PERIODIC, TX, TY>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
        else
            hipLaunchKernelGGL((tracer_tendency_tiled<OCN_BOUNDED, TX, TY>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
        OCN_CHECK_HIP(hipGetLastError());
        return OCN_SUCCESS;
    }
    static const int strip_tiles = !(getenv("OCN_STRIP_TILES") && !strcmp(getenv("OCN_STRIP_TILES"), "0"));
    if (strip_tiles && !tracer_direct && grid->tz != OCN_FLAT && wx <= 3 && wy >= 64 && wz >= 4) {  // halo-wide x-strips: 4 x 64 patches
        constexpr int TX = 4, TY = 64;
        const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
        int KZ = wz;
        while (KZ > strip_min_kz() && tiles * ((wz + KZ - 1) / KZ) < 2048) KZ = (KZ + 1) / 2;
        dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);
        if (grid->tz == OCN_PERIODIC)
            hipLaunchKernelGGL((tracer_tendency_tiled<OCN_PERIODIC, TX, TY>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
        else
            hipLaunchKernelGGL((tracer_tendency_tiled<OCN_BOUNDED, TX, TY>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, c, Gc, r, KZ, tf);
        OCN_CHECK_HIP(hipGetLastError());
        return OCN_SUCCESS;
    }
    const dim3 block = ocn::range_block(wx), nb = ocn::range_grid(block, wx, wy, wz);
    switch (grid->tz) {
        case OCN_PERIODIC: hipLaunchKernelGGL(tracer_tendency_direct<OCN_PERIODIC>, nb, block, 0, stream, g, u, v, w, c, Gc, r, tf); break;
        case OCN_BOUNDED: hipLaunchKernelGGL(tracer_tendency_direct<OCN_BOUNDED>, nb, block, 0, stream, g, u, v, w, c, Gc, r, tf); break;
        case OCN_FLAT: hipLaunchKernelGGL(tracer_tendency_direct<OCN_FLAT>, nb, block, 0, stream, g, u, v, w, c, Gc, r, tf); break;
        default: ocn::set_error("unsupported z topology %d", grid->tz); return OCN_ERR_UNSUPPORTED;
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

// both tracers must take the tiled path (the caller falls back to two single launches otherwise): returns 1 when launched
int launch_tracer_pair_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *const c[2],
                                double *const Gc[2], const int32_t *range, hipStream_t stream, const ocn::TracerFuse fuse[2], int *launched)
{
    *launched = 0;
    Range r;
    int st = make_range(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) { *launched = 1; return OCN_SUCCESS; }
    const int wx = r.i1 - r.i0 + 1, wy = r.j1 - r.j0 + 1, wz = r.k1 - r.k0 + 1;
    // Measured on MI355X (round 2): 192 VGPRs -> 2 waves / SIMD; config 5 11.7 ms / step against 11.3 with two single launches, config 4
    // 40.7 against 40.8 -- the shared loads do not pay for the lost occupancy.  Kept as an option (OCN_TRACER_PAIR=1), off by default.
    static const int off = !(getenv("OCN_TRACER_PAIR") && !strcmp(getenv("OCN_TRACER_PAIR"), "1"));
    if (off || grid->tz == OCN_FLAT || wx < 16 || wy < 8 || wz < 4) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    TracerPair tp{};
    for (int t = 0; t < 2; ++t) { tp.c[t] = c[t]; tp.G[t] = Gc[t]; tp.tf[t] = fuse[t]; }
    constexpr int TX = 32, TY = 8;
    const int tiles = ((wx + TX - 2) / (TX - 1)) * ((wy + TY - 2) / (TY - 1));
    int KZ = wz;
    while (KZ > 16 && tiles * ((wz + KZ - 1) / KZ) < min_blocks()) KZ = (KZ + 1) / 2;
    dim3 nbt((wx + TX - 2) / (TX - 1), (wy + TY - 2) / (TY - 1), (wz + KZ - 1) / KZ);
    if (grid->tz == OCN_PERIODIC)
        hipLaunchKernelGGL((tracer_pair_tendency_tiled<OCN_PERIODIC, TX, TY>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, tp, r, KZ);
    else
        hipLaunchKernelGGL((tracer_pair_tendency_tiled<OCN_BOUNDED, TX, TY>), nbt, dim3(TX * TY), 0, stream, g, u, v, w, tp, r, KZ);
    OCN_CHECK_HIP(hipGetLastError());
    *launched = 1;
    return OCN_SUCCESS;
}

}  // namespace OCN_NS
