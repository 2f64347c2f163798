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
#include "ocn_weno.h"

namespace OCN_NS {

using ocn::GridDev;
using ocn::Lay;

struct Range {
    int i0, i1, j0, j1, k0, k1;  // 1-based inclusive
    int ou, ov, ow;              // first index written for Gu (in i), Gv (in j), Gw (in k): periphery exclusion
};

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
                                                                  Range r)
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
        Gu[ocn::at(Lu, i, j, k)] = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
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
        Gv[ocn::at(Lv, i, j, k)] = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
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
        Gw[ocn::at(Lw, i, j, k)] = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
    }
}

// K4 tracer: flux = (A * U[i,j,k]) * cR   (upwind_biased_advective_fluxes.jl:99-121)
template <int TB>
__device__ __forceinline__ double tracer_flux(double area, double ut, const double *__restrict__ pc, long long sc, int idx,
                                              int N)
{
    const double cr = bias_interp<TB, false>([&](int m) { return pc[m * sc]; }, idx, N, ut > 0);
    return (area * ut) * cr;
}

template <int TZ>
__global__ __launch_bounds__(256) void tracer_tendency_direct(GridDev g, const double *__restrict__ u,
                                                              const double *__restrict__ v, const double *__restrict__ w,
                                                              const double *__restrict__ c, double *__restrict__ Gc, Range r)
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
    const double rV = 1 / (M.Az * M.dzC(k));
    Gc[ocn::at(Lc, i, j, k)] = -(rV * (((fx1 - fx0) + (fy1 - fy0)) + dzF));
}

static int make_range(const ocn_grid *grid, const int32_t *range, Range &r)
{
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

int launch_momentum_tendencies(const ocn_grid *grid, const double *u, const double *v, const double *w, double *Gu,
                               double *Gv, double *Gw, const int32_t *range, hipStream_t stream)
{
    Range r;
    int st = make_range(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    dim3 block(64, 4, 1);
    dim3 nb((r.i1 - r.i0 + 64) / 64, (r.j1 - r.j0 + 4) / 4, r.k1 - r.k0 + 1);
    switch (grid->tz) {
        case OCN_PERIODIC: hipLaunchKernelGGL(momentum_tendencies_direct<OCN_PERIODIC>, nb, block, 0, stream, g, u, v, w, Gu, Gv, Gw, r); break;
        case OCN_BOUNDED: hipLaunchKernelGGL(momentum_tendencies_direct<OCN_BOUNDED>, nb, block, 0, stream, g, u, v, w, Gu, Gv, Gw, r); break;
        case OCN_FLAT: hipLaunchKernelGGL(momentum_tendencies_direct<OCN_FLAT>, nb, block, 0, stream, g, u, v, w, Gu, Gv, Gw, r); break;
        default: ocn::set_error("unsupported z topology %d", grid->tz); return OCN_ERR_UNSUPPORTED;
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_tracer_tendency(const ocn_grid *grid, const double *u, const double *v, const double *w, const double *c,
                           double *Gc, const int32_t *range, hipStream_t stream)
{
    Range r;
    int st = make_range(grid, range, r);
    if (st != OCN_SUCCESS) return st;
    if (r.i1 < r.i0 || r.j1 < r.j0 || r.k1 < r.k0) return OCN_SUCCESS;
    GridDev g = ocn::to_dev(*grid);
    dim3 block(64, 4, 1);
    dim3 nb((r.i1 - r.i0 + 64) / 64, (r.j1 - r.j0 + 4) / 4, r.k1 - r.k0 + 1);
    switch (grid->tz) {
        case OCN_PERIODIC: hipLaunchKernelGGL(tracer_tendency_direct<OCN_PERIODIC>, nb, block, 0, stream, g, u, v, w, c, Gc, r); break;
        case OCN_BOUNDED: hipLaunchKernelGGL(tracer_tendency_direct<OCN_BOUNDED>, nb, block, 0, stream, g, u, v, w, c, Gc, r); break;
        case OCN_FLAT: hipLaunchKernelGGL(tracer_tendency_direct<OCN_FLAT>, nb, block, 0, stream, g, u, v, w, c, Gc, r); break;
        default: ocn::set_error("unsupported z topology %d", grid->tz); return OCN_ERR_UNSUPPORTED;
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

}  // namespace OCN_NS
