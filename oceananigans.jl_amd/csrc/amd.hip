// amd.hip -- SURVEY §8(f) rank 2: AnisotropicMinimumDissipation eddy viscosity / diffusivities (Cb = nothing).
//   _compute_AMD_viscosity!, _compute_AMD_diffusivity!
//       src/TurbulenceClosures/turbulence_closure_implementations/anisotropic_minimum_dissipation.jl:125-169
//   the "30 terms"  norm_uᵢₐ_uⱼₐ_Σᵢⱼᶜᶜᶜ :208-257,  norm_tr_∇uᶜᶜᶜ :263-281,  norm_uᵢⱼ_cⱼ_cᵢᶜᶜᶜ :297-323,  norm_θᵢ²ᶜᶜᶜ :325-327
//   normalised gradients  src/TurbulenceClosures/velocity_tracer_gradients.jl:66-140
// Filter widths Δᶠx = 2Δxᶜᶜᶜ etc. are the ccc values at the index they are called with, for every location
// (anisotropic_minimum_dissipation.jl:192-203).
//
// One thread per cell, x fastest across the wave; each term is written exactly as the reference spells it (operand order,
// left-associated sums) so the strict build is bit-identical to the CPU oracle.  x, y Periodic => one set of strides.
// Every normalised gradient is evaluated once per thread into registers and reused by all terms and all tracers.
#include <cstdlib>
#include "ocn_weno.h"

namespace OCN_NS {

using ocn::GridDev;
using ocn::Lay;

// GEN = 0: x, y Periodic -- one parent layout (strides, interior offset) serves u, v, w and the centre fields.  GEN = 1: walls, no Flat
// direction (per-field strides, a step in x is one element: the x offsets of the stencil stay immediates of the loads).  GEN = 2: a Bounded
// x / y gives the Face fields one more point along it, so every field has its own strides (grid_utils.jl:66-72).
template <int GEN>
struct Amd {
    const double *u, *v, *w, *c;  // pointers at the cell (i, j, k)
    long long s2, s3;             // centre fields
    long long u2, u3, v2, v3, w2, w3;
    int sa;                       // stride of a step in x: 1, or 0 along a Flat x (GEN only; a Flat y has zero j strides): differences along a
                                  // Flat direction vanish and its interpolations return the value itself (flat grids: Δ = 1)
    long long oc;                 // offset of the cell in a centre field
    double dx, dy, Fx, Fy;
    Metrics M;
    int k;  // 1-based k of the cell, for the z metrics
    __device__ __forceinline__ int SA() const { return GEN == 2 ? sa : 1; }
    __device__ __forceinline__ double U(int a, int b, int d) const { return GEN ? u[a * SA() + b * u2 + d * u3] : u[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double V(int a, int b, int d) const { return GEN ? v[a * SA() + b * v2 + d * v3] : v[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double W(int a, int b, int d) const { return GEN ? w[a * SA() + b * w2 + d * w3] : w[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double C(int a, int b, int d) const { return c[a * SA() + b * s2 + d * s3]; }
    __device__ __forceinline__ double Fz(int d) const { return 2 * M.dzC(k + d); }
};

__device__ __forceinline__ double julia_max0(double x) { return (x > 0 || x != x) ? x : 0.0; }

template <int GEN>
__device__ __forceinline__ Amd<GEN> make_amd(const GridDev &g, const double *u, const double *v, const double *w, const double *c, int i,
                                            int j, int k)
{
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long o = ocn::at(L, i, j, k);
    Amd<GEN> A;
    A.oc = o;
    A.s2 = L.s2; A.s3 = L.s3;
    A.sa = 1;
    if (GEN) {
        const Lay Lu = ocn::make_lay(g, OCN_LOC_FCC), Lv = ocn::make_lay(g, OCN_LOC_CFC), Lw = ocn::make_lay(g, OCN_LOC_CCF);
        A.u = u + ocn::at(Lu, i, j, k); A.v = v + ocn::at(Lv, i, j, k); A.w = w + ocn::at(Lw, i, j, k);
        A.u2 = Lu.s2; A.u3 = Lu.s3; A.v2 = Lv.s2; A.v3 = Lv.s3; A.w2 = Lw.s2; A.w3 = Lw.s3;
        if (GEN == 2 && g.tx == OCN_FLAT) A.sa = 0;
        if (GEN == 2 && g.ty == OCN_FLAT) A.u2 = A.v2 = A.w2 = A.s2 = 0;
    } else {
        A.u = u + o; A.v = v + o; A.w = w + o;
        A.u2 = A.v2 = A.w2 = L.s2; A.u3 = A.v3 = A.w3 = L.s3;
    }
    A.c = c ? c + o : nullptr;
    A.dx = g.dx; A.dy = g.dy; A.Fx = 2 * g.dx; A.Fy = 2 * g.dy;
    A.M = make_metrics(g);
    A.k = k;
    return A;
}

// ---------------------------------------------------------------------------------------------------
// Fused kernel: νₑ and the κₑ of up to OCN_AMD_MAX_TRACERS tracers in one pass.  Every normalised velocity gradient the 30
// terms need is evaluated ONCE per thread into registers (27 derivatives + the 2 extra ∂y w of the reference's ℑxz quirk)
// and reused by all terms and all tracers; re-evaluating a pure function gives the same bits, so this is still the
// reference's arithmetic term for term.  Arrays are indexed [inner][outer] of the interpolation they feed:
//   ffc quantities [a][b] at (i+a, j+b, k);  fcf [a][d] at (i+a, j, k+d);  cff [b][d] at (i, j+b, k+d).
// ---------------------------------------------------------------------------------------------------
constexpr int OCN_AMD_MAX_TRACERS = 4;
struct AmdTracers {
    const double *c[OCN_AMD_MAX_TRACERS];
    double *kappa_e[OCN_AMD_MAX_TRACERS];
    double Ck[OCN_AMD_MAX_TRACERS];
    int n;
};

// ℑ over a 2x2 set: 0.5 * (0.5*(f[0][0] + f[1][0]) + 0.5*(f[0][1] + f[1][1]))
#if OCN_STRICT
#define I4(f) (0.5 * (0.5 * (f[0][0] + f[1][0]) + 0.5 * (f[0][1] + f[1][1])))
#define I4SQ(f) (0.5 * (0.5 * (f[0][0] * f[0][0] + f[1][0] * f[1][0]) + 0.5 * (f[0][1] * f[0][1] + f[1][1] * f[1][1])))
#define I4PR(f, g) (0.5 * (0.5 * (f[0][0] * g[0][0] + f[1][0] * g[1][0]) + 0.5 * (f[0][1] * g[0][1] + f[1][1] * g[1][1])))
#else  // fast math: one factor 1/4 instead of three halves (two multiplies less per average; the halvings are exact, so only the sums' order changes)
#define I4(f) (0.25 * ((f[0][0] + f[1][0]) + (f[0][1] + f[1][1])))
#define I4SQ(f) (0.25 * ((f[0][0] * f[0][0] + f[1][0] * f[1][0]) + (f[0][1] * f[0][1] + f[1][1] * f[1][1])))
#define I4PR(f, g) (0.25 * ((f[0][0] * g[0][0] + f[1][0] * g[1][0]) + (f[0][1] * g[0][1] + f[1][1] * g[1][1])))
#endif

template <int GEN>
__global__ __launch_bounds__(256, 3) void amd_fused_kernel(GridDev g, double Cnu, const double *__restrict__ u,
                                                        const double *__restrict__ v, const double *__restrict__ w,
                                                        double *__restrict__ nu_e, AmdTracers tr, int i0, int i1, int KZ, int xcd)
{
    // i0..i1: 1..Nx, or a sub-range / the halo columns 0 and Nx+1 (the buffer recomputation of a distributed run,
    // compute_nonhydrostatic_buffer_tendencies.jl:55-68)
    // A workgroup marches KZ planes upward: planes k and k+1 of one iteration are planes k-1 and k of the next, so they come from
    // the L2 of the workgroup's own XCD instead of HBM (one workgroup per plane re-read every plane three times: 120 B per cell
    // measured against 64 compulsory, profiles/r02b_config4.md).  Each XCD walks a contiguous range of tiles (xcd).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd) {
        const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        const unsigned b = bx + nx * (by + ny * bz);
        const unsigned q = b & 7u, chunk = n >> 3, rem = n & 7u;
        const unsigned logical = q * chunk + (q < rem ? q : rem) + (b >> 3);
        bx = logical % nx;
        by = (logical / nx) % ny;
        bz = logical / (nx * ny);
    }
    const int i = i0 + bx * blockDim.x + threadIdx.x, j = 1 + by * blockDim.y + threadIdx.y;
    if (i > i1 || j > g.Ny) return;
    const int kb = 1 + bz * KZ, ke = min(kb + KZ - 1, g.Nz);
    // Everything at z level k+1 of one iteration is at level k of the next: the five z-staggered gradients, the own-column
    // velocities and the tracer values are carried in registers (re-evaluating them would give the same bits: same operands, same
    // operations), so an iteration loads 28 values instead of 46.  Built for 3 waves / SIMD (20 spilled VGPRs): config 4 at 512 x 512
    // x 256 steps in 39.7 ms against 40.4 (the compiler's 176 VGPRs, 2 waves) and 40.8 (one plane per iteration recomputed).
    const Amd<GEN> A0 = make_amd<GEN>(g, u, v, w, nullptr, i, j, kb);
    const double dx = A0.dx, dy = A0.dy, Fx = A0.Fx, Fy = A0.Fy;
#if OCN_STRICT
#define AMD_D(num, den) ((num) / (den))
#define AMD_G(r, num, den) ((r) * ((num) / (den)))  /* normalised gradient: filter-width ratio times the derivative */
    const double rxy = Fx / Fy, ryx = Fy / Fx;
    const double qdx = dx, qdy = dy;
#else
#define AMD_D(num, den) ((num) * (den))  /* den holds the reciprocal */
#define AMD_G(r, num, den) ((r) * ((num) * (den)))  /* (folding r * den into one factor saves 32 multiplies and costs 8 registers: slower) */
    const double rFx = fast_rcp(Fx), rFy = fast_rcp(Fy);
    const double rxy = Fx * rFy, ryx = Fy * rFx;
    const double qdx = fast_rcp(dx), qdy = fast_rcp(dy);
#endif
    double cU[2], cV[2], cW, c_dzu[2], c_dxw[2], c_dzv[2], c_dyw[2], c_dywq[2];
    double tc0[OCN_AMD_MAX_TRACERS], tgz[OCN_AMD_MAX_TRACERS];  // tracer value at level k, its norm_∂z_c at face k
    {   // level kb
        const double Fz0 = A0.Fz(0), dzf0 = A0.M.dzF(kb);
#if OCN_STRICT
        const double rxz0 = Fx / Fz0, rzx0 = Fz0 / Fx, ryz0 = Fy / Fz0, rzy0 = Fz0 / Fy, qdzf0 = dzf0;
#else
        const double rFz0 = fast_rcp(Fz0);
        const double rxz0 = Fx * rFz0, rzx0 = Fz0 * rFx, ryz0 = Fy * rFz0, rzy0 = Fz0 * rFy, qdzf0 = fast_rcp(dzf0);
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            cU[a] = A0.U(a, 0, 0);
            cV[a] = A0.V(0, a, 0);
            c_dzu[a] = AMD_G(rzx0, A0.U(a, 0, 0) - A0.U(a, 0, -1), qdzf0);
            c_dxw[a] = AMD_G(rxz0, A0.W(a, 0, 0) - A0.W(a - 1, 0, 0), qdx);
            c_dzv[a] = AMD_G(rzy0, A0.V(0, a, 0) - A0.V(0, a, -1), qdzf0);
            c_dyw[a] = AMD_G(ryz0, A0.W(0, a, 0) - A0.W(0, a - 1, 0), qdy);
            c_dywq[a] = AMD_G(ryz0, A0.W(a, 0, 0) - A0.W(a, -1, 0), qdy);
        }
        cW = A0.W(0, 0, 0);
        const long long o0 = A0.oc;
#pragma unroll
        for (int n = 0; n < OCN_AMD_MAX_TRACERS; ++n) {
            tc0[n] = tgz[n] = 0.0;
            if (n < tr.n) {
                const double *pc = tr.c[n] + o0;
                tc0[n] = pc[0];
                tgz[n] = AMD_G(Fz0, tc0[n] - pc[-A0.s3], qdzf0);
            }
        }
    }
    for (int k = kb; k <= ke; ++k) {
    const Amd<GEN> A = make_amd<GEN>(g, u, v, w, nullptr, i, j, k);
    const long long o = A.oc;
    // filter width and spacings of level k (for d2, dzw) and of level k+1 (the new gradients); k is uniform across the workgroup
    const double Fz[2] = {A.Fz(0), A.Fz(1)};
    const double dzf1 = A.M.dzF(k + 1);
    const double dzc0 = A.M.dzC(k);
#if OCN_STRICT
    const double rxz1 = Fx / Fz[1], rzx1 = Fz[1] / Fx, ryz1 = Fy / Fz[1], rzy1 = Fz[1] / Fy;
    const double qdzc = dzc0, qdzf1 = dzf1;
#else
    const double rFz[2] = {fast_rcp(Fz[0]), fast_rcp(Fz[1])};
    const double rxz1 = Fx * rFz[1], rzx1 = Fz[1] * rFx, ryz1 = Fy * rFz[1], rzy1 = Fz[1] * rFy;
    const double qdzc = fast_rcp(dzc0), qdzf1 = fast_rcp(dzf1);
#endif
    // new values: the in-plane neighbours at level k and everything at level k+1
    const double uT[2] = {A.U(0, 0, 1), A.U(1, 0, 1)}, vT[2] = {A.V(0, 0, 1), A.V(0, 1, 1)};
    const double wT[3] = {A.W(-1, 0, 1), A.W(0, 0, 1), A.W(1, 0, 1)}, wS = A.W(0, -1, 1), wN = A.W(0, 1, 1), wSE = A.W(1, -1, 1);
    double dyu[2][2], dxv[2][2], dzu[2][2], dxw[2][2], dzv[2][2], dyw[2][2], dywq[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        dyu[a][0] = AMD_G(ryx, cU[a] - A.U(a, -1, 0), qdy);       // norm_∂y_u at (i+a, j+b, k)
        dyu[a][1] = AMD_G(ryx, A.U(a, 1, 0) - cU[a], qdy);
        dzu[a][0] = c_dzu[a];                                      // norm_∂z_u at (i+a, j, k+b)
        dzu[a][1] = AMD_G(rzx1, uT[a] - cU[a], qdzf1);
        dxw[a][0] = c_dxw[a];                                      // norm_∂x_w
        dxw[a][1] = AMD_G(rxz1, wT[a + 1] - wT[a], qdx);
        dzv[a][0] = c_dzv[a];                                      // norm_∂z_v at (i, j+a, k+b)
        dzv[a][1] = AMD_G(rzy1, vT[a] - cV[a], qdzf1);
        dyw[a][0] = c_dyw[a];                                      // norm_∂y_w
        dywq[a][0] = c_dywq[a];                                    // norm_∂y_w at (i+a, j, k+b): the ℑxz quirk
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        dxv[0][b] = AMD_G(rxy, cV[b] - A.V(-1, b, 0), qdx);       // norm_∂x_v at (i+a, j+b, k)
        dxv[1][b] = AMD_G(rxy, A.V(1, b, 0) - cV[b], qdx);
    }
    dyw[0][1] = AMD_G(ryz1, wT[1] - wS, qdy);
    dyw[1][1] = AMD_G(ryz1, wN - wT[1], qdy);
    dywq[0][1] = AMD_G(ryz1, wT[1] - wS, qdy);
    dywq[1][1] = AMD_G(ryz1, wT[2] - wSE, qdy);
    const double dxu = AMD_D(cU[1] - cU[0], qdx), dyv = AMD_D(cV[1] - cV[0], qdy), dzw = AMD_D(wT[1] - cW, qdzc);
#if OCN_STRICT
    const double d2 = 3 / ((1 / (Fx * Fx) + 1 / (Fy * Fy)) + 1 / (Fz[0] * Fz[0]));
#define AMD_Q(num, den) ((num) / (den))
#else
    const double d2 = 3 * fast_rcp((rFx * rFx + rFy * rFy) + rFz[0] * rFz[0]);
#define AMD_Q(num, den) ((num) * fast_rcp(den))
#endif

    if (nu_e) {
        double s12[2][2], s13[2][2], s23[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                s12[a][b] = 0.5 * (dyu[a][b] + dxv[a][b]);
                s13[a][b] = 0.5 * (dzu[a][b] + dxw[a][b]);
                s23[a][b] = 0.5 * (dzv[a][b] + dyw[a][b]);
            }
        const double xv2 = I4SQ(dxv), yu2 = I4SQ(dyu), xw2 = I4SQ(dxw), zu2 = I4SQ(dzu), yw2 = I4SQ(dyw), zv2 = I4SQ(dzv);
        const double q = (((((((dxu * dxu + dyv * dyv) + dzw * dzw) + xv2) + yu2) + xw2) + zu2) + yw2) + zv2;
        double nu = 0.0;
        if (q != 0) {
            const double r1 = ((((dxu * (dxu * dxu) + dyv * xv2) + dzw * xw2) + 2 * dxu * I4PR(dxv, s12)) + 2 * dxu * I4PR(dxw, s13)) +
                              2 * I4(dxv) * I4(dxw) * I4(s23);
            const double r2 = ((((dxu * yu2 + dyv * (dyv * dyv)) + dzw * yw2) + 2 * dyv * I4PR(dyu, s12)) + 2 * I4(dyu) * I4(dyw) * I4(s13)) +
                              2 * dyv * I4PR(dyw, s23);
            const double r3 = ((((dxu * zu2 + dyv * zv2) + dzw * (dzw * dzw)) + 2 * I4(dzu) * I4(dzv) * I4(s12)) + 2 * dzw * I4PR(dzu, s13)) +
                              2 * dzw * I4PR(dzv, s23);
            const double r = (r1 + r2) + r3;
#if OCN_STRICT
            const double Cb_zeta = 0.0 / Fz[0];  // Cb = nothing
#else
            const double Cb_zeta = 0.0;
#endif
            nu = AMD_Q(-Cnu * d2 * (r - Cb_zeta), q);
        }
        nu_e[o] = julia_max0(nu);
    }
    if (tr.n > 0) {
        const double ixy_dxv = I4(dxv), ixz_dxw = I4(dxw), ixy_dyu = I4(dyu), ixz_dywq = I4(dywq), ixz_dzu = I4(dzu), iyz_dzv = I4(dzv);
#pragma unroll
        for (int n = 0; n < OCN_AMD_MAX_TRACERS; ++n) {
            if (n >= tr.n) break;
            const double *pc = tr.c[n] + o;
            const long long s2 = A.s2, s3 = A.s3;
            const int sa = A.SA();
            const double c0 = tc0[n], cT = pc[s3];
            const double gx0 = AMD_G(Fx, c0 - pc[-sa], qdx), gx1 = AMD_G(Fx, pc[sa] - c0, qdx);             // norm_∂x_c at i, i+1
            const double gy0 = AMD_G(Fy, c0 - pc[-s2], qdy), gy1 = AMD_G(Fy, pc[s2] - c0, qdy);             // norm_∂y_c at j, j+1
            const double gz0 = tgz[n], gz1 = AMD_G(Fz[1], cT - c0, qdzf1);                                   // norm_∂z_c at k, k+1
            tc0[n] = cT; tgz[n] = gz1;
            const double xc2 = 0.5 * (gx0 * gx0 + gx1 * gx1), yc2 = 0.5 * (gy0 * gy0 + gy1 * gy1), zc2 = 0.5 * (gz0 * gz0 + gz1 * gz1);
            const double sigma = (xc2 + yc2) + zc2;
            double kap = 0.0;
            if (sigma != 0) {
                const double cx = 0.5 * (gx0 + gx1), cy = 0.5 * (gy0 + gy1), cz = 0.5 * (gz0 + gz1);
                const double cx_ux = (dxu * xc2 + ixy_dxv * cx * cy) + ixz_dxw * cx * cz;
                const double cy_uy = (ixy_dyu * cy * cx + dyv * yc2) + ixz_dywq * cy * cz;  // ℑxz of norm_∂y_w, as the reference (:313)
                const double cz_uz = (ixz_dzu * cz * cx + iyz_dzv * cz * cy) + dzw * zc2;
                const double theta = (cx_ux + cy_uy) + cz_uz;
                kap = AMD_Q(-tr.Ck[n] * d2 * theta, sigma);
            }
            tr.kappa_e[n][o] = julia_max0(kap);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        cU[a] = uT[a]; cV[a] = vT[a];
        c_dzu[a] = dzu[a][1]; c_dxw[a] = dxw[a][1]; c_dzv[a] = dzv[a][1]; c_dyw[a] = dyw[a][1]; c_dywq[a] = dywq[a][1];
    }
    cW = wT[1];
    }  // k
#undef AMD_D
#undef AMD_G
#undef AMD_Q
}
#undef I4
#undef I4SQ
#undef I4PR

int launch_amd_fused(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e, int ntr,
                     const double *Ck, const double *const *c, double *const *kappa_e, hipStream_t stream, const int32_t *irange)
{
    if (ntr > OCN_AMD_MAX_TRACERS) {
        ocn::set_error("at most %d tracers per AMD launch, got %d", OCN_AMD_MAX_TRACERS, ntr);
        return OCN_ERR_INVALID_ARGUMENT;
    }
    AmdTracers tr{};
    tr.n = ntr;
    for (int n = 0; n < ntr; ++n) {
        tr.c[n] = c[n];
        tr.kappa_e[n] = kappa_e[n];
        tr.Ck[n] = Ck[n];
    }
    GridDev g = ocn::to_dev(*grid);
    const int i0 = irange ? irange[0] : 1, i1 = irange ? irange[1] : g.Nx;
    if (i1 < i0) return OCN_SUCCESS;
    static const int kz_env = getenv("OCN_AMD_KZ") ? atoi(getenv("OCN_AMD_KZ")) : 16;
    static const int xcd = getenv("OCN_XCD_REMAP") ? atoi(getenv("OCN_XCD_REMAP")) : 1;
    const int wx = i1 - i0 + 1;
    dim3 block = ocn::range_block(wx);
    if (block.x == 64) block = dim3(32, 8, 1);  // squarer tiles: fewer rim rows re-read per plane
    int KZ = kz_env < 1 ? 1 : kz_env;
    const long long tiles = (long long)((wx + block.x - 1) / block.x) * ((g.Ny + block.y - 1) / block.y);
    while (KZ > 1 && tiles * ((g.Nz + KZ - 1) / KZ) < 2048) KZ = (KZ + 1) / 2;  // narrow ranges: keep the chip full
    const dim3 nb = ocn::range_grid(block, wx, g.Ny, (g.Nz + KZ - 1) / KZ);
    if (grid->tx == OCN_FLAT || grid->ty == OCN_FLAT)  // zero strides along a Flat direction
        hipLaunchKernelGGL(amd_fused_kernel<2>, nb, block, 0, stream, g, Cnu, u, v, w, nu_e, tr, i0, i1, KZ, xcd);
    else if (ocn::x_wall_west(*grid) || ocn::x_wall_east(*grid) || grid->ty == OCN_BOUNDED)  // per-field parent layouts
        hipLaunchKernelGGL(amd_fused_kernel<1>, nb, block, 0, stream, g, Cnu, u, v, w, nu_e, tr, i0, i1, KZ, xcd);
    else
        hipLaunchKernelGGL(amd_fused_kernel<0>, nb, block, 0, stream, g, Cnu, u, v, w, nu_e, tr, i0, i1, KZ, xcd);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_amd_viscosity(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e,
                         hipStream_t stream)
{
    return launch_amd_fused(grid, Cnu, u, v, w, nu_e, 0, nullptr, nullptr, nullptr, stream, nullptr);
}

int launch_amd_diffusivity(const ocn_grid *grid, double Ck, const double *u, const double *v, const double *w, const double *c,
                           double *kappa_e, hipStream_t stream)
{
    return launch_amd_fused(grid, 0.0, u, v, w, nullptr, 1, &Ck, &c, &kappa_e, stream, nullptr);
}

}  // namespace OCN_NS
