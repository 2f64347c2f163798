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

struct Amd {
    const double *u, *v, *w, *c;  // pointers at the cell (i, j, k)
    long long s2, s3;
    double dx, dy, Fx, Fy;
    Metrics M;
    int k;  // 1-based k of the cell, for the z metrics
    __device__ __forceinline__ double U(int a, int b, int d) const { return u[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double V(int a, int b, int d) const { return v[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double W(int a, int b, int d) const { return w[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double C(int a, int b, int d) const { return c[a + b * s2 + d * s3]; }
    __device__ __forceinline__ double Fz(int d) const { return 2 * M.dzC(k + d); }
};

__device__ __forceinline__ double julia_max0(double x) { return (x > 0 || x != x) ? x : 0.0; }

__device__ __forceinline__ Amd make_amd(const GridDev &g, const double *u, const double *v, const double *w, const double *c, int i,
                                       int j, int k)
{
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    const long long o = ocn::at(L, i, j, k);
    Amd A;
    A.u = u + o; A.v = v + o; A.w = w + o; A.c = c ? c + o : nullptr;
    A.s2 = L.s2; A.s3 = L.s3;
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
#define I4(f) (0.5 * (0.5 * (f[0][0] + f[1][0]) + 0.5 * (f[0][1] + f[1][1])))
#define I4SQ(f) (0.5 * (0.5 * (f[0][0] * f[0][0] + f[1][0] * f[1][0]) + 0.5 * (f[0][1] * f[0][1] + f[1][1] * f[1][1])))
#define I4PR(f, g) (0.5 * (0.5 * (f[0][0] * g[0][0] + f[1][0] * g[1][0]) + 0.5 * (f[0][1] * g[0][1] + f[1][1] * g[1][1])))

__global__ __launch_bounds__(256) void amd_fused_kernel(GridDev g, double Cnu, const double *__restrict__ u,
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
    for (int k = kb; k <= ke; ++k) {
    const Amd A = make_amd(g, u, v, w, nullptr, i, j, k);
    const long long o = A.u - u;
    // filter-width ratios and spacings of the two z levels this cell touches (k is uniform across the workgroup)
    const double dx = A.dx, dy = A.dy, Fx = A.Fx, Fy = A.Fy;
    const double Fz[2] = {A.Fz(0), A.Fz(1)};
    const double dzf[2] = {A.M.dzF(k), A.M.dzF(k + 1)};
    const double dzc0 = A.M.dzC(k);
#if OCN_STRICT
#define AMD_D(num, den) ((num) / (den))
    const double rxy = Fx / Fy, ryx = Fy / Fx;
    const double rxz[2] = {Fx / Fz[0], Fx / Fz[1]}, rzx[2] = {Fz[0] / Fx, Fz[1] / Fx};
    const double ryz[2] = {Fy / Fz[0], Fy / Fz[1]}, rzy[2] = {Fz[0] / Fy, Fz[1] / Fy};
    const double qdx = dx, qdy = dy, qdzc = dzc0;
    const double qdzf[2] = {dzf[0], dzf[1]};
#else
#define AMD_D(num, den) ((num) * (den))  /* den holds the reciprocal */
    const double rFx = fast_rcp(Fx), rFy = fast_rcp(Fy), rFz[2] = {fast_rcp(Fz[0]), fast_rcp(Fz[1])};
    const double rxy = Fx * rFy, ryx = Fy * rFx;
    const double rxz[2] = {Fx * rFz[0], Fx * rFz[1]}, rzx[2] = {Fz[0] * rFx, Fz[1] * rFx};
    const double ryz[2] = {Fy * rFz[0], Fy * rFz[1]}, rzy[2] = {Fz[0] * rFy, Fz[1] * rFy};
    const double qdx = fast_rcp(dx), qdy = fast_rcp(dy), qdzc = fast_rcp(dzc0);
    const double qdzf[2] = {fast_rcp(dzf[0]), fast_rcp(dzf[1])};
#endif
    double dyu[2][2], dxv[2][2], dzu[2][2], dxw[2][2], dzv[2][2], dyw[2][2], dywq[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            dyu[a][b] = ryx * AMD_D(A.U(a, b, 0) - A.U(a, b - 1, 0), qdy);       // norm_∂y_u at (i+a, j+b, k)
            dxv[a][b] = rxy * AMD_D(A.V(a, b, 0) - A.V(a - 1, b, 0), qdx);       // norm_∂x_v
            dzu[a][b] = rzx[b] * AMD_D(A.U(a, 0, b) - A.U(a, 0, b - 1), qdzf[b]);  // norm_∂z_u at (i+a, j, k+b)
            dxw[a][b] = rxz[b] * AMD_D(A.W(a, 0, b) - A.W(a - 1, 0, b), qdx);      // norm_∂x_w
            dzv[a][b] = rzy[b] * AMD_D(A.V(0, a, b) - A.V(0, a, b - 1), qdzf[b]);  // norm_∂z_v at (i, j+a, k+b)
            dyw[a][b] = ryz[b] * AMD_D(A.W(0, a, b) - A.W(0, a - 1, b), qdy);      // norm_∂y_w
            dywq[a][b] = ryz[b] * AMD_D(A.W(a, 0, b) - A.W(a, -1, b), qdy);        // norm_∂y_w at (i+a, j, k+b): the ℑxz quirk
        }
    }
    const double dxu = AMD_D(A.U(1, 0, 0) - A.U(0, 0, 0), qdx), dyv = AMD_D(A.V(0, 1, 0) - A.V(0, 0, 0), qdy),
                 dzw = AMD_D(A.W(0, 0, 1) - A.W(0, 0, 0), qdzc);
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
            const double c0 = pc[0];
            const double gx0 = Fx * AMD_D(c0 - pc[-1], qdx), gx1 = Fx * AMD_D(pc[1] - c0, qdx);               // norm_∂x_c at i, i+1
            const double gy0 = Fy * AMD_D(c0 - pc[-s2], qdy), gy1 = Fy * AMD_D(pc[s2] - c0, qdy);             // norm_∂y_c at j, j+1
            const double gz0 = Fz[0] * AMD_D(c0 - pc[-s3], qdzf[0]), gz1 = Fz[1] * AMD_D(pc[s3] - c0, qdzf[1]);  // norm_∂z_c at k, k+1
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
    }  // k
#undef AMD_D
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
    hipLaunchKernelGGL(amd_fused_kernel, nb, block, 0, stream, g, Cnu, u, v, w, nu_e, tr, i0, i1, KZ, xcd);
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
