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

// normalised gradients at offset (a, b, d) from the cell
struct DxU { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return (A.U(a + 1, b, d) - A.U(a, b, d)) / A.dx; } };
struct DyV { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return (A.V(a, b + 1, d) - A.V(a, b, d)) / A.dy; } };
struct DzW { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return (A.W(a, b, d + 1) - A.W(a, b, d)) / A.M.dzC(A.k + d); } };
struct DxV { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fx / A.Fy * ((A.V(a, b, d) - A.V(a - 1, b, d)) / A.dx); } };
struct DyU { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fy / A.Fx * ((A.U(a, b, d) - A.U(a, b - 1, d)) / A.dy); } };
struct DxW { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fx / A.Fz(d) * ((A.W(a, b, d) - A.W(a - 1, b, d)) / A.dx); } };
struct DzU { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fz(d) / A.Fx * ((A.U(a, b, d) - A.U(a, b, d - 1)) / A.M.dzF(A.k + d)); } };
struct DyW { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fy / A.Fz(d) * ((A.W(a, b, d) - A.W(a, b - 1, d)) / A.dy); } };
struct DzV { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fz(d) / A.Fy * ((A.V(a, b, d) - A.V(a, b, d - 1)) / A.M.dzF(A.k + d)); } };
struct DxC { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fx * ((A.C(a, b, d) - A.C(a - 1, b, d)) / A.dx); } };
struct DyC { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fy * ((A.C(a, b, d) - A.C(a, b - 1, d)) / A.dy); } };
struct DzC { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return A.Fz(d) * ((A.C(a, b, d) - A.C(a, b, d - 1)) / A.M.dzF(A.k + d)); } };
struct S12 { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return 0.5 * (DyU()(A, a, b, d) + DxV()(A, a, b, d)); } };
struct S13 { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return 0.5 * (DzU()(A, a, b, d) + DxW()(A, a, b, d)); } };
struct S23 { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return 0.5 * (DzV()(A, a, b, d) + DyW()(A, a, b, d)); } };
template <class F>
struct Sq { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { const double t = F()(A, a, b, d); return t * t; } };
template <class F, class G>
struct Pr { __device__ __forceinline__ double operator()(const Amd &A, int a, int b, int d) const { return F()(A, a, b, d) * G()(A, a, b, d); } };

// ℑ of functions (interpolation_operators.jl:20-26, 44-57), evaluated at the cell
template <class F> __device__ __forceinline__ double Ix(const Amd &A, int b = 0, int d = 0) { return 0.5 * (F()(A, 0, b, d) + F()(A, 1, b, d)); }
template <class F> __device__ __forceinline__ double Iy(const Amd &A, int a = 0, int d = 0) { return 0.5 * (F()(A, a, 0, d) + F()(A, a, 1, d)); }
template <class F> __device__ __forceinline__ double Iz(const Amd &A) { return 0.5 * (F()(A, 0, 0, 0) + F()(A, 0, 0, 1)); }
template <class F> __device__ __forceinline__ double Ixy(const Amd &A) { return 0.5 * (Ix<F>(A, 0, 0) + Ix<F>(A, 1, 0)); }
template <class F> __device__ __forceinline__ double Ixz(const Amd &A) { return 0.5 * (Ix<F>(A, 0, 0) + Ix<F>(A, 0, 1)); }
template <class F> __device__ __forceinline__ double Iyz(const Amd &A) { return 0.5 * (Iy<F>(A, 0, 0) + Iy<F>(A, 0, 1)); }

__device__ __forceinline__ double amd_delta2(const Amd &A)
{
    const double Fz = A.Fz(0);
    return 3 / ((1 / (A.Fx * A.Fx) + 1 / (A.Fy * A.Fy)) + 1 / (Fz * Fz));
}
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

__global__ __launch_bounds__(256) void amd_viscosity_kernel(GridDev g, double Cnu, const double *__restrict__ u,
                                                            const double *__restrict__ v, const double *__restrict__ w,
                                                            double *__restrict__ nu_e)
{
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y, k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny) return;
    const Amd A = make_amd(g, u, v, w, nullptr, i, j, k);
    const double dxu = DxU()(A, 0, 0, 0), dyv = DyV()(A, 0, 0, 0), dzw = DzW()(A, 0, 0, 0);
    const double xv2 = Ixy<Sq<DxV>>(A), yu2 = Ixy<Sq<DyU>>(A), xw2 = Ixz<Sq<DxW>>(A), zu2 = Ixz<Sq<DzU>>(A), yw2 = Iyz<Sq<DyW>>(A),
                 zv2 = Iyz<Sq<DzV>>(A);
    const double q = (((((((dxu * dxu + dyv * dyv) + dzw * dzw) + xv2) + yu2) + xw2) + zu2) + yw2) + zv2;
    double nu = 0.0;
    if (q != 0) {
        const double r1 = ((((dxu * (dxu * dxu) + dyv * xv2) + dzw * xw2) + 2 * dxu * Ixy<Pr<DxV, S12>>(A)) + 2 * dxu * Ixz<Pr<DxW, S13>>(A)) +
                          2 * Ixy<DxV>(A) * Ixz<DxW>(A) * Iyz<S23>(A);
        const double r2 = ((((dxu * yu2 + dyv * (dyv * dyv)) + dzw * yw2) + 2 * dyv * Ixy<Pr<DyU, S12>>(A)) +
                           2 * Ixy<DyU>(A) * Iyz<DyW>(A) * Ixz<S13>(A)) + 2 * dyv * Iyz<Pr<DyW, S23>>(A);
        const double r3 = ((((dxu * zu2 + dyv * zv2) + dzw * (dzw * dzw)) + 2 * Ixz<DzU>(A) * Iyz<DzV>(A) * Ixy<S12>(A)) +
                           2 * dzw * Ixz<Pr<DzU, S13>>(A)) + 2 * dzw * Iyz<Pr<DzV, S23>>(A);
        const double r = (r1 + r2) + r3;
        const double Cb_zeta = 0.0 / A.Fz(0);  // Cb = nothing
        nu = -Cnu * amd_delta2(A) * (r - Cb_zeta) / q;
    }
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    nu_e[ocn::at(L, i, j, k)] = julia_max0(nu);
}

__global__ __launch_bounds__(256) void amd_diffusivity_kernel(GridDev g, double Ck, const double *__restrict__ u,
                                                              const double *__restrict__ v, const double *__restrict__ w,
                                                              const double *__restrict__ c, double *__restrict__ kappa_e)
{
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y, k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny) return;
    const Amd A = make_amd(g, u, v, w, c, i, j, k);
    const double xc2 = Ix<Sq<DxC>>(A), yc2 = Iy<Sq<DyC>>(A), zc2 = Iz<Sq<DzC>>(A);
    const double sigma = (xc2 + yc2) + zc2;
    double kap = 0.0;
    if (sigma != 0) {
        const double cx = Ix<DxC>(A), cy = Iy<DyC>(A), cz = Iz<DzC>(A);
        const double cx_ux = (DxU()(A, 0, 0, 0) * xc2 + Ixy<DxV>(A) * cx * cy) + Ixz<DxW>(A) * cx * cz;
        // ℑxzᶜᵃᶜ (not ℑyz) of norm_∂y_w, as the reference writes it (anisotropic_minimum_dissipation.jl:313)
        const double cy_uy = (Ixy<DyU>(A) * cy * cx + DyV()(A, 0, 0, 0) * yc2) + Ixz<DyW>(A) * cy * cz;
        const double cz_uz = (Ixz<DzU>(A) * cz * cx + Iyz<DzV>(A) * cz * cy) + DzW()(A, 0, 0, 0) * zc2;
        const double theta = (cx_ux + cy_uy) + cz_uz;
        kap = -Ck * amd_delta2(A) * theta / sigma;
    }
    const Lay L = ocn::make_lay(g, OCN_LOC_CCC);
    kappa_e[ocn::at(L, i, j, k)] = julia_max0(kap);
}

int launch_amd_viscosity(const ocn_grid *grid, double Cnu, const double *u, const double *v, const double *w, double *nu_e,
                         hipStream_t stream)
{
    GridDev g = ocn::to_dev(*grid);
    dim3 block(64, 4, 1), nb((g.Nx + 63) / 64, (g.Ny + 3) / 4, g.Nz);
    hipLaunchKernelGGL(amd_viscosity_kernel, nb, block, 0, stream, g, Cnu, u, v, w, nu_e);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_amd_diffusivity(const ocn_grid *grid, double Ck, const double *u, const double *v, const double *w, const double *c,
                           double *kappa_e, hipStream_t stream)
{
    GridDev g = ocn::to_dev(*grid);
    dim3 block(64, 4, 1), nb((g.Nx + 63) / 64, (g.Ny + 3) / 4, g.Nz);
    hipLaunchKernelGGL(amd_diffusivity_kernel, nb, block, 0, stream, g, Ck, u, v, w, c, kappa_e);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

}  // namespace OCN_NS
