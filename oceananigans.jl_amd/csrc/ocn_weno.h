// ocn_weno.h -- device arithmetic of the WENO5 flux-form momentum/tracer advection
// (src/Advection/weno_interpolants.jl, centered_reconstruction.jl, upwind_biased_advective_fluxes.jl,
// topologically_conditional_interpolation.jl).  Included by tendencies.hip, which is compiled
// twice: OCN_STRICT=1 with -ffp-contract=off (reference evaluation order, bit-reproducible
// against the CPU oracle) and OCN_STRICT=0 (FMA contraction + single-division weights).
#pragma once
#include "ocn_common.h"

#ifndef OCN_STRICT
#error "define OCN_STRICT to 0 or 1"
#endif

// OCN_UPWIND=1 builds the same kernels for advection = UpwindBiased(order=5) (upwind_biased_reconstruction.jl:41-140): the
// biased reconstructions become the fixed 5-point (3-point near Bounded walls, then 1-point) upwind stencils; the advecting
// velocity scheme Centered(order=4), the halo conditions and the flux forms are those of WENO(order=5).
#ifndef OCN_UPWIND
#define OCN_UPWIND 0
#endif
#if OCN_STRICT && OCN_UPWIND
#define OCN_NS ocn_strict_up
#elif OCN_STRICT
#define OCN_NS ocn_strict
#elif OCN_UPWIND
#define OCN_NS ocn_fast_up
#else
#define OCN_NS ocn_fast
#endif

namespace OCN_NS {

// Reconstruction coefficients: stencil_coefficients (reconstruction_coefficients.jl:100-115) evaluated as
// Julia does (Int/Int -> Float64 quotient, BigFloat accumulation, last = 1 - sum).  Same values as
// oracle/coefficients.py generates; tests/test_library_constants.py compares them.
#define OCN_C4_0 (-0.08333333333333326)
#define OCN_C4_1 (0.5833333333333333)
#define OCN_C4_2 (0.5833333333333333)
#define OCN_C4_3 (-0.08333333333333333)

#define OCN_W5P_00 (0.33333333333333337)
#define OCN_W5P_01 (0.8333333333333334)
#define OCN_W5P_02 (-0.16666666666666674)
#define OCN_W5P_10 (-0.16666666666666669)
#define OCN_W5P_11 (0.8333333333333333)
#define OCN_W5P_12 (0.3333333333333335)
#define OCN_W5P_20 (0.33333333333333326)
#define OCN_W5P_21 (-1.1666666666666667)
#define OCN_W5P_22 (1.8333333333333335)

// const ε = 1f-8, widened to Float64 where used (weno_interpolants.jl:70)
#define OCN_WENO_EPS (9.99999993922529e-09)

#define OCN_C5_0 (3.0 / 10.0)
#define OCN_C5_1 (3.0 / 5.0)
#define OCN_C5_2 (1.0 / 10.0)
#define OCN_C3_0 (2.0 / 3.0)
#define OCN_C3_1 (1.0 / 3.0)

// smoothness_operation for buffer 3 (weno_interpolants.jl:178-183, 213-225)
__device__ __forceinline__ double beta3(double p0, double p1, double p2, double c0, double c1, double c2, double c3,
                                        double c4, double c5)
{
    return p0 * ((c0 * p0 + c1 * p1) + c2 * p2) + p1 * (c3 * p1 + c4 * p2) + (p2 * p2) * c5;
}

#if !OCN_STRICT
// 1/x to fp64 round-off: v_rcp_f64 seed + two Newton steps (no div_scale / div_fixup: operands here are well inside
// the normal range: sums of squares bounded below by eps^6 ~ 1e-48 and above by |u|^12).
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// the same with ONE Newton step: v_rcp_f64 is good to about 2^-26 relative, one step squares that (~2^-52); used where the
// quotient is a convex combination of O(1) candidates (the WENO5 weights), whose value is insensitive to the last bit of 1/den
__device__ __forceinline__ double fast_rcp1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
#endif

// WENO5 reconstruction at a face from S = psi[n-3..n+2] (weno_interpolants.jl:341-348, 445-447, 475-511)
#if OCN_UPWIND
// UpwindBiased(order=5) / (order=3) stencils: calc_reconstruction_stencil(FT, buffer, :left / :right) with the coefficients of
// stencil_coefficients evaluated as Julia does (oracle/coefficients.py); n-ary + is left-associated.
__device__ __forceinline__ double weno5(double S0, double S1, double S2, double S3, double S4, double S5, bool left)
{
    const double l = (((0.033333333333333326 * S0 + -0.21666666666666667 * S1) + 0.7833333333333333 * S2) + 0.45 * S3) + -0.04999999999999998 * S4;
    const double r = (((-0.050000000000000044 * S1 + 0.45 * S2) + 0.7833333333333333 * S3) + -0.21666666666666667 * S4) + 0.03333333333333331 * S5;
    return left ? l : r;
}
__device__ __forceinline__ double weno3(double S0, double S1, double S2, double S3, bool left)
{
    const double l = (-0.16666666666666674 * S0 + 0.8333333333333334 * S1) + 0.33333333333333337 * S2;
    const double r = (0.3333333333333335 * S1 + 0.8333333333333333 * S2) + -0.16666666666666669 * S3;
    return left ? l : r;
}
#define weno5 weno5_nonlinear_unused
#define weno3 weno3_nonlinear_unused
#endif
#if !OCN_STRICT
// fast-math WENO5 from the five inputs of the selected (mirrored for a right bias) stencil
__device__ __forceinline__ double weno5_fast_core(double T0, double T1, double T2, double T3, double T4)
{
    // Everything below is built from the four first differences of the five inputs.
    const double da = T1 - T0, db = T2 - T1, dc = T3 - T2, dd = T4 - T3;
    // smoothness indicators in difference form: the reference polynomial (coefficients 10,-31,11,25,-19,4 etc.) equals
    // 13/4 D^2 + 3/4 E^2 with D the second difference and E the one-sided first difference of each sub-stencil:
    //   E0 = 3 T2 - 4 T3 + T4 = dd - 3 dc,   E1 = T1 - T3 = -(db + dc) (it only enters squared),   E2 = T0 - 4 T1 + 3 T2 = 3 db - da.
    const double D0 = dd - dc, D1 = dc - db, D2 = db - da;
    const double E0 = __builtin_fma(-3.0, dc, dd), E1 = db + dc, E2 = __builtin_fma(3.0, db, -da);
    // The weights depend on the betas only through tau/(beta + eps), which is invariant under a common scaling of
    // (beta, eps): work with b_r = beta_r / 0.75 = E^2 + (13/3) D^2 and eps' = eps / 0.75 (one multiply less per beta);
    // d_r = b_r + eps' takes eps' as the addend of the first FMA, and tau = |b0 - b2| = |d0 - d2|.
    const double d0 = __builtin_fma((13.0 / 3.0) * D0, D0, __builtin_fma(E0, E0, OCN_WENO_EPS / 0.75));
    const double d1 = __builtin_fma((13.0 / 3.0) * D1, D1, __builtin_fma(E1, E1, OCN_WENO_EPS / 0.75));
    const double d2 = __builtin_fma((13.0 / 3.0) * D2, D2, __builtin_fma(E2, E2, OCN_WENO_EPS / 0.75));
    const double tau = d0 - d2;
    // Same rational function with ONE reciprocal: alpha_r = C_r (d_r^2 + tau^2)/d_r^2;
    // multiply numerator and denominator of sum(alpha_r p_r)/sum(alpha_r) by d0^2 d1^2 d2^2:
    //   m_r = (d_r^2 + tau^2) prod_{s != r} d_s^2,  result = sum (C_r p_r) m_r / sum C_r m_r.
    const double t2 = tau * tau;
    const double e0 = d0 * d0, e1 = d1 * d1, e2 = d2 * d2;
    // m_r = (e_r + t2) e_s e_t = e0 e1 e2 + t2 e_s e_t
    const double e12 = e1 * e2, e02 = e0 * e2, e01 = e0 * e1, e012 = e0 * e12;
    const double m0 = __builtin_fma(t2, e12, e012);
    const double m2 = __builtin_fma(t2, e01, e012);
    // sum C_r = 1: den = e0 e1 e2 + t2 (C0 e1 e2 + C1 e0 e2 + C2 e0 e1)
    const double den = __builtin_fma(t2, __builtin_fma(OCN_C5_2, e01, __builtin_fma(OCN_C5_1, e02, OCN_C5_0 * e12)), e012);
    // sum w_r p_r = p1 + w0 (p0 - p1) + w2 (p2 - p1), and the differences of the candidate reconstructions are differences of the
    // second differences already at hand: p0 - p1 = (D1 - D0) / 6, p2 - p1 = (D2 - D1) / 3: the three 3-point candidates (9 operations)
    // shrink to the centre one, p1 = (-T1 + 5 T2 + 2 T3) / 6 = T2 + db / 6 + dc / 3 (2), plus 6 for the correction
    const double p1 = __builtin_fma(1.0 / 3.0, dc, __builtin_fma(1.0 / 6.0, db, T2));
    const double num = __builtin_fma(OCN_C5_2 / 3.0, m2 * (D2 - D1), (OCN_C5_0 / 6.0) * (m0 * (D1 - D0)));
    return __builtin_fma(num, fast_rcp1(den), p1);
}
// The same reconstruction for a stencil that sits in LDS (p0 = the element at offset 0 of the line, st = its element stride): the
// upwind choice selects the ADDRESSES -- element q of the selected stencil is at offset (left ? q - 3 : 2 - q) = c + q d with
// d = +-st -- so the five inputs are loaded once (5 LDS reads instead of 6) and two 32-bit selects + 4 integer operations replace
// ten v_cndmask_b32 of the 64-bit value selects.
__device__ __forceinline__ double weno5_fast_lds(const double *p0, int st, bool left)
{
    const int d = left ? st : -st;
    const double *c = p0 + (left ? -3 * st : 2 * st);
    return weno5_fast_core(c[0], c[d], c[2 * d], c[3 * d], c[4 * d]);
}
#endif
__device__ __forceinline__ double weno5(double S0, double S1, double S2, double S3, double S4, double S5, bool left)
{
#if OCN_STRICT
    // left: psi0=(S2,S3,S4) psi1=(S1,S2,S3) psi2=(S0,S1,S2); right: psi0=(S3,S2,S1) psi1=(S4,S3,S2) psi2=(S5,S4,S3)
    const double a0 = left ? S2 : S3, a1 = left ? S3 : S2, a2 = left ? S4 : S1;
    const double b0 = left ? S1 : S4, b1 = left ? S2 : S3, b2 = left ? S3 : S2;
    const double c0 = left ? S0 : S5, c1 = left ? S1 : S4, c2 = left ? S2 : S3;
    const double be0 = beta3(a0, a1, a2, 10., -31., 11., 25., -19., 4.);
    const double be1 = beta3(b0, b1, b2, 4., -13., 5., 13., -13., 4.);
    const double be2 = beta3(c0, c1, c2, 4., -19., 11., 25., -31., 10.);
    const double tau = fabs(be0 - be2);
    const double p0 = (OCN_W5P_00 * a0 + OCN_W5P_01 * a1) + OCN_W5P_02 * a2;
    const double p1 = (OCN_W5P_10 * b0 + OCN_W5P_11 * b1) + OCN_W5P_12 * b2;
    const double p2 = (OCN_W5P_20 * c0 + OCN_W5P_21 * c1) + OCN_W5P_22 * c2;
    const double q0 = tau / (be0 + OCN_WENO_EPS), q1 = tau / (be1 + OCN_WENO_EPS), q2 = tau / (be2 + OCN_WENO_EPS);
    const double al0 = OCN_C5_0 * (1 + q0 * q0), al1 = OCN_C5_1 * (1 + q1 * q1), al2 = OCN_C5_2 * (1 + q2 * q2);
    const double sa = (al0 + al1) + al2;
    const double w0 = al0 / sa, w1 = al1 / sa, w2 = al2 / sa;
    return (w0 * p0 + w1 * p1) + w2 * p2;
#else
    // The right-biased stencils are the left-biased ones of the mirrored data: select 5 inputs, then one code path.
    const double T0 = left ? S0 : S5, T1 = left ? S1 : S4, T2 = left ? S2 : S3, T3 = left ? S3 : S2, T4 = left ? S4 : S1;
    return weno5_fast_core(T0, T1, T2, T3, T4);
#endif
}

// WENO3 (buffer scheme) from S = psi[n-2..n+1]
__device__ __forceinline__ double weno3(double S0, double S1, double S2, double S3, bool left)
{
    const double a0 = left ? S1 : S2, a1 = left ? S2 : S1;
    const double b0 = left ? S0 : S3, b1 = left ? S1 : S2;
    const double be0 = a0 * (1. * a0 + -2. * a1) + (a1 * a1) * 1.;
    const double be1 = b0 * (1. * b0 + -2. * b1) + (b1 * b1) * 1.;
    const double tau = fabs(be0 - be1);
    const double q0 = tau / (be0 + OCN_WENO_EPS), q1 = tau / (be1 + OCN_WENO_EPS);
    const double al0 = OCN_C3_0 * (1 + q0 * q0), al1 = OCN_C3_1 * (1 + q1 * q1);
    const double sa = al0 + al1;
    const double w0 = al0 / sa, w1 = al1 / sa;
    const double p0 = 0.5 * a0 + 0.5 * a1;
    const double p1 = -0.5 * b0 + 1.5 * b1;
    return w0 * p0 + w1 * p1;
}

__device__ __forceinline__ double centered4(double m2, double m1, double z0, double p1)
{
    return ((OCN_C4_0 * m2 + OCN_C4_1 * m1) + OCN_C4_2 * z0) + OCN_C4_3 * p1;
}

// ---------------------------------------------------------------------------------------------------
// Topology-conditional interpolation along one line (topologically_conditional_interpolation.jl:37-128).
// `val(m)` returns the (metric-weighted) value at offset m from the face index n.
// TOPO is the topology along the line; idx the index the reference tests (i, j or k of the call);
// CENTER selects the *ᶜ variants (the caller has already shifted the line to face idx+1).
// ---------------------------------------------------------------------------------------------------
#if OCN_UPWIND
#undef weno5
#undef weno3
#endif

template <int TOPO, bool CENTER, class V>
__device__ __forceinline__ double sym_interp(V val, int idx, int N)
{
    if (TOPO == OCN_FLAT) return val(CENTER ? -1 : 0);  // flat_advective_fluxes.jl:26-44
    if (TOPO == OCN_BOUNDED) {
        const bool hi = CENTER ? (idx >= 3 && idx <= N + 1 - 3) : (idx >= 4 && idx <= N + 1 - 3);  // :46-47, H = 3
        if (!hi) return 0.5 * val(-1) + 0.5 * val(0);  // Centered(order=2) for every deeper fallback
    }
    return centered4(val(-2), val(-1), val(0), val(1));
}

// symmetric interpolation of (a * psi) for a metric `a` that is constant along the line: the reference multiplies every
// stencil value (strict); the interpolation is linear, so fast math factors the metric out.  (Folding `a` into the two distinct
// coefficients saves 4 operations per plane and costs 12 loop-invariant registers: spills in the correction-on-load kernel.)
template <int TOPO, bool CENTER, class V>
__device__ __forceinline__ double sym_interp_scaled(V val, double a, int idx, int N)
{
#if OCN_STRICT
    return sym_interp<TOPO, CENTER>([&](int m) { return a * val(m); }, idx, N);
#else
    return a * sym_interp<TOPO, CENTER>(val, idx, N);
#endif
}

// 1 / V for the tendency prefactor
__device__ __forceinline__ double recip_volume(double V)
{
#if OCN_STRICT
    return 1 / V;
#else
    return fast_rcp(V);
#endif
}

template <int TOPO, bool CENTER, class V>
__device__ __forceinline__ double bias_interp(V val, int idx, int N, bool left)
{
    if (TOPO == OCN_FLAT) return val(CENTER ? -1 : 0);
    if (TOPO == OCN_BOUNDED) {
        bool ok5, ok3;
        if (CENTER) {  // outside_biased_haloᶜ :51-52 with H = 3 and H = 2
            ok5 = (idx >= 3) && (idx <= N + 1 - 3);
            ok3 = (idx >= 2) && (idx <= N + 1 - 2);
        } else {  // outside_biased_haloᶠ :49-50
            ok5 = (idx >= 4) && (idx <= N + 1 - 3);
            ok3 = (idx >= 3) && (idx <= N + 1 - 2);
        }
        if (!ok5) {
            if (ok3) return weno3(val(-2), val(-1), val(0), val(1), left);
            return left ? val(-1) : val(0);  // UpwindBiased(order=1)
        }
    }
    return weno5(val(-3), val(-2), val(-1), val(0), val(1), val(2), left);
}

// biased reconstruction along a Periodic line whose stencil sits in LDS: p0 = the element val(0), st = element stride
#ifndef OCN_LDS_SELECT
#define OCN_LDS_SELECT 1
#endif
__device__ __forceinline__ double bias_interp_lds(const double *p0, int st, bool left)
{
#if !OCN_STRICT && !OCN_UPWIND && OCN_LDS_SELECT
    return weno5_fast_lds(p0, st, left);
#else
    return weno5(p0[-3 * st], p0[-2 * st], p0[-st], p0[0], p0[st], p0[2 * st], left);
#endif
}

// grid metrics at Center z-location (spacings_and_areas_and_volumes.jl:106-140, 263-345)
struct Metrics {
    double dx, dy, dz, Az;
    const double *dzc, *dzf;
    int Hz;
    __device__ __forceinline__ double dzC(int k) const { return dzc ? ocn::uniform_load(dzc, k + Hz - 1) : dz; }
    __device__ __forceinline__ double dzF(int k) const { return dzf ? ocn::uniform_load(dzf, k + Hz - 1) : dz; }
    __device__ __forceinline__ double Ax(int k) const { return dy * dzC(k); }  // Axᶠᶜᶜ = Δy*Δz
    __device__ __forceinline__ double Ay(int k) const { return dx * dzC(k); }  // Ayᶜᶠᶜ = Δx*Δz
};
__device__ __forceinline__ Metrics make_metrics(const ocn::GridDev &g)
{
    Metrics M;
    M.dx = g.dx; M.dy = g.dy; M.dz = g.dz; M.Az = g.dx * g.dy;
    M.dzc = g.dzc; M.dzf = g.dzf; M.Hz = g.Hz;
    return M;
}

}  // namespace OCN_NS
