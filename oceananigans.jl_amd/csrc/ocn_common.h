// ocn_common.h -- shared host/device helpers for libocn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "../../include/ocn_hip.h"

namespace ocn {

void set_error(const char *fmt, ...);

#define OCN_CHECK_HIP(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ocn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OCN_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

#define OCN_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            ocn::set_error(__VA_ARGS__);      \
            return OCN_ERR_INVALID_ARGUMENT;  \
        }                                     \
    } while (0)

// Device-side copy of the grid, plus derived scalars.  Passed by value to kernels.
struct GridDev {
    int Nx, Ny, Nz, Hx, Hy, Hz, tx, ty, tz;
    double dx, dy, dz;
    const double *dzc, *dzf;
    // walls of the x direction by side: both for Bounded, one for the first (RightConnected: west) and the last (LeftConnected: east)
    // slab of a grid whose partitioned x is Bounded; tx is OCN_BOUNDED for all three (kernels that never meet an x wall ignore these)
    int xw, xe;
};

// x walls of a host-side grid description
inline bool x_wall_west(const ocn_grid &g) { return g.tx == OCN_BOUNDED || g.tx == OCN_RIGHT_CONNECTED; }
inline bool x_wall_east(const ocn_grid &g) { return g.tx == OCN_BOUNDED || g.tx == OCN_LEFT_CONNECTED; }
// x halos that come from a neighbour rank (by side)
inline bool x_connected_west(const ocn_grid &g) { return g.tx == OCN_FULLY_CONNECTED || g.tx == OCN_LEFT_CONNECTED; }
inline bool x_connected_east(const ocn_grid &g) { return g.tx == OCN_FULLY_CONNECTED || g.tx == OCN_RIGHT_CONNECTED; }

inline GridDev to_dev(const ocn_grid &g)
{
    GridDev d;
    d.Nx = g.Nx; d.Ny = g.Ny; d.Nz = g.Nz;
    d.Hx = g.Hx; d.Hy = g.Hy; d.Hz = g.Hz;
    d.xw = x_wall_west(g) ? 1 : 0;
    d.xe = x_wall_east(g) ? 1 : 0;
    // FullyConnected: interior arithmetic identical to Periodic; the half-Bounded slabs: Bounded, qualified by xw / xe
    d.tx = g.tx == OCN_FULLY_CONNECTED ? OCN_PERIODIC : ((d.xw || d.xe) ? OCN_BOUNDED : g.tx);
    d.ty = g.ty; d.tz = g.tz;
    d.dx = g.dx; d.dy = g.dy; d.dz = g.dz;
    d.dzc = g.dzc; d.dzf = g.dzf;
    return d;
}

// Read-only vectors indexed by a wave-uniform index (the stretched-z metrics Δzᵃᵃᶜ[k], Δzᵃᵃᶠ[k]): a pointer that reaches a kernel inside a
// by-value struct is not known to be invariant, so the compiler loads through it with VECTOR memory instructions -- and every
// `s_waitcnt vmcnt(0)` in front of such a value also waits for all the prefetches issued before it, which serialised the software
// pipelines of the Bounded-z kernels (profiles/r03a_config4.md: 45-53 % of the wave cycles parked).  Reading through the CONSTANT address
// space makes them scalar loads (s_load, lgkmcnt): no vector-memory ordering, no VGPRs.  The vectors are never written by a kernel.
typedef const double __attribute__((address_space(4))) *ocn_const_ptr;
__device__ __forceinline__ double uniform_load(const double *p, int idx) { return ((ocn_const_ptr)p)[idx]; }
// A wave-uniform double that the compiler computed with vector instructions (products of kernel arguments, reciprocals) lives in a
// pair of VGPRs; this moves it to scalar registers, which vector instructions read as an operand (one per instruction).  Only for values
// that ARE the same in every lane of the wave.
__device__ __forceinline__ double to_sgpr(double x)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Compiler-level fence for software pipelines: memory operations written above it are issued above it (the loads of the NEXT plane must
// leave at the top of an iteration, not where the scheduler finds their first use).  No instruction is emitted.
#define OCN_ISSUE_LOADS_HERE() asm volatile("" ::: "memory")

// Parent-array layout of a field at location `loc` (bit0 x-face, bit1 y-face, bit2 z-face).
struct Lay {
    int sx, sy, sz;       // parent extents
    long long s2, s3;     // strides of j and k
    long long o;          // offset of interior (1,1,1)
};

__host__ __device__ inline int ocn_ext(int N, int H, int topo, int face)
{
    return N + 2 * H + ((face && topo == OCN_BOUNDED) ? 1 : 0);
}

__host__ __device__ inline Lay make_lay(const GridDev &g, int loc)
{
    Lay L;
    // (x: the extra Face point belongs to the side with the EAST wall -- Bounded, or the last slab of a partitioned Bounded x)
    L.sx = g.Nx + 2 * g.Hx + (((loc & 1) && g.xe) ? 1 : 0);
    L.sy = ocn_ext(g.Ny, g.Hy, g.ty, loc & 2);
    L.sz = ocn_ext(g.Nz, g.Hz, g.tz, loc & 4);
    L.s2 = L.sx;
    L.s3 = (long long)L.sx * L.sy;
    L.o = g.Hx + L.s2 * g.Hy + L.s3 * g.Hz;
    return L;
}
// 1-based interior index -> parent offset
__host__ __device__ inline long long at(const Lay &L, int i, int j, int k)
{
    return L.o + (i - 1) + L.s2 * (j - 1) + L.s3 * (k - 1);
}

// Optional epilogue of the tendency launch: the NEXT RK3 substep  Uo = U + dt*(gamma*G + zeta*Gm)  (runge_kutta_3.jl:194-200)
struct FuseArgs {
    const double *Gm[3];
    double *Uo[3];
    double dt, gamma, zeta;
    int on;
    int has_zeta;  // 0: first-stage form  Uo = U + (dt*gamma)*G  (runge_kutta_3.jl:202-208)
    // optional prologue: the PREVIOUS stage's pressure correction applied to every velocity value as it is loaded,
    //   u <- u - ((p[i]-p[i-1])/dx)*pc_dt  etc. (pressure_correction.jl:31-37); all-periodic grids only (indices wrap)
    const double *pc_p;
    double pc_dt;
    int pc_on;
    // x-partitioned (FullyConnected) local grid: the x indices of p are NOT wrapped -- the x halos of p hold the neighbours' planes
    // (ocn_halo_exchange_pressure) and the westmost halo column of u, whose correction would need p[-Hx], arrives corrected
    int pc_xhalo;
    // G already holds the terms that do not come from advection (momentum_extra_* in `pre` mode ran first): G <- advection + G before it is
    // stored and used by the substep.  Fast math only: the sum is the reference's with the advective term added last instead of first.
    int acc;
    // slab-x rank, correction-on-load stage: the substep results of the Hx westmost / eastmost columns are ALSO written into the send
    // buffers of the next x-halo exchange (layout of halo_pack_x_fields_kernel: field f at strip_field * f, then h + Hx * parent row), so
    // the exchange needs no pack launch; only interior rows are written -- the receiver wraps (ocn_halo_exchange_begin_packed)
    double *strip_w, *strip_e;
    long long strip_field;
};

// device-side copy of ocn_model_terms (physics.hip)
struct TermsDev {
    int coriolis, closure, buoyancy;
    double f, nu, g, alpha, beta;
    const double *T, *S, *pHY;
    const double *nu_e;  // eddy viscosity νₑ (ccc, halos filled) of an LES closure, or NULL: the number `nu`
    double cbeta;        // coriolis == 2 (BetaPlane): f = f + cbeta * y, y = yc[j + Hy - 1] at u points, yf[j + Hy - 1] at v points
    const double *yc, *yf;
};
// the Coriolis parameter at row j of a u point (face = 0) or a v point (face = 1)   (f_plane.jl:44-46, beta_plane.jl:43-57)
__device__ __forceinline__ double coriolis_f_at(const TermsDev &t, int Hy, int j, int face)
{
    if (t.coriolis != 2) return t.f;
    return t.f + t.cbeta * (face ? t.yf : t.yc)[j + Hy - 1];
}
inline TermsDev to_dev(const ocn_model_terms &m)
{
    TermsDev t;
    t.coriolis = m.coriolis; t.closure = m.closure; t.buoyancy = m.buoyancy;
    t.f = m.f; t.nu = m.nu; t.g = m.g; t.alpha = m.alpha; t.beta = m.beta;
    t.T = m.T; t.S = m.S; t.pHY = m.pHY;
    t.nu_e = m.nu_e;
    t.cbeta = m.coriolis_beta; t.yc = m.yc; t.yf = m.yf;
    return t;
}

// bottom / top boundary condition of one field (kind 0: default fill / no flux)
struct ZBc {
    int kind;
    double value, coeff;
    const double *values;
};
// value of a boundary condition at (i, j): array entry, value + coeff * c[boundary cell], or the number
__host__ __device__ inline double bc_condition(const ZBc &bc, int i, int j, int Nx, double c_int)
{
    if (bc.values) return bc.values[(i - 1) + (long long)Nx * (j - 1)];
    if (bc.coeff != 0.0) return bc.value + bc.coeff * c_int;
    return bc.value;
}
// the NEXT stage's rk3 substep of one field, as the epilogue of its tendency kernel:  out = U + dt*(gamma*G + zeta*Gm)
struct SubstepDev {
    const double *Gm;
    double *out;
};
struct SubstepCoef {
    double dt, gamma, zeta;
    int on, has_zeta;
};
// everything the tracer tendency kernel can fold in besides advection (physics.hip / tendencies.hip)
struct TracerFuse {
    int diffusion;          // add -∇_dot_qᶜ
    double kappa;
    const double *kappa_e;  // eddy diffusivity field or NULL
    ZBc bottom, top;        // flux boundary conditions (kind OCN_BC_FLUX) folded into Gc
    SubstepDev sub;
    SubstepCoef sc;
};
// flux boundary conditions of u, v and the substeps of u, v, w for the momentum "finaliser" (momentum_extra_kernel)
struct MomentumFinal {
    ZBc bottom[2], top[2];
    SubstepDev sub[3];
    SubstepCoef sc;
    int xcd;  // XCD-aware workgroup -> tile mapping (set by the launchers)
    int pre;  // the pass runs BEFORE the advective launch: G does not hold anything yet (taken as 0), no substep (FuseArgs::acc follows)
};

// what hydrostatic_momentum_tiled (physics.hip) folds in besides the tendency and the AB2 step of u, v
struct HydroFuse {
    const double *eta;  // ExplicitFreeSurface: η plane (sx, sy) for - g ∇η, or NULL (SplitExplicitFreeSurface: the gradient is 0)
    double grav;
    double *GU, *GV;    // SplitExplicitFreeSurface: barotropic forcing planes (compute_split_explicit_forcing!), or NULL
    double *Ub, *Vb;    //                           Σ Δz u*, Σ Δz v* of the stepped velocities (for the barotropic corrector)
};

int validate_grid(const ocn_grid *g);      // Periodic (or partitioned) x, Periodic y
int validate_grid_any(const ocn_grid *g);  // any of Periodic / Bounded / Flat in x and y

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// 256-thread block for a one-thread-per-cell launch over a range wx cells wide: the usual 64 x 4 (one wave = 64 consecutive
// x), but for the thin x-strips of the distributed buffer tendencies (wx = halo width) 64 lanes along x would leave most of
// every wave idle, so the block is folded towards y: (pow2 >= wx) x (256 / that).
inline dim3 range_block(int wx)
{
    int bx = 64;
    while (bx > 4 && bx / 2 >= wx) bx /= 2;
    return dim3(bx, 256 / bx, 1);
}
inline dim3 range_grid(dim3 block, int wx, int wy, int wz)
{
    return dim3((wx + block.x - 1) / block.x, (wy + block.y - 1) / block.y, wz);
}

}  // namespace ocn
