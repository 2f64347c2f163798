// rowfft.hip -- x-direction (contiguous rows) real <-> half-spectrum transforms of the Poisson solver, fused with
// their neighbours on the path:
//   rowfft_source_r2c : K8 `_compute_source_term!` (solve_for_pressure.jl:12-17) + the forward real FFT along x (K10),
//                       so the divergence never makes a round trip through HBM (u, v, w in -> half spectrum out);
//   rowfft_c2r        : the inverse real FFT along x writing the rows of the haloed pressure field directly
//                       (K10 + K13 `copy_real_component!`), including its periodic x-halo.
// A real row of length N = 2M is transformed as ONE complex FFT of length M on z[n] = x[2n] + i x[2n+1] plus an
// O(N) split / merge step (the classic packed real FFT); M = 8*8*R3 in {64,128,256,512} uses the same radix-8 stages as
// colfft.hip.  RB rows per workgroup, thread (row, t) holds the 8 elements t + (M/8) r.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ocn_internal.h"

namespace ocn {

namespace {
struct cplx {
    double x, y;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.x, -a.y}; }
template <bool INV>
__device__ __forceinline__ cplx mul_mi(cplx a)
{
    return INV ? cplx{-a.y, a.x} : cplx{a.y, -a.x};
}
template <bool INV>
__device__ __forceinline__ void radix2(cplx &a, cplx &b)
{
    const cplx t = csub(a, b);
    a = cadd(a, b);
    b = t;
}
template <bool INV>
__device__ __forceinline__ void radix4(cplx &x0, cplx &x1, cplx &x2, cplx &x3)
{
    const cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = mul_mi<INV>(csub(x1, x3));
    x0 = cadd(a, c);
    x1 = cadd(b, d);
    x2 = csub(a, c);
    x3 = csub(b, d);
}
template <bool INV>
__device__ __forceinline__ void radix8(cplx *x)
{
    cplx e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    cplx o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    radix4<INV>(e0, e1, e2, e3);
    radix4<INV>(o0, o1, o2, o3);
    const double h = 0.70710678118654752440;
    const cplx w1 = INV ? cplx{h, h} : cplx{h, -h};
    const cplx w3 = INV ? cplx{-h, h} : cplx{-h, -h};
    o1 = cmul(o1, w1);
    o2 = mul_mi<INV>(o2);
    o3 = cmul(o3, w3);
    x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
    x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
    x[2] = cadd(e2, o2); x[6] = csub(e2, o2);
    x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
}
template <int R3, bool INV>
__device__ __forceinline__ void radix_last(cplx *z)
{
    if (R3 == 8) radix8<INV>(z);
    if (R3 == 4) {
        radix4<INV>(z[0], z[1], z[2], z[3]);
        radix4<INV>(z[4], z[5], z[6], z[7]);
    }
    if (R3 == 2) {
        radix2<INV>(z[0], z[1]);
        radix2<INV>(z[2], z[3]);
        radix2<INV>(z[4], z[5]);
        radix2<INV>(z[6], z[7]);
    }
}

// stored position p (thread t owns p = 8t..8t+7 after the last stage) -> index (frequency for DIF-forward,
// time for DIF-with-conjugate-twiddles)
template <int M>
__device__ __forceinline__ int stage_index(int p)
{
    constexpr int T2 = M / 64;
    if (T2 == 1) return (p >> 3) + 8 * (p & 7);
    const int g = p / T2, q3 = p % T2;
    return (g >> 3) + 8 * (g & 7) + 64 * q3;
}

// Decimation-in-frequency FFT of length M on x[r] = in[t + (M/8) r] (natural order), twiddles exp(-/+ 2 pi i j / M) from
// the LDS table W (CONJ selects the inverse sign).  On return x[m] holds the output element stage_index(8 t + m).
// A is this row's M-element LDS scratch.  Contains two (three) __syncthreads().
template <int M, bool CONJ>
__device__ __forceinline__ void dif_fft(cplx *x, cplx *A, const cplx *W, int t)
{
    constexpr int T = M / 8, T2 = M / 64;
    const int q = t / T2, t2 = t % T2;
    radix8<CONJ>(x);
#pragma unroll
    for (int qq = 1; qq < 8; ++qq) {
        const cplx w = W[t * qq];
        x[qq] = cmul(x[qq], CONJ ? cconj(w) : w);
    }
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) A[qq * T + t] = x[qq];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = A[q * T + t2 + T2 * r];
    radix8<CONJ>(x);
    if (T2 > 1) {
#pragma unroll
        for (int qq = 1; qq < 8; ++qq) {
            const cplx w = W[8 * t2 * qq];
            x[qq] = cmul(x[qq], CONJ ? cconj(w) : w);
        }
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) A[(q * 8 + qq) * T2 + t2] = x[qq];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = A[8 * t + m];
        radix_last<T2, CONJ>(x);
    }
}
}  // namespace

struct RowFFTArgs {
    GridDev g;
    const double *u, *v, *w;  // source mode
    const double *real_in;    // alternative input: halo-free real array Nx x rows (set_source_term!), used when u == nullptr
    double dt;
    double *spec;             // half spectrum, complex interleaved, row pitch nxh = M + 1
    double *p;                // inverse mode: haloed pressure field
    const double *twM;        // exp(-2 pi i j / M), j < M
    const double *twN;        // exp(-2 pi i k / (2M)), k <= M   (split / merge factors)
    long long nrows;          // Ny * Nz
    double scale;             // inverse: applied to the output
    int scale_dz;             // source mode: rhs = (Δzᶜ div) / dt, the Fourier-tridiagonal form (solve_for_pressure.jl:33-38)
};

// ---- forward: rows of div(u,v,w)/dt -> half spectrum ------------------------------------------------------------------
template <int M, int RB>
__global__ __launch_bounds__(RB *(M / 8)) void rowfft_source_r2c_kernel(RowFFTArgs a)
{
    constexpr int T = M / 8, N = 2 * M;
    __shared__ cplx A[RB][M + 1];
    __shared__ cplx W[M];
    __shared__ cplx WN[M + 1];
    const int tid = threadIdx.x, t = tid % T, rl = tid / T;
    for (int j = tid; j < M; j += RB * T) W[j] = reinterpret_cast<const cplx *>(a.twM)[j];
    for (int j = tid; j <= M; j += RB * T) WN[j] = reinterpret_cast<const cplx *>(a.twN)[j];
    const long long row = (long long)blockIdx.x * RB + rl;
    const bool active = row < a.nrows;
    const GridDev &g = a.g;
    const int j = active ? (int)(row % g.Ny) + 1 : 1, k = active ? (int)(row / g.Ny) + 1 : 1;
    const Lay L = make_lay(g, OCN_LOC_CCC);  // x, y periodic: same strides for every location
    // div at cells i, i+1 for i = 2 (t + T r) + 1 (1-based): divᶜᶜᶜ (divergence_operators.jl:16-19), then / dt
    const double dzc = g.dzc ? uniform_load(g.dzc, k + g.Hz - 1) : g.dz;
    const double Ax = g.dy * dzc, Ay = g.dx * dzc, Az = g.dx * g.dy, rV = 1 / (Az * dzc);
    const bool flatz = g.tz == OCN_FLAT;
    cplx x[8];
    if (a.u != nullptr) {
    // The 11 values of every pair of cells are LOADED for four pairs at a time (44 doubles in flight per thread) before any divergence is
    // formed: left to itself the compiler interleaves each pair's arithmetic (two IEEE divisions) with its own loads and waits for memory
    // eight times per thread (`s_waitcnt vmcnt(0)` after every 7 loads: 4.3 TB/s, 67 % of the wave cycles parked).  Same expressions.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double uu[4][3], vv[4][4], ww[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 2 * (t + T * (4 * h + q)) + 1;
            const long long o = at(L, i, j, k);
            uu[q][0] = a.u[o]; uu[q][1] = a.u[o + 1]; uu[q][2] = a.u[o + 2];
            vv[q][0] = a.v[o]; vv[q][1] = a.v[o + 1]; vv[q][2] = a.v[o + L.s2]; vv[q][3] = a.v[o + L.s2 + 1];
            if (!flatz) {
                ww[q][0] = a.w[o]; ww[q][1] = a.w[o + 1]; ww[q][2] = a.w[o + L.s3]; ww[q][3] = a.w[o + L.s3 + 1];
            } else {
                ww[q][0] = ww[q][1] = ww[q][2] = ww[q][3] = 0.0;
            }
        }
        OCN_ISSUE_LOADS_HERE();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double u0 = uu[q][0], u1 = uu[q][1], u2 = uu[q][2];
            const double v00 = vv[q][0], v01 = vv[q][1], v10 = vv[q][2], v11 = vv[q][3];
            double dw0 = 0.0, dw1 = 0.0;
            if (!flatz) {
                dw0 = Az * ww[q][2] - Az * ww[q][0];
                dw1 = Az * ww[q][3] - Az * ww[q][1];
            }
            const double d0 = rV * (((Ax * u1 - Ax * u0) + (Ay * v10 - Ay * v00)) + dw0);
            const double d1 = rV * (((Ax * u2 - Ax * u1) + (Ay * v11 - Ay * v01)) + dw1);
            x[4 * h + q] = active ? (a.scale_dz ? cplx{(dzc * d0) / a.dt, (dzc * d1) / a.dt} : cplx{d0 / a.dt, d1 / a.dt}) : cplx{0, 0};
        }
    }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const long long o = row * N + 2 * (t + T * r);
            x[r] = active ? cplx{a.real_in[o], a.real_in[o + 1]} : cplx{0, 0};
        }
    }
    __syncthreads();
    dif_fft<M, false>(x, A[rl], W, t);
    // natural order through LDS, then the split step  X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-2 pi i k/N} (Z[k] - conj Z[M-k])
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) A[rl][stage_index<M>(8 * t + m)] = x[m];
    __syncthreads();
    if (active) {
        cplx *out = reinterpret_cast<cplx *>(a.spec) + row * (M + 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int kk = t + T * r;  // 0 .. M-1
            const cplx zk = A[rl][kk], zm = cconj(A[rl][(M - kk) % M]);
            const cplx s = cadd(zk, zm), d = csub(zk, zm);
            const cplx wd = cmul(WN[kk], d);  // e^{-2 pi i k / N} (Z[k] - conj Z[M-k])
            out[kk] = cplx{0.5 * (s.x + wd.y), 0.5 * (s.y - wd.x)};  // s/2 - (i/2) wd
        }
        if (t == 0) {  // k = M: X[M] = Re Z[0] - Im Z[0]
            const cplx z0 = A[rl][0];
            out[M] = cplx{z0.x - z0.y, 0.0};
        }
    }
    (void)N;
}

// ---- inverse: half spectrum -> rows of the haloed pressure field -----------------------------------------------------
template <int M, int RB>
__global__ __launch_bounds__(RB *(M / 8)) void rowfft_c2r_kernel(RowFFTArgs a)
{
    constexpr int T = M / 8;
    __shared__ cplx A[RB][M + 1];
    __shared__ cplx W[M];
    __shared__ cplx WN[M + 1];
    const int tid = threadIdx.x, t = tid % T, rl = tid / T;
    for (int j = tid; j < M; j += RB * T) W[j] = reinterpret_cast<const cplx *>(a.twM)[j];
    for (int j = tid; j <= M; j += RB * T) WN[j] = reinterpret_cast<const cplx *>(a.twN)[j];
    const long long row = (long long)blockIdx.x * RB + rl;
    const bool active = row < a.nrows;
    const GridDev &g = a.g;
    const int j = active ? (int)(row % g.Ny) + 1 : 1, k = active ? (int)(row / g.Ny) + 1 : 1;
    const cplx *in = reinterpret_cast<const cplx *>(a.spec) + (active ? row : 0) * (M + 1);
    __syncthreads();
    // merge step: Z[k] = (X[k] + conj X[M-k])/2 + i/2 e^{+2 pi i k/N} (X[k] - conj X[M-k]),  k = t + T r
    cplx x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int kk = t + T * r;
        const cplx xk = in[kk], xm = cconj(in[M - kk]);
        const cplx s = cadd(xk, xm), d = csub(xk, xm);
        const cplx wd = cmul(cconj(WN[kk]), d);
        x[r] = cplx{0.5 * (s.x - wd.y), 0.5 * (s.y + wd.x)};  // s/2 + (i/2) wd
    }
    dif_fft<M, true>(x, A[rl], W, t);  // unnormalised inverse; x[m] = z[stage_index(8t+m)]
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) A[rl][stage_index<M>(8 * t + m)] = x[m];
    __syncthreads();
    if (active) {
        const Lay L = make_lay(g, OCN_LOC_CCC);
        double *prow = a.p + at(L, 1, j, k);
        const int Nx = 2 * M;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = t + T * r;
            const cplx z = A[rl][n];
            const double p0 = z.x * a.scale, p1 = z.y * a.scale;  // x[2n], x[2n+1]
            prow[2 * n] = p0;
            prow[2 * n + 1] = p1;
            // periodic x-halo of the row (fill_periodic_west_and_east_halo!): images of the first / last Hx cells
            if (2 * n < g.Hx) prow[2 * n + Nx] = p0;
            if (2 * n + 1 < g.Hx) prow[2 * n + 1 + Nx] = p1;
            if (2 * n >= Nx - g.Hx) prow[2 * n - Nx] = p0;
            if (2 * n + 1 >= Nx - g.Hx) prow[2 * n + 1 - Nx] = p1;
        }
    }
}

bool rowfft_supported(int Nx) { return Nx == 128 || Nx == 256 || Nx == 512 || Nx == 1024; }

void rowfft_twiddles(int Nx, std::vector<double> &twM, std::vector<double> &twN)
{
    const int M = Nx / 2;
    const long double two_pi = 6.283185307179586476925286766559L;
    twM.resize(2 * M);
    twN.resize(2 * (M + 1));
    for (int j = 0; j < M; ++j) {
        twM[2 * j] = (double)cosl(two_pi * j / M);
        twM[2 * j + 1] = (double)(-sinl(two_pi * j / M));
    }
    for (int k = 0; k <= M; ++k) {
        twN[2 * k] = (double)cosl(two_pi * k / Nx);
        twN[2 * k + 1] = (double)(-sinl(two_pi * k / Nx));
    }
}

template <int M>
static int launch_m(int inverse, const RowFFTArgs &a, hipStream_t stream)
{
    constexpr int RB = (M >= 256) ? 8 : 16;
    const dim3 grid((unsigned)((a.nrows + RB - 1) / RB)), block(RB * (M / 8));
    if (inverse)
        hipLaunchKernelGGL((rowfft_c2r_kernel<M, RB>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((rowfft_source_r2c_kernel<M, RB>), grid, block, 0, stream, a);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_rowfft(const ocn_grid *grid, int inverse, const double *u, const double *v, const double *w, const double *real_in,
                  double dt, double *spec, double *p, const double *twM, const double *twN, double scale, hipStream_t stream,
                  int scale_dz)
{
    RowFFTArgs a;
    a.scale_dz = scale_dz;
    a.g = to_dev(*grid);
    a.u = u; a.v = v; a.w = w; a.real_in = real_in; a.dt = dt; a.spec = spec; a.p = p; a.twM = twM; a.twN = twN;
    a.nrows = (long long)grid->Ny * grid->Nz;
    a.scale = scale;
    switch (grid->Nx) {
        case 128: return launch_m<64>(inverse, a, stream);
        case 256: return launch_m<128>(inverse, a, stream);
        case 512: return launch_m<256>(inverse, a, stream);
        case 1024: return launch_m<512>(inverse, a, stream);
        default: set_error("row FFT length %d is not supported (128, 256, 512, 1024)", grid->Nx); return OCN_ERR_UNSUPPORTED;
    }
}

}  // namespace ocn
