// colfft.hip -- strided "column" FFTs for the Poisson solver (replaces rocFFT's sbcc passes for the y and z
// directions of the 3-D transform, K10, and fuses the spectral solve K12 into the z pass).
//
// Why: in the 3-D transform of a (kx fastest) half spectrum the y and z passes are FFTs over strided columns.
// rocFFT runs them as separate in-place passes (measured 27 B/cell each at 512^3 against 16 B/cell compulsory) with
// the eigenvalue division K12 as another full pass in between.  Along z the three operations
//      FFT_z  ->  phi_hat = -b_hat / (lx + ly + lz)  ->  IFFT_z
// act on one (kx, ky) column at a time, so they are fused here: the column is read once, transformed in LDS/registers,
// scaled, transformed back and written once (32 B per complex element instead of ~86).
//
// Structure: a workgroup owns CB consecutive kx (contiguous in memory) x the whole column of length N = 8*8*R3
// (R3 = 1, 2, 4, 8 -> N = 64 ... 512).  Thread (c, t) holds the 8 elements  t + (N/8) r  of column c, so global loads are
// CB*16 B contiguous runs.  Forward = radix-8 decimation in frequency in 3 stages with two LDS exchanges, output left in
// digit-reversed ("stage") order; inverse = the exact conjugate-transpose stage sequence, which consumes that order and
// returns natural order.  No reordering pass is ever needed: whoever needs the true wavenumber of a stored position (the
// eigenvalue tables) gets a permuted table from the host (ocn_colfft_position_to_wavenumber).
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ocn_internal.h"

namespace ocn {

struct cplx {
    double x, y;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.x, -a.y}; }
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ cplx mul_mi(cplx a)
{
    return INV ? cplx{-a.y, a.x} : cplx{a.y, -a.x};
}

// In-place radix-R butterfly y[q] = sum_r x[r] w^(r q), w = exp(-2 pi i / R) (INV: conjugate root). R in {2,4,8}.
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
    // y[q] = sum_r x[r] W8^(rq): even/odd split in r
    cplx e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    cplx o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    radix4<INV>(e0, e1, e2, e3);
    radix4<INV>(o0, o1, o2, o3);
    const double h = 0.70710678118654752440;
    // W8^1 = (1 - i)/sqrt2 (fwd), W8^2 = -i, W8^3 = (-1 - i)/sqrt2
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

// last stage: groups of R3 consecutive registers
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

struct ColFFTArgs {
    double *data;            // complex interleaved
    long long col_stride;    // elements between consecutive points of one column
    long long batch_stride;  // elements between consecutive batches
    int ncols;               // contiguous columns (stride 1) per batch
    int nbatch;
    const double *tw;        // W_N^j = exp(-2 pi i j / N), j = 0..N-1 (complex interleaved)
    // fused spectral solve (MODE 2): column index = kx + inner*ky;
    // phi = -b * scale / ((lx[kx] + ly[ky]) + lc[pos]); the zero mode (column 0, position 0) -> 0
    const double *lx, *ly;   // eigenvalues of the two column-index dimensions
    const double *lc;        // eigenvalue per STORED position along the column (stage order)
    double scale;
    int inner;               // number of kx per ky
};

// MODE 0: forward (natural -> stage order), 1: inverse (stage order -> natural), 2: forward, spectral solve, inverse
template <int N, int CB, int MODE>
__global__ __launch_bounds__(CB *(N / 8)) void colfft_kernel(ColFFTArgs a)
{
    constexpr int T = N / 8;    // threads per column
    constexpr int T2 = N / 64;  // length of the last stage = R3
    extern __shared__ double lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(lds_raw);          // N * CB exchange buffer, layout [e][c]
    cplx *W = reinterpret_cast<cplx *>(lds_raw) + N * CB;  // twiddle table
    const int tid = threadIdx.x, c = tid % CB, t = tid / CB;
    const int col0 = blockIdx.x * CB;
    const int batch = blockIdx.y;
    for (int j = tid; j < N; j += CB * T) W[j] = reinterpret_cast<const cplx *>(a.tw)[j];
    const bool active = (col0 + c) < a.ncols;
    cplx *base = reinterpret_cast<cplx *>(a.data) + (long long)batch * a.batch_stride + (col0 + (active ? c : 0));
    cplx x[8];
    const int q = t / T2, t2 = t % T2;  // stage-2 coordinates of this thread

    if (MODE == 0 || MODE == 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = active ? base[(long long)(t + T * r) * a.col_stride] : cplx{0, 0};
        __syncthreads();  // twiddles resident
        // ---- stage 1: radix-8 over r, twiddle W_N^(t q)
        radix8<false>(x);
#pragma unroll
        for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], W[(t * qq) % N]);
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) A[(qq * T + t) * CB + c] = x[qq];
        __syncthreads();
        // ---- stage 2: thread (q, t2) takes A1[q][t2 + T2 r2]
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = A[(q * T + t2 + T2 * r) * CB + c];
        radix8<false>(x);
        if (T2 > 1) {
#pragma unroll
            for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], W[(8 * t2 * qq) % N]);
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) A[((q * 8 + qq) * T2 + t2) * CB + c] = x[qq];
            __syncthreads();
            // ---- stage 3: thread t owns stored positions p = 8 t .. 8 t + 7 (G groups of T2)
#pragma unroll
            for (int m = 0; m < 8; ++m) x[m] = A[(8 * t + m) * CB + c];
            radix_last<T2, false>(x);
        }
        // x[m] is the spectrum at stored position p = 8 t + m
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = active ? base[(long long)(8 * t + m) * a.col_stride] : cplx{0, 0};
        __syncthreads();
    }

    if (MODE == 2) {
        const int col = col0 + c;
        const double lxy = active ? a.lx[col % a.inner] + a.ly[col / a.inner] : 1.0;
        const bool zero_col = (col == 0) && (batch == 0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int p = 8 * t + m;
            const double lam = lxy + a.lc[p];
            double s = -a.scale / lam;
            if (zero_col && p == 0) s = 0.0;  // position 0 is wavenumber 0
            x[m].x *= s;
            x[m].y *= s;
        }
    }

    if (MODE == 1 || MODE == 2) {
        // ---- inverse: conjugate-transpose stage sequence
        if (T2 > 1) {
            radix_last<T2, true>(x);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 8; ++m) A[(8 * t + m) * CB + c] = x[m];
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) x[qq] = A[((q * 8 + qq) * T2 + t2) * CB + c];
#pragma unroll
            for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], cconj(W[(8 * t2 * qq) % N]));
        }
        radix8<true>(x);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) A[(q * T + t2 + T2 * r) * CB + c] = x[r];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) x[qq] = A[(qq * T + t) * CB + c];
#pragma unroll
        for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], cconj(W[(t * qq) % N]));
        radix8<true>(x);
        if (active) {
#pragma unroll
            for (int r = 0; r < 8; ++r) base[(long long)(t + T * r) * a.col_stride] = x[r];
        }
    } else if (active) {
#pragma unroll
        for (int m = 0; m < 8; ++m) base[(long long)(8 * t + m) * a.col_stride] = x[m];
    }
}

bool colfft_supported(int N) { return N == 64 || N == 128 || N == 256 || N == 512; }

// stored position p -> true wavenumber k for the stage order produced by the forward kernel
int colfft_wavenumber(int N, int p)
{
    const int T2 = N / 64;
    if (T2 == 1) return (p / 8) + 8 * (p % 8);      // p = q*8 + q2 -> k = q + 8 q2
    const int g = p / T2, q3 = p % T2;              // p = (q*8 + q2)*T2 + q3
    return (g / 8) + 8 * (g % 8) + 64 * q3;
}

std::vector<double> colfft_twiddles(int N)
{
    std::vector<double> tw(2 * N);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int j = 0; j < N; ++j) {
        tw[2 * j] = (double)cosl(two_pi * j / N);
        tw[2 * j + 1] = (double)(-sinl(two_pi * j / N));
    }
    return tw;
}

template <int N, int CB>
static int launch_n(int mode, const ColFFTArgs &a, hipStream_t stream)
{
    const dim3 grid((a.ncols + CB - 1) / CB, a.nbatch), block(CB * (N / 8));
    const size_t lds = (size_t)(N * CB + N) * sizeof(cplx);
    if (mode == 0) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 0>), grid, block, lds, stream, a);
    } else if (mode == 1) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 1>), grid, block, lds, stream, a);
    } else {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 2>), grid, block, lds, stream, a);
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_colfft(int N, int mode, double *data, long long col_stride, long long batch_stride, int ncols, int nbatch,
                  const double *tw, const double *lx, const double *ly, const double *lc, double scale, int inner,
                  hipStream_t stream)
{
    ColFFTArgs a{data, col_stride, batch_stride, ncols, nbatch, tw, lx, ly, lc, scale, inner > 0 ? inner : 1};
    switch (N) {
        case 64: return launch_n<64, 16>(mode, a, stream);
        case 128: return launch_n<128, 16>(mode, a, stream);
        case 256: return launch_n<256, 8>(mode, a, stream);
        case 512: {
            static const int cb = getenv("OCN_COLFFT_CB") ? atoi(getenv("OCN_COLFFT_CB")) : 8;
            if (cb == 4) return launch_n<512, 4>(mode, a, stream);
            if (cb == 16) return launch_n<512, 16>(mode, a, stream);
            return launch_n<512, 8>(mode, a, stream);
        }
        default: set_error("column FFT length %d is not supported (64, 128, 256, 512)", N); return OCN_ERR_UNSUPPORTED;
    }
}

}  // namespace ocn
