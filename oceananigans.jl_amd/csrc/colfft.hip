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

#include <type_traits>

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

// ---- the exchange between the second and the third radix-8 stage without LDS (N = 512, CB = 8) ---------------------------------------
// There thread (c, t = 8 q + t2) sits in lane c + 8 t2 of wave q, and the exchange is an 8 x 8 transpose between the register index and
// t2 inside one wave: thread (q, t2) register qq  <->  thread (q, qq) register t2.  Three rounds, one per bit: register bit 2 against
// lane bit 5 and register bit 1 against lane bit 4 with gfx950's v_permlane32_swap / v_permlane16_swap (one instruction per dword, no
// select), register bit 0 against lane bit 3 with two bank-masked row_ror:8 DPP moves per dword.  80 VALU instructions per thread
// replace 8 ds_write_b128 + 8 ds_read_b128 + two barriers; the fused z pass is bound by the LDS pipeline, the
// VALU is busy a third of its cycles.  Pure data movement: results are bit-identical to the LDS exchange.
#ifndef OCN_FFT_LANE_TRANSPOSE
#define OCN_FFT_LANE_TRANSPOSE 1
#endif
template <int WHICH>  // 32, 16: permlane swaps; 8: DPP
__device__ __forceinline__ void lane_swap_dword(unsigned &a, unsigned &b)
{
    if (WHICH == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);  // lanes 32..63 of a <-> lanes 0..31 of b
        a = r[0];
        b = r[1];
    } else if (WHICH == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);  // odd rows of a <-> even rows of b
        a = r[0];
        b = r[1];
    } else {
        // row_ror:8 = lane i reads lane i ^ 8 of its row; bank_mask 0xC writes lanes 8..15 of each row, 0x3 lanes 0..7
        const unsigned na = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false);
        const unsigned nb = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);
        a = na;
        b = nb;
    }
}
template <int WHICH>
__device__ __forceinline__ void lane_swap(cplx &A, cplx &B)
{
    unsigned a0 = (unsigned)__double2loint(A.x), a1 = (unsigned)__double2hiint(A.x), a2 = (unsigned)__double2loint(A.y), a3 = (unsigned)__double2hiint(A.y);
    unsigned b0 = (unsigned)__double2loint(B.x), b1 = (unsigned)__double2hiint(B.x), b2 = (unsigned)__double2loint(B.y), b3 = (unsigned)__double2hiint(B.y);
    lane_swap_dword<WHICH>(a0, b0);
    lane_swap_dword<WHICH>(a1, b1);
    lane_swap_dword<WHICH>(a2, b2);
    lane_swap_dword<WHICH>(a3, b3);
    A = cplx{__hiloint2double((int)a1, (int)a0), __hiloint2double((int)a3, (int)a2)};
    B = cplx{__hiloint2double((int)b1, (int)b0), __hiloint2double((int)b3, (int)b2)};
}
__device__ __forceinline__ void lane_transpose8(cplx *x)
{
    lane_swap<32>(x[0], x[4]); lane_swap<32>(x[1], x[5]); lane_swap<32>(x[2], x[6]); lane_swap<32>(x[3], x[7]);
    lane_swap<16>(x[0], x[2]); lane_swap<16>(x[1], x[3]); lane_swap<16>(x[4], x[6]); lane_swap<16>(x[5], x[7]);
    lane_swap<8>(x[0], x[1]); lane_swap<8>(x[2], x[3]); lane_swap<8>(x[4], x[5]); lane_swap<8>(x[6], x[7]);
}

// N = 256, CB = 16: thread (c, t = 4 q + t2) sits in lane c + 16 t2 of wave q; the four threads of one column in a wave trade
// thread t2 register qq = 4 h1 + 2 h0 + l  <->  thread (h1 h0) register 4 l + t2: lane bit 5 against register bit 2, lane bit 4 against
// register bit 1 (permlane swaps only), then a renaming of the registers.
template <bool INV>
__device__ __forceinline__ void lane_transpose4(cplx *x)
{
    if (INV) {
        cplx y[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) y[r] = x[4 * (r & 1) + (r >> 1)];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = y[r];
    }
    lane_swap<32>(x[0], x[4]); lane_swap<32>(x[1], x[5]); lane_swap<32>(x[2], x[6]); lane_swap<32>(x[3], x[7]);
    lane_swap<16>(x[0], x[2]); lane_swap<16>(x[1], x[3]); lane_swap<16>(x[4], x[6]); lane_swap<16>(x[5], x[7]);
    if (!INV) {
        cplx y[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) y[4 * (r & 1) + (r >> 1)] = x[r];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = y[r];
    }
}

// Forward stage sequence of one column: x[r] = element t + (N/8) r (natural order) -> x[m] = spectrum at STORED position
// 8 t + m (stage order, colfft_wavenumber).  A = N*CB exchange buffer [e][c], W = W_N^j table (the first barrier also makes
// the caller's table writes visible).
// SPLIT: the one exchange that crosses waves moves the real parts, then the imaginary parts, through a buffer of N * CB DOUBLES (half
// the LDS: a third workgroup of the fused z pass fits a CU); same values, two more barriers per exchange.
#ifndef OCN_FFT_SPLIT_EXCHANGE
#define OCN_FFT_SPLIT_EXCHANGE 0
#endif
template <int N, int CB, bool SPLIT = false>
__device__ __forceinline__ void fft_fwd_stages(cplx *x, cplx *A, const cplx *W, int c, int t)
{
    constexpr int T = N / 8, T2 = N / 64;
    const int q = t / T2, t2 = t % T2;
    __syncthreads();  // twiddles resident
    // ---- stage 1: radix-8 over r, twiddle W_N^(t q)
    radix8<false>(x);
#pragma unroll
    for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], W[(t * qq) % N]);
    if constexpr (SPLIT) {
        double *Ad = reinterpret_cast<double *>(A);
        double xr[8];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) Ad[(qq * T + t) * CB + c] = x[qq].x;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) xr[r] = Ad[(q * T + t2 + T2 * r) * CB + c];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) Ad[(qq * T + t) * CB + c] = x[qq].y;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = cplx{xr[r], Ad[(q * T + t2 + T2 * r) * CB + c]};
    } else {
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) A[(qq * T + t) * CB + c] = x[qq];
        __syncthreads();
        // ---- stage 2: thread (q, t2) takes A1[q][t2 + T2 r2]
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = A[(q * T + t2 + T2 * r) * CB + c];
    }
    radix8<false>(x);
    if (T2 > 1) {
#pragma unroll
        for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], W[(8 * t2 * qq) % N]);
        // ---- stage 3: thread t owns stored positions p = 8 t .. 8 t + 7 (G groups of T2)
        if constexpr (T2 == 8 && CB == 8 && OCN_FFT_LANE_TRANSPOSE) {
            lane_transpose8(x);  // (the callers' next write into the exchange buffer comes after a barrier of their own)
        } else if constexpr (T2 == 4 && CB == 16 && OCN_FFT_LANE_TRANSPOSE) {
            lane_transpose4<false>(x);
        } else {
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) A[((q * 8 + qq) * T2 + t2) * CB + c] = x[qq];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 8; ++m) x[m] = A[(8 * t + m) * CB + c];
        }
        radix_last<T2, false>(x);
    }
}

// Inverse stage sequence: x[m] at stored position 8 t + m -> x[r] = (unnormalised) element t + (N/8) r in natural order.
template <int N, int CB, bool SPLIT = false>
__device__ __forceinline__ void fft_inv_stages(cplx *x, cplx *A, const cplx *W, int c, int t)
{
    constexpr int T = N / 8, T2 = N / 64;
    const int q = t / T2, t2 = t % T2;
    // ---- inverse: conjugate-transpose stage sequence
    if (T2 > 1) {
        radix_last<T2, true>(x);
        if constexpr (T2 == 8 && CB == 8 && OCN_FFT_LANE_TRANSPOSE) {
            lane_transpose8(x);
        } else if constexpr (T2 == 4 && CB == 16 && OCN_FFT_LANE_TRANSPOSE) {
            lane_transpose4<true>(x);
        } else {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 8; ++m) A[(8 * t + m) * CB + c] = x[m];
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) x[qq] = A[((q * 8 + qq) * T2 + t2) * CB + c];
        }
#pragma unroll
        for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], cconj(W[(8 * t2 * qq) % N]));
    }
    radix8<true>(x);
    __syncthreads();
    if constexpr (SPLIT) {
        double *Ad = reinterpret_cast<double *>(A);
        double xr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) Ad[(q * T + t2 + T2 * r) * CB + c] = x[r].x;
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) xr[qq] = Ad[(qq * T + t) * CB + c];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) Ad[(q * T + t2 + T2 * r) * CB + c] = x[r].y;
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) x[qq] = cplx{xr[qq], Ad[(qq * T + t) * CB + c]};
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) A[(q * T + t2 + T2 * r) * CB + c] = x[r];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) x[qq] = A[(qq * T + t) * CB + c];
    }
#pragma unroll
    for (int qq = 1; qq < 8; ++qq) x[qq] = cmul(x[qq], cconj(W[(t * qq) % N]));
    radix8<true>(x);
}

// MI355X dispatches consecutive workgroup ids round-robin over its 8 XCDs, each with its own L2.  Column blocks that are neighbours
// in memory share the cache lines their runs straddle (rows of Nx/2 + 1 = 257 complex numbers or of haloed reals are never line-
// aligned: a 128-byte run touches two lines, both shared with a neighbour), so with the plain mapping every such line is fetched into two
// L2s and written back from two as partial lines.  Hardware workgroup b (XCD b % 8) takes the logical block start_(b % 8) + b / 8:
// each XCD walks its own contiguous range of blocks (x-fastest) and meets its neighbours' lines in its own L2.  A bijection for any n.
__device__ __forceinline__ void xcd_block(int on, unsigned &bx, unsigned &by)
{
    bx = blockIdx.x; by = blockIdx.y;
    if (!on) return;
    const unsigned nx = gridDim.x, n = nx * gridDim.y;
    const unsigned b = bx + nx * by;
    const unsigned q = b & 7u, chunk = n >> 3, rem = n & 7u;
    const unsigned logical = q * chunk + (q < rem ? q : rem) + (b >> 3);
    bx = logical % nx;
    by = logical / nx;
}
static int fft_xcd_remap()
{
    static const int v = getenv("OCN_FFT_XCD") ? atoi(getenv("OCN_FFT_XCD")) : 1;
    return v;
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
    int zero_mode;           // MODE 2: this launch holds the (0,0,0) mode (column 0 of batch 0, position 0) and zeroes it
    int xcd;                 // XCD-contiguous block order (xcd_block)
    // MODE 4 as the LAST pass of a solve on real pairs: the result goes straight into the (haloed) pressure field instead of back into the
    // spectrum -- element `dst` of column col holds the x-adjacent cells (2 ip, 2 ip + 1): pdim = 1: (ip, j, k) = (col, dst, batch),
    // pdim = 2: (col % inner, col / inner, dst); preal + p0 is the first interior cell, ps2 / ps3 its row / plane strides.  NULL: in place.
    double *preal;
    long long p0, ps2, ps3;
    int pdim;
    const double *lx2;       // MODE 5: eigenvalue per NATURAL wavenumber along the column (lc holds the cosine-transform twiddles there)
};

template <int N>
__device__ __forceinline__ int stage_wavenumber(int p);

// MODE 0: forward (natural -> stage order), 1: inverse (stage order -> natural), 2: forward, spectral solve, inverse;
// MODE 3 / 4: the cosine transforms of the general Poisson solver (REDFT10 / REDFT01 / 2N of the real and of the imaginary part of every
// column, Makhoul 1980: poisson.hip dct_shuffle_kernel) with their permutation and twiddle passes INSIDE the column transform -- 3: gather on
// load (v[n] = x[2n], v[N-1-n] = x[2n+1]), FFT, then X[k] = Re(w_k (V[k] + conj V[N-k])) + i Im(w_k (V[k] - conj V[N-k])) through one more
// LDS exchange, stored in NATURAL wavenumber order; 4: V[k] = conj(w_k) (X[k] - i X[N-k]) / 2 from the natural-order column in LDS, inverse
// FFT scaled by a.scale, scatter on store.  a.lc = w_k = e^{-i pi k / 2N}, k < N (complex, natural order).  One pass each instead of three.
template <int N, int CB, int MODE>
constexpr bool colfft_split = (OCN_FFT_SPLIT_EXCHANGE != 0) && (OCN_FFT_LANE_TRANSPOSE != 0) && MODE == 2 && N == 512 && CB == 8;
template <int N, int CB, int MODE>
__global__ __launch_bounds__(CB *(N / 8), (colfft_split<N, CB, MODE> ? 6 : 1)) void colfft_kernel(ColFFTArgs a)
{
    constexpr int T = N / 8;    // threads per column
    constexpr bool SPLIT = colfft_split<N, CB, MODE>;
    extern __shared__ double lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(lds_raw);          // N * CB exchange buffer, layout [e][c] (SPLIT: of doubles)
    cplx *W = reinterpret_cast<cplx *>(lds_raw) + (SPLIT ? N * CB / 2 : N * CB);  // twiddle table
    const int tid = threadIdx.x, c = tid % CB, t = tid / CB;
    unsigned lbx, lby;
    xcd_block(a.xcd, lbx, lby);
    const int col0 = lbx * CB;
    const int batch = lby;
    for (int j = tid; j < N; j += CB * T) W[j] = reinterpret_cast<const cplx *>(a.tw)[j];
    const bool active = (col0 + c) < a.ncols;
    cplx *base = reinterpret_cast<cplx *>(a.data) + (long long)batch * a.batch_stride + (col0 + (active ? c : 0));
    cplx x[8];
    constexpr int HALF = (N + 1) / 2;

    if (MODE == 3 || MODE == 5) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int e = t + T * r, src = e < HALF ? 2 * e : 2 * (N - 1 - e) + 1;
            x[r] = active ? base[(long long)src * a.col_stride] : cplx{0, 0};
        }
        fft_fwd_stages<N, CB>(x, A, W, c, t);
        __syncthreads();  // the exchange buffer is free: the spectrum by natural wavenumber
#pragma unroll
        for (int m = 0; m < 8; ++m) A[stage_wavenumber<N>(8 * t + m) * CB + c] = x[m];
        __syncthreads();
        const cplx *wd = reinterpret_cast<const cplx *>(a.lc);
        if (MODE == 3) {
            if (active) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = t + T * r;
                    const cplx va = A[k * CB + c], vb = A[((N - k) % N) * CB + c], wk = wd[k];
                    const double sr = va.x + vb.x, si = va.y - vb.y, dr = va.x - vb.x, di = va.y + vb.y;
                    base[(long long)k * a.col_stride] = cplx{wk.x * sr - wk.y * si, wk.x * di + wk.y * dr};
                }
            }
            return;
        }
        // MODE 5: X[k] of both parts, -X / ((λx + λy) + λz[k]) (the zero mode := 0), back into the exchange buffer by natural wavenumber
        {
            const int col = col0 + c;
            const double lxy = active ? a.lx[col % a.inner] + a.ly[col / a.inner] : 1.0;
            const bool zero_col = a.zero_mode && (col == 0) && (batch == 0);
            cplx y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = t + T * r;
                const cplx va = A[k * CB + c], vb = A[((N - k) % N) * CB + c], wk = wd[k];
                const double sr = va.x + vb.x, si = va.y - vb.y, dr = va.x - vb.x, di = va.y + vb.y;
                double sc = -1.0 / (lxy + a.lx2[k]);
                if (zero_col && k == 0) sc = 0.0;
                y[r] = cplx{(wk.x * sr - wk.y * si) * sc, (wk.x * di + wk.y * dr) * sc};
            }
            __syncthreads();  // every V[k], V[N - k] has been read
#pragma unroll
            for (int r = 0; r < 8; ++r) A[(t + T * r) * CB + c] = y[r];
            __syncthreads();
        }
    }
    if (MODE == 4 || MODE == 5) {
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = t + T * r;
                A[k * CB + c] = active ? base[(long long)k * a.col_stride] : cplx{0, 0};
            }
            __syncthreads();
        }
        const cplx *wd = reinterpret_cast<const cplx *>(a.lc);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = stage_wavenumber<N>(8 * t + m);
            const cplx va = A[k * CB + c], vb = k == 0 ? cplx{0, 0} : A[(N - k) * CB + c], wk = wd[k];
            const double zr = va.x + vb.y, zi = va.y - vb.x;
            x[m] = cplx{0.5 * (wk.x * zr + wk.y * zi), 0.5 * (wk.x * zi - wk.y * zr)};
        }
        fft_inv_stages<N, CB>(x, A, W, c, t);  // (its first write into the exchange buffer comes after a barrier)
        if (active && a.preal) {
            const int col = col0 + c;
            const long long ip = a.pdim == 1 ? col : col % a.inner, jk = a.pdim == 1 ? batch : col / a.inner;
            double *pr = a.preal + a.p0 + 2 * ip + (a.pdim == 1 ? a.ps3 : a.ps2) * jk;
            const long long ps = a.pdim == 1 ? a.ps2 : a.ps3;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int e = t + T * r, dst = e < HALF ? 2 * e : 2 * (N - 1 - e) + 1;
                pr[ps * dst] = x[r].x * a.scale;
                pr[ps * dst + 1] = x[r].y * a.scale;
            }
        } else if (active) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int e = t + T * r, dst = e < HALF ? 2 * e : 2 * (N - 1 - e) + 1;
                base[(long long)dst * a.col_stride] = cplx{x[r].x * a.scale, x[r].y * a.scale};
            }
        }
        return;
    }

    if (MODE == 0 || MODE == 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = active ? base[(long long)(t + T * r) * a.col_stride] : cplx{0, 0};
        fft_fwd_stages<N, CB, SPLIT>(x, A, W, c, t);
        // x[m] is the spectrum at stored position p = 8 t + m
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = active ? base[(long long)(8 * t + m) * a.col_stride] : cplx{0, 0};
        __syncthreads();
    }

    if (MODE == 2) {
        const int col = col0 + c;
        const double lxy = active ? a.lx[col % a.inner] + a.ly[col / a.inner] : 1.0;
        const bool zero_col = a.zero_mode && (col == 0) && (batch == 0);
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
        fft_inv_stages<N, CB, SPLIT>(x, A, W, c, t);
        if (MODE == 1 && a.scale != 1.0) {  // normalised inverse (the general solver's line transforms: 1 / N)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x[r].x *= a.scale;
                x[r].y *= a.scale;
            }
        }
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
    } else if (mode == 3) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 3>), grid, block, lds, stream, a);
    } else if (mode == 4) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 4>), grid, block, lds, stream, a);
    } else if (mode == 5) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 5>), grid, block, lds, stream, a);
    } else {
        // (measured and rejected, round 3: a persistent variant that requests the next column set's values before transforming the current
        //  one -- 2.93 ms per 512^3 solve at 166 VGPRs / one workgroup per CU, 3.17 ms with the registers capped for two, against 2.85 ms)
        const size_t lds2 = colfft_split<N, CB, 2> ? (size_t)(N * CB / 2 + N) * sizeof(cplx) : lds;
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_kernel<N, CB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL((colfft_kernel<N, CB, 2>), grid, block, lds2, stream, a);
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

static int launch_colfft_args(int N, int mode, const ColFFTArgs &a, hipStream_t stream);

int launch_colfft(int N, int mode, double *data, long long col_stride, long long batch_stride, int ncols, int nbatch,
                  const double *tw, const double *lx, const double *ly, const double *lc, double scale, int inner,
                  hipStream_t stream, int zero_mode)
{
    ColFFTArgs a{data, col_stride, batch_stride, ncols, nbatch, tw, lx, ly, lc, scale, inner > 0 ? inner : 1, zero_mode, fft_xcd_remap(),
                 nullptr, 0, 0, 0, 0, nullptr};
    return launch_colfft_args(N, mode, a, stream);
}

// MODE 5: REDFT10 along the column, -b / ((λx + λy) + λz) with the zero mode := 0, REDFT01 / 2N back -- the whole z part of a solve on a
// (Periodic, Periodic, Bounded) grid with a REGULAR z in ONE pass over the half spectrum (the cosine-transform twin of MODE 2; the Thomas
// sweep it replaces moves 88 B per element instead of 32).  lz: eigenvalues by natural wavenumber, wd: w_k = e^{-i pi k / 2N}.
int launch_colfft_dct_solve(int N, double *data, long long col_stride, int ncols, const double *tw, const double *wd, const double *lx,
                            const double *ly, const double *lz, double scale, int inner, hipStream_t stream)
{
    ColFFTArgs a{data, col_stride, 0, ncols, 1, tw, lx, ly, wd, scale, inner > 0 ? inner : 1, 1, fft_xcd_remap(), nullptr, 0, 0, 0, 0, lz};
    return launch_colfft_args(N, 5, a, stream);
}

// the inverse cosine transform (MODE 4) of real pairs as the last pass of a solve: written into the pressure field `p` (ColFFTArgs::preal)
int launch_colfft_dct_to_field(int N, double *data, long long col_stride, long long batch_stride, int ncols, int nbatch, const double *tw,
                               const double *wd, double scale, int inner, int pdim, double *p, long long p0, long long ps2, long long ps3,
                               hipStream_t stream)
{
    ColFFTArgs a{data, col_stride, batch_stride, ncols, nbatch, tw, nullptr, nullptr, wd, scale, inner > 0 ? inner : 1, 1, fft_xcd_remap(),
                 p, p0, ps2, ps3, pdim, nullptr};
    return launch_colfft_args(N, 4, a, stream);
}

static int launch_colfft_args(int N, int mode, const ColFFTArgs &a, hipStream_t stream)
{
    switch (N) {
        case 64: return launch_n<64, 16>(mode, a, stream);
        case 128: return launch_n<128, 16>(mode, a, stream);
        case 256: {
            static const int cb = getenv("OCN_COLFFT_CB") ? atoi(getenv("OCN_COLFFT_CB")) : 16;  // 0.376 vs 0.409 ms per 256^3 solve
            if (cb == 8) return launch_n<256, 8>(mode, a, stream);
            return launch_n<256, 16>(mode, a, stream);
        }
        case 512: {
            // 8 columns per workgroup (128-B runs, two workgroups per CU whose load / transform / store phases overlap) since the
            // XCD-contiguous block order keeps the lines two neighbouring blocks share in one L2: 2.72 ms per 512^3 solve against 2.89 ms
            // with 16 columns (256-B runs, one workgroup of 1024 threads per CU) and 2.80 with 16 for the plain passes + 8 for the fused z
            // pass (the round-3 default; before the block order 8 columns lost: 2.96 against 2.87)
            static const int cb = getenv("OCN_COLFFT_CB") ? atoi(getenv("OCN_COLFFT_CB")) : 8;
            // the fused FFT-solve-IFFT pass holds 94 VGPRs and works twice as long per byte: two 8-column workgroups per CU overlap one's
            // loads / stores with the other's arithmetic (512^3 step 27.15 vs 27.35 ms, same box)
            static const int cb2 = getenv("OCN_COLFFT_CB2") ? atoi(getenv("OCN_COLFFT_CB2")) : (getenv("OCN_COLFFT_CB") ? cb : 8);
            const int c = mode == 2 ? cb2 : cb;
            if (c == 4) return launch_n<512, 4>(mode, a, stream);
            if (c == 8) return launch_n<512, 8>(mode, a, stream);
            return launch_n<512, 16>(mode, a, stream);
        }
        default: set_error("column FFT length %d is not supported (64, 128, 256, 512)", N); return OCN_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------------
// Cosine transforms along x of a REAL array (Nx, Ny, Nz), Ny even, in place: the closed boxes of the general solver (poisson.hip,
// `gallreal`).  The complex line (jp, kz) is the pair of real rows j = 2 jp (real part) and 2 jp + 1 (imaginary part) -- a cosine transform
// maps reals to reals, so two rows ride on one complex FFT (poisson.hip dct_rowpair_kernel states the four passes this replaces).
// MODE 5: REDFT10 (gather on load, FFT, twiddle through LDS, natural wavenumbers);  6: REDFT01 / 2N (pre-twiddle from LDS, inverse FFT, scatter
// on store);  7: forward, -b / (λx + λy + λz [- m]) with the mode (1, 1, 1) := 0 iff m === 0, inverse -- the whole x part of a solve in ONE pass
// over the array (16 B per cell instead of seven passes and 112).  Thread (c, t): line c of the workgroup's CB, elements t + (N / 8) r.
// MODE 8: a PERIODIC x (channels, `gpacked` without another Periodic direction): the row pair z = a + i b through one complex FFT,
// Z = A + i B with Hermitian A, B; the division scales A by s0(k) = -1 / λ(k, j0, kz) and B by s1(k) (real, even in k), so
// Z'[k] = (s0 + s1) / 2 Z[k] + (s0 - s1) / 2 conj Z[N - k], and the inverse FFT returns the two solved rows: real-to-complex transform,
// division and complex-to-real transform (three passes, 64 B per cell) in one.
// ---------------------------------------------------------------------------------------------------
struct RowDCTArgs {
    double *data;
    int Nyp, Nz;             // row pairs per plane, planes
    const double *tw;        // W_N^j
    const double *wd;        // w_k = e^{-i pi k / 2N} (complex, natural order)
    const double *lx, *ly, *lz;  // MODE 7: eigenvalues by stored position along x (natural), y, z
    double scale;            // of the inverse: 1 / N
    double shift;
    int shifted;
    int xcd;
};

template <int N, int CB, int MODE>
__global__ __launch_bounds__(CB *(N / 8)) void rowdct_kernel(RowDCTArgs a)
{
    constexpr int T = N / 8, HALF = N / 2;
    extern __shared__ double lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(lds_raw);
    cplx *W = reinterpret_cast<cplx *>(lds_raw) + N * CB;
    const int tid = threadIdx.x, c = tid % CB, t = tid / CB;
    unsigned lbx, lby;
    xcd_block(a.xcd, lbx, lby);
    const int jp0 = lbx * CB, kz = lby;
    for (int j = tid; j < N; j += CB * T) W[j] = reinterpret_cast<const cplx *>(a.tw)[j];
    const bool active = (jp0 + c) < a.Nyp;
    const int jp = active ? jp0 + c : jp0;
    double *re = a.data + (long long)N * (2 * jp + (long long)2 * a.Nyp * kz), *im = re + N;
    const cplx *wd = reinterpret_cast<const cplx *>(a.wd);
    cplx x[8];

    if (MODE == 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = active ? cplx{re[t + T * r], im[t + T * r]} : cplx{0, 0};
        fft_fwd_stages<N, CB>(x, A, W, c, t);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) A[stage_wavenumber<N>(8 * t + m) * CB + c] = x[m];
        __syncthreads();
        const double ly0 = a.ly[2 * jp], ly1 = a.ly[2 * jp + 1], lz = a.lz[kz];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = stage_wavenumber<N>(8 * t + m);
            const cplx za = A[k * CB + c], zb = A[((N - k) % N) * CB + c];
            const double lxk = a.lx[k];
            double l0 = (lxk + ly0) + lz, l1 = (lxk + ly1) + lz;
            if (a.shifted) { l0 = l0 - a.shift; l1 = l1 - a.shift; }
            double s0 = -1.0 / l0;
            const double s1 = -1.0 / l1;
            if (!a.shifted && k == 0 && jp == 0 && kz == 0) s0 = 0.0;
            const double p = 0.5 * (s0 + s1), q = 0.5 * (s0 - s1);
            x[m] = cplx{p * za.x + q * zb.x, p * za.y - q * zb.y};
        }
        fft_inv_stages<N, CB>(x, A, W, c, t);  // (its first write into the exchange buffer comes after a barrier)
        if (active) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                re[t + T * r] = x[r].x * a.scale;
                im[t + T * r] = x[r].y * a.scale;
            }
        }
        return;
    }
    if (MODE == 5 || MODE == 7) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int e = t + T * r, src = e < HALF ? 2 * e : 2 * (N - 1 - e) + 1;
            x[r] = active ? cplx{re[src], im[src]} : cplx{0, 0};
        }
        fft_fwd_stages<N, CB>(x, A, W, c, t);
        __syncthreads();  // the exchange buffer is free: the spectrum by natural wavenumber
#pragma unroll
        for (int m = 0; m < 8; ++m) A[stage_wavenumber<N>(8 * t + m) * CB + c] = x[m];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = t + T * r;
            const cplx va = A[k * CB + c], vb = A[((N - k) % N) * CB + c], wk = wd[k];
            const double sr = va.x + vb.x, si = va.y - vb.y, dr = va.x - vb.x, di = va.y + vb.y;
            x[r] = cplx{wk.x * sr - wk.y * si, wk.x * di + wk.y * dr};
        }
        if (MODE == 5) {
            if (active) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    re[t + T * r] = x[r].x;
                    im[t + T * r] = x[r].y;
                }
            }
            return;
        }
        // (the association of spectral_solve_real_kernel: (λx + λy) + λz)
        const double ly0 = a.ly[2 * jp], ly1 = a.ly[2 * jp + 1], lz = a.lz[kz];
        __syncthreads();  // every V[k], V[N - k] has been read
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = t + T * r;
            const double lxk = a.lx[k];
            double l0 = (lxk + ly0) + lz, l1 = (lxk + ly1) + lz;
            if (a.shifted) { l0 = l0 - a.shift; l1 = l1 - a.shift; }
            cplx v{-x[r].x / l0, -x[r].y / l1};
            if (!a.shifted && k == 0 && jp == 0 && kz == 0) v.x = 0.0;
            A[k * CB + c] = v;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = t + T * r;
            A[k * CB + c] = active ? cplx{re[k], im[k]} : cplx{0, 0};
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = stage_wavenumber<N>(8 * t + m);
        const cplx va = A[k * CB + c], vb = k == 0 ? cplx{0, 0} : A[(N - k) * CB + c], wk = wd[k];
        const double zr = va.x + vb.y, zi = va.y - vb.x;
        x[m] = cplx{0.5 * (wk.x * zr + wk.y * zi), 0.5 * (wk.x * zi - wk.y * zr)};
    }
    fft_inv_stages<N, CB>(x, A, W, c, t);  // (its first write into the exchange buffer comes after a barrier)
    if (active) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int e = t + T * r, dst = e < HALF ? 2 * e : 2 * (N - 1 - e) + 1;
            re[dst] = x[r].x * a.scale;
            im[dst] = x[r].y * a.scale;
        }
    }
}

template <int N, int CB>
static int launch_rowdct_n(int mode, const RowDCTArgs &a, hipStream_t stream)
{
    const dim3 grid((a.Nyp + CB - 1) / CB, a.Nz), block(CB * (N / 8));
    const size_t lds = (size_t)(N * CB + N) * sizeof(cplx);
    if (mode == 5) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)rowdct_kernel<N, CB, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rowdct_kernel<N, CB, 5>), grid, block, lds, stream, a);
    } else if (mode == 6) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)rowdct_kernel<N, CB, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rowdct_kernel<N, CB, 6>), grid, block, lds, stream, a);
    } else if (mode == 8) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)rowdct_kernel<N, CB, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rowdct_kernel<N, CB, 8>), grid, block, lds, stream, a);
    } else {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)rowdct_kernel<N, CB, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rowdct_kernel<N, CB, 7>), grid, block, lds, stream, a);
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

// mode 5 forward, 6 inverse, 7 forward + division by the eigenvalues + inverse (cosine transforms), 8 the same around the FFT of a Periodic x,
// of the x lines of the real array `data` (Nx, Ny, Nz), Ny even
int launch_rowdct(int Nx, int Ny, int Nz, int mode, double *data, const double *tw, const double *wd, const double *lx, const double *ly,
                  const double *lz, double shift, int shifted, hipStream_t stream)
{
    if (Ny % 2 != 0 || mode < 5 || mode > 8) {
        set_error("row transform: Ny = %d must be even and the mode (%d) 5 ... 8", Ny, mode);
        return OCN_ERR_INVALID_ARGUMENT;
    }
    RowDCTArgs a{data, Ny / 2, Nz, tw, wd, lx, ly, lz, 1.0 / Nx, shift, shifted, fft_xcd_remap()};
    switch (Nx) {
        case 64: return launch_rowdct_n<64, 16>(mode, a, stream);
        case 128: return launch_rowdct_n<128, 16>(mode, a, stream);
        case 256: return launch_rowdct_n<256, 8>(mode, a, stream);
        case 512: return launch_rowdct_n<512, 8>(mode, a, stream);
        default: set_error("row cosine transform length %d is not supported (64, 128, 256, 512)", Nx); return OCN_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------------
// Slab pipeline of the distributed solver (poisson.hip, ocn_dist_poisson "fast" mode): x is partitioned, so the local
// directions y and z are transformed first -- y as a REAL transform of strided columns, z as a column FFT that writes straight
// into the all-to-all send layout -- and x after the exchange with the fused FFT -> divide -> IFFT column kernel above.
//
//   rhs  real    [xl + nx (y + Ny z)]
//   A1   complex [ky + NyH (xl + nx z)],  NyH = Ny/2 + 1            (ky fastest: the transposition happens in LDS here)
//   send complex [d][ky + NyH (pz_l + cz xl)],  stored z position pz = d cz + pz_l, cz = Nz / R   (chunk d goes to rank d)
//   recv complex [xg S + (ky + NyH pz_l)],  S = NyH cz, xg = r nx + xl    (x is a clean strided column after the exchange)
// ---------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int stage_wavenumber(int p)
{
    constexpr int T2 = N / 64;
    if (T2 == 1) return (p / 8) + 8 * (p % 8);
    const int g = p / T2, q3 = p % T2;
    return (g / 8) + 8 * (g % 8) + 64 * q3;
}

struct RealYArgs {
    const double *rhs;   // forward input (or NULL: the source term is evaluated from u, v, w)
    GridDev g;           // fused source term: rhs = divᶜᶜᶜ(u, v, w) / dt  (solve_for_pressure.jl:21-27)
    const double *u, *v, *w;
    double dt;
    double *spec;        // A1 (forward output / inverse input)
    double *p;           // inverse output: first interior element of the haloed pressure field
    long long p_s2, p_s3;
    int nx, Nz;
    const double *twH;   // W_H^j, j < H
    const double *twN;   // W_Ny^k, k <= H
    // layout of the half spectrum: kc == 0: A1[ky + NyH (xl + nx z)];  kc > 0 (tridiagonal flavour, partitioned by ky in R chunks
    // of kc): [(ky / kc) chunk + (ky % kc) + kc (z + Nz xl)], i.e. the all-to-all layout itself
    int kc;
    long long chunk;
    int scale_dz;        // source term times Δzᶜ (solve_for_pressure.jl:33-38)
    double scale;        // inverse: applied to the real output
    int xcd;             // XCD-contiguous block order (xcd_block)
    __device__ __forceinline__ long long spec_at(int k, int xl, int z, int NYH) const
    {
        if (kc == 0) return k + (long long)NYH * (xl + (long long)nx * z);
        return (long long)(k / kc) * chunk + (k % kc) + (long long)kc * (z + (long long)Nz * xl);
    }
    // the same with the layout known at compile time (the run-time division by kc stays out of the unrolled A1 loops)
    template <bool A1>
    __device__ __forceinline__ long long spec_at_t(int k, int xl, int z, int NYH) const
    {
        if (A1) return k + (long long)NYH * (xl + (long long)nx * z);
        return (long long)(k / kc) * chunk + (k % kc) + (long long)kc * (z + (long long)Nz * xl);
    }
};

// Real FFT of length Ny = 2H along y for CB adjacent x columns of one z plane: z_m = s[2m] + i s[2m+1], complex FFT of length H,
// split step X[k] = E[k] + W_Ny^k O[k] with E = (Z[k] + conj Z[H-k]) / 2, O = (Z[k] - conj Z[H-k]) / (2i), k = 0..H; the H + 1
// outputs of a column are written as one contiguous run (ky fastest).

template <int H, int CB, bool SRC>
__global__ __launch_bounds__(CB *(H / 8)) void realfft_y_fwd_kernel(RealYArgs a)
{
    constexpr int T = H / 8, NYH = H + 1, NT = CB * T;
    extern __shared__ double lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(lds_raw);  // H*CB exchange buffer, then Zs[c][NYH] (natural wavenumbers)
    cplx *W = A + CB * NYH;
    cplx *WN = W + H;
    const int tid = threadIdx.x, c = tid % CB, t = tid / CB;
    unsigned lbx, lby;
    xcd_block(a.xcd, lbx, lby);
    const int col0 = lbx * CB, z = lby;
    for (int j = tid; j < H; j += NT) W[j] = reinterpret_cast<const cplx *>(a.twH)[j];
    for (int j = tid; j < NYH; j += NT) WN[j] = reinterpret_cast<const cplx *>(a.twN)[j];
    const bool active = (col0 + c) < a.nx;
    cplx x[8];
    if (SRC) {
        const Lay L = make_lay(a.g, OCN_LOC_CCC);  // x, y, z Periodic: one layout for u, v, w
        const int i = col0 + (active ? c : 0) + 1, k = z + 1;
        // The 11 values of every pair of rows are LOADED for four pairs at a time (44 loads in flight per thread) before any of them is
        // used: left to itself the compiler interleaves each divergence with its own loads and waits for memory two dozen times per thread
        // (`s_waitcnt vmcnt(0)` after every 2 - 5 loads; the kernel ran at 3.7 TB/s with 60 VGPRs of the 128 its occupancy leaves).  Same
        // expression as kernels.hip div_ccc / source_term_kernel (x, y Periodic; every field shares one set of strides).
        const GridDev &g = a.g;
        const double dzc = g.dzc ? uniform_load(g.dzc, k + g.Hz - 1) : g.dz;
        const double Ax = g.dy * dzc, Ay = g.dx * dzc, Az = g.dx * g.dy;
        const double rV = 1 / (Az * dzc);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double uu[4][4], vv[4][3], ww[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = t + T * (4 * h + q);
                const long long o = at(L, i, 2 * m + 1, k), o2 = o + L.s2;
                uu[q][0] = a.u[o]; uu[q][1] = a.u[o + 1]; uu[q][2] = a.u[o2]; uu[q][3] = a.u[o2 + 1];
                vv[q][0] = a.v[o]; vv[q][1] = a.v[o2]; vv[q][2] = a.v[o2 + L.s2];
                ww[q][0] = a.w[o]; ww[q][1] = a.w[o + L.s3]; ww[q][2] = a.w[o2]; ww[q][3] = a.w[o2 + L.s3];
            }
            OCN_ISSUE_LOADS_HERE();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double sv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double dxu = Ax * uu[q][2 * e + 1] - Ax * uu[q][2 * e];
                    const double dyv = Ay * vv[q][e + 1] - Ay * vv[q][e];
                    const double dzw = Az * ww[q][2 * e + 1] - Az * ww[q][2 * e];
                    const double d = rV * ((dxu + dyv) + dzw);
                    sv[e] = a.scale_dz ? (dzc * d) / a.dt : d / a.dt;
                }
                x[4 * h + q] = active ? cplx{sv[0], sv[1]} : cplx{0, 0};
            }
        }
    } else {
        const double *src = a.rhs + (col0 + (active ? c : 0)) + (long long)a.nx * ((long long)(2 * H) * z);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const long long m = t + T * r;
            x[r] = active ? cplx{src[(2 * m) * a.nx], src[(2 * m + 1) * a.nx]} : cplx{0, 0};
        }
    }
    fft_fwd_stages<H, CB>(x, A, W, c, t);
    __syncthreads();  // the exchange buffer is free: reuse it by natural wavenumber
#pragma unroll
    for (int m = 0; m < 8; ++m) A[c * NYH + stage_wavenumber<H>(8 * t + m)] = x[m];
    __syncthreads();
    cplx *out = reinterpret_cast<cplx *>(a.spec);
    constexpr int NIO = (CB * NYH + NT - 1) / NT;
    auto store = [&](auto a1) {
#pragma unroll
        for (int q = 0; q < NIO; ++q) {
            const int idx = tid + q * NT;
            if (idx >= CB * NYH) break;
            const int cc = idx / NYH, k = idx % NYH;
            if (col0 + cc >= a.nx) continue;
            const cplx Zk = A[cc * NYH + (k == H ? 0 : k)], Zc = cconj(A[cc * NYH + ((H - k) % H)]);
            const cplx E = {0.5 * (Zk.x + Zc.x), 0.5 * (Zk.y + Zc.y)};
            const cplx D = {0.5 * (Zk.x - Zc.x), 0.5 * (Zk.y - Zc.y)};
            const cplx O = {D.y, -D.x};  // D / i
            out[a.template spec_at_t<decltype(a1)::value>(k, col0 + cc, z, NYH)] = cadd(E, cmul(WN[k], O));
        }
    };
    if (a.kc == 0) store(std::true_type{});
    else store(std::false_type{});
}

// Inverse of the above: Hermitian half spectrum (ky fastest) -> real rows of the haloed pressure field (unnormalised: x H).
template <int H, int CB>
__global__ __launch_bounds__(CB *(H / 8)) void realfft_y_inv_kernel(RealYArgs a)
{
    constexpr int T = H / 8, NYH = H + 1, NT = CB * T;
    extern __shared__ double lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(lds_raw);
    cplx *W = A + CB * NYH;
    cplx *WN = W + H;
    const int tid = threadIdx.x, c = tid % CB, t = tid / CB;
    unsigned lbx, lby;
    xcd_block(a.xcd, lbx, lby);
    const int col0 = lbx * CB, z = lby;
    for (int j = tid; j < H; j += NT) W[j] = reinterpret_cast<const cplx *>(a.twH)[j];
    for (int j = tid; j < NYH; j += NT) WN[j] = reinterpret_cast<const cplx *>(a.twN)[j];
    const cplx *in = reinterpret_cast<const cplx *>(a.spec);
    {   // every load of the thread in flight before the first LDS store (the rolled loop waited for each one: one load in flight per thread)
        constexpr int NIO = (CB * NYH + NT - 1) / NT;
        cplx tmp[NIO];
        auto load = [&](auto a1) {
#pragma unroll
            for (int q = 0; q < NIO; ++q) {
                const int idx = tid + q * NT;
                const int cc = idx / NYH, k = idx % NYH;
                tmp[q] = (idx < CB * NYH && col0 + cc < a.nx) ? in[a.template spec_at_t<decltype(a1)::value>(k, col0 + cc, z, NYH)] : cplx{0, 0};
            }
        };
        if (a.kc == 0) load(std::true_type{});
        else load(std::false_type{});
        OCN_ISSUE_LOADS_HERE();
#pragma unroll
        for (int q = 0; q < NIO; ++q) {
            const int idx = tid + q * NT;
            if (idx < CB * NYH) A[idx] = tmp[q];
        }
    }
    __syncthreads();
    cplx x[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = stage_wavenumber<H>(8 * t + m);
        const cplx Xk = A[c * NYH + k], Xc = cconj(A[c * NYH + (H - k)]);
        const cplx E = {0.5 * (Xk.x + Xc.x), 0.5 * (Xk.y + Xc.y)};
        const cplx WO = {0.5 * (Xk.x - Xc.x), 0.5 * (Xk.y - Xc.y)};
        const cplx O = cmul(cconj(WN[k]), WO);
        x[m] = cplx{E.x - O.y, E.y + O.x};  // Z[k] = E + i O
    }
    __syncthreads();  // A becomes the exchange buffer
    fft_inv_stages<H, CB>(x, A, W, c, t);
    if ((col0 + c) < a.nx) {
        double *dst = a.p + (col0 + c) + a.p_s3 * z;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const long long m = t + T * r;
            dst[(2 * m) * a.p_s2] = x[r].x * a.scale;
            dst[(2 * m + 1) * a.p_s2] = x[r].y * a.scale;
        }
    }
}

// element address of (column, position along the column) in a two-level layout
struct ColMap {
    long long div, hi;   // column:   (col % div) + (col / div) * hi
    long long pdiv;      // position: (e % pdiv) * s1 + (e / pdiv) * s2
    long long s1, s2;
    __device__ __forceinline__ long long at(long long col, long long e) const
    {
        return (col % div) + (col / div) * hi + (e % pdiv) * s1 + (e / pdiv) * s2;
    }
};
struct ColIOArgs {
    const double *in;
    double *out;
    ColMap im, om;
    int ncols;
    const double *tw;
};

// Out-of-place column FFT between two layouts.  MODE 0: natural -> stage order, 1: stage order -> natural.
template <int N, int CB, int MODE>
__global__ __launch_bounds__(CB *(N / 8)) void colfft_io_kernel(ColIOArgs a)
{
    constexpr int T = N / 8;
    extern __shared__ double lds_raw[];
    cplx *A = reinterpret_cast<cplx *>(lds_raw);
    cplx *W = A + N * CB;
    const int tid = threadIdx.x, c = tid % CB, t = tid / CB;
    const long long col = (long long)blockIdx.x * CB + c;
    for (int j = tid; j < N; j += CB * T) W[j] = reinterpret_cast<const cplx *>(a.tw)[j];
    const bool active = col < a.ncols;
    const cplx *in = reinterpret_cast<const cplx *>(a.in);
    cplx *out = reinterpret_cast<cplx *>(a.out);
    const long long cl = active ? col : 0;
    cplx x[8];
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = active ? in[a.im.at(cl, t + T * r)] : cplx{0, 0};
        fft_fwd_stages<N, CB>(x, A, W, c, t);
        if (active) {
#pragma unroll
            for (int m = 0; m < 8; ++m) out[a.om.at(cl, 8 * t + m)] = x[m];
        }
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = active ? in[a.im.at(cl, 8 * t + m)] : cplx{0, 0};
        __syncthreads();
        fft_inv_stages<N, CB>(x, A, W, c, t);
        if (active) {
#pragma unroll
            for (int r = 0; r < 8; ++r) out[a.om.at(cl, t + T * r)] = x[r];
        }
    }
}

template <int H, int CB>
static int launch_realy(int inverse, const RealYArgs &a, hipStream_t stream)
{
    const dim3 grid((a.nx + CB - 1) / CB, a.Nz), block(CB * (H / 8));
    const size_t lds = (size_t)(CB * (H + 1) + H + (H + 1)) * sizeof(cplx);
    if (!inverse && a.rhs) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)realfft_y_fwd_kernel<H, CB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((realfft_y_fwd_kernel<H, CB, false>), grid, block, lds, stream, a);
    } else if (!inverse) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)realfft_y_fwd_kernel<H, CB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((realfft_y_fwd_kernel<H, CB, true>), grid, block, lds, stream, a);
    } else {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)realfft_y_inv_kernel<H, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((realfft_y_inv_kernel<H, CB>), grid, block, lds, stream, a);
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

bool realfft_y_supported(int Ny) { return Ny % 2 == 0 && colfft_supported(Ny / 2); }

// real y transform of the slab: forward rhs -> A1, inverse A1 -> p (first interior element, row / plane strides p_s2 / p_s3)
int launch_realfft_y(int Ny, int inverse, const double *rhs, double *spec, double *p, long long p_s2, long long p_s3, int nx, int Nz,
                     const double *twH, const double *twN, hipStream_t stream, const ocn_grid *grid, const double *u, const double *v,
                     const double *w, double dt, int kc, long long chunk, int scale_dz, double scale)
{
    RealYArgs a{rhs, GridDev{}, u, v, w, dt, spec, p, p_s2, p_s3, nx, Nz, twH, twN, kc, chunk, scale_dz, scale, fft_xcd_remap()};
    if (grid) a.g = to_dev(*grid);
    if (!inverse && !rhs && !(grid && u && v && w)) {
        set_error("launch_realfft_y: the forward transform needs either rhs or (grid, u, v, w)");
        return OCN_ERR_INVALID_ARGUMENT;
    }
    switch (Ny / 2) {
        case 64: return launch_realy<64, 16>(inverse, a, stream);
        case 128: return launch_realy<128, 16>(inverse, a, stream);
        case 256: {  // (32 columns per workgroup, 256-byte runs, one workgroup per CU: no faster, round 4)
            static const int cb = getenv("OCN_REALY_CB") ? atoi(getenv("OCN_REALY_CB")) : 16;
            if (cb == 8) return launch_realy<256, 8>(inverse, a, stream);
            return launch_realy<256, 16>(inverse, a, stream);
        }
        case 512: return launch_realy<512, 8>(inverse, a, stream);
        default: set_error("real y transform of length %d is not supported (128, 256, 512, 1024)", Ny); return OCN_ERR_UNSUPPORTED;
    }
}

template <int N, int CB>
static int launch_io_n(int mode, const ColIOArgs &a, hipStream_t stream)
{
    const dim3 grid((a.ncols + CB - 1) / CB), block(CB * (N / 8));
    const size_t lds = (size_t)(N * CB + N) * sizeof(cplx);
    if (mode == 0) {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_io_kernel<N, CB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_io_kernel<N, CB, 0>), grid, block, lds, stream, a);
    } else {
        OCN_CHECK_HIP(hipFuncSetAttribute((const void *)colfft_io_kernel<N, CB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((colfft_io_kernel<N, CB, 1>), grid, block, lds, stream, a);
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

// z transform of the slab between A1 and the all-to-all layout (see the layout table above); inverse = 0: A1 -> send (stage
// order in z), 1: send -> A1 (natural z)
int launch_colfft_slab_z(int Nz, int inverse, const double *in, double *out, int nx, int NyH, int R, const double *tw, hipStream_t stream)
{
    const long long cz = Nz / R, S = (long long)NyH * cz, plane = (long long)NyH * nx;
    const long long big = 1LL << 62;
    const ColMap a1{big, 0, big, plane, 0};                  // col = ky + NyH xl (contiguous), z stride NyH nx
    const ColMap snd{NyH, S, cz, NyH, S * nx};               // ky + S xl;  (pz % cz) NyH + (pz / cz) S nx
    ColIOArgs a{in, out, inverse ? snd : a1, inverse ? a1 : snd, (int)plane, tw};
    switch (Nz) {
        case 64: return launch_io_n<64, 16>(inverse, a, stream);
        case 128: return launch_io_n<128, 16>(inverse, a, stream);
        case 256: {
            static const int cb = getenv("OCN_COLFFT_IO_CB") ? atoi(getenv("OCN_COLFFT_IO_CB")) : 8;
            return cb == 8 ? launch_io_n<256, 8>(inverse, a, stream) : launch_io_n<256, 16>(inverse, a, stream);
        }
        case 512: {
            static const int cb = getenv("OCN_COLFFT_IO_CB") ? atoi(getenv("OCN_COLFFT_IO_CB")) : 8;
            return cb == 8 ? launch_io_n<512, 8>(inverse, a, stream) : launch_io_n<512, 16>(inverse, a, stream);
        }
        default: set_error("column FFT length %d is not supported (64, 128, 256, 512)", Nz); return OCN_ERR_UNSUPPORTED;
    }
}

}  // namespace ocn
