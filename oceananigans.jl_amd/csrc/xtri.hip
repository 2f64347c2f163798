// xtri.hip -- the x direction of the distributed (slab-x) pressure solve WITHOUT transposes.
//
// The reference's DistributedFFTBasedPoissonSolver (distributed_fft_based_poisson_solver.jl:141-178) transposes the whole
// spectrum twice per solve (y-local <-> x-local all-to-all, distributed_transpose.jl:185-191) because it diagonalises the
// x direction with an FFT as well.  On xGMI every pair of GPUs shares ONE link, so those all-to-alls are per-link bound: at
// 512^3 a rank of a 2-GPU run sends 270 MB per transpose over a single link (~5 ms each, six per RK3 step), more than the
// whole local compute.
//
// After the local y and z transforms the operator is, for every (ky, kz), the periodic second-difference in x:
//      p[i-1] - (2 + mu) p[i] + p[i+1] = dx^2 F[i],     mu = dx^2 (ly[ky] + lz[kz]) >= 0,
// whose eigenvalues are exactly the lx[kx] the reference divides by (poisson_eigenvalues.jl:8-31): solving this cyclic
// tridiagonal system IS the reference's FFT_x -> divide -> IFFT_x, up to rounding.  It is solved by the partition (SPIKE /
// Wang) method: every rank forward-eliminates its local Toeplitz block T = tridiag(1, -(2 + mu), 1) (Thomas, in place) and obtains
// the first and last entry of the local solution g = T^-1 F from that one pass (xtri_forward_kernel), the ranks exchange only
// those (2 complex numbers per mode: 4 MB per rank at 512^3 instead of 2 x 135-540 MB), every rank solves the circulant
// 2R x 2R interface system of each mode redundantly (a length-R DFT over the ranks decouples it into 2 x 2 systems) and back-
// substitutes its block with the neighbours' values folded into the right-hand side (xtri_backward_kernel): two passes over the
// spectrum in all.  The spike vectors have the closed form
//      v_i = T^-1 e_1 = -(r^i - r^(2(n+1)-i)) / (1 - r^(2(n+1))),   w_i = T^-1 e_n = v_(n+1-i),   r + 1/r = 2 + mu, 0 < r < 1.
// The (ky, kz) = (0, 0) mode (mu = 0, singular: the mean of p is free) is the one line the reference zeroes at kx = 0
// (distributed_fft_based_poisson_solver.jl:162-164): its right-hand side is gathered whole (nx numbers per rank) and solved
// by two prefix sums with the mean of F and of p removed.
//
// Layout: the half spectrum A1[ky + NyH (xl + nx pz)] of colfft.hip (ky fastest, z in the column kernels' stage order): a
// thread owns one (ky, pz) mode and marches in xl, so a wave reads 64 consecutive ky = 1 KB runs.
#include <cmath>

#include "ocn_internal.h"

namespace ocn {

namespace {

struct cx {
    double x, y;
};
__device__ __forceinline__ cx operator+(cx a, cx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cx operator-(cx a, cx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cx operator*(double s, cx a) { return {s * a.x, s * a.y}; }
__device__ __forceinline__ cx cmul(cx a, cx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cx cdiv(cx a, cx b)
{
    const double d = 1.0 / (b.x * b.x + b.y * b.y);
    return {(a.x * b.x + a.y * b.y) * d, (a.y * b.x - a.x * b.y) * d};
}

constexpr int XTRI_MAX_R = 8;

struct XTriArgs {
    double *a1;             // complex, A1[ky + NyH (xl + nx pz)]
    const double *ly, *lz;  // eigenvalues: ky natural (NyH entries), pz in stored (stage) order (Nz entries)
    int NyH, nx, Nz;
    double dx2, scale;      // F <- (dx^2 scale) F: the normalisation of the y, z transforms rides on the right-hand side
    double *gsend;          // complex [M][2]: g_1, g_n of every mode, then nx entries: the scaled F line of mode (0, 0)
    const double *grecv;    // complex [R][2 M + nx]: gsend of every rank
    int rank;
    double wr[XTRI_MAX_R], wi[XTRI_MAX_R];  // exp(2 pi i j / R)
};

// constants of one mode: r, 1 - r^2, ln r (mu > 0)
struct ModeConst {
    double b, r, omr2, lnr;
};
__device__ __forceinline__ ModeConst mode_const(double mu)
{
    const double sq = sqrt(mu * (mu + 4.0));
    const double S = (2.0 + mu) + sq;
    ModeConst c;
    c.b = -(2.0 + mu);
    c.r = 2.0 / S;
    const double omr = (mu + sq) / S;  // 1 - r without cancellation
    c.omr2 = omr * (1.0 + c.r);
    c.lnr = log1p(-omr);
    return c;
}

// Pass 1 of 2: forward elimination of the local block, d_i = c_i (F_i - d_(i-1)), c_i = 1 / (b - c_(i-1)), stored in place -- and the two
// numbers the interface system needs, WITHOUT the back substitution: g_n = d_n, and g_1 = e_1' T^-1 F = sum_j v_j F_j (T is symmetric, v
// the spike of the header), accumulated with P = r^j by products from r and Q = r^(2(n+1)-j) by quotients from r^(2n+1) (where that
// underflows, Q stays 0 and every term it stands for is below 1e-154).  16 B read + 16 B written per complex element.
__global__ __launch_bounds__(256) void xtri_forward_kernel(XTriArgs a)
{
    const long long M = (long long)a.NyH * a.Nz;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int ky = (int)(m % a.NyH), pz = (int)(m / a.NyH);
    const int n = a.nx;
    const long long st = a.NyH;  // complex elements between consecutive xl
    cx *p = reinterpret_cast<cx *>(a.a1) + ky + st * ((long long)n * pz);
    cx *gs = reinterpret_cast<cx *>(a.gsend);
    // mode (0, 0): any positive mu keeps the arithmetic finite; its line is replaced by xtri_zero_mode_kernel
    const double mu = (m == 0) ? 1.0 : a.dx2 * (a.ly[ky] + a.lz[pz]);
    const ModeConst mc = mode_const(mu);
    const double fs = a.dx2 * a.scale;
    const double iden = -1.0 / expm1(2.0 * (n + 1) * mc.lnr);  // 1 / (1 - r^(2(n+1)))
    const double rinv = 1.0 / mc.r;
    double P = mc.r, Q = exp((2.0 * n + 1.0) * mc.lnr);
    constexpr int B = 8;
    double c = 0.0;
    cx d = {0.0, 0.0}, g1 = {0.0, 0.0};
    for (int i0 = 0; i0 < n; i0 += B) {
        cx f[B];
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 + q < n) f[q] = p[st * (i0 + q)];
        if (m == 0) {
#pragma unroll
            for (int q = 0; q < B; ++q)
                if (i0 + q < n) gs[2 * M + i0 + q] = fs * f[q];
        }
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 + q < n) {
                const cx F = fs * f[q];
                c = 1.0 / (mc.b - c);
                d = c * (F - d);
                f[q] = d;
                g1 = g1 + (-(P - Q) * iden) * F;
                P *= mc.r;
                Q *= rinv;
            }
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 + q < n) p[st * (i0 + q)] = f[q];
    }
    gs[2 * m] = g1;
    gs[2 * m + 1] = d;  // g_n = d_n
}

// Pass 2 of 2: interface system, then the back substitution of the CORRECTED block in one descending sweep.  With the neighbours'
// values x0 = x_n of rank - 1 and xn1 = x_1 of rank + 1 the block solves T x = F - e_1 x0 - e_n xn1; forward elimination is linear, so
// its eliminated right-hand side is d_i - x0 dv_i - xn1 dw_i with dv_i = (-1)^(i-1) prod_(k<=i) c_k = -r^i (1 - r^2) / (1 - r^(2i+2)),
// dw = c_n e_n, and  x_n = d_n - x0 dv_n - xn1 c_n,  x_i = (d_i - x0 dv_i) - c_i x_(i+1),  c_i = -r (1 - r^2i) / (1 - r^(2i+2)):
// the spike correction costs no pass of its own (16 B read + 16 B written per complex element; the separate correction was 32 more).
template <int R>
__global__ __launch_bounds__(256) void xtri_backward_kernel(XTriArgs a)
{
    const long long M = (long long)a.NyH * a.Nz;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int ky = (int)(m % a.NyH), pz = (int)(m / a.NyH);
    const int n = a.nx;
    const long long st = a.NyH;
    cx *p = reinterpret_cast<cx *>(a.a1) + ky + st * ((long long)n * pz);
    const cx *gr = reinterpret_cast<const cx *>(a.grecv);
    const long long chunk = 2 * M + n;
    if (m == 0) return;  // the singular mode has its own kernel (xtri_zero_mode_kernel)
    const double mu = a.dx2 * (a.ly[ky] + a.lz[pz]);
    const ModeConst mc = mode_const(mu);
    const double den = -expm1(2.0 * (n + 1) * mc.lnr);  // 1 - r^(2(n+1))
    const double rn = exp(n * mc.lnr);
    const double iden = 1.0 / den;
    const double v1 = -mc.r * (-expm1(2.0 * n * mc.lnr)) * iden;  // -(r - r^(2n+1)) / den
    const double vn = -rn * mc.omr2 * iden;                       // -(r^n - r^(n+2)) / den
    // interface unknowns a_s = x_1, z_s = x_n of rank s:
    //   a_s + v1 z_(s-1) + vn a_(s+1) = g1_s,    z_s + vn z_(s-1) + v1 a_(s+1) = gn_s      (cyclic in s)
    // with x_s = sum_k xh_k w^(k s), w = exp(2 pi i / R):  (1 + vn w^k) ah_k + v1 w^-k zh_k = g1h_k,  v1 w^k ah_k + (1 + vn w^-k) zh_k = gnh_k
    cx g1[R], gn[R];
#pragma unroll
    for (int s = 0; s < R; ++s) {
        g1[s] = gr[s * chunk + 2 * m];
        gn[s] = gr[s * chunk + 2 * m + 1];
    }
    cx x0 = {0.0, 0.0}, xn1 = {0.0, 0.0};  // z_(rank-1), a_(rank+1)
    const int sm = (a.rank + R - 1) % R, sp = (a.rank + 1) % R;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        cx h1 = {0.0, 0.0}, hn = {0.0, 0.0};
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int j = (R - (k * s) % R) % R;  // w^(-k s)
            const cx w = {a.wr[j], a.wi[j]};
            h1 = h1 + cmul(w, g1[s]);
            hn = hn + cmul(w, gn[s]);
        }
        h1 = (1.0 / R) * h1;
        hn = (1.0 / R) * hn;
        const cx wk = {a.wr[k], a.wi[k]}, wmk = {a.wr[(R - k) % R], a.wi[(R - k) % R]};
        const cx A11 = {1.0 + vn * wk.x, vn * wk.y}, A12 = v1 * wmk, A21 = v1 * wk, A22 = {1.0 + vn * wmk.x, vn * wmk.y};
        const cx det = cmul(A11, A22) - cmul(A12, A21);
        const cx ah = cdiv(cmul(A22, h1) - cmul(A12, hn), det);
        const cx zh = cdiv(cmul(A11, hn) - cmul(A21, h1), det);
        const int jm = (k * sm) % R, jp = (k * sp) % R;
        x0 = x0 + cmul(cx{a.wr[jm], a.wi[jm]}, zh);
        xn1 = xn1 + cmul(cx{a.wr[jp], a.wi[jp]}, ah);
    }
    // ---- descending sweep.  em = r^i - 1 by expm1 (no cancellation in 1 - r^2i = -em (2 + em) for the long waves, r -> 1)
    constexpr int B = 8;
    cx x = {0.0, 0.0};
    for (int i0 = n; i0 >= 1; i0 -= B) {  // i0: 1-based index of the first element of this batch (descending)
        cx f[B];
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 - q >= 1) f[q] = p[st * (i0 - q - 1)];
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 - q >= 1) {
                const int i = i0 - q;
                const double em = expm1(mc.lnr * (double)i), E = em + 1.0;  // r^i - 1, r^i
                const double qi = -em * (2.0 + em);                         // 1 - r^2i
                const double inv = 1.0 / (qi + (E * E) * mc.omr2);          // 1 / (1 - r^(2i+2))
                const double ci = -mc.r * qi * inv, dvi = -E * mc.omr2 * inv;
                if (i == n) x = (f[q] - dvi * x0) - ci * xn1;
                else x = (f[q] - dvi * x0) - ci * x;
                f[q] = x;
            }
#pragma unroll
        for (int q = 0; q < B; ++q)
            if (i0 - q >= 1) p[st * (i0 - q - 1)] = f[q];
    }
}


// The (ky, kz) = (0, 0) line: p[i-1] - 2 p[i] + p[i+1] = F'[i] over the N = R n points of the global periodic line, F' = F - mean(F),
// mean(p) = 0.  With S = inclusive prefix sums of F:  s_i = S_i - (i + 1) mean(F) (prefix sums of F'),  d_i = p[i+1] - p[i] = d_-1 + s_i,
// sum(d) = 0  =>  d_-1 = -sum(s) / N,  p_i = i d_-1 + T_(i-1) with T = inclusive prefix sums of s (p_0 = 0), then minus mean(p).
// One workgroup, two Hillis-Steele scans in LDS.
constexpr int XTRI_ZERO_MAX_N = 2048;
__device__ __forceinline__ void block_scan(cx *buf, cx *tmp, int N)
{
    // inclusive scan of buf[0..N) (result in buf), all threads of the workgroup
    cx *src = buf, *dst = tmp;
    for (int off = 1; off < N; off <<= 1) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) dst[i] = (i >= off) ? src[i] + src[i - off] : src[i];
        cx *t = src; src = dst; dst = t;
    }
    __syncthreads();
    if (src != buf) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) buf[i] = src[i];
        __syncthreads();
    }
}
template <int R>
__global__ __launch_bounds__(256) void xtri_zero_mode_kernel(XTriArgs a)
{
    __shared__ cx buf[XTRI_ZERO_MAX_N], tmp[XTRI_ZERO_MAX_N];
    __shared__ cx red[256];
    const long long M = (long long)a.NyH * a.Nz;
    const int n = a.nx, N = R * n;
    const cx *gr = reinterpret_cast<const cx *>(a.grecv);
    const long long chunk = 2 * M + n;
    for (int i = threadIdx.x; i < N; i += blockDim.x) buf[i] = gr[(long long)(i / n) * chunk + 2 * M + (i % n)];
    block_scan(buf, tmp, N);  // S
    const cx mean = (1.0 / N) * buf[N - 1];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) buf[i] = buf[i] - (double)(i + 1) * mean;  // s
    block_scan(buf, tmp, N);  // T
    const cx dm1 = (-1.0 / N) * buf[N - 1];
    // p_i (before the mean is removed) into tmp, its sum by a tree reduction
    __syncthreads();
    cx part = {0.0, 0.0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const cx pi = (i == 0) ? cx{0.0, 0.0} : (double)i * dm1 + buf[i - 1];
        tmp[i] = pi;
        part = part + pi;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    const cx pmean = (1.0 / N) * red[0];
    cx *p = reinterpret_cast<cx *>(a.a1);  // mode (0, 0): element xl at NyH xl
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[(long long)a.NyH * i] = tmp[a.rank * n + i] - pmean;
}

}  // namespace

bool xtri_supported(int R, int Nxg) { return R >= 1 && R <= XTRI_MAX_R && Nxg <= XTRI_ZERO_MAX_N; }

static XTriArgs make_args(double *a1, const double *ly, const double *lz, int NyH, int nx, int Nz, double dx, double scale, double *gsend,
                          const double *grecv, int rank, int R)
{
    XTriArgs a{};
    a.a1 = a1; a.ly = ly; a.lz = lz;
    a.NyH = NyH; a.nx = nx; a.Nz = Nz;
    a.dx2 = dx * dx; a.scale = scale;
    a.gsend = gsend; a.grecv = grecv; a.rank = rank;
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int j = 0; j < R && j < XTRI_MAX_R; ++j) {
        a.wr[j] = (double)cosl(two_pi * j / R);
        a.wi[j] = (double)sinl(two_pi * j / R);
    }
    return a;
}

int launch_xtri_sweep(double *a1, const double *ly, const double *lz, int NyH, int nx, int Nz, double dx, double scale, double *gsend,
                      hipStream_t stream)
{
    const XTriArgs a = make_args(a1, ly, lz, NyH, nx, Nz, dx, scale, gsend, nullptr, 0, 1);
    const long long M = (long long)NyH * Nz;
    hipLaunchKernelGGL(xtri_forward_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, a);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

int launch_xtri_finish(double *a1, const double *ly, const double *lz, int NyH, int nx, int Nz, double dx, const double *grecv, int rank,
                       int R, hipStream_t stream)
{
    if (!xtri_supported(R, R * nx)) {
        set_error("the transpose-free x solve supports 1..%d ranks, got %d", XTRI_MAX_R, R);
        return OCN_ERR_UNSUPPORTED;
    }
    const XTriArgs a = make_args(a1, ly, lz, NyH, nx, Nz, dx, 1.0, nullptr, grecv, rank, R);
    const long long M = (long long)NyH * Nz;
    const dim3 grid((unsigned)((M + 255) / 256)), block(256);
    switch (R) {
#define OCN_XTRI_CASE(RV)                                                             \
    case RV:                                                                          \
        hipLaunchKernelGGL(xtri_backward_kernel<RV>, grid, block, 0, stream, a);        \
        hipLaunchKernelGGL(xtri_zero_mode_kernel<RV>, dim3(1), block, 0, stream, a);  \
        break;
        OCN_XTRI_CASE(1) OCN_XTRI_CASE(2) OCN_XTRI_CASE(3) OCN_XTRI_CASE(4) OCN_XTRI_CASE(5) OCN_XTRI_CASE(6) OCN_XTRI_CASE(7) OCN_XTRI_CASE(8)
#undef OCN_XTRI_CASE
    }
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

}  // namespace ocn
