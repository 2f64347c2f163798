// poisson.hip -- direct Poisson solvers on rocFFT.
//   FFTBasedPoissonSolver          src/Solvers/fft_based_poisson_solver.jl:5-125   (z regular, Periodic or Flat)
//   FourierTridiagonalPoissonSolver src/Solvers/fourier_tridiagonal_poisson_solver.jl:6-147 (z Bounded)
//   DistributedFFTBasedPoissonSolver src/DistributedComputations/distributed_fft_based_poisson_solver.jl:10-188 (slab-x)
// Eigenvalues: src/Solvers/poisson_eigenvalues.jl:8-31.  Transforms: plan_transforms.jl:36-146.
//
// MI355X design: the divergence of a real velocity field is real, so the transforms are real-to-complex /
// complex-to-real on the Hermitian half spectrum (Nx/2+1 modes in x): half the HBM traffic of the reference's
// in-place C2C (fft_based_poisson_solver.jl:65) with identical results up to FFT round-off.  The inverse
// transform writes straight into the interior of the haloed pressure field (custom output strides), which
// removes the copy_real_component! pass (K13).  OCN_POISSON_C2C=1 selects the literal complex path instead.
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "ocn_internal.h"

#define OCN_CHECK_FFT(expr)                                                                        \
    do {                                                                                           \
        rocfft_status _s = (expr);                                                                 \
        if (_s != rocfft_status_success) {                                                         \
            ocn::set_error("%s failed with rocfft_status %d (%s:%d)", #expr, (int)_s, __FILE__, __LINE__); \
            return OCN_ERR_ROCFFT;                                                                 \
        }                                                                                          \
    } while (0)

namespace {

struct FFTSetup {
    FFTSetup() { rocfft_setup(); }
    ~FFTSetup() { rocfft_cleanup(); }
};
void ensure_rocfft()
{
    static FFTSetup s;
    (void)s;
}

// poisson_eigenvalues (poisson_eigenvalues.jl:8-31)
std::vector<double> eigenvalues(int N, double L, int topo)
{
    std::vector<double> lam(N, 0.0);
    const double pi = 3.141592653589793;
    for (int q = 0; q < N; ++q) {
        if (topo == OCN_PERIODIC || topo == OCN_FULLY_CONNECTED) {
            const double s = 2 * std::sin(q * pi / N) / (L / N);
            lam[q] = s * s;
        } else if (topo == OCN_BOUNDED) {
            const double s = 2 * std::sin(q * pi / (2 * N)) / (L / N);
            lam[q] = s * s;
        }
    }
    return lam;
}

int upload(const std::vector<double> &h, double **d)
{
    OCN_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(d), h.size() * sizeof(double)));
    OCN_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return OCN_SUCCESS;
}

struct Plan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
    int finish()
    {
        OCN_CHECK_FFT(rocfft_plan_get_work_buffer_size(plan, &work_bytes));
        OCN_CHECK_FFT(rocfft_execution_info_create(&info));
        if (work_bytes) {
            OCN_CHECK_HIP(hipMalloc(&work, work_bytes));
            OCN_CHECK_FFT(rocfft_execution_info_set_work_buffer(info, work, work_bytes));
        }
        return OCN_SUCCESS;
    }
    int exec(void *in, void *out, hipStream_t stream)
    {
        OCN_CHECK_FFT(rocfft_execution_info_set_stream(info, stream));
        void *ib[1] = {in};
        void *ob[1] = {out};
        OCN_CHECK_FFT(rocfft_execute(plan, ib, out ? ob : nullptr, info));
        return OCN_SUCCESS;
    }
    void destroy()
    {
        // Plans are destroyed eagerly.  rocFFT (ROCm 7.2) real-transform plans collide with OTHER LIVE real plans: a power-of-two
        // pair whose 2-D (x, y) kernel lengths are the transpose of an existing plan's -- (Nx/2, Ny) = (Ny', Nx'/2), e.g. 16 x 16 x k
        // alive, then 32 x 8 x k, or 64^3 then 128 x 32 -- comes out wrong (forward spectrum, inverse, or both; the earlier plan stays
        // correct), and destroying the first plan BEFORE creating the second cures it (tools/rocfft_repro.hip,
        // profiles/r02_rocfft_repro.md).  Keeping dead plans alive therefore only widens the exposure; every new pair is additionally
        // verified against known answers over its full spectrum at creation (poisson_plans_self_test) and replaced by complex plans
        // if it fails.  OCN_ROCFFT_RETIRE_PLANS=1 restores round 1's behaviour (objects kept until the process ends).
        static const bool retire = std::getenv("OCN_ROCFFT_RETIRE_PLANS") && std::getenv("OCN_ROCFFT_RETIRE_PLANS")[0] == '1';
        if (!retire) {
            if (plan) rocfft_plan_destroy(plan);
            if (info) rocfft_execution_info_destroy(info);
        }
        if (work) (void)hipFree(work);
        plan = nullptr; info = nullptr; work = nullptr;
    }
};

// Generic plan helper.  dims-long arrays; strides in elements.
int make_plan(Plan &P, rocfft_result_placement placement, rocfft_transform_type type, int dims, const size_t *lengths,
              size_t batch, rocfft_array_type in_type, rocfft_array_type out_type, const size_t *in_strides, size_t in_dist,
              const size_t *out_strides, size_t out_dist, double scale)
{
    rocfft_plan_description desc = nullptr;
    OCN_CHECK_FFT(rocfft_plan_description_create(&desc));
    OCN_CHECK_FFT(rocfft_plan_description_set_data_layout(desc, in_type, out_type, nullptr, nullptr, dims, in_strides, in_dist,
                                                          dims, out_strides, out_dist));
    if (scale != 1.0) OCN_CHECK_FFT(rocfft_plan_description_set_scale_factor(desc, scale));
    OCN_CHECK_FFT(rocfft_plan_create(&P.plan, placement, type, rocfft_precision_double, dims, lengths, batch, desc));
    OCN_CHECK_FFT(rocfft_plan_description_destroy(desc));
    return P.finish();
}

}  // namespace

// ===================================================================================================
// Single-device solver
// ===================================================================================================
struct ocn_poisson {
    ocn_grid grid{};
    int kind = 0;       // 0 FFT-based, 1 Fourier-tridiagonal (z)
    bool c2c = false;   // literal complex path
    int nxh = 0;        // stored x modes
    double *dzc = nullptr, *dzf = nullptr;  // handle-owned copies of the stretched spacings
    double *lx = nullptr, *ly = nullptr, *lz = nullptr;
    double *rhs = nullptr;      // real Nx*Ny*Nz (r2c) -- source term in physical space
    double *spec = nullptr;     // complex nxh*Ny*Nz -- spectrum / storage
    double *spec2 = nullptr;    // tridiagonal solution (complex nxh*Ny*Nz)
    double *diag = nullptr, *tscr = nullptr, *lower = nullptr;
    Plan fwd, bwd;
    bool fused_z = false;       // FFT_z + spectral solve + IFFT_z in one column kernel (colfft.hip); rocFFT does (x, y)
    double *tw = nullptr, *lz_stage = nullptr;
    bool custom_xy = false;     // x passes by rowfft.hip (fused with the source term / the write into p), y passes by colfft.hip
    double *twMx = nullptr, *twNx = nullptr, *twy = nullptr, *ly_stage = nullptr;
    bool custom_tri = false;    // Fourier-tridiagonal flavour of custom_xy: row / column kernels for (x, y), Thomas sweep in z, no rocFFT
    // ... and on a REGULAR Bounded z of a supported length the sweep is replaced by its exact spectral twin: cosine transform, division by
    // the eigenvalues, inverse cosine transform in ONE column pass (colfft.hip MODE 5) -- what the reference's FFTBasedPoissonSolver does on
    // such a grid (plan_transforms.jl:129-140), 32 instead of 88 B per element  [OCN_POISSON_DCT_Z=0: the Thomas sweep]
    bool dct_z = false;
    double *wdz = nullptr, *lz_bounded = nullptr;
    bool source_in_rhs = false; // custom_xy: the source was given as a real array (set_source_term!) and still needs its x transform
    bool direct_out = true;  // r2c path: inverse transform writes straight into the haloed pressure interior
    bool source_set = false;
    double shift = 0.0;         // ocn_poisson_solve_shifted: (∇² + m) ϕ = b for this solve
    bool shifted = false;
    // kind 2: FFTBasedPoissonSolver for ANY regular (Periodic | Bounded | Flat)^3 topology: separable transforms evaluated as direct sums
    // (DFT along Periodic, REDFT10 / REDFT01 along Bounded dimensions, plan_transforms.jl:16-34) -- the reference's K11 path
    double *tab[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // per dimension: cos / sin tables
    // ... or, by default, from complex FFTs of the same length with index permutations and twiddle factors (the reference's K11,
    // index_permutations.jl:38-90, discrete_transforms.jl:141-176): O(N log N) per line
    Plan gfwd[3], gbwd[3];
    double *gtw[3] = {nullptr, nullptr, nullptr};  // w_k = exp(-i π k / 2N), k < N, of the Bounded dimensions
    // strided lines (y, z) of the supported lengths go through the column FFT kernels (csrc/colfft.hip: ONE launch per pass; rocFFT's
    // strided plan needs one launch per z plane for the y lines): their spectra are then in STAGE order along that direction --
    // eigenvalues and twiddles are stored permuted, gpartner[d][p] = stored position of wavenumber N - k(p) for the cosine transforms
    bool gathered = false;  // compute_source_term stored the source already permuted along the first cosine-transform dimension
    bool gcol[3] = {false, false, false};
    // ... and a Bounded dimension on the column kernel runs its cosine transform in ONE pass (colfft.hip MODE 3 / 4: permutation and twiddle
    // inside the kernel, natural wavenumber order on both sides, so eigenvalues and twiddles stay natural)  [OCN_POISSON_FUSED_DCT=0: three passes]
    bool gdct[3] = {false, false, false};
    // ... and the x lines of an all-real closed box (row pairs) with Nx = 64 ... 512 run in colfft.hip's row kernel: forward transform,
    // division by the eigenvalues and inverse in ONE in-place pass (two around the Thomas sweep of a stretched z)  [OCN_POISSON_ROW_DCT=0:
    // gather, rocFFT lines, twiddle, solve, twiddle, lines, scatter]
    bool growdct = false;
    double *gcoltw[3] = {nullptr, nullptr, nullptr};
    int *gpartner[3] = {nullptr, nullptr, nullptr};
    bool fft_dct = false;
    // ... with the z direction solved by the batched Thomas sweep instead of a third transform: FourierTridiagonalPoissonSolver on grids
    // with a Bounded / Flat x or y (XYRegularRG with any (x, y) topology: fourier_tridiagonal_poisson_solver.jl:82-147) -- the only solver
    // of a channel with a stretched z
    bool gtri = false;
    // x Periodic (even Nx) next to a Bounded y / z: the source term is REAL and a cosine transform maps reals to reals, so the transforms
    // along the Bounded y / z run on the real array VIEWED as complex numbers of x-adjacent pairs (Nx / 2 complex columns: the complex
    // transform of a line is the transform of its real and of its imaginary part, separated by the Hermitian symmetry the twiddle passes
    // already apply) -- half the bytes per pass -- then x is a real-to-complex transform to the half spectrum (Nx / 2 + 1), on which the
    // remaining Periodic direction, the solve and the way back run.  Same arithmetic per line as the complex path.
    bool gpacked = false;
    Plan xr2c, xc2r;
    // ... and the closed box / the y-z wall slices (x AND y Bounded, z Bounded or Flat, even Nx and Ny): nothing ever becomes complex -- the
    // transforms along y / z on x-adjacent pairs as above, the one along x on pairs of ROWS (dct_rowpair_kernel), a real division by the
    // eigenvalues or (stretched z) the Thomas sweep on reals
    bool gallreal = false;
};

static void free_all(ocn_poisson *s)
{
    s->fwd.destroy();
    s->bwd.destroy();
    s->xr2c.destroy();
    s->xc2r.destroy();
    for (int d = 0; d < 3; ++d) {
        s->gfwd[d].destroy();
        s->gbwd[d].destroy();
        if (s->gtw[d]) (void)hipFree(s->gtw[d]);
        s->gtw[d] = nullptr;
        if (s->gcoltw[d]) (void)hipFree(s->gcoltw[d]);
        s->gcoltw[d] = nullptr;
        if (s->gpartner[d]) (void)hipFree(s->gpartner[d]);
        s->gpartner[d] = nullptr;
    }
    double **ptrs[] = {&s->dzc, &s->dzf, &s->lx, &s->ly, &s->lz, &s->rhs, &s->spec, &s->spec2, &s->diag, &s->tscr, &s->lower, &s->tw, &s->lz_stage, &s->twMx, &s->twNx, &s->twy, &s->ly_stage, &s->wdz, &s->lz_bounded,
                       &s->tab[0][0], &s->tab[0][1], &s->tab[1][0], &s->tab[1][1], &s->tab[2][0], &s->tab[2][1]};
    for (auto p : ptrs)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
}

static int poisson_plans_self_test(ocn_poisson *s);

// ---------------------------------------------------------------------------------------------------------------------------------
// kind 2: FFTBasedPoissonSolver on ANY regular topology (fft_based_poisson_solver.jl:5-125 with the transforms of plan_transforms.jl:16-34,
// 129-140: Bounded dimensions first, then Periodic ones; backward in the opposite order).  The reference's GPU path builds the cosine
// transforms from FFTs with index permutations and twiddle factors (K11: index_permutations.jl:38-90, discrete_transforms.jl:141-176);
// here every 1-D transform is the direct sum of its definition over a table of cos / sin values computed on the host in fp64:
//   Periodic forward  X[k] = Σ_n x[n] e^{-2πi k n / N}                      backward x[n] = (1/N) Σ_k X[k] e^{+2πi k n / N}
//   Bounded  forward  X[k] = 2 Σ_n x[n] cos(π (n + 1/2) k / N)  (REDFT10)    backward x[n] = (1/2N) (X[0] + 2 Σ_{k>=1} X[k] cos(π (n + 1/2) k / N))
// O(N) work per output element: this path serves the Bounded-x / Bounded-y topologies of test_poisson_solvers.jl:58-106, which no
// BASELINE configuration uses (their fields are not implemented by the tendency kernels either); it is exact to N eps, not tuned.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void naive_transform_kernel(int Nx, int Ny, int Nz, int dim, int topo, int inverse, const double2 *__restrict__ in,
                                                              double2 *__restrict__ out, const double *__restrict__ ctab,
                                                              const double *__restrict__ stab)
{
    const long long n = (long long)Nx * Ny * Nz, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = (int)(t % Nx), j = (int)((t / Nx) % Ny), k = (int)(t / ((long long)Nx * Ny));
    const int N = dim == 0 ? Nx : dim == 1 ? Ny : Nz, q = dim == 0 ? i : dim == 1 ? j : k;  // q: output index along the transformed dimension
    const long long stride = dim == 0 ? 1 : dim == 1 ? Nx : (long long)Nx * Ny;
    const double2 *line = in + (t - q * stride);
    double re = 0.0, im = 0.0;
    if (topo == OCN_PERIODIC) {
        for (int m = 0; m < N; ++m) {
            const int idx = (int)(((long long)q * m) % N);
            const double c = ctab[idx], sn = inverse ? stab[idx] : -stab[idx];
            const double2 v = line[m * stride];
            re += v.x * c - v.y * sn;
            im += v.x * sn + v.y * c;
        }
        if (inverse) { re /= N; im /= N; }
    } else {  // Bounded: cosine transforms of the real and imaginary parts; table index (2 n + 1) k mod 4N of cos(π idx / 2N)
        if (!inverse) {
            for (int m = 0; m < N; ++m) {
                const double c = ctab[(int)(((long long)(2 * m + 1) * q) % (4 * N))];
                const double2 v = line[m * stride];
                re += v.x * c;
                im += v.y * c;
            }
            re *= 2; im *= 2;
        } else {
            for (int m = 1; m < N; ++m) {
                const double c = ctab[(int)(((long long)(2 * q + 1) * m) % (4 * N))];
                const double2 v = line[m * stride];
                re += v.x * c;
                im += v.y * c;
            }
            const double2 v0 = line[0];
            re = (v0.x + 2 * re) / (2 * N);
            im = (v0.y + 2 * im) / (2 * N);
        }
    }
    out[t] = make_double2(re, im);
}

// ---- the same transforms from FFTs (Makhoul 1980; the reference's GPU path K11) ------------------------------------------------
// REDFT10 of a line x[0..N):  v[n] = x[2n] (n < ceil(N/2)),  v[N-1-n] = x[2n+1];  V = FFT_N(v);  X[k] = 2 Re(w_k V[k]), w_k = e^{-iπk/2N}.
// The lines here are complex (earlier dimensions are already in spectral space): the real and imaginary parts are two real lines
// transformed by ONE complex FFT and separated through the Hermitian symmetry of their spectra,
//     X[k] = Re(w_k (V[k] + conj V[N-k]))  +  i Im(w_k (V[k] - conj V[N-k])).
// REDFT01 / 2N (the inverse):  V[k] = (1/2) conj(w_k) (X[k] - i X[N-k]),  X[N] := 0;  v = IFFT_N(V) / N;  x[2n] = v[n], x[2n+1] = v[N-1-n].
// mode 0: gather (x -> v), 1: forward post-twiddle (V -> X), 2: inverse pre-twiddle (X -> V), 3: scatter (v -> x); out of place.
// partner != NULL: the spectrum along `dim` is in the column kernels' stage order (position p holds wavenumber k(p)): w[p] = w_k(p),
// partner[p] = position of wavenumber N - k(p) (position 0 is wavenumber 0 in either order).
__global__ __launch_bounds__(256) void dct_shuffle_kernel(int Nx, int Ny, int Nz, int dim, int mode, const double2 *__restrict__ in,
                                                          double2 *__restrict__ out, const double2 *__restrict__ w,
                                                          const int *__restrict__ partner)
{
    const long long n = (long long)Nx * Ny * Nz, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = (int)(t % Nx), j = (int)((t / Nx) % Ny), k = (int)(t / ((long long)Nx * Ny));
    const int N = dim == 0 ? Nx : dim == 1 ? Ny : Nz, q = dim == 0 ? i : dim == 1 ? j : k;
    const long long stride = dim == 0 ? 1 : dim == 1 ? Nx : (long long)Nx * Ny;
    const double2 *line = in + (t - q * stride);
    const int half = (N + 1) / 2;
    if (mode == 0) {         // v[q]
        out[t] = q < half ? line[(long long)(2 * q) * stride] : line[(long long)(2 * (N - 1 - q) + 1) * stride];
    } else if (mode == 3) {  // x[q]
        out[t] = (q & 1) ? line[(long long)(N - 1 - (q - 1) / 2) * stride] : line[(long long)(q / 2) * stride];
    } else if (mode == 1) {
        const int qp = partner ? partner[q] : (N - q) % N;
        const double2 a = line[(long long)q * stride], b = line[(long long)qp * stride], wk = w[q];
        const double sr = a.x + b.x, si = a.y - b.y;  // V[k] + conj V[N-k]
        const double dr = a.x - b.x, di = a.y + b.y;  // V[k] - conj V[N-k]
        out[t] = make_double2(wk.x * sr - wk.y * si, wk.x * di + wk.y * dr);
    } else {
        const double2 a = line[(long long)q * stride];
        const double2 b = q == 0 ? make_double2(0.0, 0.0) : line[(long long)(partner ? partner[q] : N - q) * stride];
        const double2 wk = w[q];
        const double zr = a.x + b.y, zi = a.y - b.x;  // X[k] - i X[N-k]
        out[t] = make_double2(0.5 * (wk.x * zr + wk.y * zi), 0.5 * (wk.x * zi - wk.y * zr));  // (1/2) conj(w_k) z
    }
}

// The same four passes for the lines ALONG x of a REAL array (Nx, Ny, Nz), Ny even: the complex line (i, jp, k), jp < Ny / 2, is the pair of real
// rows j = 2 jp (real part) and 2 jp + 1 (imaginary part) -- a cosine transform maps reals to reals, and the complex transform of a line is the
// transform of its real and of its imaginary part, so two rows ride on one complex FFT.  Modes 0 and 2 read the real array (split: the two
// parts Nx doubles apart) and write an interleaved complex array (Nx, Ny / 2, Nz) for the line FFTs; modes 1 and 3 read that and write the
// real array.  Natural wavenumber order (rocFFT lines).
__global__ __launch_bounds__(256) void dct_rowpair_kernel(int Nx, int Nyp, int Nz, int mode, const double *__restrict__ in, double *__restrict__ out,
                                                          const double2 *__restrict__ w)
{
    const long long n = (long long)Nx * Nyp * Nz, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int q = (int)(t % Nx), jp = (int)((t / Nx) % Nyp), k = (int)(t / ((long long)Nx * Nyp));
    const int N = Nx, half = (N + 1) / 2;
    const long long row = (long long)Nx * (2 * jp + (long long)2 * Nyp * k);  // the real row j = 2 jp of plane k; the odd row follows at + Nx
    const double2 *cin = reinterpret_cast<const double2 *>(in) + (t - q);      // the interleaved complex line
    double2 *cout = reinterpret_cast<double2 *>(out);
    auto split_in = [&](int m) { return make_double2(in[row + m], in[row + Nx + m]); };
    auto split_out = [&](double2 v) { out[row + q] = v.x; out[row + Nx + q] = v.y; };
    if (mode == 0) {         // v[q], real rows -> complex
        cout[t] = q < half ? split_in(2 * q) : split_in(2 * (N - 1 - q) + 1);
    } else if (mode == 3) {  // x[q], complex -> real rows
        split_out((q & 1) ? cin[N - 1 - (q - 1) / 2] : cin[q / 2]);
    } else if (mode == 1) {  // post-twiddle, complex -> real rows
        const double2 a = cin[q], b = cin[(N - q) % N], wk = w[q];
        const double sr = a.x + b.x, si = a.y - b.y, dr = a.x - b.x, di = a.y + b.y;
        split_out(make_double2(wk.x * sr - wk.y * si, wk.x * di + wk.y * dr));
    } else {                 // pre-twiddle, real rows -> complex
        const double2 a = split_in(q);
        const double2 b = q == 0 ? make_double2(0.0, 0.0) : split_in(N - q);
        const double2 wk = w[q];
        const double zr = a.x + b.y, zi = a.y - b.x;
        cout[t] = make_double2(0.5 * (wk.x * zr + wk.y * zi), 0.5 * (wk.x * zi - wk.y * zr));
    }
}

// -b / (λx + λy + λz [- m]) of a REAL spectrum (every direction a cosine transform), mode (1, 1, 1) := 0 iff m === 0
__global__ __launch_bounds__(256) void spectral_solve_real_kernel(int Nx, int Ny, int Nz, const double *__restrict__ lx, const double *__restrict__ ly,
                                                                  const double *__restrict__ lz, double *__restrict__ b, double m, int shifted)
{
    const long long n = (long long)Nx * Ny * Nz, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = (int)(t % Nx), j = (int)((t / Nx) % Ny), k = (int)(t / ((long long)Nx * Ny));
    double lam = (lx[i] + ly[j]) + lz[k];
    if (shifted) lam = lam - m;
    double val = -b[t] / lam;
    if (!shifted && t == 0) val = 0.0;
    b[t] = val;
}

// A twiddle pass along `dt` (mode mt = 1 forward post-twiddle, 2 inverse pre-twiddle) and a pure permutation along ANOTHER dimension
// `dp` (mode mp = 0 gather, 3 scatter) in ONE pass: the permutation only relabels whole lines of the twiddle direction, so
//   forward:  out = gather_dp(twiddle_dt(in)),   inverse:  out = pretwiddle_dt(scatter_dp(in))
// both read `in` at the permuted position of the output point (and at its N - k partner along dt).
__global__ __launch_bounds__(256) void dct_twiddle_permute_kernel(int Nx, int Ny, int Nz, int dt, int mt, int dp, int mp,
                                                                  const double2 *__restrict__ in, double2 *__restrict__ out,
                                                                  const double2 *__restrict__ w, const int *__restrict__ partner)
{
    const long long n = (long long)Nx * Ny * Nz, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int idx[3] = {(int)(t % Nx), (int)((t / Nx) % Ny), (int)(t / ((long long)Nx * Ny))};
    const int Ns[3] = {Nx, Ny, Nz};
    const long long strides[3] = {1, Nx, (long long)Nx * Ny};
    // the permuted source position along dp
    const int Np = Ns[dp], qp = idx[dp], half = (Np + 1) / 2;
    const int sp = mp == 0 ? (qp < half ? 2 * qp : 2 * (Np - 1 - qp) + 1) : ((qp & 1) ? Np - 1 - (qp - 1) / 2 : qp / 2);
    const long long src = t + (long long)(sp - qp) * strides[dp];
    const int N = Ns[dt], q = idx[dt];
    const long long st = strides[dt];
    const double2 *line = in + (src - q * st);
    const double2 a = line[(long long)q * st], wk = w[q];
    if (mt == 1) {
        const int qq = partner ? partner[q] : (N - q) % N;
        const double2 b = line[(long long)qq * st];
        const double sr = a.x + b.x, si = a.y - b.y, dr = a.x - b.x, di = a.y + b.y;
        out[t] = make_double2(wk.x * sr - wk.y * si, wk.x * di + wk.y * dr);
    } else {
        const double2 b = q == 0 ? make_double2(0.0, 0.0) : line[(long long)(partner ? partner[q] : N - q) * st];
        const double zr = a.x + b.y, zi = a.y - b.x;
        out[t] = make_double2(0.5 * (wk.x * zr + wk.y * zi), 0.5 * (wk.x * zi - wk.y * zr));
    }
}

// batched complex FFT plans along dimension d of a contiguous Nx x Ny x Nz array (x fastest), in place; d = 1 is planned per z plane
static int make_line_plans(ocn_poisson *s, int d, const int N[3])
{
    const size_t len[1] = {(size_t)N[d]};
    size_t stride[1], dist, batch;
    if (d == 0) { stride[0] = 1; dist = (size_t)N[0]; batch = (size_t)N[1] * N[2]; }
    else if (d == 1) { stride[0] = (size_t)N[0]; dist = 1; batch = (size_t)N[0]; }
    else { stride[0] = (size_t)N[0] * N[1]; dist = 1; batch = (size_t)N[0] * N[1]; }
    int st = make_plan(s->gfwd[d], rocfft_placement_inplace, rocfft_transform_type_complex_forward, 1, len, batch,
                       rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, stride, dist, stride, dist, 1.0);
    if (st != OCN_SUCCESS) return st;
    return make_plan(s->gbwd[d], rocfft_placement_inplace, rocfft_transform_type_complex_inverse, 1, len, batch,
                     rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, stride, dist, stride, dist, 1.0 / N[d]);
}

static int exec_line_plan(ocn_poisson *s, int d, int inverse, double *a, const int N[3], hipStream_t stream, bool dct = false)
{
    if (s->gcol[d]) {  // one launch of the column kernel; forward: natural -> stage order, inverse: stage order -> natural, scaled 1 / N
        // (a whole cosine transform, natural order on both sides, when `dct`)
        const long long plane = (long long)N[0] * N[1];
        const int mode = dct ? (inverse ? 4 : 3) : (inverse ? 1 : 0);
        const double *wd = dct ? s->gtw[d] : nullptr;
        if (d == 1) return ocn::launch_colfft(N[1], mode, a, N[0], plane, N[0], N[2], s->gcoltw[d], nullptr, nullptr, wd,
                                              inverse ? 1.0 / N[1] : 1.0, 1, stream);
        return ocn::launch_colfft(N[2], mode, a, plane, 0, (int)plane, 1, s->gcoltw[d], nullptr, nullptr, wd,
                                  inverse ? 1.0 / N[2] : 1.0, 1, stream);
    }
    Plan &P = inverse ? s->gbwd[d] : s->gfwd[d];
    if (d != 1) return P.exec(a, nullptr, stream);
    for (int k = 0; k < N[2]; ++k) {
        int st = P.exec(a + (size_t)2 * N[0] * N[1] * k, nullptr, stream);
        if (st != OCN_SUCCESS) return st;
    }
    return OCN_SUCCESS;
}

static thread_local bool g_no_packed = false;  // the retry of poisson_create_general after a failed self test of the real x plans
static int packed_plans_self_test(ocn_poisson *s);

static int poisson_create_general(ocn_poisson_t *out, const ocn_grid *grid)
{
    // a stretched z (or OCN_POISSON_GENERAL_TRI=1 on a regular Bounded z): transforms along x and y, tridiagonal solve along z
    const char *etri = std::getenv("OCN_POISSON_GENERAL_TRI");
    const bool gtri = grid->dzc != nullptr || (etri && etri[0] == '1' && grid->tz == OCN_BOUNDED);
    if (gtri && (grid->tz != OCN_BOUNDED || grid->Nz < 2)) {
        ocn::set_error("`FourierTridiagonalPoissonSolver` can only be used when the stretched direction's topology is `Bounded` (and Nz >= 2)");
        return OCN_ERR_UNSUPPORTED;
    }
    for (int t : {grid->tx, grid->ty, grid->tz})
        if (t != OCN_PERIODIC && t != OCN_BOUNDED && t != OCN_FLAT) {
            ocn::set_error("ocn_poisson_create: topology code %d is not supported by the general solver", t);
            return OCN_ERR_UNSUPPORTED;
        }
    ocn_poisson *s = new ocn_poisson();
    s->grid = *grid;
    s->kind = 2;
    s->gtri = gtri;
    s->c2c = true;
    s->direct_out = false;
    const int N[3] = {grid->Nx, grid->Ny, grid->Nz}, topo[3] = {grid->tx, grid->ty, gtri ? OCN_FLAT : grid->tz};  // (no transform along z)
    const double Ls[3] = {grid->Lx, grid->Ly, grid->Lz};
    s->nxh = N[0];
    const size_t n = (size_t)N[0] * N[1] * N[2];
    const double pi = 3.14159265358979323846;
    double **lam[3] = {&s->lx, &s->ly, &s->lz};
    int st = OCN_SUCCESS;
    for (int d = 0; d < 3 && st == OCN_SUCCESS; ++d) {
        st = upload(eigenvalues(N[d], Ls[d], topo[d]), lam[d]);
        if (st != OCN_SUCCESS || topo[d] == OCN_FLAT) continue;
        const int M = topo[d] == OCN_PERIODIC ? N[d] : 4 * N[d];
        std::vector<double> c(M), sn(M);
        for (int m = 0; m < M; ++m) {
            const double a = topo[d] == OCN_PERIODIC ? 2 * pi * m / N[d] : pi * m / (2.0 * N[d]);
            c[m] = std::cos(a);
            sn[m] = std::sin(a);
        }
        st = upload(c, &s->tab[d][0]);
        if (st == OCN_SUCCESS) st = upload(sn, &s->tab[d][1]);
    }
    {   // FFT-based transforms (default); OCN_POISSON_NAIVE_DCT=1 keeps the direct sums of the definitions (the checker of the tests)
        const char *nv = std::getenv("OCN_POISSON_NAIVE_DCT");
        s->fft_dct = !(nv && nv[0] == '1');
        {
            const char *ep = std::getenv("OCN_POISSON_PACKED");
            s->gpacked = s->fft_dct && !g_no_packed && topo[0] == OCN_PERIODIC && N[0] % 2 == 0 && N[0] >= 4 &&
                         (topo[1] == OCN_BOUNDED || topo[2] == OCN_BOUNDED) && !(ep && ep[0] == '0');
            s->gallreal = s->fft_dct && topo[0] == OCN_BOUNDED && topo[1] == OCN_BOUNDED && (topo[2] == OCN_BOUNDED || topo[2] == OCN_FLAT) &&
                          N[0] % 2 == 0 && N[1] % 2 == 0 && !(ep && ep[0] == '0');
        }
        if (s->fft_dct && st == OCN_SUCCESS) {
            ensure_rocfft();
            const char *gc = std::getenv("OCN_POISSON_GENERAL_COLFFT");
            for (int d = 0; d < 3 && st == OCN_SUCCESS; ++d) {
                if (topo[d] == OCN_FLAT) continue;
                s->gcol[d] = d > 0 && ocn::colfft_supported(N[d]) && !(gc && gc[0] == '0');
                const char *fd = std::getenv("OCN_POISSON_FUSED_DCT");
                s->gdct[d] = s->gcol[d] && topo[d] == OCN_BOUNDED && !(fd && fd[0] == '0');
                std::vector<int> kofp(N[d]), pofk(N[d]);
                for (int p = 0; p < N[d]; ++p) {
                    kofp[p] = (s->gcol[d] && !s->gdct[d]) ? ocn::colfft_wavenumber(N[d], p) : p;
                    pofk[kofp[p]] = p;
                }
                if (s->gdct[d]) {
                    st = upload(ocn::colfft_twiddles(N[d]), &s->gcoltw[d]);
                    if (st != OCN_SUCCESS) break;
                } else if (s->gcol[d]) {
                    st = upload(ocn::colfft_twiddles(N[d]), &s->gcoltw[d]);
                    if (st != OCN_SUCCESS) break;
                    // eigenvalues in stored order
                    std::vector<double> ln = eigenvalues(N[d], Ls[d], topo[d]), lp(N[d]);
                    for (int p = 0; p < N[d]; ++p) lp[p] = ln[kofp[p]];
                    (void)hipFree(*lam[d]);
                    *lam[d] = nullptr;
                    st = upload(lp, lam[d]);
                    if (st != OCN_SUCCESS) break;
                } else if (s->gallreal && d == 0 && ocn::colfft_supported(N[0]) && !(gc && gc[0] == '0') &&
                           !(std::getenv("OCN_POISSON_ROW_DCT") && std::getenv("OCN_POISSON_ROW_DCT")[0] == '0')) {
                    s->growdct = true;
                    st = upload(ocn::colfft_twiddles(N[0]), &s->gcoltw[0]);
                } else if (s->gallreal) {
                    // x lines of row pairs (Nx, Ny / 2, Nz); y / z lines of x-adjacent pairs (Nx / 2, Ny, Nz)
                    const int Nd[3] = {d == 0 ? N[0] : N[0] / 2, d == 0 ? N[1] / 2 : N[1], N[2]};
                    st = make_line_plans(s, d, Nd);
                } else if (s->gpacked && d == 0) {
                    // x: real-to-complex / complex-to-real lines (below) -- or, on a regular (Periodic, Bounded / Flat, Bounded / Flat) channel
                    // with Nx = 64 ... 512 and an even Ny, transform, division and inverse in one pass of the row kernel (colfft.hip MODE 8)
                    if (!gtri && topo[1] != OCN_PERIODIC && topo[2] != OCN_PERIODIC && N[1] % 2 == 0 && ocn::colfft_supported(N[0]) &&
                        !(gc && gc[0] == '0') && !(std::getenv("OCN_POISSON_ROW_DCT") && std::getenv("OCN_POISSON_ROW_DCT")[0] == '0')) {
                        s->growdct = true;
                        st = upload(ocn::colfft_twiddles(N[0]), &s->gcoltw[0]);
                    }
                } else if (s->gpacked) {
                    // y / z lines of the packed views: Nx / 2 complex columns under a cosine transform, the half spectrum under an FFT
                    const int Nd[3] = {topo[d] == OCN_BOUNDED ? N[0] / 2 : N[0] / 2 + 1, N[1], N[2]};
                    st = make_line_plans(s, d, Nd);
                } else {
                    st = make_line_plans(s, d, N);
                }
                if (st != OCN_SUCCESS || topo[d] != OCN_BOUNDED) continue;
                std::vector<double> w(2 * (size_t)N[d]);
                for (int p = 0; p < N[d]; ++p) {
                    const long double a = 3.14159265358979323846264338327950288L * kofp[p] / (2.0L * N[d]);
                    w[2 * p] = (double)cosl(a);
                    w[2 * p + 1] = (double)(-sinl(a));
                }
                st = upload(w, &s->gtw[d]);
                if (st == OCN_SUCCESS && s->gcol[d] && !s->gdct[d]) {
                    std::vector<int> partner(N[d]);
                    for (int p = 0; p < N[d]; ++p) partner[p] = pofk[(N[d] - kofp[p]) % N[d]];
                    if (hipMalloc((void **)&s->gpartner[d], N[d] * sizeof(int)) != hipSuccess ||
                        hipMemcpy(s->gpartner[d], partner.data(), N[d] * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                        ocn::set_error("ocn_poisson_create: partner table upload failed");
                        st = OCN_ERR_ALLOC;
                    }
                }
            }
        }
    }
    if (st == OCN_SUCCESS && (hipMalloc((void **)&s->spec, n * 2 * sizeof(double)) != hipSuccess ||
                              hipMalloc((void **)&s->spec2, n * 2 * sizeof(double)) != hipSuccess)) {
        ocn::set_error("ocn_poisson_create: out of device memory");
        st = OCN_ERR_ALLOC;
    }
    if (st == OCN_SUCCESS && s->gpacked && !s->growdct) {
        const size_t len[1] = {(size_t)N[0]}, one[1] = {1};
        const size_t nxh = (size_t)N[0] / 2 + 1, batch = (size_t)N[1] * N[2];
        st = make_plan(s->xr2c, rocfft_placement_notinplace, rocfft_transform_type_real_forward, 1, len, batch, rocfft_array_type_real,
                       rocfft_array_type_hermitian_interleaved, one, (size_t)N[0], one, nxh, 1.0);
        if (st == OCN_SUCCESS)
            st = make_plan(s->xc2r, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, 1, len, batch,
                           rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, one, nxh, one, (size_t)N[0], 1.0 / N[0]);
        if (st == OCN_SUCCESS) st = packed_plans_self_test(s);
        if (st != OCN_SUCCESS) {  // rocFFT's real plans are verified at creation like the periodic solver's (see Plan::destroy): complex path instead
            free_all(s);
            delete s;
            g_no_packed = true;
            const int st2 = poisson_create_general(out, grid);
            g_no_packed = false;
            return st2;
        }
    }
    if (st == OCN_SUCCESS && gtri) {
        // own copies of the spacings, lower = upper = 1/Δzᶠ[q], q = 2..Nz, main diagonal with the (stored-order) eigenvalues of x and y
        // (fourier_tridiagonal_poisson_solver.jl:41-51, 97-99)
        const int Nz = grid->Nz, Hz = grid->Hz;
        const size_t nf = (size_t)Nz + 2 * Hz;
        std::vector<double> hf(nf, grid->dz);
        bool ok = true;
        if (grid->dzc) {
            ok = hipMalloc((void **)&s->dzc, nf * sizeof(double)) == hipSuccess && hipMalloc((void **)&s->dzf, nf * sizeof(double)) == hipSuccess &&
                 hipMemcpy(s->dzc, grid->dzc, nf * sizeof(double), hipMemcpyDeviceToDevice) == hipSuccess &&
                 hipMemcpy(s->dzf, grid->dzf, nf * sizeof(double), hipMemcpyDeviceToDevice) == hipSuccess &&
                 hipMemcpy(hf.data(), grid->dzf, nf * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
            s->grid.dzc = s->dzc;
            s->grid.dzf = s->dzf;
        }
        ok = ok && hipMalloc((void **)&s->diag, n * sizeof(double)) == hipSuccess && hipMalloc((void **)&s->tscr, n * sizeof(double)) == hipSuccess;
        if (!ok) {
            ocn::set_error("ocn_poisson_create: out of device memory (tridiagonal arrays)");
            st = OCN_ERR_ALLOC;
        }
        if (st == OCN_SUCCESS) {
            std::vector<double> low(Nz - 1, 0.0);
            for (int q = 2; q <= Nz; ++q) low[q - 2] = 1 / hf[q + Hz - 1];
            st = upload(low, &s->lower);
        }
        if (st == OCN_SUCCESS) st = ocn::launch_main_diagonal(&s->grid, s->gpacked ? N[0] / 2 + 1 : N[0], s->lx, s->ly, s->diag, nullptr);
        if (st == OCN_SUCCESS && hipDeviceSynchronize() != hipSuccess) {
            ocn::set_error("ocn_poisson_create: main diagonal kernel failed");
            st = OCN_ERR_HIP;
        }
    }
    if (st != OCN_SUCCESS) {
        free_all(s);
        delete s;
        return st;
    }
    (void)hipMemset(s->spec, 0, n * 2 * sizeof(double));
    (void)hipMemset(s->spec2, 0, n * 2 * sizeof(double));
    *out = s;
    return OCN_SUCCESS;
}

static bool general_fuse_shuffles()
{
    static const bool fuse = !(std::getenv("OCN_POISSON_FUSE_SHUFFLES") && std::getenv("OCN_POISSON_FUSE_SHUFFLES")[0] == '0');
    return fuse;
}

// forward transforms Bounded first, then Periodic (plan_transforms.jl:53-57, 129-140); backward in the opposite order
static int poisson_solve_general(ocn_poisson *s, double *p, hipStream_t stream)
{
    const ocn_grid *g = &s->grid;
    const int N[3] = {g->Nx, g->Ny, g->Nz}, topo[3] = {g->tx, g->ty, s->gtri ? OCN_FLAT : g->tz};
    double *a = s->spec, *b = s->spec2;
    int order[3], no = 0;
    for (int d = 0; d < 3; ++d) if (topo[d] == OCN_BOUNDED) order[no++] = d;
    for (int d = 0; d < 3; ++d) if (topo[d] == OCN_PERIODIC) order[no++] = d;
    int pst = OCN_SUCCESS;
    // the extents of the (complex) array the passes see: the grid's, or -- packed -- Nx / 2 complex columns of x-adjacent real pairs under the
    // cosine transforms and the half spectrum (Nx / 2 + 1) between the real x transforms
    const int nxh = N[0] / 2 + 1;
    const int Nfull[3] = {N[0], N[1], N[2]}, Npair[3] = {N[0] / 2, N[1], N[2]}, Nhalf[3] = {nxh, N[1], N[2]};
    const int *Nv = Nfull;
    auto shuffle = [&](int d, int mode) {
        const long long n = (long long)Nv[0] * Nv[1] * Nv[2];
        hipLaunchKernelGGL(dct_shuffle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Nv[0], Nv[1], Nv[2], d, mode,
                           reinterpret_cast<const double2 *>(a), reinterpret_cast<double2 *>(b), reinterpret_cast<const double2 *>(s->gtw[d]),
                           (mode == 1 || mode == 2) ? s->gpartner[d] : nullptr);
        std::swap(a, b);
    };
    auto shuffle2 = [&](int dt, int mt, int dp, int mp) {
        const long long n = (long long)Nv[0] * Nv[1] * Nv[2];
        hipLaunchKernelGGL(dct_twiddle_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Nv[0], Nv[1], Nv[2], dt, mt, dp, mp,
                           reinterpret_cast<const double2 *>(a), reinterpret_cast<double2 *>(b), reinterpret_cast<const double2 *>(s->gtw[dt]),
                           s->gpartner[dt]);
        std::swap(a, b);
    };
    // the passes as a list of operations (kind 0 gather, 1 post-twiddle, 2 pre-twiddle, 3 scatter, 4 line FFT forward, 5 inverse, 6 direct sums,
    // 8 / 9 a whole forward / inverse cosine transform in the column kernel),
    // then a twiddle followed by a permutation along another dimension (forward: 1 then 0; inverse: 3 then 2) runs as ONE pass
    struct Op { int kind, d; };
    const bool fuse = general_fuse_shuffles();
    auto run = [&](const Op *ops, int nops) {
        for (int q = 0; q < nops && pst == OCN_SUCCESS; ++q) {
            const Op o = ops[q];
            if (fuse && q + 1 < nops && o.d != ops[q + 1].d && ((o.kind == 1 && ops[q + 1].kind == 0) || (o.kind == 3 && ops[q + 1].kind == 2))) {
                if (o.kind == 1) shuffle2(o.d, 1, ops[q + 1].d, 0);
                else shuffle2(ops[q + 1].d, 2, o.d, 3);
                ++q;
            } else if (o.kind <= 3) {
                shuffle(o.d, o.kind);
            } else if (o.kind == 6 || o.kind == 7) {
                const long long n = (long long)N[0] * N[1] * N[2];
                hipLaunchKernelGGL(naive_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, N[0], N[1], N[2], o.d, topo[o.d],
                                   o.kind == 7, reinterpret_cast<const double2 *>(a), reinterpret_cast<double2 *>(b), s->tab[o.d][0], s->tab[o.d][1]);
                std::swap(a, b);
            } else {
                pst = exec_line_plan(s, o.d, o.kind == 5 || o.kind == 9, a, Nv, stream, o.kind >= 8);
            }
        }
    };
    auto solve = [&](int nx) {  // the division by the eigenvalues, or the Thomas sweep along z, on an (nx, Ny, Nz) spectrum
        int st;
        if (s->gtri) {  // batched Thomas sweep along z, then the zero-mean gauge on the (kx, ky) = (0, 0) column (stored position 0 in either order)
            st = ocn::launch_tridiag_z(nx, N[1], N[2], s->lower, s->diag, s->lower, a, s->tscr, b, stream);
            if (st != OCN_SUCCESS) return st;
            std::swap(a, b);
            return ocn::launch_remove_mean_mode((long long)nx * N[1], N[2], a, stream);
        }
        // -b / (λx + λy + λz [- m]), mode (1,1,1) := 0 iff m === 0
        return ocn::launch_spectral_solve(nx, N[1], N[2], s->lx, s->ly, s->lz, a, 1, 0, 0, stream, s->shift, s->shifted);
    };
    // the inverse cosine transforms of a solve on real pairs (Nv == Npair) and the way into the pressure field: when the last of them is a
    // one-pass column transform it stores straight into p (colfft.hip MODE 4, ColFFTArgs::preal) -- no copy_real_component! pass;
    // otherwise the last scatter is folded into the copy  [OCN_POISSON_DCT_TO_FIELD=0: always the copy]
    auto finish_real = [&](const Op *ops, int nops) -> int {
        static const bool direct = !(std::getenv("OCN_POISSON_DCT_TO_FIELD") && std::getenv("OCN_POISSON_DCT_TO_FIELD")[0] == '0');
        if (direct && nops > 0 && ops[nops - 1].kind == 9) {
            run(ops, nops - 1);
            if (pst != OCN_SUCCESS) return pst;
            const int d = ops[nops - 1].d;
            const ocn::GridDev gd = ocn::to_dev(*g);
            const ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
            const long long p0 = ocn::at(Lp, 1, 1, 1), plane = (long long)Nv[0] * Nv[1];
            int st = d == 1 ? ocn::launch_colfft_dct_to_field(N[1], a, Nv[0], plane, Nv[0], Nv[2], s->gcoltw[1], s->gtw[1], 1.0 / N[1], Nv[0], 1, p,
                                                              p0, Lp.s2, Lp.s3, stream)
                            : ocn::launch_colfft_dct_to_field(N[2], a, plane, 0, (int)plane, 1, s->gcoltw[2], s->gtw[2], 1.0 / N[2], Nv[0], 2, p, p0,
                                                              Lp.s2, Lp.s3, stream);
            if (a != s->spec) std::swap(s->spec, s->spec2);
            return st;
        }
        const int last_scatter = (fuse && nops > 0 && ops[nops - 1].kind == 3) ? ops[nops - 1].d : -1;
        run(ops, nops - (last_scatter >= 0 ? 1 : 0));
        if (pst != OCN_SUCCESS) return pst;
        OCN_CHECK_HIP(hipGetLastError());
        if (a != s->spec) std::swap(s->spec, s->spec2);
        return ocn::launch_copy_real(g, s->spec, p, stream, /*real_source=*/1, last_scatter);
    };
    Op fwd[9], bwd[9];
    int nf = 0, nb = 0;
    auto push_fwd = [&](int d) {  // REDFT10: gather, FFT, twiddle -- or the one fused pass
        if (s->gdct[d]) { fwd[nf++] = Op{8, d}; return; }
        fwd[nf++] = Op{0, d}; fwd[nf++] = Op{4, d}; fwd[nf++] = Op{1, d};
    };
    auto push_bwd = [&](int d) {  // REDFT01 / 2N: twiddle, inverse FFT (scaled 1 / N), scatter -- or the one fused pass
        if (s->gdct[d]) { bwd[nb++] = Op{9, d}; return; }
        bwd[nb++] = Op{2, d}; bwd[nb++] = Op{5, d}; bwd[nb++] = Op{3, d};
    };
    if (s->gallreal) {
        // ---- y, z on the pair view (the source's rows are already gathered along y when compute_source_term stored it that way)
        Nv = Npair;
        for (int d = 1; d < 3; ++d)
            if (topo[d] == OCN_BOUNDED) push_fwd(d);
        for (int d = 2; d >= 1; --d)
            if (topo[d] == OCN_BOUNDED) push_bwd(d);
        const bool skip_gather = s->gathered && nf > 0 && fwd[0].kind == 0;
        s->gathered = false;
        run(fwd + (skip_gather ? 1 : 0), nf - (skip_gather ? 1 : 0));
        if (pst != OCN_SUCCESS) return pst;
        // ---- x on pairs of rows: gather, line FFT, twiddle back to the real array; the division; and back
        const int Nrow[3] = {N[0], N[1] / 2, N[2]};
        const long long nr = (long long)Nrow[0] * Nrow[1] * Nrow[2], nreal = (long long)N[0] * N[1] * N[2];
        auto rowpair = [&](int mode) {
            hipLaunchKernelGGL(dct_rowpair_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, stream, Nrow[0], Nrow[1], Nrow[2], mode, a, b,
                               reinterpret_cast<const double2 *>(s->gtw[0]));
            std::swap(a, b);
        };
        auto xrow = [&](int mode) {  // 5 forward, 6 inverse, 7 forward + division + inverse: the row kernel, in place
            return ocn::launch_rowdct(N[0], N[1], N[2], mode, a, s->gcoltw[0], s->gtw[0], s->lx, s->ly, s->lz, s->shift, s->shifted ? 1 : 0, stream);
        };
        if (s->growdct && !s->gtri) {
            pst = xrow(7);
            if (pst != OCN_SUCCESS) return pst;
        } else {
            if (s->growdct) {
                pst = xrow(5);
            } else {
                rowpair(0);
                pst = exec_line_plan(s, 0, 0, a, Nrow, stream);
                if (pst != OCN_SUCCESS) return pst;
                rowpair(1);
            }
            if (pst != OCN_SUCCESS) return pst;
            if (s->gtri) {  // the Thomas sweep along the stretched z of a closed box, on reals, and the zero-mean gauge on the (0, 0) column
                int st = ocn::launch_tridiag_z_real(N[0], N[1], N[2], s->lower, s->diag, s->lower, a, s->tscr, b, stream);
                if (st != OCN_SUCCESS) return st;
                std::swap(a, b);
                st = ocn::launch_remove_mean_mode_real((long long)N[0] * N[1], N[2], a, stream);
                if (st != OCN_SUCCESS) return st;
            } else {
                hipLaunchKernelGGL(spectral_solve_real_kernel, dim3((unsigned)((nreal + 255) / 256)), dim3(256), 0, stream, N[0], N[1], N[2], s->lx,
                                   s->ly, s->lz, a, s->shift, s->shifted ? 1 : 0);
            }
            if (s->growdct) {
                pst = xrow(6);
            } else {
                rowpair(2);
                pst = exec_line_plan(s, 0, 1, a, Nrow, stream);
                if (pst != OCN_SUCCESS) return pst;
                rowpair(3);
            }
            if (pst != OCN_SUCCESS) return pst;
        }
        // ---- z, y back on the pair view; the last scatter is folded into the copy into the pressure field
        return finish_real(bwd, nb);
    }
    if (s->gpacked) {
        // ---- cosine transforms along the Bounded y / z on the real array viewed as Nx / 2 complex columns
        Nv = Npair;
        for (int q = 0; q < no; ++q) {
            const int d = order[q];
            if (topo[d] == OCN_BOUNDED) push_fwd(d);
        }
        for (int q = no - 1; q >= 0; --q) {
            const int d = order[q];
            if (topo[d] == OCN_BOUNDED) push_bwd(d);
        }
        const bool skip_gather = s->gathered && nf > 0 && fwd[0].kind == 0;
        s->gathered = false;
        run(fwd + (skip_gather ? 1 : 0), nf - (skip_gather ? 1 : 0));
        if (pst != OCN_SUCCESS) return pst;
        if (s->growdct) {  // x: transform, division and inverse of row pairs in place, then straight to the inverse cosine transforms
            int st = ocn::launch_rowdct(N[0], N[1], N[2], 8, a, s->gcoltw[0], nullptr, s->lx, s->ly, s->lz, s->shift, s->shifted ? 1 : 0, stream);
            if (st != OCN_SUCCESS) return st;
            return finish_real(bwd, nb);
        }
        // ---- x: real rows -> half spectrum; the other Periodic direction on the half spectrum
        int st = s->xr2c.exec(a, b, stream);
        if (st != OCN_SUCCESS) return st;
        std::swap(a, b);
        Nv = Nhalf;
        Op pf[2], pb[2];
        int np = 0;
        for (int d = 1; d < 3; ++d)
            if (topo[d] == OCN_PERIODIC) { pf[np] = Op{4, d}; pb[np] = Op{5, d}; ++np; }
        run(pf, np);
        if (pst != OCN_SUCCESS) return pst;
        st = solve(nxh);
        if (st != OCN_SUCCESS) return st;
        for (int q = np - 1; q >= 0 && pst == OCN_SUCCESS; --q) run(pb + q, 1);
        if (pst != OCN_SUCCESS) return pst;
        st = s->xc2r.exec(a, b, stream);  // (scaled 1 / Nx)
        if (st != OCN_SUCCESS) return st;
        std::swap(a, b);
        // ---- inverse cosine transforms on the pair view; the last scatter is folded into the copy into the pressure field
        Nv = Npair;
        return finish_real(bwd, nb);
    }
    for (int q = 0; q < no; ++q) {
        const int d = order[q];
        if (!s->fft_dct) fwd[nf++] = Op{6, d};
        else if (topo[d] == OCN_PERIODIC) fwd[nf++] = Op{4, d};
        else push_fwd(d);
    }
    for (int q = no - 1; q >= 0; --q) {
        const int d = order[q];
        if (!s->fft_dct) bwd[nb++] = Op{7, d};
        else if (topo[d] == OCN_PERIODIC) bwd[nb++] = Op{5, d};
        else push_bwd(d);
    }
    const bool skip_gather = s->gathered && nf > 0 && fwd[0].kind == 0;
    s->gathered = false;
    run(fwd + (skip_gather ? 1 : 0), nf - (skip_gather ? 1 : 0));
    if (pst != OCN_SUCCESS) return pst;
    int st = solve(N[0]);
    if (st != OCN_SUCCESS) return st;
    // the scatter pass of the last inverse cosine transform is folded into the read of copy_real_component!
    const int last_scatter = (fuse && nb > 0 && bwd[nb - 1].kind == 3) ? bwd[nb - 1].d : -1;
    run(bwd, nb - (last_scatter >= 0 ? 1 : 0));
    if (pst != OCN_SUCCESS) return pst;
    OCN_CHECK_HIP(hipGetLastError());
    if (a != s->spec) std::swap(s->spec, s->spec2);  // the result lives in `a`; keep the handle's roles consistent
    return ocn::launch_copy_real(g, s->spec, p, stream, 0, last_scatter);
}

// The real x plans of the packed general solver against known answers: the spectrum of one pseudo-random row against the DFT of its
// definition, and the round trip of a block of rows (see Plan::destroy for why rocFFT's real plans are not trusted unverified).
static int packed_plans_self_test(ocn_poisson *s)
{
    const ocn_grid *g = &s->grid;
    const int N0 = g->Nx, nxh = N0 / 2 + 1;
    const size_t rows = (size_t)g->Ny * g->Nz, n = (size_t)N0 * rows;
    const size_t m = std::min(n, (size_t)1 << 18) / N0 * N0;  // whole rows
    std::vector<double> in(m), back(m), spec(2 * (size_t)nxh);
    unsigned long long x = 88172645463325252ULL;
    for (size_t q = 0; q < m; ++q) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        in[q] = (double)(x >> 11) / 9007199254740992.0 - 0.5;
    }
    OCN_CHECK_HIP(hipMemset(s->spec, 0, n * sizeof(double)));
    OCN_CHECK_HIP(hipMemcpy(s->spec, in.data(), m * sizeof(double), hipMemcpyHostToDevice));
    int st = s->xr2c.exec(s->spec, s->spec2, nullptr);
    if (st != OCN_SUCCESS) return st;
    OCN_CHECK_HIP(hipDeviceSynchronize());
    OCN_CHECK_HIP(hipMemcpy(spec.data(), s->spec2, spec.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double two_pi = 6.283185307179586476925286766559;
    double err = 0.0, scale = 0.0;
    for (int k = 0; k < nxh; ++k) {
        double re = 0.0, im = 0.0;
        for (int q = 0; q < N0; ++q) {
            const double ang = two_pi * (double)(((long long)k * q) % N0) / N0;
            re += in[q] * std::cos(ang);
            im -= in[q] * std::sin(ang);
        }
        err = std::max(err, std::max(std::fabs(re - spec[2 * k]), std::fabs(im - spec[2 * k + 1])));
        scale = std::max(scale, std::max(std::fabs(re), std::fabs(im)));
    }
    st = s->xc2r.exec(s->spec2, s->spec, nullptr);
    if (st != OCN_SUCCESS) return st;
    OCN_CHECK_HIP(hipDeviceSynchronize());
    OCN_CHECK_HIP(hipMemcpy(back.data(), s->spec, m * sizeof(double), hipMemcpyDeviceToHost));
    double rt = 0.0;
    for (size_t q = 0; q < m; ++q) rt = std::max(rt, std::fabs(back[q] - in[q]));
    OCN_CHECK_HIP(hipMemset(s->spec, 0, n * 2 * sizeof(double)));
    if (!(err <= 1e-11 * std::max(scale, 1.0) * N0 && rt <= 1e-12)) {
        ocn::set_error("rocFFT real x plans of the packed general solver failed their self test (spectrum %.3e, round trip %.3e)", err, rt);
        return OCN_ERR_ROCFFT;
    }
    return OCN_SUCCESS;
}

static int poisson_create_impl(ocn_poisson_t *out, const ocn_grid *grid, bool force_c2c)
{
    OCN_REQUIRE(out && grid, "ocn_poisson_create: null argument");
    {
        const char *eg = std::getenv("OCN_POISSON_GENERAL");
        const bool bounded_xy = grid->tx == OCN_BOUNDED || grid->ty == OCN_BOUNDED || grid->tx == OCN_FLAT || grid->ty == OCN_FLAT;
        if (bounded_xy || (eg && eg[0] == '1' && grid->dzc == nullptr)) {
            // (the solver only touches a Center field, whose layout does not depend on the topology: its own light validation)
            OCN_REQUIRE(grid->Nx >= 1 && grid->Ny >= 1 && grid->Nz >= 1 && grid->Hx >= 0 && grid->Hy >= 0 && grid->Hz >= 0, "bad grid size / halo");
            OCN_REQUIRE(grid->dx > 0 && grid->dy > 0 && (grid->dzc || grid->dz > 0) && grid->Lx > 0 && grid->Ly > 0 && grid->Lz > 0, "spacings and extents must be positive");
            OCN_REQUIRE((grid->dzc == nullptr) == (grid->dzf == nullptr), "dzc and dzf must both be set or both be NULL");
            return poisson_create_general(out, grid);
        }
    }
    int st = ocn::validate_grid(grid);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(grid->tx == OCN_PERIODIC && grid->ty == OCN_PERIODIC,
                "ocn_poisson_create: x and y must be Periodic or Bounded (distributed grids use ocn_dist_poisson_create)");
    ensure_rocfft();
    ocn_poisson *s = new ocn_poisson();
    s->grid = *grid;
    const int Nx = grid->Nx, Ny = grid->Ny, Nz = grid->Nz, Hz = grid->Hz;
    const char *env = std::getenv("OCN_POISSON_C2C");
    s->c2c = force_c2c || (env && env[0] == '1');
    if (grid->tz == OCN_BOUNDED) {
        s->kind = 1;  // nonhydrostatic_pressure_solver dispatch (NonhydrostaticModels.jl:25-62); see DESIGN.md for z regular+Bounded
    } else {
        if (grid->dzc != nullptr) {
            delete s;
            ocn::set_error("FFTBasedPoissonSolver requires a regular z direction");
            return OCN_ERR_UNSUPPORTED;
        }
        s->kind = 0;
    }
    s->nxh = s->c2c ? Nx : Nx / 2 + 1;
    const size_t nspec = (size_t)s->nxh * Ny * Nz;

#define TRY(expr)            \
    do {                     \
        int _st = (expr);    \
        if (_st != OCN_SUCCESS) { free_all(s); delete s; return _st; } \
    } while (0)
#define TRY_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ocn::set_error("%s failed: %s", #expr, hipGetErrorString(_e));                    \
            free_all(s); delete s; return OCN_ERR_ALLOC;                                      \
        }                                                                                     \
    } while (0)

    if (grid->dzc) {  // own copies: the handle must not depend on caller arrays staying alive
        const size_t nc = Nz + 2 * Hz, nf = Nz + 2 * Hz;
        TRY_HIP(hipMalloc((void **)&s->dzc, nc * sizeof(double)));
        TRY_HIP(hipMalloc((void **)&s->dzf, nf * sizeof(double)));
        TRY_HIP(hipMemcpy(s->dzc, grid->dzc, nc * sizeof(double), hipMemcpyDeviceToDevice));
        TRY_HIP(hipMemcpy(s->dzf, grid->dzf, nf * sizeof(double), hipMemcpyDeviceToDevice));
        s->grid.dzc = s->dzc;
        s->grid.dzf = s->dzf;
    }
    TRY(upload(eigenvalues(Nx, grid->Lx, grid->tx), &s->lx));
    TRY(upload(eigenvalues(Ny, grid->Ly, grid->ty), &s->ly));
    TRY(upload(eigenvalues(Nz, grid->Lz, s->kind == 0 ? grid->tz : OCN_FLAT), &s->lz));
    TRY_HIP(hipMalloc((void **)&s->spec, nspec * 2 * sizeof(double)));
    TRY_HIP(hipMemset(s->spec, 0, nspec * 2 * sizeof(double)));
    if (!s->c2c) {
        TRY_HIP(hipMalloc((void **)&s->rhs, (size_t)Nx * Ny * Nz * sizeof(double)));
        TRY_HIP(hipMemset(s->rhs, 0, (size_t)Nx * Ny * Nz * sizeof(double)));
    }
    if (s->kind == 1) {
        TRY_HIP(hipMalloc((void **)&s->spec2, nspec * 2 * sizeof(double)));
        TRY_HIP(hipMemset(s->spec2, 0, nspec * 2 * sizeof(double)));
        TRY_HIP(hipMalloc((void **)&s->diag, nspec * sizeof(double)));
        TRY_HIP(hipMalloc((void **)&s->tscr, nspec * sizeof(double)));
        // lower = upper = 1/Δzᶠ[q], q = 2..Nz (fourier_tridiagonal_poisson_solver.jl:97-99)
        std::vector<double> hf(Nz + 2 * Hz, grid->dz);
        if (grid->dzf) TRY_HIP(hipMemcpy(hf.data(), grid->dzf, hf.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> low(Nz > 1 ? Nz - 1 : 1, 0.0);
        for (int q = 2; q <= Nz; ++q) low[q - 2] = 1 / hf[q + Hz - 1];
        TRY(upload(low, &s->lower));
        {
            const char *e = std::getenv("OCN_POISSON_CUSTOM_XY");
            s->custom_tri = !s->c2c && ocn::rowfft_supported(Nx) && ocn::colfft_supported(Ny) && Nz > 1 && !(e && e[0] == '0');
        }
        if (s->custom_tri) {  // ky stays in the column kernel's stage order between its two passes: permuted eigenvalues
            std::vector<double> a, b;
            ocn::rowfft_twiddles(Nx, a, b);
            TRY(upload(a, &s->twMx));
            TRY(upload(b, &s->twNx));
            TRY(upload(ocn::colfft_twiddles(Ny), &s->twy));
            std::vector<double> lyn = eigenvalues(Ny, grid->Ly, grid->ty), lys(Ny);
            for (int p = 0; p < Ny; ++p) lys[p] = lyn[ocn::colfft_wavenumber(Ny, p)];
            TRY(upload(lys, &s->ly_stage));
            s->custom_xy = true;
            const char *ez = std::getenv("OCN_POISSON_DCT_Z");
            s->dct_z = grid->dzc == nullptr && ocn::colfft_supported(Nz) && !(ez && ez[0] == '0');
            if (s->dct_z) {
                TRY(upload(ocn::colfft_twiddles(Nz), &s->tw));
                std::vector<double> w(2 * (size_t)Nz);
                for (int k = 0; k < Nz; ++k) {
                    const long double ang = 3.14159265358979323846264338327950288L * k / (2.0L * Nz);
                    w[2 * k] = (double)cosl(ang);
                    w[2 * k + 1] = (double)(-sinl(ang));
                }
                TRY(upload(w, &s->wdz));
                TRY(upload(eigenvalues(Nz, grid->Lz, OCN_BOUNDED), &s->lz_bounded));
            }
        }
        TRY(ocn::launch_main_diagonal(&s->grid, s->nxh, s->lx, s->custom_tri ? s->ly_stage : s->ly, s->diag, nullptr));
    }

    // Layout of the pressure field interior as an FFT output (r2c path): strides (1, sx, sx*sy)
    ocn::GridDev gd = ocn::to_dev(*grid);
    ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
    {
        const char *e = std::getenv("OCN_POISSON_FUSED_Z");
        s->fused_z = s->kind == 0 && grid->tz == OCN_PERIODIC && !s->c2c && ocn::colfft_supported(Nz) && !(e && e[0] == '0');
    }
    if (s->fused_z) {
        TRY(upload(ocn::colfft_twiddles(Nz), &s->tw));
        std::vector<double> lzn = eigenvalues(Nz, grid->Lz, grid->tz), lzs(Nz);
        for (int p = 0; p < Nz; ++p) lzs[p] = lzn[ocn::colfft_wavenumber(Nz, p)];  // eigenvalue of each stored position
        TRY(upload(lzs, &s->lz_stage));
    }
    {
        const char *e = std::getenv("OCN_POISSON_CUSTOM_XY");
        s->custom_xy = s->custom_tri || (s->fused_z && ocn::rowfft_supported(Nx) && ocn::colfft_supported(Ny) && !(e && e[0] == '0'));
    }
    if (s->custom_xy && !s->custom_tri) {
        std::vector<double> a, b;
        ocn::rowfft_twiddles(Nx, a, b);
        TRY(upload(a, &s->twMx));
        TRY(upload(b, &s->twNx));
        TRY(upload(ocn::colfft_twiddles(Ny), &s->twy));
        std::vector<double> lyn = eigenvalues(Ny, grid->Ly, grid->ty), lys(Ny);
        for (int p = 0; p < Ny; ++p) lys[p] = lyn[ocn::colfft_wavenumber(Ny, p)];  // y is kept in stage order between its two passes
        TRY(upload(lys, &s->ly_stage));
    }
    if (!s->custom_tri) {
    const int fft_dims = (s->kind == 0 && grid->tz == OCN_PERIODIC && !s->fused_z) ? 3 : 2;
    const size_t batch = (fft_dims == 3) ? 1 : (size_t)Nz;
    const size_t len[3] = {(size_t)Nx, (size_t)Ny, (size_t)Nz};
    // fused z path: the whole 1/(Nx Ny Nz) normalisation is applied in the column kernel
    const double scale = s->fused_z ? 1.0 : (fft_dims == 3) ? 1.0 / ((double)Nx * Ny * Nz) : 1.0 / ((double)Nx * Ny);
    if (s->c2c) {
        const size_t str[3] = {1, (size_t)Nx, (size_t)Nx * Ny};
        const size_t dist = (size_t)Nx * Ny * (fft_dims == 3 ? Nz : 1);
        TRY(make_plan(s->fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, fft_dims, len, batch,
                      rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, dist, str, dist, 1.0));
        TRY(make_plan(s->bwd, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, fft_dims, len, batch,
                      rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, dist, str, dist, scale));
    } else {
        const size_t rstr[3] = {1, (size_t)Nx, (size_t)Nx * Ny};
        const size_t cstr[3] = {1, (size_t)s->nxh, (size_t)s->nxh * Ny};
        const size_t pstr[3] = {1, (size_t)Lp.s2, (size_t)Lp.s3};
        const size_t rdist = (size_t)Nx * Ny * (fft_dims == 3 ? Nz : 1);
        const size_t cdist = (size_t)s->nxh * Ny * (fft_dims == 3 ? Nz : 1);
        const size_t pdist = (fft_dims == 3) ? (size_t)Lp.s3 * Lp.sz : (size_t)Lp.s3;
        TRY(make_plan(s->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, fft_dims, len, batch,
                      rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, rstr, rdist, cstr, cdist, 1.0));
        const char *ed = std::getenv("OCN_POISSON_DIRECT_OUT");
        int bst = (ed && ed[0] == '0') ? OCN_ERR_UNSUPPORTED
                                       : make_plan(s->bwd, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, fft_dims, len, batch,
                                                   rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, cstr, cdist, pstr, pdist, scale);
        if (bst != OCN_SUCCESS) {
            // rocFFT has no kernel for this strided-output shape: transform into the contiguous real buffer and
            // copy into the pressure interior (the reference's copy_real_component! pass).
            s->bwd.destroy();
            s->direct_out = false;
            TRY(make_plan(s->bwd, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, fft_dims, len, batch,
                          rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, cstr, cdist, rstr, rdist, scale));
        }
    }
    }
#undef TRY
#undef TRY_HIP
    // rocFFT real-transform plans have been observed to return wrong results when certain other real plans already exist in
    // the process (e.g. a 16^3 3-D R2C plan followed by a batched 32x8 2-D one, rocFFT of ROCm 7.2): every plan pair is
    // verified by a forward + inverse round trip before the handle is handed out; a failing real pair is replaced by
    // complex plans, a failing complex pair is an error.
    if (!s->custom_xy) {
        st = poisson_plans_self_test(s);
        if (st != OCN_SUCCESS) {
            const bool was_c2c = s->c2c;
            free_all(s);
            delete s;
            if (was_c2c) return st;
            return poisson_create_impl(out, grid, true);
        }
    }
    *out = s;
    return OCN_SUCCESS;
}

// Known-answer checks of a forward transform over the FULL spectrum.  The round trip alone does not catch a forward / inverse pair
// that is wrong in a mutually consistent way, and a sample of entries misses permuted spectra.
//  (a) plane waves, any grid size, O(n): the input is a sum of NW real plane waves with distinct amplitudes and phases whose DFT is
//      known in closed form (two conjugate deltas per wave, folded into the half spectrum); every entry of the device spectrum is
//      compared with it.
//  (b) grids of at most 2^16 points: a pseudo-random field against a separable DFT evaluated on the host.
// spec(kx, ky, kz) at kx + nxh (ky + Ny kz); dims3 = the plan transforms z as well (otherwise z is the batch index).
struct Wave {
    int kx, ky, kz;
    double amp, phase;
};
static std::vector<Wave> test_waves(int Nx, int Ny, int Nz, bool dims3)
{
    std::vector<Wave> w;
    unsigned long long x = 0xD1B54A32D192ED03ull;
    auto next = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(x >> 33); };
    for (int q = 0; q < 12; ++q)
        w.push_back(Wave{(int)(next() % (unsigned)Nx), (int)(next() % (unsigned)Ny), dims3 ? (int)(next() % (unsigned)Nz) : 0, 0.5 + 0.125 * q,
                         0.37 * q + 0.11});
    w.push_back(Wave{1 % Nx, 0, 0, 1.7, 0.3});         // the axes, the Nyquist lines and the mean, each with its own amplitude
    w.push_back(Wave{0, 1 % Ny, 0, 2.3, -0.4});
    w.push_back(Wave{Nx / 2, Ny / 2, dims3 ? Nz / 2 : 0, 2.9, 0.0});
    w.push_back(Wave{0, 0, 0, 3.1, 0.0});
    if (dims3) w.push_back(Wave{0, 0, 1 % Nz, 3.7, 0.9});
    return w;
}
// in[i,j,k] = sum_w amp cos(2 pi (kx i / Nx + ky j / Ny + kz k / Nz) + phase)  [2-D plans: every z plane gets (1 + k) x the field]
static void fill_waves(std::vector<double> &in, const std::vector<Wave> &W, int Nx, int Ny, int Nz, bool dims3)
{
    const double two_pi = 6.283185307179586476925286766559;
    std::fill(in.begin(), in.end(), 0.0);
    for (const Wave &w : W) {
        std::vector<double> cx(Nx), sx(Nx), cy(Ny), sy(Ny), cz(Nz), sz(Nz);
        for (int i = 0; i < Nx; ++i) { const double a = two_pi * (double)((long long)w.kx * i % Nx) / Nx; cx[i] = std::cos(a); sx[i] = std::sin(a); }
        for (int j = 0; j < Ny; ++j) { const double a = two_pi * (double)((long long)w.ky * j % Ny) / Ny; cy[j] = std::cos(a); sy[j] = std::sin(a); }
        for (int k = 0; k < Nz; ++k) { const double a = dims3 ? two_pi * (double)((long long)w.kz * k % Nz) / Nz : 0.0; cz[k] = std::cos(a); sz[k] = std::sin(a); }
        const double cp = std::cos(w.phase), sp = std::sin(w.phase);
        for (int k = 0; k < Nz; ++k) {
            const double scale = dims3 ? 1.0 : 1.0 + k;
            for (int j = 0; j < Ny; ++j) {
                // cos(a + b + c + phase) with a = x angle, (b, c, phase) folded into (C, S): cos(a) C - sin(a) S
                const double cyz = cy[j] * cz[k] - sy[j] * sz[k], syz = sy[j] * cz[k] + cy[j] * sz[k];
                const double Cc = cyz * cp - syz * sp, Ss = syz * cp + cyz * sp;
                double *row = &in[(size_t)Nx * (j + (size_t)Ny * k)];
                for (int i = 0; i < Nx; ++i) row[i] += scale * w.amp * (cx[i] * Cc - sx[i] * Ss);
            }
        }
    }
}
// max |device spectrum - closed form| over every stored entry
static double wave_spectrum_error(const std::vector<double> &spec, const std::vector<Wave> &W, int Nx, int Ny, int Nz, int nxh, bool dims3)
{
    std::vector<double> ex(spec.size(), 0.0);
    const double count = (double)Nx * Ny * (dims3 ? Nz : 1);
    for (const Wave &w : W) {
        // amp cos(theta + phase) = amp/2 (e^{i phase} e^{i theta} + c.c.): delta at +k with amp/2 e^{i phase}, at -k with the conjugate
        for (int sgn = 0; sgn < 2; ++sgn) {
            const int kx = sgn ? (Nx - w.kx) % Nx : w.kx, ky = sgn ? (Ny - w.ky) % Ny : w.ky, kz = sgn ? (Nz - w.kz) % Nz : w.kz;
            if (kx >= nxh) continue;  // the other half of the Hermitian spectrum is not stored
            const double re = 0.5 * w.amp * std::cos(w.phase) * count, im = (sgn ? -1.0 : 1.0) * 0.5 * w.amp * std::sin(w.phase) * count;
            for (int k = 0; k < (dims3 ? 1 : Nz); ++k) {
                const size_t o = 2 * ((size_t)kx + (size_t)nxh * (ky + (size_t)Ny * (dims3 ? kz : k)));
                const double scale = dims3 ? 1.0 : 1.0 + k;
                ex[o] += scale * re;
                ex[o + 1] += scale * im;
            }
        }
    }
    double worst = 0.0;
    for (size_t q = 0; q < ex.size(); ++q) worst = std::fmax(worst, std::fabs(ex[q] - spec[q]));
    return worst;
}
// full separable host DFT of a small real field -> max error over every stored entry
static double full_spectrum_error(const std::vector<double> &in, const std::vector<double> &spec, int Nx, int Ny, int Nz, int nxh, bool dims3)
{
    const double two_pi = 6.283185307179586476925286766559;
    const size_t m = (size_t)nxh * Ny * Nz;
    std::vector<double> a(2 * m), b(2 * m);
    for (int k = 0; k < Nz; ++k)
        for (int j = 0; j < Ny; ++j)
            for (int kx = 0; kx < nxh; ++kx) {
                double re = 0, im = 0;
                for (int i = 0; i < Nx; ++i) {
                    const double ph = -two_pi * (double)((long long)kx * i % Nx) / Nx, v = in[i + (size_t)Nx * (j + (size_t)Ny * k)];
                    re += v * std::cos(ph);
                    im += v * std::sin(ph);
                }
                const size_t o = 2 * (kx + (size_t)nxh * (j + (size_t)Ny * k));
                a[o] = re; a[o + 1] = im;
            }
    auto pass = [&](std::vector<double> &src, std::vector<double> &dst, int N, size_t stride, size_t outer_stride, size_t n_outer, size_t n_inner) {
        for (size_t oo = 0; oo < n_outer; ++oo)
            for (size_t ii = 0; ii < n_inner; ++ii)
                for (int kk = 0; kk < N; ++kk) {
                    double re = 0, im = 0;
                    for (int q = 0; q < N; ++q) {
                        const double ph = -two_pi * (double)((long long)kk * q % N) / N, c = std::cos(ph), sn = std::sin(ph);
                        const size_t o = 2 * (ii + stride * q + outer_stride * oo);
                        re += src[o] * c - src[o + 1] * sn;
                        im += src[o] * sn + src[o + 1] * c;
                    }
                    const size_t o = 2 * (ii + stride * kk + outer_stride * oo);
                    dst[o] = re; dst[o + 1] = im;
                }
    };
    pass(a, b, Ny, nxh, (size_t)nxh * Ny, Nz, nxh);              // along y
    if (dims3) { a = b; pass(a, b, Nz, (size_t)nxh * Ny, 0, 1, (size_t)nxh * Ny); }  // along z
    double worst = 0.0;
    for (size_t q = 0; q < 2 * m; ++q) worst = std::fmax(worst, std::fabs(b[q] - spec[q]));
    return worst;
}

// forward then inverse transform of a pseudo-random field must reproduce it (times the plans' net scale), and the forward
// spectrum must be the DFT of the input
static int poisson_plans_self_test(ocn_poisson *s)
{
    const ocn_grid *g = &s->grid;
    const int Nx = g->Nx, Ny = g->Ny, Nz = g->Nz;
    const size_t n = (size_t)Nx * Ny * Nz;
    const bool three_d = (s->kind == 0 && g->tz == OCN_PERIODIC && !s->fused_z);
    const double count = (double)Nx * Ny * (three_d ? Nz : 1);
    const double net = s->fused_z ? count : 1.0;  // inverse plans carry 1/count unless the column kernel applies it
    std::vector<double> in(n), outv;
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (size_t q = 0; q < n; ++q) {  // splitmix-style generator, values in (-1, 1)
        x += 0x9E3779B97F4A7C15ull;
        unsigned long long z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        in[q] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    double err = 0.0, ferr = 0.0;
    if (s->c2c) {
        std::vector<double> c(2 * n, 0.0);
        for (size_t q = 0; q < n; ++q) c[2 * q] = in[q];
        OCN_CHECK_HIP(hipMemcpy(s->spec, c.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice));
        int st = s->fwd.exec(s->spec, nullptr, nullptr);
        if (st != OCN_SUCCESS) return st;
        OCN_CHECK_HIP(hipDeviceSynchronize());
        OCN_CHECK_HIP(hipMemcpy(c.data(), s->spec, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
        if (n <= (size_t)1 << 16) ferr = full_spectrum_error(in, c, Nx, Ny, Nz, Nx, three_d);
        st = s->bwd.exec(s->spec, nullptr, nullptr);
        if (st != OCN_SUCCESS) return st;
        OCN_CHECK_HIP(hipDeviceSynchronize());
        OCN_CHECK_HIP(hipMemcpy(c.data(), s->spec, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < n; ++q) {
            err = std::fmax(err, std::fabs(c[2 * q] - net * in[q]));
            err = std::fmax(err, std::fabs(c[2 * q + 1]));
        }
        {   // plane waves: closed-form answer for every entry, any grid size
            const std::vector<Wave> W = test_waves(Nx, Ny, Nz, three_d);
            fill_waves(in, W, Nx, Ny, Nz, three_d);
            std::fill(c.begin(), c.end(), 0.0);
            for (size_t q = 0; q < n; ++q) c[2 * q] = in[q];
            OCN_CHECK_HIP(hipMemcpy(s->spec, c.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice));
            st = s->fwd.exec(s->spec, nullptr, nullptr);
            if (st != OCN_SUCCESS) return st;
            OCN_CHECK_HIP(hipDeviceSynchronize());
            OCN_CHECK_HIP(hipMemcpy(c.data(), s->spec, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
            ferr = std::fmax(ferr, wave_spectrum_error(c, W, Nx, Ny, Nz, Nx, three_d));
        }
        OCN_CHECK_HIP(hipMemset(s->spec, 0, 2 * n * sizeof(double)));
    } else {
        ocn::GridDev gd = ocn::to_dev(*g);
        ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
        const size_t np = (size_t)Lp.sx * Lp.sy * Lp.sz;
        double *tmp = nullptr;
        OCN_CHECK_HIP(hipMemcpy(s->rhs, in.data(), n * sizeof(double), hipMemcpyHostToDevice));
        int st = s->fwd.exec(s->rhs, s->spec, nullptr);
        if (st != OCN_SUCCESS) return st;
        {
            OCN_CHECK_HIP(hipDeviceSynchronize());
            std::vector<double> sp((size_t)s->nxh * Ny * Nz * 2);
            OCN_CHECK_HIP(hipMemcpy(sp.data(), s->spec, sp.size() * sizeof(double), hipMemcpyDeviceToHost));
            if (n <= (size_t)1 << 16) ferr = full_spectrum_error(in, sp, Nx, Ny, Nz, s->nxh, three_d);
        }
        if (s->direct_out) {
            OCN_CHECK_HIP(hipMalloc((void **)&tmp, np * sizeof(double)));
            OCN_CHECK_HIP(hipMemset(tmp, 0, np * sizeof(double)));
            st = s->bwd.exec(s->spec, tmp + Lp.o, nullptr);
        } else {
            st = s->bwd.exec(s->spec, s->rhs, nullptr);
        }
        if (st != OCN_SUCCESS) {
            if (tmp) (void)hipFree(tmp);
            return st;
        }
        OCN_CHECK_HIP(hipDeviceSynchronize());
        if (s->direct_out) {
            outv.resize(np);
            OCN_CHECK_HIP(hipMemcpy(outv.data(), tmp, np * sizeof(double), hipMemcpyDeviceToHost));
            (void)hipFree(tmp);
            for (int k = 0; k < Nz; ++k)
                for (int j = 0; j < Ny; ++j)
                    for (int i = 0; i < Nx; ++i)
                        err = std::fmax(err, std::fabs(outv[Lp.o + i + Lp.s2 * j + Lp.s3 * k] - net * in[i + (size_t)Nx * (j + (size_t)Ny * k)]));
        } else {
            outv.resize(n);
            OCN_CHECK_HIP(hipMemcpy(outv.data(), s->rhs, n * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t q = 0; q < n; ++q) err = std::fmax(err, std::fabs(outv[q] - net * in[q]));
        }
        {   // plane waves: closed-form answer for every entry, any grid size
            const std::vector<Wave> W = test_waves(Nx, Ny, Nz, three_d);
            fill_waves(in, W, Nx, Ny, Nz, three_d);
            OCN_CHECK_HIP(hipMemcpy(s->rhs, in.data(), n * sizeof(double), hipMemcpyHostToDevice));
            st = s->fwd.exec(s->rhs, s->spec, nullptr);
            if (st != OCN_SUCCESS) return st;
            OCN_CHECK_HIP(hipDeviceSynchronize());
            std::vector<double> sp((size_t)s->nxh * Ny * Nz * 2);
            OCN_CHECK_HIP(hipMemcpy(sp.data(), s->spec, sp.size() * sizeof(double), hipMemcpyDeviceToHost));
            ferr = std::fmax(ferr, wave_spectrum_error(sp, W, Nx, Ny, Nz, s->nxh, three_d));
        }
        OCN_CHECK_HIP(hipMemset(s->rhs, 0, n * sizeof(double)));
        OCN_CHECK_HIP(hipMemset(s->spec, 0, (size_t)s->nxh * Ny * Nz * 2 * sizeof(double)));
    }
    // (amplitudes of the test waves sum to < 40: the same absolute tolerance scale as the unit random field, times the wave amplitudes)
    if (!(err <= 1e-9 * net) || !(ferr <= 4e-8 * count * (three_d ? 1 : Nz))) {
        ocn::set_error("rocFFT %s plan pair for %dx%dx%d failed its self test (round trip: max error %.3e; forward spectrum vs direct DFT: %.3e)",
                       s->c2c ? "complex" : "real", Nx, Ny, Nz, err, ferr);
        return OCN_ERR_ROCFFT;
    }
    return OCN_SUCCESS;
}

extern "C" int ocn_poisson_create(ocn_poisson_t *out, const ocn_grid *grid) { return poisson_create_impl(out, grid, false); }

extern "C" int ocn_poisson_destroy(ocn_poisson_t s)
{
    if (!s) return OCN_SUCCESS;
    free_all(s);
    delete s;
    return OCN_SUCCESS;
}

extern "C" int ocn_poisson_info(ocn_poisson_t s, int32_t *kind, int32_t *r2c, int32_t *direct_out)
{
    OCN_REQUIRE(s, "ocn_poisson_info: null solver");
    if (kind) *kind = s->gtri ? 3 : s->kind;  // 3: Fourier-tridiagonal on a grid with a Bounded / Flat x or y
    if (r2c) *r2c = !s->c2c;
    if (direct_out) *direct_out = (!s->c2c && s->direct_out) + 2 * (s->fused_z ? 1 : 0) + 4 * (s->custom_xy ? 1 : 0) + 8 * (s->dct_z ? 1 : 0);
    return OCN_SUCCESS;
}

extern "C" int ocn_poisson_compute_source_term(ocn_poisson_t s, const double *u, const double *v, const double *w, double dt,
                                               void *stream)
{
    OCN_REQUIRE(s && u && v && w, "ocn_poisson_compute_source_term: null argument");
    const ocn_grid *g = &s->grid;
    int st;
    if (s->kind == 2) {
        // (divᶜᶜᶜ reads u, v, w through their own parent layouts and treats Flat directions: any topology)
        // the gather pass of the first cosine transform (the first Bounded dimension, transformed first: plan_transforms.jl:53-57) is
        // folded into this store
        int first = -1;
        if (s->fft_dct && general_fuse_shuffles())
            for (int d = s->gtri ? 1 : 2; d >= (s->gallreal ? 1 : 0); --d)  // (all-real boxes transform y first: their x lines pair rows)
                if ((d == 0 ? g->tx : d == 1 ? g->ty : g->tz) == OCN_BOUNDED) first = d;
        if (first >= 0 && s->gdct[first]) first = -1;  // (the fused cosine transform gathers on its own load)
        // (the tridiagonal flavour's right-hand side carries Δzᶜ: _fourier_tridiagonal_source_term!, solve_for_pressure.jl:33-38)
        // (packed: the source stays a REAL array -- modes 3 / 4 -- whose x-adjacent pairs the cosine transforms read as complex numbers)
        const int mode = (s->gpacked || s->gallreal) ? (s->gtri ? 4 : 3) : (s->gtri ? 2 : 1);
        st = ocn::launch_source_term(g, u, v, w, dt, mode, s->spec, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream), first);
        s->gathered = (st == OCN_SUCCESS) && first >= 0;
        s->source_set = (st == OCN_SUCCESS);
        return st;
    }
    if (s->custom_xy) {  // K8 fused with the forward x transform: the divergence goes straight into the half spectrum
        st = ocn::launch_rowfft(g, 0, u, v, w, nullptr, dt, s->spec, nullptr, s->twMx, s->twNx, 1.0, ocn::as_stream(stream),
                                (s->custom_tri && !s->dct_z) ? 1 : 0);  // (Δzᶜ rides on the tridiagonal system only)
        s->source_in_rhs = false;
        s->source_set = (st == OCN_SUCCESS);
        return st;
    }
    if (s->c2c)
        st = ocn::launch_source_term(g, u, v, w, dt, s->kind == 1 ? 2 : 1, s->spec, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
    else
        st = ocn::launch_source_term(g, u, v, w, dt, s->kind == 1 ? 4 : 3, s->rhs, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
    s->source_set = (st == OCN_SUCCESS);
    return st;
}

extern "C" int ocn_poisson_set_source_term(ocn_poisson_t s, const double *R, void *stream)
{
    OCN_REQUIRE(s && R, "ocn_poisson_set_source_term: null argument");
    const ocn_grid *g = &s->grid;
    // set_source_term! multiplies by Δzᶜ for the Fourier-tridiagonal solver (fourier_tridiagonal_poisson_solver.jl:155-177)
    const bool tri = (s->kind == 1 && !s->dct_z) || s->gtri;
    const double *dzc = tri ? g->dzc : nullptr;
    int st;
    if (tri && !g->dzc) {
        // regular Bounded z: Δz is a scalar; scale on the fly through a constant array is not needed -- use a tiny device array
        std::vector<double> h(g->Nz + 2 * g->Hz, g->dz);
        double *d = nullptr;
        OCN_CHECK_HIP(hipMalloc((void **)&d, h.size() * sizeof(double)));
        OCN_CHECK_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, ocn::as_stream(stream)));
        st = ocn::launch_set_source(g->Nx, g->Ny, g->Nz, R, d, g->Hz, s->c2c ? s->spec : s->rhs, s->c2c && !s->gpacked && !s->gallreal, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
        OCN_CHECK_HIP(hipStreamSynchronize(ocn::as_stream(stream)));
        OCN_CHECK_HIP(hipFree(d));
    } else {
        st = ocn::launch_set_source(g->Nx, g->Ny, g->Nz, R, dzc, g->Hz, s->c2c ? s->spec : s->rhs, s->c2c && !s->gpacked && !s->gallreal, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
    }
    s->source_set = (st == OCN_SUCCESS);
    s->source_in_rhs = s->custom_xy;
    s->gathered = false;  // natural order
    return st;
}

// solve!(ϕ, solver, b, m) with m != 0 (fft_based_poisson_solver.jl:95-125): the screened equation (∇² + m) ϕ = b, no zero-mode gauge
extern "C" int ocn_poisson_solve_shifted(ocn_poisson_t s, double *p, double m, void *stream_)
{
    OCN_REQUIRE(s && p, "ocn_poisson_solve_shifted: null argument");
    OCN_REQUIRE(s->kind != 1 && !s->gtri, "ocn_poisson_solve_shifted: FFT-based solvers only");
    OCN_REQUIRE(!s->custom_xy && !s->fused_z, "ocn_poisson_solve_shifted: not available on the fused transform pipelines (a Flat or small z "
                "uses the plain path; OCN_POISSON_CUSTOM_XY=0 OCN_POISSON_FUSED_Z=0 select it elsewhere)");
    s->shift = m;
    s->shifted = true;
    const int st = ocn_poisson_solve(s, p, stream_);
    s->shift = 0.0;
    s->shifted = false;
    return st;
}

extern "C" int ocn_poisson_solve(ocn_poisson_t s, double *p, void *stream_)
{
    OCN_REQUIRE(s && p, "ocn_poisson_solve: null argument");
    OCN_REQUIRE(s->source_set, "ocn_poisson_solve: source term not set (call ocn_poisson_compute_source_term first)");
    hipStream_t stream = ocn::as_stream(stream_);
    const ocn_grid *g = &s->grid;
    int st;
    if (s->kind == 2) return poisson_solve_general(s, p, stream);
    if (s->custom_xy) {
        const long long plane = (long long)s->nxh * g->Ny;
        if (s->source_in_rhs) {  // set_source_term! path: x transform of the real array
            st = ocn::launch_rowfft(g, 0, nullptr, nullptr, nullptr, s->rhs, 1.0, s->spec, nullptr, s->twMx, s->twNx, 1.0, stream);
            if (st != OCN_SUCCESS) return st;
            s->source_in_rhs = false;
        }
        // FFT_y (stage order out) -> FFT_z + solve + IFFT_z -> IFFT_y -> inverse x transform into the rows of p
        st = ocn::launch_colfft(g->Ny, 0, s->spec, s->nxh, plane, s->nxh, g->Nz, s->twy, nullptr, nullptr, nullptr, 1.0, 1, stream);
        if (st != OCN_SUCCESS) return st;
        if (s->custom_tri && s->dct_z) {  // ... -> REDFT10_z, division, REDFT01_z in one column pass (regular z) -> IFFT_y -> x
            st = ocn::launch_colfft_dct_solve(g->Nz, s->spec, plane, (int)plane, s->tw, s->wdz, s->lx, s->ly_stage, s->lz_bounded,
                                              1.0 / g->Nz, s->nxh, stream);
            if (st != OCN_SUCCESS) return st;
            st = ocn::launch_colfft(g->Ny, 1, s->spec, s->nxh, plane, s->nxh, g->Nz, s->twy, nullptr, nullptr, nullptr, 1.0, 1, stream);
            if (st != OCN_SUCCESS) return st;
            return ocn::launch_rowfft(g, 1, nullptr, nullptr, nullptr, nullptr, 1.0, s->spec, p, s->twMx, s->twNx,
                                      2.0 / ((double)g->Nx * g->Ny), stream);
        }
        if (s->custom_tri) {  // ... -> Thomas sweep in z (diagonal built with the stage-ordered λy) + zero-mean gauge -> IFFT_y -> x
            st = ocn::launch_tridiag_z(s->nxh, g->Ny, g->Nz, s->lower, s->diag, s->lower, s->spec, s->tscr, s->spec2, stream);
            if (st != OCN_SUCCESS) return st;
            st = ocn::launch_remove_mean_mode(plane, g->Nz, s->spec2, stream);  // column (kx, ky position) = (0, 0) is wavenumber (0, 0)
            if (st != OCN_SUCCESS) return st;
            st = ocn::launch_colfft(g->Ny, 1, s->spec2, s->nxh, plane, s->nxh, g->Nz, s->twy, nullptr, nullptr, nullptr, 1.0, 1, stream);
            if (st != OCN_SUCCESS) return st;
            // the packed real inverse returns (Nx/2) x; y contributes Ny
            return ocn::launch_rowfft(g, 1, nullptr, nullptr, nullptr, nullptr, 1.0, s->spec2, p, s->twMx, s->twNx,
                                      2.0 / ((double)g->Nx * g->Ny), stream);
        }
        st = ocn::launch_colfft(g->Nz, 2, s->spec, plane, 0, (int)plane, 1, s->tw, s->lx, s->ly_stage, s->lz_stage,
                                1.0 / ((double)g->Nx * g->Ny * g->Nz), s->nxh, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn::launch_colfft(g->Ny, 1, s->spec, s->nxh, plane, s->nxh, g->Nz, s->twy, nullptr, nullptr, nullptr, 1.0, 1, stream);
        if (st != OCN_SUCCESS) return st;
        // the packed real inverse returns (Nx/2) x, the normalisation above assumed Nx x: scale by 2
        return ocn::launch_rowfft(g, 1, nullptr, nullptr, nullptr, nullptr, 1.0, s->spec, p, s->twMx, s->twNx, 2.0, stream);
    }
    // forward transforms (solve! :104-107)
    if (s->c2c)
        st = s->fwd.exec(s->spec, nullptr, stream);
    else
        st = s->fwd.exec(s->rhs, s->spec, stream);
    if (st != OCN_SUCCESS) return st;
    double *sol = s->spec;
    if (s->kind == 0 && s->fused_z) {
        const long long plane = (long long)s->nxh * g->Ny;
        st = ocn::launch_colfft(g->Nz, 2, s->spec, plane, 0, (int)plane, 1, s->tw, s->lx, s->ly, s->lz_stage,
                                1.0 / ((double)g->Nx * g->Ny * g->Nz), s->nxh, stream);
    } else if (s->kind == 0) {
        st = ocn::launch_spectral_solve(s->nxh, g->Ny, g->Nz, s->lx, s->ly, s->lz, s->spec, 1, 0, 0, stream, s->shift, s->shifted);
    } else {
        st = ocn::launch_tridiag_z(s->nxh, g->Ny, g->Nz, s->lower, s->diag, s->lower, s->spec, s->tscr, s->spec2, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn::launch_remove_mean_mode((long long)s->nxh * g->Ny, g->Nz, s->spec2, stream);
        sol = s->spec2;
    }
    if (st != OCN_SUCCESS) return st;
    // backward transforms (:118-121) and real part into the pressure interior (:123)
    if (s->c2c) {
        st = s->bwd.exec(sol, nullptr, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn::launch_copy_real(g, sol, p, stream, 0);
    } else {
        if (s->direct_out) {
            ocn::GridDev gd = ocn::to_dev(*g);
            ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
            st = s->bwd.exec(sol, p + Lp.o, stream);
        } else {
            st = s->bwd.exec(sol, s->rhs, stream);
            if (st != OCN_SUCCESS) return st;
            st = ocn::launch_copy_real(g, s->rhs, p, stream, /*real_source=*/1);
        }
    }
    return st;
}

extern "C" int ocn_solve_for_pressure(ocn_poisson_t s, double *p, const double *u, const double *v, const double *w, double dt,
                                      void *stream)
{
    int st = ocn_poisson_compute_source_term(s, u, v, w, dt, stream);
    if (st != OCN_SUCCESS) return st;
    return ocn_poisson_solve(s, p, stream);
}

// ===================================================================================================
// Distributed slab-x solver pieces (complex-to-complex, layouts of transposable_field.jl:50-78)
// ===================================================================================================
struct ocn_dist_poisson {
    ocn_grid grid{};  // local grid
    int rank = 0, R = 1;
    int nx = 0, Nxg = 0;
    bool r2c = true;   // real-to-complex over (y, z): half the all-to-all volume of the reference's C2C
    int nyt = 0;       // y extent of the transposed (complex) data: Ny (c2c) or Ny/2+1 padded to a multiple of R (r2c)
    int ny = 0;        // nyt / R: y extent of the x-local field
    double *lx = nullptr, *ly = nullptr, *lz = nullptr;
    double *rhs = nullptr;  // real source term (r2c)
    double *yfield = nullptr, *xfield = nullptr, *send = nullptr, *recv = nullptr;
    Plan fyz, byz, fx, bx;
    // z Bounded: DistributedFourierTridiagonalPoissonSolver (distributed_fft_tridiagonal_solver.jl:149-351), ZStretched flavour.
    // FFT in y on the slab, transpose, FFT in x, Thomas sweep in z (z is local in the x-local layout as well, so the
    // reference's extra transpose pair to a z-local pencil is not needed), inverse x, transpose back, inverse y.
    bool tri = false;
    bool ycol = false;             // y transforms by the column-FFT kernel (stage-ordered ky) instead of rocFFT
    bool ybounded = false;         // Bounded y: REDFT10 / REDFT01 from the complex FFT (gather, FFT, twiddle: dct_shuffle_kernel), scratch = send
    double *ytw = nullptr;         // e^{-i pi k / 2 Ny} by stored position
    int *ypartner = nullptr;       // stored position of wavenumber Ny - k (stage order only)
    bool xbounded = false;         // the partitioned x is Bounded: cosine transforms along x of the x-local field (natural order, rocFFT lines)
    double *xtw = nullptr;         // e^{-i pi k / 2 Nxg}
    double *tw_y = nullptr;        // column-FFT twiddles
    double *xsol = nullptr;        // tridiagonal solution (x-local layout)
    double *diag = nullptr, *lower = nullptr, *tscr = nullptr;
    double *dzc = nullptr, *dzf = nullptr;
    // "fast" slab pipeline (colfft.hip; periodic z): real y transform -> z column FFT writing the all-to-all layout -> exchange ->
    // fused FFT_x / divide / IFFT_x column kernel -> exchange -> inverse z -> inverse real y into p.  No rocFFT plans, no pack /
    // unpack passes; yfield aliases recv (the half spectrum A1 lives there between the exchanges' uses of it).
    bool fast = false;
    double *tw_h = nullptr, *tw_z = nullptr, *tw_x = nullptr;  // W_{Ny/2}, W_Nz, W_Nxg  (tw_y = W_Ny)
    // the source term is evaluated inside the real y transform: ocn_dist_poisson_source_term only records its arguments
    const double *src_u = nullptr, *src_v = nullptr, *src_w = nullptr;
    double src_dt = 1.0;
    // transpose-free flavour of the fast pipeline (xtri.hip): y and z transforms in place in `recv`, the x direction as a cyclic
    // tridiagonal solve whose ranks exchange two numbers per mode (one all-gather of gsend into grecv) instead of the spectrum
    bool xtri = false;
    double *gsend = nullptr, *grecv = nullptr;
    size_t gchunk = 0;  // doubles per rank in grecv
};

static void free_all(ocn_dist_poisson *s)
{
    s->fyz.destroy(); s->byz.destroy(); s->fx.destroy(); s->bx.destroy();
    if (s->fast) s->yfield = nullptr;  // alias of recv
    double **ptrs[] = {&s->lx, &s->ly, &s->lz, &s->rhs, &s->yfield, &s->xfield, &s->send, &s->recv,
                       &s->tw_y, &s->xsol, &s->diag, &s->lower, &s->tscr, &s->dzc, &s->dzf, &s->tw_h, &s->tw_z, &s->tw_x, &s->gsend, &s->grecv,
                       &s->ytw, &s->xtw};
    for (auto p : ptrs)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    if (s->ypartner) {
        (void)hipFree(s->ypartner);
        s->ypartner = nullptr;
    }
}

static int dist_create_impl(ocn_dist_poisson_t *out, const ocn_grid *lg, int32_t rank, int32_t R, double global_Lx, bool force_c2c,
                            int global_tx = OCN_PERIODIC);

extern "C" int ocn_dist_poisson_create(ocn_dist_poisson_t *out, const ocn_grid *lg, int32_t rank, int32_t R, double global_Lx)
{
    return dist_create_impl(out, lg, rank, R, global_Lx, false);
}

// ... of a grid whose partitioned x is Bounded (global_tx = OCN_BOUNDED; the local grids are RightConnected / FullyConnected /
// LeftConnected slabs, so the global topology has to be said): (Bounded, Bounded, Bounded) only (distributed_fft_based_poisson_solver.jl:62-66)
extern "C" int ocn_dist_poisson_create_global(ocn_dist_poisson_t *out, const ocn_grid *lg, int32_t rank, int32_t R, double global_Lx,
                                              int32_t global_tx)
{
    OCN_REQUIRE(global_tx == OCN_PERIODIC || global_tx == OCN_BOUNDED, "ocn_dist_poisson_create_global: global_tx must be Periodic or Bounded");
    return dist_create_impl(out, lg, rank, R, global_Lx, false, global_tx);
}

// real (y, z) plan pair of the slab: forward + inverse must reproduce a pseudo-random field (see poisson_plans_self_test)
static int dist_real_plans_self_test(ocn_dist_poisson *s)
{
    const ocn_grid *g = &s->grid;
    const int nx = g->Nx, Ny = g->Ny, Nz = g->Nz;
    const size_t n = (size_t)nx * Ny * Nz;
    std::vector<double> in(n);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (size_t q = 0; q < n; ++q) {
        x += 0x9E3779B97F4A7C15ull;
        unsigned long long z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        in[q] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    ocn::GridDev gd = ocn::to_dev(*g);
    ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
    const size_t np = (size_t)Lp.sx * Lp.sy * Lp.sz;
    double *tmp = nullptr;
    OCN_CHECK_HIP(hipMalloc((void **)&tmp, np * sizeof(double)));
    OCN_CHECK_HIP(hipMemset(tmp, 0, np * sizeof(double)));
    OCN_CHECK_HIP(hipMemcpy(s->rhs, in.data(), n * sizeof(double), hipMemcpyHostToDevice));
    int st = s->fyz.exec(s->rhs, s->yfield, nullptr);
    double ferr = 0.0;
    if (st == OCN_SUCCESS) {  // known-answer check of the forward (y, z) transform (samples; the serial handles check every entry): column xl, entry (ky, kz)
        if (hipDeviceSynchronize() != hipSuccess) st = OCN_ERR_ROCFFT;
        std::vector<double> sp((size_t)nx * s->nyt * Nz * 2);
        if (st == OCN_SUCCESS && hipMemcpy(sp.data(), s->yfield, sp.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = OCN_ERR_ROCFFT;
        const double two_pi = 6.283185307179586476925286766559;
        const int xs[3] = {0, nx / 2, nx - 1}, kys[3] = {1 % (Ny / 2 + 1), 0, Ny / 2}, kzs[3] = {Nz - 1, 1 % Nz, 0};
        for (int q = 0; q < 3 && st == OCN_SUCCESS; ++q) {
            double re = 0.0, im = 0.0;
            for (int k = 0; k < Nz; ++k)
                for (int j = 0; j < Ny; ++j) {
                    const double ph = -two_pi * ((double)kys[q] * j / Ny + (double)kzs[q] * k / Nz);
                    const double v = in[xs[q] + (size_t)nx * (j + (size_t)Ny * k)];
                    re += v * std::cos(ph);
                    im += v * std::sin(ph);
                }
            const size_t o = 2 * ((size_t)xs[q] + (size_t)nx * (kys[q] + (size_t)s->nyt * kzs[q]));
            ferr = std::fmax(ferr, std::fmax(std::fabs(sp[o] - re), std::fabs(sp[o + 1] - im)));
        }
    }
    if (st == OCN_SUCCESS) st = s->byz.exec(s->yfield, tmp + Lp.o, nullptr);
    if (st != OCN_SUCCESS) {
        (void)hipFree(tmp);
        return st;
    }
    OCN_CHECK_HIP(hipDeviceSynchronize());
    std::vector<double> outv(np);
    OCN_CHECK_HIP(hipMemcpy(outv.data(), tmp, np * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(tmp);
    double err = 0.0;
    for (int k = 0; k < Nz; ++k)
        for (int j = 0; j < Ny; ++j)
            for (int i = 0; i < nx; ++i)
                err = std::fmax(err, std::fabs(outv[Lp.o + i + Lp.s2 * j + Lp.s3 * k] - in[i + (size_t)nx * (j + (size_t)Ny * k)]));
    OCN_CHECK_HIP(hipMemset(s->rhs, 0, n * sizeof(double)));
    OCN_CHECK_HIP(hipMemset(s->yfield, 0, (size_t)nx * s->nyt * Nz * 2 * sizeof(double)));
    if (!(err <= 1e-9) || !(ferr <= 1e-9 * Ny * Nz)) {
        ocn::set_error("rocFFT real (y, z) plan pair for the %dx%dx%d slab failed its self test (round trip: max error %.3e; forward spectrum vs direct DFT: %.3e)",
                       nx, Ny, Nz, err, ferr);
        return OCN_ERR_ROCFFT;
    }
    return OCN_SUCCESS;
}

static int dist_create_impl(ocn_dist_poisson_t *out, const ocn_grid *lg, int32_t rank, int32_t R, double global_Lx, bool force_c2c, int global_tx)
{
    OCN_REQUIRE(out && lg, "ocn_dist_poisson_create: null argument");
    const bool xbounded = global_tx == OCN_BOUNDED;
    {   // the slab's x topology must be the one its rank has in that global grid
        const int want = !xbounded ? (lg->tx == OCN_PERIODIC ? OCN_PERIODIC : OCN_FULLY_CONNECTED)
                                   : (R == 1 ? OCN_BOUNDED : rank == 0 ? OCN_RIGHT_CONNECTED : rank == R - 1 ? OCN_LEFT_CONNECTED : OCN_FULLY_CONNECTED);
        OCN_REQUIRE(lg->tx == want, "ocn_dist_poisson_create: rank %d of %d of a %s x must hold a slab of x topology %d, got %d", rank, R,
                    xbounded ? "Bounded" : "Periodic", want, lg->tx);
        OCN_REQUIRE(!xbounded || (lg->ty == OCN_BOUNDED && lg->tz == OCN_BOUNDED),
                    "ocn_dist_poisson_create: a Bounded x needs Bounded y and z (distributed_fft_based_poisson_solver.jl:62-66)");
    }
    int st = ocn::validate_grid_any(lg);
    if (st != OCN_SUCCESS) return st;
    OCN_REQUIRE(R >= 1 && rank >= 0 && rank < R, "ocn_dist_poisson_create: bad rank %d of %d", rank, R);
    OCN_REQUIRE(global_Lx > 0, "ocn_dist_poisson_create: global_Lx must be positive");
    // (distributed_fft_based_poisson_solver.jl:62-66: a Periodic z needs a Periodic y)
    OCN_REQUIRE((lg->ty == OCN_PERIODIC && ((lg->tz == OCN_PERIODIC && lg->dzc == nullptr) || lg->tz == OCN_BOUNDED)) ||
                    (lg->ty == OCN_BOUNDED && lg->tz == OCN_BOUNDED),
                "ocn_dist_poisson_create: supports (x-partitioned, Periodic, Periodic) regular grids, (x-partitioned, Periodic, Bounded) and "
                "(x-partitioned, Bounded, Bounded)");
    // validate_poisson_solver_distributed_grid (distributed_fft_based_poisson_solver.jl:211-229)
    OCN_REQUIRE(lg->Ny % R == 0, "ocn_dist_poisson_create: Ny = %d must be divisible by the number of ranks %d", lg->Ny, R);
    ensure_rocfft();
    ocn_dist_poisson *s = new ocn_dist_poisson();
    s->grid = *lg;
    s->rank = rank; s->R = R;
    s->nx = lg->Nx; s->Nxg = lg->Nx * R;
    const int nx = s->nx, Ny = lg->Ny, Nz = lg->Nz, Nxg = s->Nxg;
    const char *env = std::getenv("OCN_POISSON_C2C");
    s->r2c = !force_c2c && !(env && env[0] == '1');
#define TRY(expr) do { int _st = (expr); if (_st != OCN_SUCCESS) { free_all(s); delete s; return _st; } } while (0)
#define TRY_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ocn::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); free_all(s); delete s; return OCN_ERR_ALLOC; } } while (0)
    const char *efast = std::getenv("OCN_DIST_POISSON_FAST");
    if (lg->tz == OCN_BOUNDED && lg->ty == OCN_PERIODIC && !xbounded && !(efast && efast[0] == '0') && R >= 1 && Nz > 1 && ocn::realfft_y_supported(Ny) &&
        ocn::colfft_supported(Nxg)) {
        // Slab pipeline, tridiagonal flavour: real y transform (source term x Δzᶜ evaluated on load) writing the all-to-all layout
        // [d][ky_l + c (z + Nz xl)] (ky = d c + ky_l, c = ceil((Ny/2+1) / R), padded entries stay 0) -> exchange -> x is a strided
        // column: FFT_x -> Thomas sweep along z (stride c) with this rank's ky range -> IFFT_x -> exchange -> inverse real y into p.
        s->tri = true;
        s->fast = true;
        s->r2c = true;
        const int NyH = Ny / 2 + 1, c = (NyH + R - 1) / R, Hz = lg->Hz;
        s->ny = c;
        s->nyt = c * R;
        const size_t n = (size_t)s->nyt * Nz * nx;
        if (lg->dzc) {
            const size_t nf = (size_t)Nz + 2 * Hz;
            TRY_HIP(hipMalloc((void **)&s->dzc, nf * sizeof(double)));
            TRY_HIP(hipMalloc((void **)&s->dzf, nf * sizeof(double)));
            TRY_HIP(hipMemcpy(s->dzc, lg->dzc, nf * sizeof(double), hipMemcpyDeviceToDevice));
            TRY_HIP(hipMemcpy(s->dzf, lg->dzf, nf * sizeof(double), hipMemcpyDeviceToDevice));
            s->grid.dzc = s->dzc;
            s->grid.dzf = s->dzf;
        }
        for (double **p : {&s->send, &s->recv}) {
            TRY_HIP(hipMalloc((void **)p, n * 2 * sizeof(double)));
            TRY_HIP(hipMemset(*p, 0, n * 2 * sizeof(double)));
        }
        s->yfield = s->recv;
        TRY_HIP(hipMalloc((void **)&s->diag, n * sizeof(double)));
        TRY_HIP(hipMalloc((void **)&s->tscr, n * sizeof(double)));
        TRY(upload(ocn::colfft_twiddles(Ny / 2), &s->tw_h));
        TRY(upload(ocn::colfft_twiddles(Ny), &s->tw_y));
        TRY(upload(ocn::colfft_twiddles(Nxg), &s->tw_x));
        std::vector<double> lyn = eigenvalues(Ny, lg->Ly, OCN_PERIODIC), lxn = eigenvalues(Nxg, global_Lx, OCN_PERIODIC), lxs(Nxg);
        lyn.resize(NyH);
        lyn.resize(s->nyt, 1.0);  // padded ky: their data is 0, any nonzero eigenvalue keeps the sweep finite
        for (int q = 0; q < Nxg; ++q) lxs[q] = lxn[ocn::colfft_wavenumber(Nxg, q)];
        TRY(upload(lyn, &s->ly));
        TRY(upload(lxs, &s->lx));
        std::vector<double> hf(Nz + 2 * Hz, lg->dz);
        if (lg->dzf) TRY_HIP(hipMemcpy(hf.data(), lg->dzf, hf.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> low(Nz - 1, 0.0);
        for (int q = 2; q <= Nz; ++q) low[q - 2] = 1 / hf[q + Hz - 1];
        TRY(upload(low, &s->lower));
        // main diagonal in the exchanged layout: column (ky_l, kx position) at ky_l + c Nz kx, planes c apart
        TRY(ocn::launch_main_diagonal_strided(&s->grid, c, Nxg, (long long)c * Nz, c, s->ly + (size_t)rank * c, s->lx, s->diag, nullptr));
        TRY_HIP(hipDeviceSynchronize());
        *out = s;
        return OCN_SUCCESS;
    }
    if (lg->tz == OCN_BOUNDED) {
        s->tri = true;
        s->r2c = false;
        s->nyt = Ny;
        s->ny = Ny / R;
        const int ny = s->ny, Hz = lg->Hz;
        const size_t n = (size_t)nx * Ny * Nz;
        if (lg->dzc) {  // private copies of the z spacings, like the single-GPU solver
            const size_t nf = (size_t)Nz + 2 * Hz;
            TRY_HIP(hipMalloc((void **)&s->dzc, nf * sizeof(double)));
            TRY_HIP(hipMalloc((void **)&s->dzf, nf * sizeof(double)));
            TRY_HIP(hipMemcpy(s->dzc, lg->dzc, nf * sizeof(double), hipMemcpyDeviceToDevice));
            TRY_HIP(hipMemcpy(s->dzf, lg->dzf, nf * sizeof(double), hipMemcpyDeviceToDevice));
            s->grid.dzc = s->dzc;
            s->grid.dzf = s->dzf;
        }
        for (double **p : {&s->yfield, &s->xfield, &s->send, &s->recv, &s->xsol}) {
            TRY_HIP(hipMalloc((void **)p, n * 2 * sizeof(double)));
            TRY_HIP(hipMemset(*p, 0, n * 2 * sizeof(double)));
        }
        TRY_HIP(hipMalloc((void **)&s->diag, n * sizeof(double)));
        TRY_HIP(hipMalloc((void **)&s->tscr, n * sizeof(double)));
        TRY(upload(eigenvalues(Nxg, global_Lx, xbounded ? OCN_BOUNDED : OCN_PERIODIC), &s->lx));
        s->xbounded = xbounded;
        if (xbounded) {
            std::vector<double> w(2 * (size_t)Nxg);
            for (int q = 0; q < Nxg; ++q) {
                const long double a = 3.14159265358979323846264338327950288L * q / (2.0L * Nxg);
                w[2 * q] = (double)cosl(a);
                w[2 * q + 1] = (double)(-sinl(a));
            }
            TRY(upload(w, &s->xtw));
        }
        // y transforms: the column-FFT kernel leaves ky in stage order, so the eigenvalue of a stored position is permuted
        s->ycol = ocn::colfft_supported(Ny);
        s->ybounded = lg->ty == OCN_BOUNDED;
        std::vector<double> lyn = eigenvalues(Ny, lg->Ly, lg->ty), lys(Ny);
        for (int q = 0; q < Ny; ++q) lys[q] = s->ycol ? lyn[ocn::colfft_wavenumber(Ny, q)] : lyn[q];
        TRY(upload(lys, &s->ly));
        if (s->ybounded) {  // the twiddles of the cosine transforms by stored position, and where wavenumber Ny - k is stored
            std::vector<int> kofp(Ny), pofk(Ny), partner(Ny);
            for (int q = 0; q < Ny; ++q) {
                kofp[q] = s->ycol ? ocn::colfft_wavenumber(Ny, q) : q;
                pofk[kofp[q]] = q;
            }
            std::vector<double> w(2 * (size_t)Ny);
            for (int q = 0; q < Ny; ++q) {
                const long double a = 3.14159265358979323846264338327950288L * kofp[q] / (2.0L * Ny);
                w[2 * q] = (double)cosl(a);
                w[2 * q + 1] = (double)(-sinl(a));
                partner[q] = pofk[(Ny - kofp[q]) % Ny];
            }
            TRY(upload(w, &s->ytw));
            TRY_HIP(hipMalloc((void **)&s->ypartner, Ny * sizeof(int)));
            TRY_HIP(hipMemcpy(s->ypartner, partner.data(), Ny * sizeof(int), hipMemcpyHostToDevice));
        }
        if (s->ycol) {
            TRY(upload(ocn::colfft_twiddles(Ny), &s->tw_y));
        } else {  // rocFFT, one z-plane per execution: (nx columns at distance 1) x (Ny points at stride nx)
            const size_t len[1] = {(size_t)Ny};
            const size_t str[1] = {(size_t)nx};
            TRY(make_plan(s->fyz, rocfft_placement_inplace, rocfft_transform_type_complex_forward, 1, len, nx,
                          rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, 1, str, 1, 1.0));
            TRY(make_plan(s->byz, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, 1, len, nx,
                          rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, 1, str, 1, 1.0));
        }
        {   // x transforms of the x-local field (Nxg, ny, Nz); the inverse reads the tridiagonal solution and carries 1/(Nxg Ny)
            const size_t len[1] = {(size_t)Nxg};
            const size_t str[1] = {1};
            TRY(make_plan(s->fx, rocfft_placement_inplace, rocfft_transform_type_complex_forward, 1, len, (size_t)ny * Nz,
                          rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, Nxg, str, Nxg, 1.0));
            TRY(make_plan(s->bx, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse, 1, len, (size_t)ny * Nz,
                          rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, Nxg, str, Nxg,
                          1.0 / ((double)Nxg * Ny)));
        }
        // lower = upper = 1/Δzᶠ[q], q = 2..Nz; main diagonal with this rank's ky range (fourier_tridiagonal_poisson_solver.jl:41-51, 97-99)
        std::vector<double> hf(Nz + 2 * Hz, lg->dz);
        if (lg->dzf) TRY_HIP(hipMemcpy(hf.data(), lg->dzf, hf.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> low(Nz > 1 ? Nz - 1 : 1, 0.0);
        for (int q = 2; q <= Nz; ++q) low[q - 2] = 1 / hf[q + Hz - 1];
        TRY(upload(low, &s->lower));
        ocn_grid xg = s->grid;  // the x-local layout as a "grid": only Ny, Nz and the z spacings are read
        xg.Nx = Nxg;
        xg.Ny = ny;
        TRY(ocn::launch_main_diagonal(&xg, Nxg, s->lx, s->ly + (size_t)rank * ny, s->diag, nullptr));
        TRY_HIP(hipDeviceSynchronize());
        *out = s;
        return OCN_SUCCESS;
    }
    {
        const char *ef = std::getenv("OCN_DIST_POISSON_FAST");
        const bool local_ok = !force_c2c && s->r2c && !(ef && ef[0] == '0') && R >= 1 && ocn::realfft_y_supported(Ny) && ocn::colfft_supported(Nz);
        // the transpose-free x solve: default for R > 1 (one rank has nothing to exchange and the fused FFT_x kernel moves fewer
        // bytes); OCN_DIST_POISSON_XTRI=1 forces it, =0 keeps the all-to-all pipeline
        const char *ex = std::getenv("OCN_DIST_POISSON_XTRI");
        const bool want = ex ? ex[0] == '1' : R > 1;
        s->xtri = local_ok && want && ocn::xtri_supported(R, Nxg);
        s->fast = s->xtri || (local_ok && ocn::colfft_supported(Nxg) && Nz % R == 0);
    }
    if (s->fast) {
        const int NyH = Ny / 2 + 1;
        s->nyt = NyH;
        s->ny = 0;  // the transposed layout is partitioned in (stored) kz, not in ky
        const size_t n = (size_t)NyH * nx * Nz;
        TRY_HIP(hipMalloc((void **)&s->rhs, (size_t)nx * Ny * Nz * sizeof(double)));
        TRY_HIP(hipMemset(s->rhs, 0, (size_t)nx * Ny * Nz * sizeof(double)));
        for (double **p : {&s->send, &s->recv}) {
            if (s->xtri && p == &s->send) continue;  // no exchange layout: the spectrum stays in `recv`
            TRY_HIP(hipMalloc((void **)p, n * 2 * sizeof(double)));
            TRY_HIP(hipMemset(*p, 0, n * 2 * sizeof(double)));
        }
        if (s->xtri) {
            s->gchunk = 2 * (2 * (size_t)NyH * Nz + nx);
            TRY_HIP(hipMalloc((void **)&s->gsend, s->gchunk * sizeof(double)));
            TRY_HIP(hipMalloc((void **)&s->grecv, s->gchunk * R * sizeof(double)));
            TRY_HIP(hipMemset(s->gsend, 0, s->gchunk * sizeof(double)));
            TRY_HIP(hipMemset(s->grecv, 0, s->gchunk * R * sizeof(double)));
        }
        s->yfield = s->recv;
        TRY(upload(ocn::colfft_twiddles(Ny / 2), &s->tw_h));
        TRY(upload(ocn::colfft_twiddles(Ny), &s->tw_y));
        TRY(upload(ocn::colfft_twiddles(Nz), &s->tw_z));
        if (!s->xtri) TRY(upload(ocn::colfft_twiddles(Nxg), &s->tw_x));
        // eigenvalues: ky natural (0..Ny/2); kz and kx by STORED position of the column kernels' stage order
        std::vector<double> lyn = eigenvalues(Ny, lg->Ly, OCN_PERIODIC), lzn = eigenvalues(Nz, lg->Lz, OCN_PERIODIC),
                            lxn = eigenvalues(Nxg, global_Lx, OCN_PERIODIC);
        lyn.resize(NyH);
        std::vector<double> lzs(Nz), lxs(Nxg);
        for (int q = 0; q < Nz; ++q) lzs[q] = lzn[ocn::colfft_wavenumber(Nz, q)];
        for (int q = 0; q < Nxg && !s->xtri; ++q) lxs[q] = lxn[ocn::colfft_wavenumber(Nxg, q)];
        TRY(upload(lyn, &s->ly));
        TRY(upload(lzs, &s->lz));
        TRY(upload(lxs, &s->lx));
        TRY_HIP(hipDeviceSynchronize());
        *out = s;
        return OCN_SUCCESS;
    }
    ocn::GridDev gd = ocn::to_dev(*lg);
    ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int nyh = Ny / 2 + 1;
        s->nyt = s->r2c ? ((nyh + R - 1) / R) * R : Ny;
        s->ny = s->nyt / R;
        const size_t len[2] = {(size_t)Ny, (size_t)Nz};
        const size_t cstr[2] = {(size_t)nx, (size_t)nx * s->nyt};
        int pst;
        if (s->r2c) {  // real (nx,Ny,Nz) -> Hermitian (nx,nyt,Nz), batched over the nx local columns (distance 1)
            const size_t rstr[2] = {(size_t)nx, (size_t)nx * Ny};
            const size_t pstr[2] = {(size_t)Lp.s2, (size_t)Lp.s3};
            pst = make_plan(s->fyz, rocfft_placement_notinplace, rocfft_transform_type_real_forward, 2, len, nx, rocfft_array_type_real,
                            rocfft_array_type_hermitian_interleaved, rstr, 1, cstr, 1, 1.0);
            if (pst == OCN_SUCCESS)  // inverse writes straight into the local pressure interior
                pst = make_plan(s->byz, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, 2, len, nx,
                                rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, cstr, 1, pstr, 1, 1.0 / ((double)Ny * Nz));
            if (pst != OCN_SUCCESS) {  // rocFFT has no kernel for this strided real layout: complex path
                s->fyz.destroy(); s->byz.destroy();
                s->r2c = false;
                continue;
            }
        } else {  // FFT over (y, z) of the y-local complex field, batched over the nx local columns (no permutedims)
            TRY(make_plan(s->fyz, rocfft_placement_inplace, rocfft_transform_type_complex_forward, 2, len, nx,
                          rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, cstr, 1, cstr, 1, 1.0));
            TRY(make_plan(s->byz, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, 2, len, nx,
                          rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, cstr, 1, cstr, 1,
                          1.0 / ((double)Ny * Nz)));
        }
        break;
    }
    const int ny = s->ny;
    const size_t n = (size_t)nx * s->nyt * Nz;  // complex elements of either layout
    // global eigenvalues (distributed_fft_based_poisson_solver.jl:104-106); padded ky (r2c) get 1 (their data is 0)
    TRY(upload(eigenvalues(Nxg, global_Lx, OCN_PERIODIC), &s->lx));
    {
        std::vector<double> ly = eigenvalues(Ny, lg->Ly, OCN_PERIODIC);
        ly.resize(s->nyt > Ny ? s->nyt : Ny, 1.0);
        for (int q = Ny / 2 + 1; s->r2c && q < s->nyt; ++q) ly[q] = 1.0;
        TRY(upload(ly, &s->ly));
    }
    TRY(upload(eigenvalues(Nz, lg->Lz, OCN_PERIODIC), &s->lz));
    for (double **p : {&s->yfield, &s->xfield, &s->send, &s->recv}) {
        TRY_HIP(hipMalloc((void **)p, n * 2 * sizeof(double)));
        TRY_HIP(hipMemset(*p, 0, n * 2 * sizeof(double)));
    }
    if (s->r2c) {
        TRY_HIP(hipMalloc((void **)&s->rhs, (size_t)nx * Ny * Nz * sizeof(double)));
        TRY_HIP(hipMemset(s->rhs, 0, (size_t)nx * Ny * Nz * sizeof(double)));
    }
    {   // FFT over x of the x-local field (Nxg, ny, Nz)
        const size_t len[1] = {(size_t)Nxg};
        const size_t str[1] = {1};
        TRY(make_plan(s->fx, rocfft_placement_inplace, rocfft_transform_type_complex_forward, 1, len, (size_t)ny * Nz,
                      rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, Nxg, str, Nxg, 1.0));
        TRY(make_plan(s->bx, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, 1, len, (size_t)ny * Nz,
                      rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, str, Nxg, str, Nxg,
                      1.0 / (double)Nxg));
    }
#undef TRY
#undef TRY_HIP
    if (s->r2c && dist_real_plans_self_test(s) != OCN_SUCCESS) {  // see poisson_create_impl: fall back to complex plans
        free_all(s);
        delete s;
        return dist_create_impl(out, lg, rank, R, global_Lx, true, global_tx);
    }
    *out = s;
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_destroy(ocn_dist_poisson_t s)
{
    if (!s) return OCN_SUCCESS;
    free_all(s);
    delete s;
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_buffers(ocn_dist_poisson_t s, double **yfield, double **xfield, double **send, double **recv)
{
    OCN_REQUIRE(s, "ocn_dist_poisson_buffers: null solver");
    if (yfield) *yfield = s->yfield;
    if (xfield) *xfield = s->xfield;
    if (send) *send = s->send;
    if (recv) *recv = s->recv;
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_layout(ocn_dist_poisson_t s, int32_t *ny_transposed, int64_t *complex_elements, int32_t *r2c)
{
    OCN_REQUIRE(s, "ocn_dist_poisson_layout: null solver");
    if (ny_transposed) *ny_transposed = s->nyt;
    if (complex_elements) *complex_elements = (int64_t)s->nx * s->nyt * s->grid.Nz;
    if (r2c) *r2c = s->r2c;
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_gather_buffers(ocn_dist_poisson_t s, double **send, double **recv, int64_t *doubles_per_rank)
{
    OCN_REQUIRE(s, "ocn_dist_poisson_gather_buffers: null solver");
    if (send) *send = s->gsend;
    if (recv) *recv = s->grecv;
    if (doubles_per_rank) *doubles_per_rank = (int64_t)s->gchunk;
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_pipeline(ocn_dist_poisson_t s, int32_t *fast)
{
    OCN_REQUIRE(s && fast, "ocn_dist_poisson_pipeline: null argument");
    *fast = s->fast ? (s->tri ? 2 : (s->xtri ? 3 : 1)) : 0;
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_source_term(ocn_dist_poisson_t s, const double *u, const double *v, const double *w, double dt,
                                            void *stream)
{
    OCN_REQUIRE(s && u && v && w, "ocn_dist_poisson_source_term: null argument");
    const ocn_grid *g = &s->grid;
    if (s->fast) {
        // evaluating the source term inside the real y transform saves a pass (with the XCD-contiguous block order of round 4 also on
        // thin slabs: 4.14 against 4.18 ms per rank-step at nx = 64; OCN_DIST_FUSED_SOURCE=0 keeps the separate pass)
        static const char *env = std::getenv("OCN_DIST_FUSED_SOURCE");
        const bool fused = s->tri || !(env && env[0] == '0');
        s->src_u = fused ? u : nullptr; s->src_v = v; s->src_w = w; s->src_dt = dt;
        if (fused) return OCN_SUCCESS;  // evaluated by forward_yz from the same (unchanged) velocity arrays
    }
    // _fourier_tridiagonal_source_term! (solve_for_pressure.jl:33-38): rhs = Δzᶜ div(U) / Δt, complex
    if (s->tri) return ocn::launch_source_term(g, u, v, w, dt, 2, s->yfield, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
    if (s->r2c) return ocn::launch_source_term(g, u, v, w, dt, 3, s->rhs, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
    return ocn::launch_source_term(g, u, v, w, dt, 1, s->yfield, g->Nx, (long long)g->Nx * g->Ny, ocn::as_stream(stream));
}

// y transform of the slab (nx, Ny, Nz) for the tridiagonal flavour: column-FFT kernel (stage-ordered ky) or rocFFT per plane
static int dist_tri_y_fft(ocn_dist_poisson *s, int inverse, double *a, hipStream_t stream)
{
    const int nx = s->nx, Ny = s->grid.Ny, Nz = s->grid.Nz;
    if (s->ycol)
        return ocn::launch_colfft(Ny, inverse ? 1 : 0, a, nx, (long long)nx * Ny, nx, Nz, s->tw_y, nullptr, nullptr, nullptr, 1.0, 1, stream);
    Plan &P = inverse ? s->byz : s->fyz;
    for (int k = 0; k < Nz; ++k) {
        int st = P.exec(a + 2 * (size_t)nx * Ny * k, nullptr, stream);
        if (st != OCN_SUCCESS) return st;
    }
    return OCN_SUCCESS;
}
// Bounded y: the cosine transforms of the single-process solver (dct_shuffle_kernel: Makhoul's gather / twiddle around the complex FFT),
// out of place through `send`, which is idle while the spectrum is y-local.  The inverse's 1 / Ny rides on the x inverse like the FFT's.
static int dist_tri_y_transform(ocn_dist_poisson *s, int inverse, hipStream_t stream)
{
    if (!s->ybounded) return dist_tri_y_fft(s, inverse, s->yfield, stream);
    const int nx = s->nx, Ny = s->grid.Ny, Nz = s->grid.Nz;
    const long long n = (long long)nx * Ny * Nz;
    auto shuffle = [&](int mode, const double *in, double *out) {
        hipLaunchKernelGGL(dct_shuffle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, nx, Ny, Nz, 1, mode,
                           reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out), reinterpret_cast<const double2 *>(s->ytw),
                           (mode == 1 || mode == 2) ? s->ypartner : nullptr);
    };
    shuffle(inverse ? 2 : 0, s->yfield, s->send);
    int st = dist_tri_y_fft(s, inverse, s->send, stream);
    if (st != OCN_SUCCESS) return st;
    shuffle(inverse ? 3 : 1, s->send, s->yfield);
    OCN_CHECK_HIP(hipGetLastError());
    return OCN_SUCCESS;
}

extern "C" int ocn_dist_poisson_forward_yz(ocn_dist_poisson_t s, void *stream)
{
    OCN_REQUIRE(s, "ocn_dist_poisson_forward_yz: null solver");
    if (s->tri && s->fast) {  // (Δzᶜ div / Δt)(u, v, w) -> half spectrum in the exchange layout
        const ocn_grid *g = &s->grid;
        OCN_REQUIRE(s->src_u, "ocn_dist_poisson_forward_yz: call ocn_dist_poisson_source_term first");
        return ocn::launch_realfft_y(g->Ny, 0, nullptr, s->send, nullptr, 0, 0, s->nx, g->Nz, s->tw_h, s->tw_y, ocn::as_stream(stream), g,
                                     s->src_u, s->src_v, s->src_w, s->src_dt, s->ny, (long long)s->ny * g->Nz * s->nx, 1, 1.0);
    }
    if (s->tri) return dist_tri_y_transform(s, 0, ocn::as_stream(stream));
    if (s->fast) {  // rhs -> A1 (in recv) -> send, ready for the exchange
        const ocn_grid *g = &s->grid;
        int st = s->src_u ? ocn::launch_realfft_y(g->Ny, 0, nullptr, s->recv, nullptr, 0, 0, s->nx, g->Nz, s->tw_h, s->tw_y,
                                                   ocn::as_stream(stream), g, s->src_u, s->src_v, s->src_w, s->src_dt)
                          : ocn::launch_realfft_y(g->Ny, 0, s->rhs, s->recv, nullptr, 0, 0, s->nx, g->Nz, s->tw_h, s->tw_y, ocn::as_stream(stream));
        if (st != OCN_SUCCESS) return st;
        if (s->xtri) {  // z in place (stage order), then the local x solves; gsend is ready for the all-gather
            const long long cols = (long long)s->nyt * s->nx;
            st = ocn::launch_colfft(g->Nz, 0, s->recv, cols, 0, (int)cols, 1, s->tw_z, nullptr, nullptr, nullptr, 1.0, 1, ocn::as_stream(stream));
            if (st != OCN_SUCCESS) return st;
            const double scale = 1.0 / ((double)(g->Ny / 2) * g->Nz);
            return ocn::launch_xtri_sweep(s->recv, s->ly, s->lz, s->nyt, s->nx, g->Nz, g->dx, scale, s->gsend, ocn::as_stream(stream));
        }
        return ocn::launch_colfft_slab_z(g->Nz, 0, s->recv, s->send, s->nx, s->nyt, s->R, s->tw_z, ocn::as_stream(stream));
    }
    if (s->r2c) return s->fyz.exec(s->rhs, s->yfield, ocn::as_stream(stream));
    return s->fyz.exec(s->yfield, nullptr, ocn::as_stream(stream));
}

extern "C" int ocn_dist_poisson_solve_x(ocn_dist_poisson_t s, void *stream_)
{
    OCN_REQUIRE(s, "ocn_dist_poisson_solve_x: null solver");
    hipStream_t stream = ocn::as_stream(stream_);
    if (s->fast && s->tri) {  // recv = [xg S + (ky_l + c z)], S = c Nz
        const int Nz = s->grid.Nz, c = s->ny;
        const long long S = (long long)c * Nz;
        int st = ocn::launch_colfft(s->Nxg, 0, s->recv, S, 0, (int)S, 1, s->tw_x, nullptr, nullptr, nullptr, 1.0, 1, stream);
        if (st != OCN_SUCCESS) return st;
        st = ocn::launch_tridiag_z_strided(c, s->Nxg, S, c, Nz, s->lower, s->diag, s->lower, s->recv, s->tscr, s->send, stream);
        if (st != OCN_SUCCESS) return st;
        if (s->rank == 0) {  // zero-mean gauge on the (kx, ky) = (0, 0) column: stored position 0 of both
            st = ocn::launch_remove_mean_mode(c, Nz, s->send, stream);
            if (st != OCN_SUCCESS) return st;
        }
        return ocn::launch_colfft(s->Nxg, 1, s->send, S, 0, (int)S, 1, s->tw_x, nullptr, nullptr, nullptr, 1.0, 1, stream);
    }
    if (s->xtri)  // grecv holds every rank's interface values: interface systems, spike correction, the mean mode's line
        return ocn::launch_xtri_finish(s->recv, s->ly, s->lz, s->nyt, s->nx, s->grid.Nz, s->grid.dx, s->grecv, s->rank, s->R, stream);
    if (s->fast) {  // recv = [xg S + (ky + NyH pz_l)]: FFT_x -> -b / ((λy + λz) + λx), rank 0 zeroes the mean mode -> IFFT_x, in place
        const int Nz = s->grid.Nz, cz = Nz / s->R, NyH = s->nyt;
        const long long S = (long long)NyH * cz;
        const double scale = 1.0 / ((double)(s->grid.Ny / 2) * Nz * s->Nxg);
        return ocn::launch_colfft(s->Nxg, 2, s->recv, S, 0, (int)S, 1, s->tw_x, s->ly, s->lz + (size_t)s->rank * cz, s->lx, scale, NyH,
                                  stream, s->rank == 0);
    }
    int st;
    // Bounded x: REDFT10 / REDFT01 along the lines of the x-local field (Nxg, ny, Nz) -- gather / twiddle passes of the single-process
    // solver around the same complex line FFTs, out of place between xfield and xsol; the inverse's 1 / Nxg rides on the plan's scale
    auto xshuffle = [&](int mode, const double *in, double *outp) {
        const long long n = (long long)s->Nxg * s->ny * s->grid.Nz;
        hipLaunchKernelGGL(dct_shuffle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, s->Nxg, s->ny, s->grid.Nz, 0, mode,
                           reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(outp), reinterpret_cast<const double2 *>(s->xtw),
                           (const int *)nullptr);
    };
    if (s->xbounded) {
        OCN_REQUIRE(s->tri, "ocn_dist_poisson_solve_x: a Bounded x runs the tridiagonal flavour");
        xshuffle(0, s->xfield, s->xsol);
        st = s->fx.exec(s->xsol, nullptr, stream);
        if (st != OCN_SUCCESS) return st;
        xshuffle(1, s->xsol, s->xfield);
    } else {
        st = s->fx.exec(s->xfield, nullptr, stream);
        if (st != OCN_SUCCESS) return st;
    }
    if (s->tri) {
        const int Nz = s->grid.Nz;
        st = ocn::launch_tridiag_z(s->Nxg, s->ny, Nz, s->lower, s->diag, s->lower, s->xfield, s->tscr, s->xsol, stream);
        if (st != OCN_SUCCESS) return st;
        // zero-mean gauge on the (kx, ky) = (0, 0) column, which lives on rank 0 (stored position 0 is wavenumber 0):
        // the single-process solver's ϕ .- mean(ϕ) (fourier_tridiagonal_poisson_solver.jl:142)
        if (s->rank == 0) {
            st = ocn::launch_remove_mean_mode((long long)s->Nxg * s->ny, Nz, s->xsol, stream);
            if (st != OCN_SUCCESS) return st;
        }
        if (s->xbounded) {
            xshuffle(2, s->xsol, s->xfield);
            st = s->bx.exec(s->xfield, s->xsol, stream);
            if (st != OCN_SUCCESS) return st;
            xshuffle(3, s->xsol, s->xfield);
            OCN_CHECK_HIP(hipGetLastError());
            return OCN_SUCCESS;
        }
        return s->bx.exec(s->xsol, s->xfield, stream);
    }
    // λy partitioned to this rank's j range; rank 0 zeroes mode (1,1,1) (distributed_fft_based_poisson_solver.jl:104-116,162-164)
    st = ocn::launch_spectral_solve(s->Nxg, s->ny, s->grid.Nz, s->lx, s->ly, s->lz, s->xfield, s->rank == 0, s->rank * s->ny, 0, stream);
    if (st != OCN_SUCCESS) return st;
    return s->bx.exec(s->xfield, nullptr, stream);
}

extern "C" int ocn_dist_poisson_backward_yz(ocn_dist_poisson_t s, double *p, void *stream_)
{
    OCN_REQUIRE(s && p, "ocn_dist_poisson_backward_yz: null argument");
    hipStream_t stream = ocn::as_stream(stream_);
    if (s->tri && s->fast) {  // recv (after the exchange back) -> real rows of p; the packed inverse carries Ny/2, FFT_x Nxg
        const ocn_grid *g = &s->grid;
        ocn::GridDev gd = ocn::to_dev(*g);
        ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
        return ocn::launch_realfft_y(g->Ny, 1, nullptr, s->recv, p + Lp.o, Lp.s2, Lp.s3, s->nx, g->Nz, s->tw_h, s->tw_y, stream, nullptr,
                                     nullptr, nullptr, nullptr, 1.0, s->ny, (long long)s->ny * g->Nz * s->nx, 0,
                                     1.0 / ((double)(g->Ny / 2) * s->Nxg));
    }
    if (s->tri) {
        int st = dist_tri_y_transform(s, 1, stream);
        if (st != OCN_SUCCESS) return st;
        return ocn::launch_copy_real(&s->grid, s->yfield, p, stream, 0);
    }
    if (s->fast) {  // send (after the exchange back) -> A1 (in recv) -> real rows of the local pressure interior
        const ocn_grid *g = &s->grid;
        ocn::GridDev gd = ocn::to_dev(*g);
        ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
        const long long cols = (long long)s->nyt * s->nx;
        int st = s->xtri ? ocn::launch_colfft(g->Nz, 1, s->recv, cols, 0, (int)cols, 1, s->tw_z, nullptr, nullptr, nullptr, 1.0, 1, stream)
                         : ocn::launch_colfft_slab_z(g->Nz, 1, s->send, s->recv, s->nx, s->nyt, s->R, s->tw_z, stream);
        if (st != OCN_SUCCESS) return st;
        return ocn::launch_realfft_y(g->Ny, 1, nullptr, s->recv, p + Lp.o, Lp.s2, Lp.s3, s->nx, g->Nz, s->tw_h, s->tw_y, stream);
    }
    if (s->r2c) {
        ocn::GridDev gd = ocn::to_dev(s->grid);
        ocn::Lay Lp = ocn::make_lay(gd, OCN_LOC_CCC);
        return s->byz.exec(s->yfield, p + Lp.o, stream);
    }
    int st = s->byz.exec(s->yfield, nullptr, stream);
    if (st != OCN_SUCCESS) return st;
    // copy_real_component! into the local pressure interior; the local grid's parent layout
    return ocn::launch_copy_real(&s->grid, s->yfield, p, stream, 0);
}
