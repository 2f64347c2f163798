// tools/rocfft_repro.hip -- standalone reproduction of the order-dependent wrong spectra of rocFFT real-transform plans
// (DESIGN.md "rocFFT plan self-test").  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/rocfft_repro.hip -o /tmp/rocfft_repro -lrocfft && /tmp/rocfft_repro <scenario>
// Each scenario runs in its own process (the failure depends on process-wide rocFFT state).  A plan pair is described exactly as
// csrc/poisson.hip builds it: real (Nx, Ny[, Nz]) -> hermitian (Nx/2+1, Ny[, Nz]) not in place with explicit strides; the
// inverse writes either a contiguous real array or straight into the interior of a haloed field (strides sx, sx*sy).
// For every executed forward plan the FULL spectrum is compared with a host DFT.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAILED %s -> %d (line %d)\n", #x, (int)e_, __LINE__); exit(2); } } while (0)

struct Pair {
    int Nx, Ny, Nz, dims;  // dims = 3: one 3-D transform; 2: (x, y) batched over z
    int H;                 // halo of the field the inverse writes into (0: contiguous output)
    rocfft_plan fwd = nullptr, bwd = nullptr;
    rocfft_execution_info fi = nullptr, bi = nullptr;
    void *fw = nullptr, *bw = nullptr;
    double *rhs = nullptr, *spec = nullptr, *p = nullptr;
    size_t nxh() const { return Nx / 2 + 1; }
};

static rocfft_plan make(rocfft_transform_type type, int dims, const size_t *len, size_t batch, rocfft_array_type it, rocfft_array_type ot,
                        const size_t *is, size_t id, const size_t *os, size_t od, bool describe)
{
    rocfft_plan p = nullptr;
    rocfft_plan_description d = nullptr;
    if (describe) {
        CK(rocfft_plan_description_create(&d));
        CK(rocfft_plan_description_set_data_layout(d, it, ot, nullptr, nullptr, dims, is, id, dims, os, od));
    }
    CK(rocfft_plan_create(&p, rocfft_placement_notinplace, type, rocfft_precision_double, dims, len, batch, d));
    if (d) CK(rocfft_plan_description_destroy(d));
    return p;
}

static void finish(rocfft_plan p, rocfft_execution_info *info, void **work, void *shared_work, size_t shared_bytes)
{
    size_t wb = 0;
    CK(rocfft_plan_get_work_buffer_size(p, &wb));
    CK(rocfft_execution_info_create(info));
    if (wb) {
        if (shared_work && wb <= shared_bytes) {
            CK(rocfft_execution_info_set_work_buffer(*info, shared_work, shared_bytes));
        } else {
            CK(hipMalloc(work, wb));
            CK(rocfft_execution_info_set_work_buffer(*info, *work, wb));
        }
    }
    printf("    work buffer %zu bytes\n", wb);
}

static void create(Pair &P, bool describe, bool inverse, void *shared_work = nullptr, size_t shared_bytes = 0)
{
    const int Nx = P.Nx, Ny = P.Ny, Nz = P.Nz;
    const size_t nxh = P.nxh();
    const size_t len[3] = {(size_t)Nx, (size_t)Ny, (size_t)Nz};
    const size_t batch = P.dims == 3 ? 1 : Nz;
    const size_t rstr[3] = {1, (size_t)Nx, (size_t)Nx * Ny}, cstr[3] = {1, nxh, nxh * Ny};
    const size_t rdist = (size_t)Nx * Ny * (P.dims == 3 ? Nz : 1), cdist = nxh * Ny * (P.dims == 3 ? Nz : 1);
    const size_t sx = Nx + 2 * P.H, sy = Ny + 2 * P.H, sz = Nz + 2 * P.H;
    const size_t pstr[3] = {1, sx, sx * sy};
    const size_t pdist = P.dims == 3 ? sx * sy * sz : sx * sy;
    printf("  plan pair %dx%dx%d dims=%d halo=%d describe=%d\n", Nx, Ny, Nz, P.dims, P.H, (int)describe);
    P.fwd = make(rocfft_transform_type_real_forward, P.dims, len, batch, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, rstr,
                 rdist, cstr, cdist, describe);
    finish(P.fwd, &P.fi, &P.fw, shared_work, shared_bytes);
    if (inverse) {
        P.bwd = make(rocfft_transform_type_real_inverse, P.dims, len, batch, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, cstr,
                     cdist, P.H ? pstr : rstr, P.H ? pdist : rdist, describe || P.H);
        finish(P.bwd, &P.bi, &P.bw, shared_work, shared_bytes);
    }
    CK(hipMalloc(&P.rhs, sizeof(double) * Nx * Ny * Nz));
    CK(hipMalloc(&P.spec, sizeof(double) * 2 * nxh * Ny * Nz));
    CK(hipMalloc(&P.p, sizeof(double) * sx * sy * sz));
}

static std::vector<double> random_field(size_t n, unsigned long long seed)
{
    std::vector<double> a(n);
    unsigned long long x = seed;
    for (size_t q = 0; q < n; ++q) {
        x += 0x9E3779B97F4A7C15ull;
        unsigned long long z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        a[q] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    return a;
}

// max |device spectrum - host DFT| over the FULL half spectrum, and the round-trip error
static void check(Pair &P, const char *name)
{
    const int Nx = P.Nx, Ny = P.Ny, Nz = P.Nz;
    const size_t nxh = P.nxh(), n = (size_t)Nx * Ny * Nz;
    std::vector<double> in = random_field(n, 0x1234 + Nx * 131 + Ny * 17 + Nz);
    CK(hipMemcpy(P.rhs, in.data(), n * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemset(P.spec, 0, sizeof(double) * 2 * nxh * Ny * Nz));
    void *ib[1] = {P.rhs}, *ob[1] = {P.spec};
    CK(rocfft_execute(P.fwd, ib, ob, P.fi));
    CK(hipDeviceSynchronize());
    std::vector<double> sp(2 * nxh * Ny * Nz);
    CK(hipMemcpy(sp.data(), P.spec, sp.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double tp = 6.283185307179586476925286766559;
    double worst = 0.0;
    size_t bad = 0;
    // separable host DFT: x, then y, then (dims == 3) z
    std::vector<double> a(2 * nxh * Ny * Nz), b(a.size());
    for (int k = 0; k < Nz; ++k)
        for (int j = 0; j < Ny; ++j)
            for (size_t kx = 0; kx < nxh; ++kx) {
                double re = 0, im = 0;
                for (int i = 0; i < Nx; ++i) {
                    const double ph = -tp * (double)((kx * i) % Nx) / Nx, v = in[i + (size_t)Nx * (j + (size_t)Ny * k)];
                    re += v * cos(ph); im += v * sin(ph);
                }
                const size_t o = 2 * (kx + nxh * (j + (size_t)Ny * k));
                a[o] = re; a[o + 1] = im;
            }
    for (int k = 0; k < Nz; ++k)
        for (int ky = 0; ky < Ny; ++ky)
            for (size_t kx = 0; kx < nxh; ++kx) {
                double re = 0, im = 0;
                for (int j = 0; j < Ny; ++j) {
                    const double ph = -tp * (double)((ky * j) % Ny) / Ny;
                    const size_t o = 2 * (kx + nxh * (j + (size_t)Ny * k));
                    re += a[o] * cos(ph) - a[o + 1] * sin(ph); im += a[o] * sin(ph) + a[o + 1] * cos(ph);
                }
                const size_t o = 2 * (kx + nxh * (ky + (size_t)Ny * k));
                b[o] = re; b[o + 1] = im;
            }
    if (P.dims == 3) {
        a = b;
        for (int kz = 0; kz < Nz; ++kz)
            for (int ky = 0; ky < Ny; ++ky)
                for (size_t kx = 0; kx < nxh; ++kx) {
                    double re = 0, im = 0;
                    for (int k = 0; k < Nz; ++k) {
                        const double ph = -tp * (double)((kz * k) % Nz) / Nz;
                        const size_t o = 2 * (kx + nxh * (ky + (size_t)Ny * k));
                        re += a[o] * cos(ph) - a[o + 1] * sin(ph); im += a[o] * sin(ph) + a[o + 1] * cos(ph);
                    }
                    const size_t o = 2 * (kx + nxh * (ky + (size_t)Ny * kz));
                    b[o] = re; b[o + 1] = im;
                }
    }
    for (size_t q = 0; q < b.size(); ++q) {
        const double e = fabs(b[q] - sp[q]);
        if (e > 1e-9 * Nx * Ny) ++bad;
        worst = fmax(worst, e);
    }
    double rt = -1.0;
    if (P.bwd) {
        const size_t sx = Nx + 2 * P.H, sy = Ny + 2 * P.H, sz = Nz + 2 * P.H;
        CK(hipMemset(P.p, 0, sizeof(double) * sx * sy * sz));
        double *outp = P.H ? P.p + P.H + sx * (P.H + sy * (P.dims == 3 || true ? P.H : 0)) : P.rhs;
        void *ib2[1] = {P.spec}, *ob2[1] = {outp};
        CK(rocfft_execute(P.bwd, ib2, ob2, P.bi));
        CK(hipDeviceSynchronize());
        std::vector<double> out(P.H ? sx * sy * sz : n);
        CK(hipMemcpy(out.data(), P.H ? P.p : P.rhs, out.size() * sizeof(double), hipMemcpyDeviceToHost));
        const double cnt = (double)Nx * Ny * (P.dims == 3 ? Nz : 1);
        rt = 0.0;
        for (int k = 0; k < Nz; ++k)
            for (int j = 0; j < Ny; ++j)
                for (int i = 0; i < Nx; ++i) {
                    const double v = P.H ? out[(i + P.H) + sx * ((j + P.H) + sy * (k + P.H))] : out[i + (size_t)Nx * (j + (size_t)Ny * k)];
                    rt = fmax(rt, fabs(v / cnt - in[i + (size_t)Nx * (j + (size_t)Ny * k)]));
                }
    }
    printf("  CHECK %-28s %dx%dx%d dims=%d: forward spectrum max err %.3e (%zu of %zu entries wrong)   round trip err %.3e  => %s\n", name, Nx, Ny,
           Nz, P.dims, worst, bad, b.size(), rt, (bad == 0 && (rt < 0 || rt < 1e-9)) ? "ok" : "WRONG");
}

// usage: rocfft_repro <op> <op> ...   op = c<P> create forward + inverse pair, f<P> create the forward plan only, x<P> execute + check,
//        d<P> destroy the plans of P;  P = A (16^3 3-D, inverse into a haloed field), a (16^3 3-D contiguous), B (32x8 2-D x 8 planes),
//        C (16x16 2-D x 64 planes), D (16x16 2-D x 16 planes), E (32x8 2-D x 8 planes as ONE 3-D-described batch: same as B)
int main(int argc, char **argv)
{
    CK(rocfft_setup());
    Pair P[128];
    P['A'] = Pair{16, 16, 16, 3, 3};
    P['a'] = Pair{16, 16, 16, 3, 0};
    P['B'] = Pair{32, 8, 8, 2, 0};
    P['C'] = Pair{16, 16, 64, 2, 0};
    P['D'] = Pair{16, 16, 16, 2, 0};
    P['F'] = Pair{32, 8, 8, 3, 0};
    printf("sequence:");
    for (int q = 1; q < argc; ++q) printf(" %s", argv[q]);
    printf("\n");
    for (int q = 1; q < argc; ++q) {
        // "=X:NxxNyxNz:dims" defines pair X (halo 0), e.g. =G:48x8x4:2
        if (argv[q][0] == '=') {
            int nx, ny, nz, dm;
            if (sscanf(argv[q] + 3, "%dx%dx%d:%d", &nx, &ny, &nz, &dm) == 4) P[(int)argv[q][1]] = Pair{nx, ny, nz, dm, 0};
            continue;
        }
        const char op = argv[q][0], id = argv[q][1];
        Pair &p = P[(int)id];
        char name[32];
        snprintf(name, sizeof name, "%c after op %d", id, q);
        if (op == 'c') create(p, true, true);
        else if (op == 'f') create(p, true, false);
        else if (op == 'x') check(p, name);
        else if (op == 'd') {
            if (p.fwd) rocfft_plan_destroy(p.fwd);
            if (p.bwd) rocfft_plan_destroy(p.bwd);
            p.fwd = p.bwd = nullptr;
        }
    }
    return 0;
}
