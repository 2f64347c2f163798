/* c_abi_rk3.c -- a host WITHOUT Python or torch: plain C99 against include/ocn_hip.h (libocn_hip.so).
 *
 * What a Julia extension does through ccall, done here through the same symbols: device memory from ocn_malloc,
 * the grid description, set! (halo fills + the dt = 1 projection, set_nonhydrostatic_model.jl:52-57), then full RK3 time_step!s
 * of the WENO5 NonhydrostaticModel through ocn_rk3_driver_* (runge_kutta_3.jl:77-151).  Writes the final u, v, w interiors to a
 * file so that tests/test_gpu_model.py can compare them bit for bit with the Python host driving the same library.
 *
 *   c_abi_rk3 N steps dt strict|fast out.bin          (N^3 periodic box of extent (2 pi)^3, halo 3)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ocn_hip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int st_ = (call);                                                                \
        if (st_ != OCN_SUCCESS) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, st_, ocn_last_error());      \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

/* the initial condition both hosts use: a 64-bit LCG mapped to (-1, 1), drawn in the order i fastest, then j, k, for u, v, w */
static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double lcg_uniform(void)
{
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return ((double)(lcg_state >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 32, steps = argc > 2 ? atoi(argv[2]) : 3, H = 3;
    const double dt = argc > 3 ? atof(argv[3]) : 1e-3;
    const int strict = argc > 4 ? !strcmp(argv[4], "strict") : 1;
    const char *out = argc > 5 ? argv[5] : NULL;
    const double two_pi = 6.283185307179586;
    int ndev = 0;
    CHECK(ocn_device_count(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no GPU\n");
        return 2;
    }
    CHECK(ocn_set_device(0));
    CHECK(ocn_set_math_mode(strict ? OCN_MATH_STRICT : OCN_MATH_FAST));

    ocn_grid g;
    memset(&g, 0, sizeof g);
    g.Nx = g.Ny = g.Nz = N;
    g.Hx = g.Hy = g.Hz = H;
    g.tx = g.ty = g.tz = OCN_PERIODIC;
    g.dx = g.dy = g.dz = two_pi / N;
    g.Lx = g.Ly = g.Lz = two_pi;
    g.dzc = g.dzf = NULL;

    const size_t s = (size_t)N + 2 * H, n = s * s * s, bytes = n * sizeof(double);
    double *host = (double *)calloc(n, sizeof(double));
    double *dev[4]; /* u, v, w, p: the OffsetArray parents, x fastest */
    for (int f = 0; f < 4; ++f) CHECK(ocn_malloc((void **)&dev[f], bytes));
    for (int f = 0; f < 3; ++f) {
        memset(host, 0, bytes);
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) host[(i + H) + s * ((j + H) + s * (size_t)(k + H))] = lcg_uniform();
        CHECK(ocn_memcpy_h2d(dev[f], host, bytes, NULL));
    }

    /* set!(model; u, v, w): fill halos, project with dt = 1, fill halos */
    const int32_t locs[3] = {OCN_LOC_FCC, OCN_LOC_CFC, OCN_LOC_CCF};
    double *vel[3] = {dev[0], dev[1], dev[2]};
    ocn_poisson_t solver = NULL;
    CHECK(ocn_poisson_create(&solver, &g));
    CHECK(ocn_fill_halo_regions(&g, vel, locs, 3, 1, NULL));
    CHECK(ocn_solve_for_pressure(solver, dev[3], dev[0], dev[1], dev[2], 1.0, NULL));
    {
        double *pp[1] = {dev[3]};
        const int32_t pl[1] = {OCN_LOC_CCC};
        CHECK(ocn_fill_halo_regions(&g, pp, pl, 1, 1, NULL));
    }
    CHECK(ocn_pressure_correct_velocities(&g, dev[0], dev[1], dev[2], dev[3], 1.0, NULL));
    CHECK(ocn_fill_halo_regions(&g, vel, locs, 3, 0, NULL));

    ocn_rk3_driver_t drv = NULL;
    CHECK(ocn_rk3_driver_create(&drv, &g, dev[0], dev[1], dev[2], dev[3], solver, NULL));
    for (int n_ = 0; n_ < steps; ++n_) CHECK(ocn_rk3_driver_time_step(drv, dt, NULL));
    CHECK(ocn_rk3_driver_flush(drv, NULL));
    CHECK(ocn_sync(NULL));

    /* discrete divergence of the result and the interiors for the comparison */
    double *div = NULL;
    CHECK(ocn_malloc((void **)&div, (size_t)N * N * N * sizeof(double)));
    CHECK(ocn_divergence(&g, dev[0], dev[1], dev[2], div, NULL));
    double *hdiv = (double *)malloc((size_t)N * N * N * sizeof(double));
    CHECK(ocn_memcpy_d2h(hdiv, div, (size_t)N * N * N * sizeof(double), NULL));
    CHECK(ocn_sync(NULL));
    double dmax = 0.0;
    for (size_t q = 0; q < (size_t)N * N * N; ++q) dmax = fmax(dmax, fabs(hdiv[q]));

    FILE *fo = out ? fopen(out, "wb") : NULL;
    double umax = 0.0;
    for (int f = 0; f < 3; ++f) {
        CHECK(ocn_memcpy_d2h(host, dev[f], bytes, NULL));
        CHECK(ocn_sync(NULL));
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j) {
                const double *row = host + H + s * ((j + H) + s * (size_t)(k + H));
                for (int i = 0; i < N; ++i) umax = fmax(umax, fabs(row[i]));
                if (fo) fwrite(row, sizeof(double), (size_t)N, fo);
            }
    }
    if (fo) fclose(fo);
    printf("c_abi_rk3: %s, N = %d, %d RK3 steps of dt = %g (%s math): max|u| = %.6f, max|div u| = %.3e\n", ocn_version(), N, steps, dt,
           strict ? "strict" : "fast", umax, dmax);

    CHECK(ocn_rk3_driver_destroy(drv));
    CHECK(ocn_poisson_destroy(solver));
    CHECK(ocn_free(div));
    for (int f = 0; f < 4; ++f) CHECK(ocn_free(dev[f]));
    free(host);
    free(hdiv);
    return (dmax < 1e-9 && isfinite(umax)) ? 0 : 3;
}
