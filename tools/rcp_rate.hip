// Issue cost of v_rcp_f64 against v_fma_f64 on one SIMD (gfx950): N independent chains per lane, one wave per SIMD and four per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/rcp_rate tools/rcp_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters)
{
    double a[8];
    for (int q = 0; q < 8; ++q) a[q] = 1.0 + threadIdx.x * 1e-3 + q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) a[q] = __builtin_fma(a[q], 1.0000001, 1e-9);
            if (MODE == 1) a[q] = __builtin_amdgcn_rcp(a[q]);
            if (MODE == 2) a[q] = a[q] * 1.0000001;
            if (MODE == 3) a[q] = a[q] + 1e-9;
        }
    }
    double s = 0;
    for (int q = 0; q < 8; ++q) s += a[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
float run(int blocks, int threads, int iters, double *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    double *d;
    hipMalloc(&d, 256 * 4 * 1024 * sizeof(double));
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = 256 * 4, threads = 64 * waves;  // one workgroup per SIMD-ish: waves per workgroup share a CU
        const float f = run<0>(blocks, threads, iters, d), r = run<1>(blocks, threads, iters, d), m = run<2>(blocks, threads, iters, d),
                    ad = run<3>(blocks, threads, iters, d);
        printf("waves/wg %d: fma %.3f ms, rcp %.3f ms, mul %.3f ms, add %.3f ms  -> rcp / fma = %.2f\n", waves, f, r, m, ad, r / f);
    }
    return 0;
}
