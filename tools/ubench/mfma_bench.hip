// issue-rate probes for the matrix instructions the receiver uses (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int KIND>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int iters)
{
    f64x4 a64[NACC]; f32x4 a32[NACC];
    for (int i = 0; i < NACC; i++) { a64[i] = (f64x4){0, 0, 0, 0}; a32[i] = (f32x4){0, 0, 0, 0}; }
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6; float xf = (float)x, yf = (float)y;
    double v = x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (KIND == 0) a64[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a64[i], 0, 0, 0);
            if (KIND == 1) a32[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, yf, a32[i], 0, 0, 0);
            if (KIND == 2) { v = fma(v, y, x); }                 // dependent f64 FMA chain
        }
        if (KIND == 3) {                                          // independent f64 FMAs
#pragma unroll
            for (int i = 0; i < NACC; i++) a64[i][0] = fma(a64[i][0], y, x);
        }
    }
    const long long t1 = clock64();
    double s = v; for (int i = 0; i < NACC; i++) s += a64[i][0] + a64[i][1] + a32[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int KIND> void run(const char *name, int threads)
{
    double *out; long long *cyc; hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NACC, KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s threads %3d chains %d: %.1f cycles per op per wave\n", name, threads, NACC, (double)c / (iters * NACC));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<1, 0>("mfma_f64_16x16x4", 64); run<2, 0>("mfma_f64_16x16x4", 64); run<4, 0>("mfma_f64_16x16x4", 64);
    run<1, 0>("mfma_f64_16x16x4", 256); run<4, 0>("mfma_f64_16x16x4", 256); run<4, 0>("mfma_f64_16x16x4", 512);
    run<1, 1>("mfma_f32_16x16x4", 64); run<4, 1>("mfma_f32_16x16x4", 64); run<4, 1>("mfma_f32_16x16x4", 512);
    run<1, 2>("f64 fma dependent", 64); run<8, 3>("f64 fma independent", 64); run<8, 3>("f64 fma independent", 512);
    return 0;
}
