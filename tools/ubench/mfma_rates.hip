// developer aid: issue rates of the instructions the receiver kernel leans on, measured on one CU (one workgroup) with 1 or 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rates tools/ubench/mfma_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define N 2000
template <int KIND, int NACC>
__global__ void k(long long *out, double seed)
{
    const int lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    if (KIND == 0) {           // v_mfma_f64_16x16x4_f64
        f64x4 acc[NACC]; for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0, 0, 0, 0};
        double a = seed + lane, b = seed - lane;
        __syncthreads(); t0 = clock64();
        for (int it = 0; it < N; it++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        double s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
        t1 = clock64(); if (s == 1.2345) out[100] = 1;
    } else if (KIND == 1) {    // v_mfma_f32_16x16x32_f16
        f32x4 acc[NACC]; for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0, 0, 0, 0};
        f16x8 a, b; for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(lane + i); }
        __syncthreads(); t0 = clock64();
        for (int it = 0; it < N; it++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
        t1 = clock64(); if (s == 1.2345f) out[100] = 1;
    } else if (KIND == 2) {    // v_pk_fma_f32
        f32x2 acc[NACC]; for (int i = 0; i < NACC; i++) acc[i] = (f32x2){(float)seed, (float)lane};
        f32x2 a = {(float)seed, 1.0f}, b = {0.5f, (float)lane};
        __syncthreads(); t0 = clock64();
        for (int it = 0; it < N; it++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_elementwise_fma(a, acc[i], b);
        float s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1];
        t1 = clock64(); if (s == 1.2345f) out[100] = 1;
    } else if (KIND == 3) {    // v_fma_f32
        float acc[NACC]; for (int i = 0; i < NACC; i++) acc[i] = (float)seed + lane + i;
        float a = (float)seed, b = (float)lane;
        __syncthreads(); t0 = clock64();
        for (int it = 0; it < N; it++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = fmaf(a, acc[i], b);
        float s = 0; for (int i = 0; i < NACC; i++) s += acc[i];
        t1 = clock64(); if (s == 1.2345f) out[100] = 1;
    } else if (KIND == 4) {    // v_fma_f64
        double acc[NACC]; for (int i = 0; i < NACC; i++) acc[i] = seed + lane + i;
        double a = seed, b = (double)lane;
        __syncthreads(); t0 = clock64();
        for (int it = 0; it < N; it++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = fma(a, acc[i], b);
        double s = 0; for (int i = 0; i < NACC; i++) s += acc[i];
        t1 = clock64(); if (s == 1.2345) out[100] = 1;
    } else if (KIND == 5) {    // v_mfma_f32_32x32x16_f16
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 acc[NACC]; for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
        f16x8 a, b; for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(lane + i); }
        __syncthreads(); t0 = clock64();
        for (int it = 0; it < N; it++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
        t1 = clock64(); if (s == 1.2345f) out[100] = 1;
    }
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND, int NACC> void run(const char *name, long long *d)
{
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL((k<KIND, NACC>), dim3(1), dim3(threads), 0, 0, d, 1.0);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        long long mx = 0; for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
        printf("%-28s nacc %d  %d waves/CU: %.1f cycles per instruction per wave, %.1f per SIMD-slot\n", name, NACC, threads / 64, (double)mx / (N * NACC), (double)mx / (N * NACC) / (threads >= 256 ? threads / 256 : 1));
    }
}
int main()
{
    long long *d; hipMalloc(&d, 1024);
    run<0, 1>("mfma_f64_16x16x4 (dep chain)", d); run<0, 4>("mfma_f64_16x16x4", d);
    run<1, 1>("mfma_f32_16x16x32_f16 (dep)", d); run<1, 4>("mfma_f32_16x16x32_f16", d);
    run<5, 1>("mfma_f32_32x32x16_f16 (dep)", d); run<5, 4>("mfma_f32_32x32x16_f16", d);
    run<2, 8>("v_pk_fma_f32", d); run<3, 8>("v_fma_f32", d); run<4, 8>("v_fma_f64", d);
    return 0;
}
