// Do a latency-bound kernel (one wave per SIMD, like the GRU scans) and a matrix-core kernel (like k_gemm16) overlap when they
// are launched on two HIP streams?   (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_serial(float *out, int iters)
{   // dependent FMA chain + a barrier per step: VALU mostly idle
    __shared__ float s[256];
    float v = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) { s[threadIdx.x] = v; __syncthreads(); v = fmaf(v, 0.999f, s[(threadIdx.x + 1) & 255]); __syncthreads(); }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ __launch_bounds__(64) void k_mfma(float *out, int iters)
{
    f32x16 acc[6]; for (int i = 0; i < 6; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    f16x8 a, b; for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.01f * (threadIdx.x + j)); b[j] = (_Float16)(0.02f * j); }
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    float s = 0; for (int i = 0; i < 6; i++) s += acc[i][0];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main()
{
    float *o1, *o2; hipMalloc(&o1, 256 * 256 * 4); hipMalloc(&o2, 2048 * 64 * 4);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](bool A, bool B, const char *name) {
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(e0, sa); hipStreamWaitEvent(sb, e0, 0);
            if (A) hipLaunchKernelGGL(k_serial, dim3(256), dim3(256), 0, sa, o1, 2000);
            if (B) hipLaunchKernelGGL(k_mfma, dim3(1024), dim3(64), 0, sb, o2, 3000);
            hipEventRecord(e1, sb); hipStreamWaitEvent(sa, e1, 0);
            hipEvent_t e2; hipEventCreate(&e2); hipEventRecord(e2, sa); hipEventSynchronize(e2);
            float ms; hipEventElapsedTime(&ms, e0, e2);
            if (rep) printf("%-28s %.3f ms\n", name, ms);
            hipEventDestroy(e2);
        }
    };
    run(true, false, "serial kernel alone");
    run(false, true, "matrix kernel alone");
    run(true, true, "both, two streams");
    return 0;
}
