// micro-benchmark of refine_tile() exactly as compiled into the library (tools only)
#include "../../radae_amd/csrc/rade_kernels.hip"
#include <cstdio>
__global__ __launch_bounds__(NT_RX) void k_refine_bench(double *out, long long *cyc, int iters, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    RxShared *sh = (RxShared *)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < RD_RXBUF; i += NT_RX) sh->rxb[i] = make_float2(0.01f * (i % 37), 0.02f * (i % 11));
    for (int i = tid; i < RD_M; i += NT_RX) sh->pd[i] = make_double2(0.1 + i * 1e-3, 0.2 - i * 1e-3);
    for (int i = tid; i < 80; i += NT_RX) { sh->rtw[i] = make_double2(cos(0.01 * (i + 1)), -sin(0.01 * (i + 1))); sh->rt80[i] = make_double2(cos(0.8 * (i + 1)), -sin(0.8 * (i + 1))); sh->rrot[i] = sh->rtw[i]; }
    for (int i = tid; i < 2 * 176 * 4; i += NT_RX) ((double *)&sh->xm[0])[i] = 0.001 * (i % 91);
    __syncthreads();
    f64x4 acc = { 0, 0, 0, 0 };
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (mode == 0) { if (wave < 6) acc += refine_tile(sh, wave >> 1, wave & 1, 0, 40, 20, 16, lane); if (wave < 4 || wave >= 6) acc += refine_tile(sh, wave & 1, 1, 40, 40, 20, 16, lane); }
        if (mode == 1) { if (wave == 0) acc += refine_tile(sh, 0, 0, 0, 80, 20, 16, lane); }
        __syncthreads();
    }
    const long long t1 = clock64();
    out[blockIdx.x * NT_RX + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 256 * NT_RX * 8); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void *)k_refine_bench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RxShared));
    const int iters = 50;
    for (int mode = 0; mode < 2; mode++)
      for (int nb : {1, 256}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_refine_bench, dim3(nb), dim3(NT_RX), sizeof(RxShared), 0, out, cyc, iters, mode); hipDeviceSynchronize(); }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d blocks %3d: %.0f cycles per refine grid (mode 0: 12 pieces of 40 steps on 8 waves; mode 1: one wave, 80 steps)\n", mode, nb, (double)c / iters);
      }
    return 0;
}
