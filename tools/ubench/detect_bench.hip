// micro-benchmark of rx_detect_fft() exactly as compiled into the library (tools only)
#include "../../radae_amd/csrc/rade_kernels.hip"
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(NT_RX) void k_detect_bench(const float *G, const float *tw, float *cache, long long *cyc, int iters, int cached)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    RxShared *sh = (RxShared *)smem_raw;
    for (int i = threadIdx.x; i < RD_RXBUF; i += NT_RX) sh->rxb[i] = make_float2(0.01f * (i % 37), 0.02f * (i % 11));
    __syncthreads();
    const long long t0 = clock64();
    float best = -1.0f; int bt = 0, bfi = 0;
    for (int it = 0; it < iters; it++) rx_detect_fft(sh, G, tw, cache + (size_t)blockIdx.x * 2 * RD_NFC * RD_NMF, cached, it & 1, 1 - (it & 1), best, bt, bfi);
    if (best == 12345.0f) cyc[1] = bt + bfi;
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    const int B = 256, iters = 20;
    float *G, *tw, *cache; long long *cyc;
    hipMalloc(&G, 40 * 2048 * 8); hipMalloc(&tw, (2048 + 64) * 8); hipMalloc(&cache, (size_t)B * 2 * 40 * 960 * 4); hipMalloc(&cyc, B * 8);
    std::vector<float> h(40 * 2048 * 2, 0.01f);
    hipMemcpy(G, h.data(), 40 * 2048 * 8, hipMemcpyHostToDevice); hipMemcpy(tw, h.data(), (2048 + 64) * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_detect_bench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RxShared));
    for (int nb : {1, 256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_detect_bench, dim3(nb), dim3(NT_RX), sizeof(RxShared), 0, G, tw, cache, cyc, iters, 1);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        printf("blocks %3d: %.3f ms, block0 %lld cycles => %.0f cycles per surface, %.2f us per surface\n", nb, ms, c0, (double)c0 / iters, ms * 1e3 / iters);
    }
    return 0;
}
