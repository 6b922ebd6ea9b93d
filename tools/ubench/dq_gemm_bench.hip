// micro-benchmark of the decoder stage's merged phase (conv tiles beside the next layer's input projection) exactly as compiled
// into the library: per-wave cycles of dq_gemm_tiles<1> (conv, waves 0-1) and dq_gemm_tiles<3> (projection, waves 2-7), one
// workgroup alone vs 256 at once (tools only)
#include "../../radae_amd/csrc/rade_kernels.hip"
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(NT_RX) void k_dq_bench(const unsigned short *wconv, const unsigned short *wgm, const float *scale, const float *bias, long long *cyc, int iters, int cin, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    DecShared *sh = (DecShared *)smem_raw;
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 26 * 768; i += NT_RX) { (&sh->xh[0][0])[i] = (_Float16)(0.01f * (i % 13)); (&sh->xl[0][0])[i] = (_Float16)(0.001f * (i % 7)); }
    __syncthreads();
    const DqGemm gc = (DqGemm){ wconv, 2, bias, 32, scale, 0, cin / 32, 0, 2 * cin / 32, 0, DQ_OUT_X, cin, 1, nullptr, 0 };
    const DqGemm gm = (DqGemm){ wgm, 18, bias, 288, scale, 0, 0, 0, cin / 32, 0, DQ_OUT_GI, 0, 0, nullptr, 0 };
    long long tw = 0, tp = 0;
    for (int it = 0; it < iters; it++) {
        __syncthreads();
        const long long t0 = clock64();
        if (wave < 2) { if (mode != 2) dq_gemm_tiles<1>(sh, gc, wave, 24, 0u); }
        else if (mode != 1) dq_gemm_tiles<3>(sh, gm, 3 * (wave - 2), 24, 0u);
        const long long t1 = clock64();
        __syncthreads();
        const long long t2 = clock64();
        tw += t1 - t0; tp += t2 - t0;
    }
    if ((tid & 63) == 0 && blockIdx.x == 0) { cyc[wave] = tw; cyc[8 + wave] = tp; }
}
int main()
{
    const int iters = 50;
    unsigned short *wc, *wg; float *sc, *bi; long long *cyc;
    const size_t nwc = (size_t)48 * 2 * 64 * 8, nwg = (size_t)24 * 18 * 64 * 8;
    hipMalloc(&wc, nwc * 2); hipMalloc(&wg, nwg * 2); hipMalloc(&sc, 512 * 4); hipMalloc(&bi, 512 * 4); hipMalloc(&cyc, 16 * 8);
    std::vector<unsigned short> h(nwg, 0x3c00); std::vector<float> f(512, 0.01f);
    hipMemcpy(wc, h.data(), nwc * 2, hipMemcpyHostToDevice); hipMemcpy(wg, h.data(), nwg * 2, hipMemcpyHostToDevice);
    hipMemcpy(sc, f.data(), 512 * 4, hipMemcpyHostToDevice); hipMemcpy(bi, f.data(), 512 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_dq_bench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DecShared));
    for (int cin : {32, 64, 128, 320, 576, 704})
        for (int mode : {0, 1, 2})
            for (int nb : {1, 256}) {
                for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_dq_bench, dim3(nb), dim3(NT_RX), sizeof(DecShared), 0, wc, wg, sc, bi, cyc, iters, cin, mode); hipDeviceSynchronize(); }
                long long c[16]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
                printf("cin %3d (%2d / %2d k-steps) %-9s blocks %3d: conv wave %6.0f, projection wave %6.0f (w7 %6.0f), phase %6.0f cycles\n", cin, 2 * cin / 32, cin / 32,
                       mode == 0 ? "both" : (mode == 1 ? "conv only" : "proj only"), nb, (double)c[0] / iters, (double)c[2] / iters, (double)c[7] / iters, (double)c[8] / iters);
            }
    return 0;
}
