// micro-benchmark of the per-wave 2048-point FFT used by the pilot correlator (tools only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FFT_N 2048
#define FFT_SCR (32*65)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x); }

__device__ static constexpr float C32[16] = { 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f, 6.123233996e-17f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f };
__device__ static constexpr float S32[16] = { 0.000000000e+00f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f, 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f };
__device__ static constexpr float C64[32] = { 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 6.123233996e-17f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f };
__device__ static constexpr float S64[32] = { 0.000000000e+00f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f };
__host__ __device__ constexpr int brev5(int x) { return ((x & 1) << 4) | ((x & 2) << 2) | (x & 4) | ((x & 8) >> 2) | ((x & 16) >> 4); }
__device__ __forceinline__ float lane_swap1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]

__device__ __forceinline__ void dft32_inlane(float2 (&v)[32])
{   // forward DFT, decimation in frequency: output X[q] is left in v[brev5(q)]
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int half = 16 >> s;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
#pragma unroll
            for (int k = 0; k < half; k++) {
                const int i = g * 2 * half + k, j = i + half, e = k << s;
                const float2 x = v[i], y = v[j];
                v[i] = make_float2(x.x + y.x, x.y + y.y);
                const float dr = x.x - y.x, di = x.y - y.y;
                if (e == 0) v[j] = make_float2(dr, di);
                else if (e == 8) v[j] = make_float2(di, -dr);                       // times -j
                else if (e == 4) v[j] = make_float2((dr + di) * C32[4], (di - dr) * C32[4]);
                else if (e == 12) v[j] = make_float2((di - dr) * C32[4], -(dr + di) * C32[4]);
                else v[j] = make_float2(dr * C32[e] + di * S32[e], di * C32[e] - dr * S32[e]);   // times e^{-j2pi e/32}
            }
        }
    }
}

// v[k2] = x[lane + 64 k2] in, X[q + 32(2p + h)] (q = lane>>1, h = lane&1) in v[brev5(p)] out
__device__ __forceinline__ void fft2048_wave(float2 (&v)[32], float *scr, const float2 *__restrict__ tw, int lane)
{
    dft32_inlane(v);
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int q = 8 * c; q < 8 * c + 8; q++) if (q) { const float2 w = tw[q * 64 + lane]; v[brev5(q)] = cmul(v[brev5(q)], w); }
        __builtin_amdgcn_sched_barrier(0);
    }
    const int q2 = lane >> 1, h = lane & 1;
    float ur[32], ui[32];
#pragma unroll
    for (int q = 0; q < 32; q++) scr[q * 65 + lane] = v[brev5(q)].x;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < 32; l++) ur[l] = scr[q2 * 65 + 32 * h + l];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 32; q++) scr[q * 65 + lane] = v[brev5(q)].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < 32; l++) ui[l] = scr[q2 * 65 + 32 * h + l];
    __builtin_amdgcn_wave_barrier();
    // radix-2 DIF stage over l <-> l + 32 (the partner lane): even outputs on h = 0, odd outputs (twiddled) on h = 1
    const float sg = h ? -1.0f : 1.0f;
    const float2 *w64 = tw + 2048 + h * 32;                                          // h = 0: ones, h = 1: e^{-j2pi l/64}
#pragma unroll
    for (int l = 0; l < 32; l++) {
        const float sr = fmaf(ur[l], sg, lane_swap1(ur[l])), si = fmaf(ui[l], sg, lane_swap1(ui[l]));   // h = 0: u + o;  h = 1: o - u
        if (l == 0) v[l] = make_float2(sr, si);
        else { const float2 w = w64[l]; v[l] = cmul(make_float2(sr, si), w); }
    }
    dft32_inlane(v);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_test(const float2 *X, const float2 *G, const float2 *tw, float *dst, long long *cyc, int iters)
{
    __shared__ float scr_all[8][FFT_SCR];
    __shared__ float2 fftX[FFT_N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q2 = lane >> 1, h = lane & 1;
    float *scr = scr_all[wave];
    for (int i = threadIdx.x; i < FFT_N; i += 512) fftX[i] = X[i];
    __syncthreads();
    float2 v[32];
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll 1
    for (int fi = 0; fi < 5; fi++) {
        const int f = wave * 5 + fi;
        const float2 *Gf = G + (size_t)(MODE == 1 ? 0 : f) * FFT_N;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int k2 = 8 * c; k2 < 8 * c + 8; k2++) { const float2 g = MODE == 2 ? make_float2(0.5f, 0.25f) : Gf[lane + 64 * k2]; const float2 y = cmul(fftX[lane + 64 * k2], g); v[k2] = make_float2(y.y, y.x); }
            __builtin_amdgcn_sched_barrier(0);
        }
        fft2048_wave(v, scr, tw, lane);
#pragma unroll
        for (int p = 0; p < 15; p++) { const float2 c = v[brev5(p)]; dst[((size_t)blockIdx.x * 40 + f) * 960 + q2 + 64 * p + 32 * h] = __builtin_amdgcn_sqrtf(fmaf(c.x, c.x, c.y * c.y)); }
    }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    const int B = 256, iters = 20;
    float2 *X, *G, *tw; float *dst; long long *cyc;
    hipMalloc(&X, FFT_N * 8); hipMalloc(&G, 40 * FFT_N * 8); hipMalloc(&tw, (2048 + 64) * 8); hipMalloc(&dst, (size_t)B * 40 * 960 * 4); hipMalloc(&cyc, B * 8);
    std::vector<float2> h(40 * FFT_N, make_float2(0.01f, 0.02f));
    hipMemcpy(X, h.data(), FFT_N * 8, hipMemcpyHostToDevice); hipMemcpy(G, h.data(), 40 * FFT_N * 8, hipMemcpyHostToDevice); hipMemcpy(tw, h.data(), (2048 + 64) * 8, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode++)
      for (int nb : {1, 256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_test<0>, dim3(nb), dim3(512), 0, 0, X, G, tw, dst, cyc, iters);
            if (mode == 1) hipLaunchKernelGGL(k_test<1>, dim3(nb), dim3(512), 0, 0, X, G, tw, dst, cyc, iters);
            if (mode == 2) hipLaunchKernelGGL(k_test<2>, dim3(nb), dim3(512), 0, 0, X, G, tw, dst, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d blocks %3d: %.3f ms, block0 %lld cycles => %.0f cycles per FFT per wave (5 FFTs x %d iters), %.2f us per surface\n", mode, nb, ms, c0, (double)c0 / (5.0 * iters), iters, ms * 1e3 / iters);
      }
    return 0;
}
