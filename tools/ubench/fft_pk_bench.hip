// packed-f32 (v_pk_*) variant of the per-wave 2048-point FFT vs the scalar one: correctness + speed (tools only)
#include "../../radae_amd/csrc/rade_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>

typedef float v2f __attribute__((ext_vector_type(2)));
// d * (c - j s) with K = {c, s} in an SGPR pair: two VOP3P instructions, no register shuffles
__device__ __forceinline__ v2f pk_twid(v2f d, v2f K)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(d), "s"(K));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(d), "s"(K), "v"(t));
    return r;
}
// a * b (complex), b in VGPRs
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ void dft32_pk(v2f (&v)[32])
{
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int half = 16 >> s;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
#pragma unroll
            for (int k = 0; k < half; k++) {
                const int i = g * 2 * half + k, j = i + half, e = k << s;
                const v2f x = v[i], y = v[j];
                v[i] = x + y;
                const v2f d = x - y;
                if (e == 0) v[j] = d;
                else { const v2f K = { C32[e], S32[e] }; v[j] = pk_twid(d, K); }
            }
        }
    }
}
__device__ __forceinline__ void fft2048_wave_pk(v2f (&v)[32], lds_float *scr, const glb_float *__restrict__ tw, int lane)
{
    dft32_pk(v);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v2f w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = 8 * c + u; w[u] = (v2f){ tw[2 * (q * 64 + lane)], tw[2 * (q * 64 + lane) + 1] }; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = 8 * c + u; if (q) v[brev5(q)] = pk_cmul(v[brev5(q)], w[u]); }
        __builtin_amdgcn_sched_barrier(0);
    }
    const int q2 = lane >> 1, h = lane & 1;
    float ur[32], ui[32];
#pragma unroll
    for (int q = 0; q < 32; q++) scr[FFT_WR(q, lane)] = v[brev5(q)].x;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < 32; l++) ur[l] = scr[FFT_RD(q2, h, l)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 32; q++) scr[FFT_WR(q, lane)] = v[brev5(q)].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int l = 0; l < 32; l++) ui[l] = scr[FFT_RD(q2, h, l)];
    __builtin_amdgcn_wave_barrier();
    const float sg = h ? -1.0f : 1.0f;
    const v2f sg2 = { sg, sg };
    const glb_float *w64 = tw + 2 * (2048 + h * 32);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v2f w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = (v2f){ w64[2 * (8 * c + u)], w64[2 * (8 * c + u) + 1] };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int l = 8 * c + u;
            const v2f mine = { ur[l], ui[l] }, other = { lane_swap1(ur[l]), lane_swap1(ui[l]) };
            const v2f sd = __builtin_elementwise_fma(mine, sg2, other);
            v[l] = l ? pk_cmul(sd, w[u]) : sd;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    dft32_pk(v);
}

template <int PK>
__global__ __launch_bounds__(512) void k_test(const float2 *X, const float *tw, float2 *out, long long *cyc, int iters)
{
    __shared__ float scr_all[8][FFT_SCR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q2 = lane >> 1, h = lane & 1;
    lds_float *scr = (lds_float *)&scr_all[wave][0];
    const glb_float *twg = (const glb_float *)tw;
    float2 acc[32];
    for (int k = 0; k < 32; k++) acc[k] = make_float2(0.f, 0.f);
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (PK) {
            v2f v[32];
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) { const float2 x = X[(wave * 2048 + lane + 64 * k2)]; v[k2] = (v2f){ x.x + 1e-3f * it, x.y }; }
            fft2048_wave_pk(v, scr, twg, lane);
#pragma unroll
            for (int k = 0; k < 32; k++) { acc[k].x += v[k].x; acc[k].y += v[k].y; }
        } else {
            float2 v[32];
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) { const float2 x = X[(wave * 2048 + lane + 64 * k2)]; v[k2] = make_float2(x.x + 1e-3f * it, x.y); }
            fft2048_wave(v, scr, twg, lane);
#pragma unroll
            for (int k = 0; k < 32; k++) { acc[k].x += v[k].x; acc[k].y += v[k].y; }
        }
    }
    const long long t1 = clock64();
#pragma unroll
    for (int p = 0; p < 32; p++) out[(size_t)blockIdx.x * 8 * 2048 + wave * 2048 + q2 + 64 * p + 32 * h] = acc[brev5(p)];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    const int nb = 256, iters = 50;
    float2 *X, *o0, *o1; float *tw; long long *cyc;
    hipMalloc(&X, 8 * 2048 * 8); hipMalloc(&tw, (2048 + 64) * 8); hipMalloc(&o0, (size_t)nb * 8 * 2048 * 8); hipMalloc(&o1, (size_t)nb * 8 * 2048 * 8); hipMalloc(&cyc, nb * 8);
    std::vector<float> hx(8 * 2048 * 2), htw((2048 + 64) * 2);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)std::sin(0.37 * i) * 0.5f;
    for (int q = 0; q < 32; q++) for (int l = 0; l < 64; l++) { const double a = -2.0 * M_PI * l * q / 2048.0; htw[(q * 64 + l) * 2] = (float)std::cos(a); htw[(q * 64 + l) * 2 + 1] = (float)std::sin(a); }
    for (int l = 0; l < 32; l++) { htw[(2048 + l) * 2] = 1.f; htw[(2048 + l) * 2 + 1] = 0.f; const double a = -2.0 * M_PI * l / 64.0; htw[(2048 + 32 + l) * 2] = (float)std::cos(a); htw[(2048 + 32 + l) * 2 + 1] = (float)std::sin(a); }
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(tw, htw.data(), htw.size() * 4, hipMemcpyHostToDevice);
    long long c0 = 0, c1 = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_test<0>, dim3(nb), dim3(512), 0, 0, X, tw, o0, cyc, iters); hipDeviceSynchronize(); hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_test<1>, dim3(nb), dim3(512), 0, 0, X, tw, o1, cyc, iters); hipDeviceSynchronize(); hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
    }
    std::vector<float> a(8 * 2048 * 2), b(8 * 2048 * 2);
    hipMemcpy(a.data(), o0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0; for (size_t i = 0; i < a.size(); i++) { md = std::fmax(md, std::fabs((double)a[i] - b[i])); mx = std::fmax(mx, std::fabs((double)a[i])); }
    // reference DFT of wave 0 (it-sum): sum over it of FFT(x + 1e-3 it) -> iters*FFT(x) + 1e-3*sum(it)*FFT(1): check bin 5 only
    printf("scalar %.0f cycles/FFT/wave, packed %.0f cycles/FFT/wave; max |scalar - packed| = %.3g (max |value| %.3g)\n", (double)c0 / iters, (double)c1 / iters, md, mx);
    return 0;
}
