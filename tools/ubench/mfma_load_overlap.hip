// developer aid (microbenchmark, gfx950): do a wavefront's vector loads and its matrix instructions overlap?  The encoder GEMM's knock-out builds say its loads (0.15 ms),
// matrix instructions (0.135 ms) and stores ADD UP to its time; this kernel has the same loop shape -- per k-block 2 activation loads of 1 KB (private, HBM / Infinity Cache) +
// 3 weight loads of 1 KB (shared, L2), 6 x v_mfma_f32_32x32x16_f16, operands in a ring of ST k-blocks -- and runs it with the loads only, the matrix instructions only, or both.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_load_overlap mfma_load_overlap.hip ; run: ./mfma_load_overlap [waves_per_simd] [k-blocks] [distinct private streams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int ST>      // MODE bit 0: loads, bit 1: matrix instructions
__global__ __launch_bounds__(64) void k(const _Float16 *x, const _Float16 *w, float *out, int nkb, int wrap)
{
    const int lane = threadIdx.x;
    const _Float16 *px = x + (size_t)(blockIdx.x % wrap) * nkb * 1024 + lane * 8;   // this wavefront's private stream: 2 KB per k-block (wrap < gridDim.x: wavefronts share streams, the footprint shrinks into L2)
    const _Float16 *pw = w + lane * 8;                                         // shared by everybody: 3 KB per k-block
    f32x16 acc[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    f16x8 xh[ST], xl[ST], wh[ST][3];
    for (int s = 0; s < ST; s++) { xh[s] = (f16x8)(_Float16)1.0f; xl[s] = xh[s]; for (int i = 0; i < 3; i++) wh[s][i] = xh[s]; }
    auto fetch = [&](int s, int kb_) {
        const int kb = min(kb_, nkb - 1);
        if (MODE & 1) {
            xh[s] = *(const f16x8 *)(px + (size_t)kb * 1024); xl[s] = *(const f16x8 *)(px + (size_t)kb * 1024 + 512);
            for (int i = 0; i < 3; i++) wh[s][i] = *(const f16x8 *)(pw + (size_t)kb * 1536 + i * 512);
        }
    };
    auto products = [&](int s) {
        if (MODE & 2) {
            for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][i], xl[s], acc[i], 0, 0, 0);
            for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][i], xh[s], acc[i], 0, 0, 0);
        } else {   // keep the loaded registers alive without instructions
            asm volatile("" :: "v"(xh[s]), "v"(xl[s]), "v"(wh[s][0]), "v"(wh[s][1]), "v"(wh[s][2]));
        }
    };
#pragma unroll
    for (int s = 0; s < ST; s++) fetch(s, s);
    int kb = 0;
#pragma unroll 1
    for (; kb + ST <= nkb; kb += ST) {
#pragma unroll
        for (int s = 0; s < ST; s++) {
            __builtin_amdgcn_sched_barrier(0);
            products(s);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s, kb + ST + s);
        }
    }
    float v = 0.0f;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 16; j++) v += acc[i][j];
    if (v == 12345.0f) out[blockIdx.x * 64 + lane] = v;
}
template <int MODE, int ST> float run(const _Float16 *x, const _Float16 *w, float *out, int waves, int nkb, int reps, int wrap)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, ST>), dim3(waves), dim3(64), 0, 0, x, w, out, nkb, wrap);
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<MODE, ST>), dim3(waves), dim3(64), 0, 0, x, w, out, nkb, wrap);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 2, nkb = argc > 2 ? atoi(argv[2]) : 48, waves = 1024 * wps, reps = 20, wrap = argc > 3 ? atoi(argv[3]) : 1024 * wps;
    _Float16 *x, *w; float *out;
    const size_t xb = (size_t)waves * nkb * 2048, wb = (size_t)nkb * 3072;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&out, (size_t)waves * 256);
    hipMemset(x, 0, xb); hipMemset(w, 0, wb);
    printf("%d wavefronts (%d per SIMD), %d k-blocks each; private stream %.0f MB (footprint %.0f MB), shared weights %.0f KB; per launch, us:\n", waves, wps, nkb, xb / 1e6, (double)wrap * nkb * 2048 / 1e6, wb / 1e3);
    printf("  3 stages: loads only %.1f   matrix instructions only %.1f   both %.1f\n", run<1, 3>(x, w, out, waves, nkb, reps, wrap), run<2, 3>(x, w, out, waves, nkb, reps, wrap), run<3, 3>(x, w, out, waves, nkb, reps, wrap));
    printf("  6 stages: loads only %.1f   matrix instructions only %.1f   both %.1f\n", run<1, 6>(x, w, out, waves, nkb, reps, wrap), run<2, 6>(x, w, out, waves, nkb, reps, wrap), run<3, 6>(x, w, out, waves, nkb, reps, wrap));
    const double flop = (double)waves * nkb * 6 * 32 * 32 * 16 * 2;
    printf("  (matrix work %.1f GFLOP per launch: %.0f us at 2.5 PFLOP/s; bytes through the L1s %.0f MB)\n", flop / 1e9, flop / 2.5e15 * 1e6, (double)waves * nkb * 5120 / 1e6);
    return 0;
}
