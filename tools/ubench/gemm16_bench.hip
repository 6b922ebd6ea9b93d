// Where does the batched split-binary16 GEMM (k_gemm16) lose its time?  Variants of the wave-per-row-tile kernel on the
// encoder's largest shapes: MODE 0 = as shipped, 1 = A operand not loaded (W stream only), 2 = W not loaded (A stream only),
// 3 = two k-blocks of A fetched together (every 128-byte line of a row touched by back-to-back loads).   (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Args { const float *a1, *a0; const unsigned short *w; float *y; int rows, st, K0, K1, N; long long *clk; };

template <int NT, int RT, int MODE, bool TILED = false, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void k16(Args a)
{
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int r0 = (blockIdx.x * WPB + (threadIdx.x >> 6)) * 32 * RT, ntt = (a.N + 31) >> 5, nt0 = blockIdx.y * NT;
    const float *p1[RT], *p0[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) {
        int r = r0 + 32 * q + (lane & 31);
        if (r >= a.rows) r = a.rows - 1;
        p1[q] = a.a1 + (size_t)r * a.st + 8 * half;
        p0[q] = a.a0 + (size_t)r * a.st + 8 * half;
        if (TILED) {           // x[tile][k/4][32 rows][4]: a k-block of 16 floats = 4 chunks of 512 bytes, lane (row, half) takes chunks 2 half, 2 half + 1
            const int v1 = r + 2, v0 = r;                  // the shifted operand sits two rows earlier
            p1[q] = a.a1 + (size_t)(v1 >> 5) * 32 * a.st + (v1 & 31) * 4 + 2 * half * 128;
            p0[q] = a.a1 + (size_t)(v0 >> 5) * 32 * a.st + (v0 & 31) * 4 + 2 * half * 128;
        }
    }
    f32x16 acc[RT][NT];
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[q][i][j] = 0.0f;
    const int nkb0 = a.K0 >> 4, nkb = nkb0 + (a.K1 >> 4);
    const unsigned short *wbase = a.w + ((size_t)nt0 * 2 * 64 + lane) * 8;
    const size_t wstep = (size_t)ntt * 2 * 64 * 8;
    constexpr int U = MODE == 3 ? 2 : 1;                 // k-blocks per iteration
    f32x4 a4[U][RT][2]; f16x8 bh[U][NT], bl[U][NT];
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
        for (int q = 0; q < RT; q++) { a4[u][q][0] = (f32x4){0.25f, -0.5f, 0.125f, 0.3f} * (float)(lane + 1) * 0.01f; a4[u][q][1] = a4[u][q][0] * 0.5f; }
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) { bh[u][i][j] = (_Float16)(0.01f * (lane + j + i)); bl[u][i][j] = (_Float16)(1e-5f * (lane - j)); }
    }
    auto fetch = [&](int kb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (MODE != 1 && MODE != 5) {
#pragma unroll
                for (int q = 0; q < RT; q++) {
                    const float *p = kb + u < nkb0 ? p0[q] + (kb + u) * (TILED ? 512 : 16) : p1[q] + (kb + u - nkb0) * (TILED ? 512 : 16);
                    a4[u][q][0] = *(const f32x4 *)p; a4[u][q][1] = *(const f32x4 *)(p + (TILED ? 128 : 4));
                }
            }
            if (MODE != 2 && MODE != 5) {
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    bh[u][i] = *(const f16x8 *)(wbase + (kb + u) * wstep + (size_t)i * 2 * 64 * 8);
                    bl[u][i] = *(const f16x8 *)(wbase + (kb + u) * wstep + (size_t)i * 2 * 64 * 8 + 64 * 8);
                }
            }
        }
    };
    fetch(0);
    const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += U) {
        f16x8 ah[U][RT], al[U][RT], ch[U][NT], cl[U][NT];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float x = 256.0f * a4[u][q][j >> 2][j & 3];
                    const _Float16 hi = (_Float16)x;
                    ah[u][q][j] = hi; al[u][q][j] = (_Float16)(x - (float)hi);
                }
#pragma unroll
            for (int i = 0; i < NT; i++) { ch[u][i] = bh[u][i]; cl[u][i] = bl[u][i]; }
        }
        if (kb + U < nkb) fetch(kb + U);
        if (MODE == 5 || MODE == 1) { for (int q = 0; q < RT; q++) { a4[0][q][0][0] += 1e-3f; asm volatile("" : "+v"(a4[0][q][0][1]), "+v"(a4[0][q][1][2])); } }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u][q], ch[u][i], acc[q][i], 0, 0, 0);
                    acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][q], cl[u][i], acc[q][i], 0, 0, 0);
                    acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][q], ch[u][i], acc[q][i], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a.clk && blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0) { a.clk[0] = clock64() - c0; a.clk[1] = wall_clock64() - w0; }
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int col = (nt0 + i) * 32 + (lane & 31);
            if (col >= a.N) continue;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int rr = r0 + 32 * q + (j & 3) + 8 * (j >> 2) + 4 * half;
                if (rr < a.rows) a.y[(size_t)rr * a.N + col] = acc[q][i][j] * 0x1p-18f;
            }
        }
}

// block of WV waves sharing each W k-block through LDS (double buffered), every wave RT row tiles; A fetched two k-blocks at a time
template <int NT, int RT, int WV, bool TILED = false>
__global__ __launch_bounds__(64 * WV) void k16s(Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned short wl[2][2][NT][2][64 * 8];     // [buf][u][nt][plane][lane*8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int r0 = (blockIdx.x * WV + wave) * 32 * RT, ntt = (a.N + 31) >> 5, nt0 = blockIdx.y * NT;
    const float *p1[RT], *p0[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) {
        int r = r0 + 32 * q + (lane & 31);
        if (r >= a.rows) r = a.rows - 1;
        p1[q] = a.a1 + (size_t)r * a.st + 8 * half;
        p0[q] = a.a0 + (size_t)r * a.st + 8 * half;
        if (TILED) {
            const int v1 = r + 2, v0 = r;
            p1[q] = a.a1 + (size_t)(v1 >> 5) * 32 * a.st + (v1 & 31) * 4 + 2 * half * 128;
            p0[q] = a.a1 + (size_t)(v0 >> 5) * 32 * a.st + (v0 & 31) * 4 + 2 * half * 128;
        }
    }
    f32x16 acc[RT][NT];
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[q][i][j] = 0.0f;
    const int nkb0 = a.K0 >> 4, nkb = nkb0 + (a.K1 >> 4);
    const size_t wstep = (size_t)ntt * 2 * 64 * 8;
    // cooperative W copy: 2 k-blocks x NT x 2 planes x 64 lanes of 16 bytes = 2*NT*2*64 chunks over 64*WV threads
    constexpr int CH = 2 * NT * 2 * 64, PER = (CH + 64 * WV - 1) / (64 * WV);
    f32x4 wreg[PER];
    auto wfetch = [&](int kb) {
#pragma unroll
        for (int c = 0; c < PER; c++) {
            const int ch = tid + c * 64 * WV;
            if (ch < CH) {
                const int u = ch / (NT * 128), rem = ch - u * NT * 128;     // rem = (nt*2 + plane)*64 + lane
                wreg[c] = *(const f32x4 *)(a.w + (size_t)(kb + u) * wstep + ((size_t)nt0 * 2 * 64 + rem) * 8);
            }
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < PER; c++) {
            const int ch = tid + c * 64 * WV;
            if (ch < CH) *(f32x4 *)(&wl[buf][0][0][0][0] + (size_t)ch * 8) = wreg[c];
        }
    };
    f32x4 a4[2][RT][2];
    auto afetch = [&](int kb) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int q = 0; q < RT; q++) {
                const float *p = kb + u < nkb0 ? p0[q] + (kb + u) * (TILED ? 512 : 16) : p1[q] + (kb + u - nkb0) * (TILED ? 512 : 16);
                a4[u][q][0] = *(const f32x4 *)p; a4[u][q][1] = *(const f32x4 *)(p + (TILED ? 128 : 4));
            }
    };
    wfetch(0); afetch(0); wstore(0);
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += 2) {
        f16x8 ah[2][RT], al[2][RT];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float x = 256.0f * a4[u][q][j >> 2][j & 3];
                    const _Float16 hi = (_Float16)x;
                    ah[u][q][j] = hi; al[u][q][j] = (_Float16)(x - (float)hi);
                }
        const bool more = kb + 2 < nkb;
        if (more) { wfetch(kb + 2); afetch(kb + 2); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            f16x8 ch[NT], cl[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) { ch[i] = *(const f16x8 *)&wl[buf][u][i][0][lane * 8]; cl[i] = *(const f16x8 *)&wl[buf][u][i][1][lane * 8]; }
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u][q], ch[i], acc[q][i], 0, 0, 0);
                    acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][q], cl[i], acc[q][i], 0, 0, 0);
                    acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u][q], ch[i], acc[q][i], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) wstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int col = (nt0 + i) * 32 + (lane & 31);
            if (col >= a.N) continue;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int rr = r0 + 32 * q + (j & 3) + 8 * (j >> 2) + 4 * half;
                if (rr < a.rows) a.y[(size_t)rr * a.N + col] = acc[q][i][j] * 0x1p-18f;
            }
        }
}


// Software-pipelined variant: one wave, two stages in flight.  During the matrix instructions of k-block s the wave splits
// the activations of k-block s+1 into binary16 planes (VALU in the MFMA shadow) while the loads of W(s+1) and A(s+2) fly.
template <int NT, int RT, bool TILED, int VPG>
__global__ __launch_bounds__(64) void k16p(Args a)
{
    const int lane = threadIdx.x, half = lane >> 5;
    const int r0 = blockIdx.x * 32 * RT, ntt = (a.N + 31) >> 5, nt0 = blockIdx.y * NT;
    const float *p1[RT], *p0[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) {
        int r = r0 + 32 * q + (lane & 31);
        if (r >= a.rows) r = a.rows - 1;
        p1[q] = a.a1 + (size_t)r * a.st + 8 * half;
        p0[q] = a.a0 + (size_t)r * a.st + 8 * half;
        if (TILED) {
            const int v1 = r + 2, v0 = r;
            p1[q] = a.a1 + (size_t)(v1 >> 5) * 32 * a.st + (v1 & 31) * 4 + 2 * half * 128;
            p0[q] = a.a1 + (size_t)(v0 >> 5) * 32 * a.st + (v0 & 31) * 4 + 2 * half * 128;
        }
    }
    f32x16 acc[RT][NT];
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[q][i][j] = 0.0f;
    const int nkb0 = a.K0 >> 4, nkb = nkb0 + (a.K1 >> 4);
    const unsigned short *wbase = a.w + ((size_t)nt0 * 2 * 64 + lane) * 8;
    const size_t wstep = (size_t)ntt * 2 * 64 * 8;
    f32x4 raw[2][RT][2]; f16x8 wh[2][NT], wl[2][NT], ah[2][RT], al[2][RT];
    auto loadA = [&](int kb, int s) {
        kb = min(kb, nkb - 1);
#pragma unroll
        for (int q = 0; q < RT; q++) {
            const float *p = kb < nkb0 ? p0[q] + kb * (TILED ? 512 : 16) : p1[q] + (kb - nkb0) * (TILED ? 512 : 16);
            raw[s][q][0] = *(const f32x4 *)p; raw[s][q][1] = *(const f32x4 *)(p + (TILED ? 128 : 4));
        }
    };
    auto loadW = [&](int kb, int s) {
        kb = min(kb, nkb - 1);
#pragma unroll
        for (int i = 0; i < NT; i++) { wh[s][i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * 2 * 64 * 8); wl[s][i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * 2 * 64 * 8 + 64 * 8); }
    };
    auto convert = [&](int s) {
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = 256.0f * raw[s][q][j >> 2][j & 3];
                const _Float16 hi = (_Float16)x;
                ah[s][q][j] = hi; al[s][q][j] = (_Float16)(x - (float)hi);
            }
    };
    auto mfmas = [&](int s) {
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][q], wh[s][i], acc[q][i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][q], wl[s][i], acc[q][i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][q], wh[s][i], acc[q][i], 0, 0, 0);
    };
    loadW(0, 0); loadA(0, 0); loadA(1, 1);
    convert(0);
#define STAGE(KB, CUR, NXT) \
    loadW((KB) + 1, NXT); loadA((KB) + 2, CUR); \
    __builtin_amdgcn_sched_barrier(0); \
    convert(NXT); mfmas(CUR); \
    _Pragma("unroll") for (int g = 0; g < RT * NT * 3; g++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0); } \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += 2) {
        STAGE(kb, 0, 1)
        STAGE(kb + 1, 1, 0)
    }
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int col = (nt0 + i) * 32 + (lane & 31);
            if (col >= a.N) continue;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int rr = r0 + 32 * q + (j & 3) + 8 * (j >> 2) + 4 * half;
                if (rr < a.rows) a.y[(size_t)rr * a.N + col] = acc[q][i][j] * 0x1p-18f;
            }
        }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; std::memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }

template <typename F> static float timeit(F launch, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

static void shape(const char *name, int rows, int K0, int K1, int N)
{
    const int st = 864, K = K0 + K1, ntt = (N + 31) / 32, nkb = K / 16;
    std::vector<float> A((size_t)(rows + 2) * st), W((size_t)N * K);
    srand(1);
    for (auto &v : A) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto &v : W) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
    std::vector<unsigned short> P((size_t)nkb * ntt * 2 * 64 * 8);
    for (int kb = 0; kb < nkb; kb++) for (int nt = 0; nt < ntt; nt++) for (int lane = 0; lane < 64; lane++) for (int j = 0; j < 8; j++) {
        const int nn = nt * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
        const float w = nn < N ? 1024.0f * W[(size_t)nn * K + k] : 0.0f;
        const unsigned short hi = f2h(w), lo = f2h(w - h2f(hi));
        unsigned short *o = &P[((((size_t)kb * ntt + nt) * 2) * 64 + lane) * 8 + j];
        o[0] = hi; o[64 * 8] = lo;
    }
    float *dA, *dY, *dY2; unsigned short *dW;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, P.size() * 2); hipMalloc(&dY, (size_t)rows * N * 4); hipMalloc(&dY2, (size_t)rows * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, P.data(), P.size() * 2, hipMemcpyHostToDevice);
    long long *dclk; hipMalloc(&dclk, 16); hipMemset(dclk, 0, 16);
    Args a = { dA + 2 * st, dA, dW, dY, rows, st, K0, K1, N, dclk };
    Args a2 = a; a2.y = dY2;
    // tiled copy of A: virtual row v = r + 2 (two history rows in front), x[v/32][k/4][v%32][k%4]
    std::vector<float> At((size_t)((rows + 2 + 31) / 32) * 32 * st);
    for (int v = 0; v < rows + 2; v++) for (int k = 0; k < st; k++) At[(size_t)(v >> 5) * 32 * st + (size_t)(k >> 2) * 128 + (v & 31) * 4 + (k & 3)] = A[(size_t)v * st + k];
    float *dAt; hipMalloc(&dAt, At.size() * 4); hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice);
    Args at = a2; at.a1 = dAt; at.a0 = dAt;
    const int NTv = ntt % 3 == 0 ? 3 : 2;
    const double mfma_floor_ms = (double)rows * N * K * 3 / (256.0 * 2048 * 2.4e9) * 1e3;
    printf("%s rows %d K %d+%d N %d   (matrix-core floor %.3f ms)\n", name, rows, K0, K1, N, mfma_floor_ms);
    auto cmp = [&]() {
        std::vector<float> y((size_t)rows * N), y2((size_t)rows * N);
        hipMemcpy(y.data(), dY, y.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(y2.data(), dY2, y2.size() * 4, hipMemcpyDeviceToHost);
        double m = 0; for (size_t i = 0; i < y.size(); i++) m = fmax(m, fabs((double)y[i] - y2[i]));
        return m;
    };
    if (NTv == 3) {
        dim3 g((rows + 63) / 64, ntt / 3);
        printf("  shipped <3,2>           %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 0>), g, dim3(64), 0, 0, a); }, 20));
        printf("  no loads at all         %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 5>), g, dim3(64), 0, 0, a2); }, 20));
        { long long c[2]; hipMemcpy(c, dclk, 16, hipMemcpyDeviceToHost); printf("     k loop of one wave: %lld shader cycles, %lld ticks of the 100 MHz clock -> %.2f GHz, %.0f cycles per k-block\n", c[0], c[1], c[0] / (c[1] * 10.0), (double)c[0] / nkb); }
        printf("  no loads, 4 waves/WG    %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 5, false, 4>), dim3((rows + 255) / 256, ntt / 3), dim3(256), 0, 0, a2); }, 20));
        printf("  shipped, 4 waves/WG     %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 0, false, 4>), dim3((rows + 255) / 256, ntt / 3), dim3(256), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A, 4 waves/WG     %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 0, true, 4>), dim3((rows + 255) / 256, ntt / 3), dim3(256), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  shipped, 2 waves/WG     %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 0, false, 2>), dim3((rows + 127) / 128, ntt / 3), dim3(128), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  W stream only           %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 1>), g, dim3(64), 0, 0, a2); }, 20));
        printf("  A stream only           %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 2>), g, dim3(64), 0, 0, a2); }, 20));
        printf("  two k-blocks per fetch  %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 3>), g, dim3(64), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  <3,1>                   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 1, 0>), dim3((rows + 31) / 32, ntt / 3), dim3(64), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  LDS W, 4 waves x RT2    %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 2, 4>), dim3((rows + 255) / 256, ntt / 3), dim3(256), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  LDS W, 8 waves x RT1    %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 1, 8>), dim3((rows + 255) / 256, ntt / 3), dim3(512), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  LDS W, 4 waves x RT1    %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 1, 4>), dim3((rows + 127) / 128, ntt / 3), dim3(256), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A <3,2>           %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 2, 0, true>), g, dim3(64), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A <3,1>           %.4f ms", timeit([&] { hipLaunchKernelGGL((k16<3, 1, 0, true>), dim3((rows + 31) / 32, ntt / 3), dim3(64), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A, LDS W 4w RT2   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 2, 4, true>), dim3((rows + 255) / 256, ntt / 3), dim3(256), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A, LDS W 8w RT1   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 1, 8, true>), dim3((rows + 255) / 256, ntt / 3), dim3(512), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A, LDS W 4w RT1   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 1, 4, true>), dim3((rows + 127) / 128, ntt / 3), dim3(256), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  tiled A, LDS W 8w RT2   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 2, 8, true>), dim3((rows + 511) / 512, ntt / 3), dim3(512), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  pipelined <3,2> vpg4    %.4f ms", timeit([&] { hipLaunchKernelGGL((k16p<3, 2, false, 4>), g, dim3(64), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  pipelined <3,2> vpg5    %.4f ms", timeit([&] { hipLaunchKernelGGL((k16p<3, 2, false, 5>), g, dim3(64), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  pipelined tiled <3,2>   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16p<3, 2, true, 4>), g, dim3(64), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  pipelined tiled <3,1>   %.4f ms", timeit([&] { hipLaunchKernelGGL((k16p<3, 1, true, 7>), dim3((rows + 31) / 32, ntt / 3), dim3(64), 0, 0, at); }, 20)); printf("   maxdiff %.3g\n", cmp());
        printf("  LDS W, 2 waves x RT2    %.4f ms", timeit([&] { hipLaunchKernelGGL((k16s<3, 2, 2>), dim3((rows + 127) / 128, ntt / 3), dim3(128), 0, 0, a2); }, 20)); printf("   maxdiff %.3g\n", cmp());
    }
    hipFree(dAt); hipFree(dA); hipFree(dW); hipFree(dY); hipFree(dY2);
}

int main()
{
    shape("conv4", 64512, 768, 768, 96);
    shape("gin4 ", 64512, 0, 704, 192);
    shape("conv1", 64512, 288, 288, 96);
    shape("gin1 ", 64512, 0, 224, 192);
    return 0;
}
