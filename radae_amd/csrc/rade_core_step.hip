// rade_core_step.hip -- one 40 ms step of CoreEncoderStatefull / CoreDecoderStatefull for ONE stream as ONE kernel launch
// (the call granularity of /root/reference/src/rade_enc.c:55-114 and rade_dec.c:50-102; radae_base.py:260-286, :400-416).
//
// A single stream is a chain of dependent mat-vec stages with M = 1: there is nothing to batch, so the layer-wise GEMM engine (one
// launch per layer, rade_engine.c:encode_core) spends its time in launch gaps.  Here one workgroup of 8 wavefronts (256 registers per lane: a whole stage of weight fragments in flight) walks the whole
// stack in one launch:
//  * every activation lives in LDS; GRU / conv state persists in HBM between calls; input and output are device-visible host buffers;
//  * weights stream once per step from L2, CHUNK-MAJOR: W[c][n][8] (chunk c = 8 consecutive k, rows padded to a multiple of 64), the int8
//    layers of the blob as ONE binary16 plane of the integers q plus a row scale (exact, half the bytes of float32).  Lane = output row,
//    wavefront w takes chunks w, w + 8, ..: a wave-load is 1 KB contiguous, every byte used, the x chunk is a wave-uniform LDS
//    broadcast, and no cross-lane reduction exists -- the 8 per-wave partial sums of a row meet in LDS and the row's own thread adds
//    them in wave order (deterministic), applies scale / bias / activation and writes the activation;
//  * the chain is latency-bound (one L2 round trip per stage on a GPU whose clocks idle low between single-stream calls), so a stage's
//    weight loads are ALL issued before the barriers that finish the stage before it: the layer shapes are template parameters, the
//    fragments of the next stage sit in registers while this stage's partial sums are combined.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include "rade_devutil.h"

#define CS_THREADS 512
#define CS_WAVES (CS_THREADS / 64)
#define CS_PS 320                 // row stride of a partial-sum area: 288 rows padded to 5 x 64
#define CS_WMAX 864

// weights are read through pointers that k_tx_frame / k_tx_frame3 load from a record in device memory: generic pointers, i.e. flat loads, which count on
// the LDS wait counter too -- every LDS wait of a stage then also waited for the weight prefetch of the next.  They ARE global memory: say so.
template <typename T> __device__ __forceinline__ T cs_gload(const T *p) { return *(const __attribute__((address_space(1))) T *)p; }
__device__ __forceinline__ float cs_clamp1(float x) { return fminf(fmaxf(x, -1.0f), 1.0f); }
__device__ __forceinline__ float cs_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct CsShared {
    __attribute__((aligned(16))) float x[CS_WMAX];            // the DenseNet concat row of this step
    __attribute__((aligned(16))) float hist[2][CS_WMAX];      // rows t-1, t-2 (conv taps, dilation 1 or 2)
    __attribute__((aligned(16))) float vin[96];               // input vector, zero padded to dense1's K
    __attribute__((aligned(16))) float h[5][96];              // GRU states at entry
    __attribute__((aligned(16))) float hc[96];                // clamp(h') of the layer at hand (decoder GLU operand)
    __attribute__((aligned(16))) float gh[288];               // W_hh h + b_hh of the layer at hand
    __attribute__((aligned(16))) float pa[CS_WAVES][CS_PS];   // per-wave partial sums, product A of a stage
    __attribute__((aligned(16))) float pb[CS_WAVES][CS_PS];   // product B (the next layer's W_hh h, riding along)
};

// fragments of one binary16 product: NRB row blocks of 64 x NCI chunks per wavefront
template <int NRB, int NCI> struct WQ { f16x8 w[NRB][NCI]; };
template <int NRB, int NCI>
__device__ __forceinline__ void cs_issue(const rd_mv &L, WQ<NRB, NCI> &r)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = L.K >> 3;
#pragma unroll
    for (int i = 0; i < NCI; i++) {
        const int c = min(wave + CS_WAVES * i, nch - 1);                      // chunks past the end re-read the last one (times zero below)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) r.w[rb][i] = cs_gload((const f16x8 *)(L.wq + ((size_t)c * (NRB * 64) + rb * 64 + lane) * 8));
    }
}
// partial sums of this wavefront's chunks into part[wave][row]; v = [v0[0..K0) | v1[..]] in LDS
template <int NRB, int NCI>
__device__ __forceinline__ void cs_consume(const rd_mv &L, const WQ<NRB, NCI> &r, const float *v0, int K0, const float *v1, float (*part)[CS_PS])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = L.K >> 3;
    float acc[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; rb++) acc[rb] = 0.0f;
#pragma unroll
    for (int i = 0; i < NCI; i++) {
        const int c = wave + CS_WAVES * i, k = 8 * min(c, nch - 1);
        const float *src = k < K0 ? v0 + k : v1 + (k - K0);
        f32x4 a = *(const f32x4 *)src, b = *(const f32x4 *)(src + 4);        // wave-uniform address: an LDS broadcast
        if (c >= nch) { a = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; b = a; }
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) {
            float s0 = acc[rb], s1 = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                s0 = fmaf((float)r.w[rb][i][j], a[j], s0); s1 = fmaf((float)r.w[rb][i][j + 1], a[j + 1], s1);
                s0 = fmaf((float)r.w[rb][i][4 + j], b[j], s0); s1 = fmaf((float)r.w[rb][i][4 + j + 1], b[j + 1], s1);
            }
            acc[rb] = s0 + s1;
        }
    }
#pragma unroll
    for (int rb = 0; rb < NRB; rb++) part[wave][rb * 64 + lane] = acc[rb];
}
// row n of a product: the per-wave partials in wave order, times the row scale, plus bias
__device__ __forceinline__ float cs_row(const rd_mv &L, const float (*part)[CS_PS], int n)
{
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; w++) s += part[w][n];
    return s * (L.scale ? cs_gload(L.scale + n) : 1.0f) + (L.bias ? cs_gload(L.bias + n) : 0.0f);
}

// float32 layers (dense1, the output layer: raw features / received symbols are unbounded, and the blob holds these as floats): the same
// chunk-major scheme with two 16-byte loads per chunk, one row block at a time
template <int NCI>
__device__ __forceinline__ void cs_f32_product(const rd_mv &L, int nrb, const float *v, float (*part)[CS_PS])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = L.K >> 3;
    for (int rb = 0; rb < nrb; rb++) {
        f32x4 w[NCI][2];
#pragma unroll
        for (int i = 0; i < NCI; i++) {
            const int c = min(wave + CS_WAVES * i, nch - 1);
            const float *p = L.wf + ((size_t)c * (nrb * 64) + rb * 64 + lane) * 8;
            w[i][0] = cs_gload((const f32x4 *)p); w[i][1] = cs_gload((const f32x4 *)(p + 4));
        }
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < NCI; i++) {
            const int c = wave + CS_WAVES * i, k = 8 * min(c, nch - 1);
            f32x4 a = *(const f32x4 *)(v + k), b = *(const f32x4 *)(v + k + 4);
            if (c >= nch) { a = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; b = a; }
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                s0 = fmaf(w[i][0][j], a[j], s0); s1 = fmaf(w[i][0][j + 1], a[j + 1], s1);
                s0 = fmaf(w[i][1][j], b[j], s0); s1 = fmaf(w[i][1][j + 1], b[j + 1], s1);
            }
        }
        part[wave][rb * 64 + lane] = s0 + s1;
    }
}

// GRU gates of hidden unit j (torch order r, z, n; radae_base.py:97-108) from the input projection's partial sums and gh
__device__ __forceinline__ float cs_gru_unit(const rd_mv &G, const float (*part)[CS_PS], const float *gh, float hj, int H, int j)
{
    const float gr = cs_row(G, part, j), gz = cs_row(G, part, H + j), gn = cs_row(G, part, 2 * H + j);
    const float r = cs_sigmoid(gh[j] + gr);
    const float z = cs_sigmoid(gh[H + j] + gz);
    const float nn = tanhf(gn + gh[2 * H + j] * r);
    return (hj - nn) * z + nn;
}

#define CS_SYNC() __syncthreads()

// ---- encoder: dense1 | 5 x (GRU 64, conv 96, dilation 1,2,2,2,2) | z_dense ---------------------------------------------------
// one layer; the fragments of its input projection (gq) were issued by the stage before, the next layer's (gnext) are issued here
template <int GCI, int CCI, int GNEXT, bool LAST>
__device__ __forceinline__ void cs_enc_layer(CsShared *sh, const rd_core_args &a, int l, int n, WQ<3, GCI> &gq, WQ<3, GNEXT> &gnext)
{
    const int tid = threadIdx.x, H = 64;
    WQ<2, CCI> cq; WQ<3, 1> hq;
    cs_consume<3, GCI>(a.gin[l], gq, sh->x, n, sh->x, sh->pa);
    cs_issue<2, CCI>(a.conv[l], cq);                               // this layer's conv and the next layer's W_hh h: in flight across the gate stage
    if (!LAST) cs_issue<3, 1>(a.ghh[l + 1], hq);
    CS_SYNC();
    if (tid < H) {
        const float hn = cs_gru_unit(a.gin[l], sh->pa, sh->gh, sh->h[l][tid], H, tid);
        a.h[l * H + tid] = hn;
        sh->h[l][tid] = hn;                                          // (this layer's W_hh h was formed a stage ago: nobody reads the old value any more)
        sh->x[n + tid] = cs_clamp1(hn);
    }
    CS_SYNC();
    const int cin = n + H;
    cs_consume<2, CCI>(a.conv[l], cq, sh->hist[a.dil[l] - 1], cin, sh->x, sh->pa);
    if (!LAST) { cs_consume<3, 1>(a.ghh[l + 1], hq, sh->h[l + 1], H, sh->h[l + 1], sh->pb); cs_issue<3, GNEXT>(a.gin[l + 1], gnext); }
    CS_SYNC();
    if (tid < 96) sh->x[cin + tid] = cs_clamp1(tanhf(cs_row(a.conv[l], sh->pa, tid)));       // Conv1d k=2 + tanh (radae_base.py:110-134)
    else if (!LAST && tid >= 128 && tid < 128 + 3 * H) sh->gh[tid - 128] = cs_row(a.ghh[l + 1], sh->pb, tid - 128);
    CS_SYNC();
}

// one encoder step; FIRST: conv history and GRU states come from HBM (else they are in LDS, left there by the step before); LASTSTEP: they go back.
// in: the step's n_in inputs; zout: its 80 latents (any address space the workgroup can write)
__device__ __forceinline__ void cs_enc_step(CsShared *sh, const rd_core_args &a, const float *in, float *zout, const bool FIRST, const bool LASTSTEP)
{
    const int tid = threadIdx.x, W = 864, H = 64;
    WQ<3, 1> g0, h0;
    cs_issue<3, 1>(a.gin[0], g0); cs_issue<3, 1>(a.ghh[0], h0);
    if (FIRST) {
        for (int i = tid; i < 2 * W; i += CS_THREADS) sh->hist[i / W][i % W] = a.hist[i];
        if (tid < 5 * H) sh->h[tid / H][tid % H] = a.h[tid];
    }
    if (tid < 96) sh->vin[tid] = tid < a.n_in ? in[tid] : 0.0f;
    CS_SYNC();
    cs_f32_product<2>(a.dense1, 1, sh->vin, sh->pa);                // dense1 + tanh (radae_base.py:263)
    cs_consume<3, 1>(a.ghh[0], h0, sh->h[0], H, sh->h[0], sh->pb);
    CS_SYNC();
    if (tid < 64) sh->x[tid] = cs_clamp1(tanhf(cs_row(a.dense1, sh->pa, tid)));
    else if (tid >= 128 && tid < 128 + 3 * H) sh->gh[tid - 128] = cs_row(a.ghh[0], sh->pb, tid - 128);
    CS_SYNC();
    WQ<3, 4> g1; WQ<3, 6> g2; WQ<3, 9> g3; WQ<3, 11> g4;
    cs_enc_layer<1, 4, 4, false>(sh, a, 0, 64, g0, g1);
    cs_enc_layer<4, 9, 6, false>(sh, a, 1, 224, g1, g2);
    cs_enc_layer<6, 14, 9, false>(sh, a, 2, 384, g2, g3);
    cs_enc_layer<9, 19, 11, false>(sh, a, 3, 544, g3, g4);
    cs_enc_layer<11, 24, 11, true>(sh, a, 4, 704, g4, g4);
    cs_f32_product<14>(a.out, 2, sh->x, sh->pa);                    // z_dense, linear (bottleneck 3; the tanh of bottleneck 1 is the caller's)
    if (LASTSTEP) { for (int i = tid; i < W; i += CS_THREADS) { a.hist[W + i] = sh->hist[0][i]; a.hist[i] = sh->x[i]; } }      // history of the next step
    CS_SYNC();
    if (tid < a.n_out) zout[tid] = cs_row(a.out, sh->pa, tid);
    if (!LASTSTEP) { for (int i = tid; i < W; i += CS_THREADS) { const float v = sh->hist[0][i]; sh->hist[1][i] = v; sh->hist[0][i] = sh->x[i]; } }   // (each thread moves its own columns)
    CS_SYNC();
}

// the same step as a REAL function for k_tx_frame's loop (one copy of the code with its own register allocation; inlined into the loop the
// three steps' weight prefetches overlapped and 400 registers spilled).  The layer table is read through a pointer to device memory: a by-value
// struct handed to a real function would live in scratch memory.
__device__ __attribute__((noinline)) void cs_enc_step_fn(CsShared *sh, const rd_core_args *ap, int st, float *zs)
{
    cs_enc_step(sh, *ap, ap->in + st * ap->n_in, zs + st * RD_LATENT, st == 0, st == 2);
}

__global__ __launch_bounds__(CS_THREADS) void k_core_enc_step(rd_core_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_raw[];
    CsShared *sh = (CsShared *)cs_raw;
    const int tid = threadIdx.x;
    cs_enc_step(sh, a, a.in, a.out_vec, true, true);
    if (tid < a.n_out) __threadfence_system();
    CS_SYNC();
    // completion word in the caller's (pinned host) memory: the host polls it instead of going through a stream synchronisation,
    // whose interrupt / wake-up path costs more than this whole kernel
    if (tid == 0 && a.done) { __threadfence_system(); *(volatile unsigned *)a.done = a.seq; }
}

// ---- decoder: dense1 | 5 x (GRU 96, GLU 96, conv 32) | output ---------------------------------------------------------------------
template <int GCI, int CCI, int GNEXT, bool LAST>
__device__ __forceinline__ void cs_dec_layer(CsShared *sh, const rd_core_args &a, int l, int n, WQ<5, GCI> &gq, WQ<5, GNEXT> &gnext)
{
    const int tid = threadIdx.x, H = 96;
    WQ<2, 2> uq; WQ<1, CCI> cq; WQ<5, 2> hq;
    cs_consume<5, GCI>(a.gin[l], gq, sh->x, n, sh->x, sh->pa);
    cs_issue<2, 2>(a.glu[l], uq); cs_issue<1, CCI>(a.conv[l], cq);
    CS_SYNC();
    if (tid < H) {
        const float hn = cs_gru_unit(a.gin[l], sh->pa, sh->gh, sh->h[l][tid], H, tid);
        a.h[l * H + tid] = hn;
        sh->hc[tid] = cs_clamp1(hn);
    }
    CS_SYNC();
    cs_consume<2, 2>(a.glu[l], uq, sh->hc, H, sh->hc, sh->pa);      // GLU: x * sigmoid(W x), no bias (radae_base.py:149-153)
    if (!LAST) cs_issue<5, 2>(a.ghh[l + 1], hq);
    CS_SYNC();
    if (tid < H) sh->x[n + tid] = cs_clamp1(sh->hc[tid] * cs_sigmoid(cs_row(a.glu[l], sh->pa, tid)));
    CS_SYNC();
    const int cin = n + H;
    cs_consume<1, CCI>(a.conv[l], cq, sh->hist[0], cin, sh->x, sh->pa);
    if (!LAST) { cs_consume<5, 2>(a.ghh[l + 1], hq, sh->h[l + 1], H, sh->h[l + 1], sh->pb); cs_issue<5, GNEXT>(a.gin[l + 1], gnext); }
    CS_SYNC();
    if (tid < 32) sh->x[cin + tid] = cs_clamp1(tanhf(cs_row(a.conv[l], sh->pa, tid)));
    else if (!LAST && tid >= 64 && tid < 64 + 3 * H) sh->gh[tid - 64] = cs_row(a.ghh[l + 1], sh->pb, tid - 64);
    CS_SYNC();
}

__global__ __launch_bounds__(CS_THREADS) void k_core_dec_step(rd_core_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_raw[];
    CsShared *sh = (CsShared *)cs_raw;
    const int tid = threadIdx.x, W = 736, H = 96;
    WQ<5, 2> g0, h0;
    cs_issue<5, 2>(a.gin[0], g0); cs_issue<5, 2>(a.ghh[0], h0);
    for (int i = tid; i < W; i += CS_THREADS) sh->hist[0][i] = a.hist[i];
    if (tid < 96) sh->vin[tid] = tid < a.n_in ? a.in[tid] : 0.0f;
    if (tid < 5 * H) sh->h[tid / H][tid % H] = a.h[tid];
    CS_SYNC();
    cs_f32_product<2>(a.dense1, 2, sh->vin, sh->pa);                // dense1 + tanh (radae_base.py:403)
    cs_consume<5, 2>(a.ghh[0], h0, sh->h[0], H, sh->h[0], sh->pb);
    CS_SYNC();
    if (tid < 96) sh->x[tid] = cs_clamp1(tanhf(cs_row(a.dense1, sh->pa, tid)));
    else if (tid >= 128 && tid < 128 + 3 * H) sh->gh[tid - 128] = cs_row(a.ghh[0], sh->pb, tid - 128);
    CS_SYNC();
    WQ<5, 4> g1; WQ<5, 6> g2; WQ<5, 8> g3; WQ<5, 10> g4;
    cs_dec_layer<2, 6, 4, false>(sh, a, 0, 96, g0, g1);
    cs_dec_layer<4, 10, 6, false>(sh, a, 1, 224, g1, g2);
    cs_dec_layer<6, 14, 8, false>(sh, a, 2, 352, g2, g3);
    cs_dec_layer<8, 18, 10, false>(sh, a, 3, 480, g3, g4);
    cs_dec_layer<10, 22, 10, true>(sh, a, 4, 608, g4, g4);
    cs_f32_product<12>(a.out, 2, sh->x, sh->pa);                    // output layer, linear
    for (int i = tid; i < W; i += CS_THREADS) a.hist[i] = sh->x[i];
    CS_SYNC();
    if (tid < a.n_out) { a.out_vec[tid] = cs_row(a.out, sh->pa, tid); __threadfence_system(); }
    CS_SYNC();
    // completion word in the caller's (pinned host) memory: the host polls it instead of going through a stream synchronisation,
    // whose interrupt / wake-up path costs more than this whole kernel
    if (tid == 0 && a.done) { __threadfence_system(); *(volatile unsigned *)a.done = a.seq; }
}

// ---- rade_tx() as ONE launch: the three encoder steps of a modem frame (state in LDS between them) and the OFDM modulator (dsp.py:340-378:
// pilot row + four data rows, 30 -> 160 IDFT, cyclic prefix, tanh limiter) in the same workgroup; a.in = 3 x 84 packed features, a.iq_out = 960
// complex samples (both pinned host memory the kernel reads / writes directly), then the completion word.  (The reference's rade_tx runs the
// same sequence through CPython: rade_api.c:403-445, radae_txe.py:108-135.)
__global__ __launch_bounds__(CS_THREADS) void k_tx_frame(const rd_core_args *ap, unsigned seq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_raw[];
    CsShared *sh = (CsShared *)cs_raw;
    __shared__ float zs[RD_ZMF];
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int st = 0; st < 3; st++) cs_enc_step_fn(sh, ap, st, zs);
    const rd_tables *tab = ap->tab;
    float2 *out = (float2 *)ap->iq_out;
    if (tid < RD_M) {
        f32x2 acc[RD_NS + 1];
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) acc[s] = (f32x2){ 0.0f, 0.0f };
#pragma unroll 6
        for (int c = 0; c < RD_NC; c++) {
            const float2 w = ld2(tab->Winv[c], tid);
            acc[0] = idft_term(acc[0], make_float2(tab->P[c] * tab->pilot_gain, 0.0f * tab->pilot_gain), w);
#pragma unroll
            for (int s = 1; s <= RD_NS; s++) { const int k = (s - 1) * RD_NC + c; acc[s] = idft_term(acc[s], make_float2(zs[2 * k], zs[2 * k + 1]), w); }
        }
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) {
            const float2 v = pa_limit(make_float2(acc[s][0], acc[s][1]));                  // tanh(|x|) e^{j angle(x)} (radae.py:218, dsp.py:377)
            out[s * RD_SYM + RD_NCP + tid] = v;
            if (tid >= RD_M - RD_NCP) out[s * RD_SYM + tid - (RD_M - RD_NCP)] = v;
        }
        __threadfence_system();
    }
    CS_SYNC();
    if (tid == 0 && ap->done) { __threadfence_system(); *(volatile unsigned *)ap->done = seq; }
}
// ---- rade_tx() with the frame's three encoder steps taken through every layer TOGETHER -----------------------------------------------------------
// k_tx_frame above runs the step three times: 3 x 17 dependent stages, each worth an L2 round trip for its weights.  Here a layer's weight
// fragments are fetched ONCE and applied to the three rows of the frame (the feed-forward pieces: dense1, the five input projections, the five
// convs, z_dense -- 12 weight stages per frame instead of 51); only the W_hh recurrences stay serial, three short steps per layer on fragments
// that are already in registers.  Same chunk -> wavefront assignment and the same order of partial sums per row as the step kernel, so the
// latents are the step kernel's, bit for bit.
struct CsShared3 {
    __attribute__((aligned(16))) float x[3][CS_WMAX];         // the DenseNet concat rows of the frame's three steps
    __attribute__((aligned(16))) float hist[2][CS_WMAX];      // rows -1, -2 (the previous frame's last two)
    __attribute__((aligned(16))) float vin[3][96];
    __attribute__((aligned(16))) float h[5][96];              // GRU states, advanced step by step
    __attribute__((aligned(16))) float gh[288];
    __attribute__((aligned(16))) float pa[CS_WAVES][3][CS_PS];   // per-wave partial sums of a product, per row
    __attribute__((aligned(16))) float pb[CS_WAVES][CS_PS];      // W_hh h of one step
    __attribute__((aligned(16))) float zs[RD_ZMF];
};
struct Rows3 { const float *r[3]; };
// cs_issue / cs_consume for three input rows sharing the fragments, in RANGES of a wavefront's chunks [I0, I1): a product's fragments need not all be in
// registers at once (conv 5 has 24 chunks per wavefront = 192 registers; with the three rows' operands beside them the allocator spilled its working
// set hundreds of times).  acc[row][row block] accumulates over the ranges in chunk order: the same sums as the step kernel, bit for bit.
template <int NRB, int NCI, int I0, int I1>
__device__ __forceinline__ void cs_issue_range(const rd_mv &L, WQ<NRB, NCI> &r)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = L.K >> 3;
#pragma unroll
    for (int i = I0; i < I1; i++) {
        const int c = min(wave + CS_WAVES * i, nch - 1);                      // chunks past the end re-read the last one (times zero below)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) r.w[rb][i] = cs_gload((const f16x8 *)(L.wq + ((size_t)c * (NRB * 64) + rb * 64 + lane) * 8));
    }
}
template <int NRB, int NCI, int I0, int I1>
__device__ __forceinline__ void cs_acc3_range(const rd_mv &L, const WQ<NRB, NCI> &r, const Rows3 &v0, int K0, const Rows3 &v1, float (&acc)[3][NRB])
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nch = L.K >> 3;      // (wave-uniform in SGPRs: chunk addresses stay out of the vector registers)
#pragma unroll
    for (int i = I0; i < I1; i++) {
        const int c = wave + CS_WAVES * i, k = 8 * min(c, nch - 1);
#pragma unroll
        for (int q = 0; q < 3; q++) {                               // (row by row: one row's operands in registers at a time)
            const float *src = k < K0 ? v0.r[q] + k : v1.r[q] + (k - K0);
            f32x4 a = *(const f32x4 *)src, b = *(const f32x4 *)(src + 4);            // wave-uniform address: an LDS broadcast
            if (c >= nch) { a = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; b = a; }
#pragma unroll
            for (int rb = 0; rb < NRB; rb++) {
                float s0 = acc[q][rb], s1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    s0 = fmaf((float)r.w[rb][i][j], a[j], s0); s1 = fmaf((float)r.w[rb][i][j + 1], a[j + 1], s1);
                    s0 = fmaf((float)r.w[rb][i][4 + j], b[j], s0); s1 = fmaf((float)r.w[rb][i][4 + j + 1], b[j + 1], s1);
                }
                acc[q][rb] = s0 + s1;
            }
        }
    }
}
template <int NRB>
__device__ __forceinline__ void cs_store3(const float (&acc)[3][NRB], float (*part)[3][CS_PS])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) part[wave][q][rb * 64 + lane] = acc[q][rb];
}
template <int NRB, int NCI>
__device__ __forceinline__ void cs_consume3(const rd_mv &L, const WQ<NRB, NCI> &r, const Rows3 &v0, int K0, const Rows3 &v1, float (*part)[3][CS_PS])
{
    float acc[3][NRB];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) acc[q][rb] = 0.0f;
    cs_acc3_range<NRB, NCI, 0, NCI>(L, r, v0, K0, v1, acc);
    cs_store3<NRB>(acc, part);
}
__device__ __forceinline__ float cs_row3(const rd_mv &L, const float (*part)[3][CS_PS], int q, int n)
{
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; w++) s += part[w][q][n];
    return s * (L.scale ? cs_gload(L.scale + n) : 1.0f) + (L.bias ? cs_gload(L.bias + n) : 0.0f);
}
// cs_f32_product for three rows
template <int NCI>
__device__ __forceinline__ void cs_f32_product3(const rd_mv &L, int nrb, const Rows3 &v, float (*part)[3][CS_PS])
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nch = L.K >> 3;      // (wave-uniform in SGPRs: chunk addresses stay out of the vector registers)
    for (int rb = 0; rb < nrb; rb++) {
        f32x4 w[NCI][2];
#pragma unroll
        for (int i = 0; i < NCI; i++) {
            const int c = min(wave + CS_WAVES * i, nch - 1);
            const float *p = L.wf + ((size_t)c * (nrb * 64) + rb * 64 + lane) * 8;
            w[i][0] = cs_gload((const f32x4 *)p); w[i][1] = cs_gload((const f32x4 *)(p + 4));
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int i = 0; i < NCI; i++) {
                const int c = wave + CS_WAVES * i, k = 8 * min(c, nch - 1);
                f32x4 a = *(const f32x4 *)(v.r[q] + k), b = *(const f32x4 *)(v.r[q] + k + 4);
                if (c >= nch) { a = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; b = a; }
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    s0 = fmaf(w[i][0][j], a[j], s0); s1 = fmaf(w[i][0][j + 1], a[j + 1], s1);
                    s0 = fmaf(w[i][1][j], b[j], s0); s1 = fmaf(w[i][1][j + 1], b[j + 1], s1);
                }
            }
            part[wave][q][rb * 64 + lane] = s0 + s1;
        }
    }
}
__device__ __forceinline__ float cs_gru_unit3(const rd_mv &G, const float (*part)[3][CS_PS], int q, const float *gh, float hj, int H, int j)
{
    const float gr = cs_row3(G, part, q, j), gz = cs_row3(G, part, q, H + j), gn = cs_row3(G, part, q, 2 * H + j);
    const float r = cs_sigmoid(gh[j] + gr);
    const float z = cs_sigmoid(gh[H + j] + gz);
    const float nn = tanhf(gn + gh[2 * H + j] * r);
    return (hj - nn) * z + nn;
}
// one layer for the three rows; gq: its input projection's fragments (issued by the stage before); gnext: the next layer's, issued here
template <int GCI, int CCI, int GNEXT, bool LAST>
__device__ __forceinline__ void cs_enc_layer3(CsShared3 *sh, const rd_core_args &a, int l, int n, WQ<3, GCI> &gq, WQ<3, GNEXT> &gnext)
{
    const int tid = threadIdx.x, H = 64, cin = n + H, d = a.dil[l];
    WQ<2, CCI> cq; WQ<3, 1> hq, hn;
    const Rows3 xr = { { sh->x[0], sh->x[1], sh->x[2] } };
    // the input projection: chunks [0, G1) per wavefront were requested by the stage before (at most 6: 72 registers across its end), the rest here
    constexpr int G1 = GCI < 6 ? GCI : 6;
    cs_issue_range<3, GCI, G1, GCI>(a.gin[l], gq);
    float gacc[3][3];
#pragma unroll
    for (int q = 0; q < 3; q++) { gacc[q][0] = 0.0f; gacc[q][1] = 0.0f; gacc[q][2] = 0.0f; }
    cs_acc3_range<3, GCI, 0, G1>(a.gin[l], gq, xr, n, xr, gacc);
    __builtin_amdgcn_sched_barrier(0);
    cs_acc3_range<3, GCI, G1, GCI>(a.gin[l], gq, xr, n, xr, gacc);
    cs_store3<3>(gacc, sh->pa);
    __builtin_amdgcn_sched_barrier(0);                             // (the fragments below reuse gq's registers: keep their loads behind its last use)
    cs_issue<3, 1>(a.ghh[l], hq);                                  // this layer's W_hh: steps 1 and 2 of the frame
    // the conv's fragments in up to three ranges of 8 chunks per wavefront (64 registers each): two ranges in flight across the recurrence steps, the
    // third requested when the first has been used
    constexpr int C1 = CCI < 8 ? CCI : 8, C2 = CCI < 16 ? CCI : 16;
    cs_issue_range<2, CCI, 0, C2>(a.conv[l], cq);
    CS_SYNC();
#pragma unroll 1
    for (int st = 0; st < 3; st++) {
        if (tid < H) {
            const float hv = cs_gru_unit3(a.gin[l], sh->pa, st, sh->gh, sh->h[l][tid], H, tid);
            sh->h[l][tid] = hv;
            sh->x[st][n + tid] = cs_clamp1(hv);
        }
        CS_SYNC();
        if (st < 2) {
            cs_consume<3, 1>(a.ghh[l], hq, sh->h[l], H, sh->h[l], sh->pb);
            CS_SYNC();
            if (tid >= 128 && tid < 128 + 3 * H) sh->gh[tid - 128] = cs_row(a.ghh[l], sh->pb, tid - 128);
            CS_SYNC();
        }
    }
    // Conv1d k=2, dilation d: tap 0 of row q reads row q - d (the frame's own earlier row, or the history)
    Rows3 t0;
#pragma unroll
    for (int q = 0; q < 3; q++) t0.r[q] = q - d >= 0 ? sh->x[q - d] : sh->hist[d - q - 1];
    float cacc[3][2];
#pragma unroll
    for (int q = 0; q < 3; q++) { cacc[q][0] = 0.0f; cacc[q][1] = 0.0f; }
    cs_acc3_range<2, CCI, 0, C1>(a.conv[l], cq, t0, cin, xr, cacc);
    __builtin_amdgcn_sched_barrier(0);
    cs_issue_range<2, CCI, C2, CCI>(a.conv[l], cq);
    if (!LAST) cs_issue<3, 1>(a.ghh[l + 1], hn);                   // the next layer's W_hh, for its step 0
    __builtin_amdgcn_sched_barrier(0);
    cs_acc3_range<2, CCI, C1, C2>(a.conv[l], cq, t0, cin, xr, cacc);
    __builtin_amdgcn_sched_barrier(0);
    cs_acc3_range<2, CCI, C2, CCI>(a.conv[l], cq, t0, cin, xr, cacc);
    cs_store3<2>(cacc, sh->pa);
    if (!LAST) cs_consume<3, 1>(a.ghh[l + 1], hn, sh->h[l + 1], H, sh->h[l + 1], sh->pb);
    __builtin_amdgcn_sched_barrier(0);
    if (!LAST) cs_issue_range<3, GNEXT, 0, (GNEXT < 6 ? GNEXT : 6)>(a.gin[l + 1], gnext);
    CS_SYNC();
    if (tid < 96) {
#pragma unroll
        for (int q = 0; q < 3; q++) sh->x[q][cin + tid] = cs_clamp1(tanhf(cs_row3(a.conv[l], sh->pa, q, tid)));
    } else if (!LAST && tid >= 128 && tid < 128 + 3 * H) sh->gh[tid - 128] = cs_row(a.ghh[l + 1], sh->pb, tid - 128);
    CS_SYNC();
}

__global__ __launch_bounds__(CS_THREADS) void k_tx_frame3(const rd_core_args *ap, unsigned seq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_raw[];
    CsShared3 *sh = (CsShared3 *)cs_raw;
    const rd_core_args &a = *ap;
    const int tid = threadIdx.x, W = 864, H = 64;
    WQ<3, 1> g0, h0;
    cs_issue<3, 1>(a.gin[0], g0); cs_issue<3, 1>(a.ghh[0], h0);
    for (int i = tid; i < 2 * W; i += CS_THREADS) sh->hist[i / W][i % W] = a.hist[i];
    if (tid < 5 * H) sh->h[tid / H][tid % H] = a.h[tid];
    if (tid < 3 * 96) { const int q = tid / 96, c = tid % 96; sh->vin[q][c] = c < a.n_in ? a.in[q * a.n_in + c] : 0.0f; }
    CS_SYNC();
    const Rows3 vr = { { sh->vin[0], sh->vin[1], sh->vin[2] } };
    cs_f32_product3<2>(a.dense1, 1, vr, sh->pa);                   // dense1 + tanh (radae_base.py:263)
    cs_consume<3, 1>(a.ghh[0], h0, sh->h[0], H, sh->h[0], sh->pb);
    CS_SYNC();
    if (tid < 64) {
#pragma unroll
        for (int q = 0; q < 3; q++) sh->x[q][tid] = cs_clamp1(tanhf(cs_row3(a.dense1, sh->pa, q, tid)));
    } else if (tid >= 128 && tid < 128 + 3 * H) sh->gh[tid - 128] = cs_row(a.ghh[0], sh->pb, tid - 128);
    CS_SYNC();
    WQ<3, 4> g1; WQ<3, 6> g2; WQ<3, 9> g3; WQ<3, 11> g4;
    cs_enc_layer3<1, 4, 4, false>(sh, a, 0, 64, g0, g1);
    cs_enc_layer3<4, 9, 6, false>(sh, a, 1, 224, g1, g2);
    cs_enc_layer3<6, 14, 9, false>(sh, a, 2, 384, g2, g3);
    cs_enc_layer3<9, 19, 11, false>(sh, a, 3, 544, g3, g4);
    cs_enc_layer3<11, 24, 11, true>(sh, a, 4, 704, g4, g4);
    const Rows3 xr = { { sh->x[0], sh->x[1], sh->x[2] } };
    cs_f32_product3<14>(a.out, 2, xr, sh->pa);                     // z_dense, linear (bottleneck 3)
    for (int i = tid; i < W; i += CS_THREADS) { a.hist[i] = sh->x[2][i]; a.hist[W + i] = sh->x[1][i]; }      // history of the next frame: rows -1, -2
    if (tid < 5 * H) a.h[tid] = sh->h[tid / H][tid % H];
    CS_SYNC();
    if (tid < 3 * RD_LATENT) { const int q = tid / RD_LATENT, c = tid % RD_LATENT; sh->zs[q * RD_LATENT + c] = cs_row3(a.out, sh->pa, q, c); }
    CS_SYNC();
    const float *zs = sh->zs;
    const rd_tables *tab = ap->tab;
    float2 *out = (float2 *)ap->iq_out;
    if (tid < RD_M) {
        f32x2 acc[RD_NS + 1];
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) acc[s] = (f32x2){ 0.0f, 0.0f };
#pragma unroll 6
        for (int c = 0; c < RD_NC; c++) {
            const float2 w = ld2(tab->Winv[c], tid);
            acc[0] = idft_term(acc[0], make_float2(tab->P[c] * tab->pilot_gain, 0.0f * tab->pilot_gain), w);
#pragma unroll
            for (int s = 1; s <= RD_NS; s++) { const int k = (s - 1) * RD_NC + c; acc[s] = idft_term(acc[s], make_float2(zs[2 * k], zs[2 * k + 1]), w); }
        }
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) {
            const float2 v = pa_limit(make_float2(acc[s][0], acc[s][1]));                  // tanh(|x|) e^{j angle(x)} (radae.py:218, dsp.py:377)
            out[s * RD_SYM + RD_NCP + tid] = v;
            if (tid >= RD_M - RD_NCP) out[s * RD_SYM + tid - (RD_M - RD_NCP)] = v;
        }
        __threadfence_system();
    }
    CS_SYNC();
    if (tid == 0 && ap->done) { __threadfence_system(); *(volatile unsigned *)ap->done = seq; }
}
/* a_dev: the rd_core_args record in DEVICE memory (written once by the caller: every pointer in it is fixed for the life of the state) */
extern "C" int rd_launch_tx_frame(const rd_core_args *a_dev, unsigned seq, rd_stream_t s)
{
    static int attr_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    static int by_step = -1; if (by_step < 0) by_step = getenv("RADE_TX_FRAME_BY_STEP") ? 1 : 0;      /* A/B: the three-launches-in-one form (k_tx_frame) */
    if (!attr_dev[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)k_tx_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CsShared));
        (void)hipFuncSetAttribute((const void *)k_tx_frame3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CsShared3));
        attr_dev[dev_ & 63] = 1;
    }
    if (by_step) hipLaunchKernelGGL(k_tx_frame, dim3(1), dim3(CS_THREADS), sizeof(CsShared), (hipStream_t)s, a_dev, seq);
    else hipLaunchKernelGGL(k_tx_frame3, dim3(1), dim3(CS_THREADS), sizeof(CsShared3), (hipStream_t)s, a_dev, seq);
    return (int)hipGetLastError();
}

extern "C" int rd_launch_core_step(const rd_core_args *a, rd_stream_t s)
{
    static int attr_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (!attr_dev[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)k_core_enc_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CsShared));
        (void)hipFuncSetAttribute((const void *)k_core_dec_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CsShared));
        attr_dev[dev_ & 63] = 1;
    }
    if (a->is_enc) hipLaunchKernelGGL(k_core_enc_step, dim3(1), dim3(CS_THREADS), sizeof(CsShared), (hipStream_t)s, *a);
    else hipLaunchKernelGGL(k_core_dec_step, dim3(1), dim3(CS_THREADS), sizeof(CsShared), (hipStream_t)s, *a);
    return (int)hipGetLastError();
}
