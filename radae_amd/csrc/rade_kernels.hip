// rade_kernels.hip -- gfx950 (MI355X) kernels of the RADE hot path + the C launch shims.
//
// Kernel inventory (reference op each one replaces):
//   k_gemm<NT>      f32 MFMA (v_mfma_f32_32x32x2_f32) skinny-N GEMM with fused bias/tanh/GLU epilogue:
//                   every Linear / GRU-input / Conv1d(k=2) / GLU layer of CoreEncoder / CoreDecoder,
//                   evaluated for all streams and all time steps of a chunk at once
//                   (radae_base.py:260-286, :400-416; src/rade_enc.c:55-114; src/rade_dec.c:50-102)
//   k_gru_scan<H>   the serial part of a GRU layer: h_t = f(gi_t, W_hh h_{t-1}); one workgroup per
//                   stream, W_hh rows held in VGPRs, h in LDS (radae_base.py:97-108)
//   k_enc_pack      12x36 feature frames -> 3x(4x21) encoder input rows, aux symbol -1 (radae_txe.py:114-121)
//   k_ofdm_mod      QPSK map, pilot row, 30->160 IDFT, cyclic prefix, tanh PA limiter (dsp.py:340-378)
//   k_eoo_build     end-of-over frame with 180 data bits (radae.py:208-219, :441-455)
//   k_chan_power / k_chan_apply   rate-Fs two-path multipath, power normalisation, freq offset, AWGN,
//                   EOO / noise framing (radae.py:529-589, inference.py:263-284)
//   k_rx_sync       one workgroup per stream runs do_radae_rx (radae_rxe.py:171-330): BPF (dsp.py:63-102),
//                   detect_pilots / refine / check_pilots (dsp.py:178-320), sync state machine, frequency
//                   correction, OFDM demod + 3-pilot LS EQ (dsp.py:418-526)
//   k_rx_post       decoder output -> 36-float feature frames, aux-bit (UW) error accounting
//                   (rade_api.c:480-513, radae_rxe.py:300-319)
//
// Written for gfx950 only: 64-lane wavefronts, MFMA f32 32x32x2, LDS-resident per-stream working sets.
#include "rade_devutil.h"

__device__ float g_zero_row[2048];   // tap-0 source of a conv row whose decoder state was just reset

// =====================================================================================================
// GEMM: one wavefront = 32 rows x (32*NT) columns; A and packed-W fragments stream straight from
// global/L2 into VGPRs as 16-byte loads (no LDS: each A row is read by exactly one wave, W is
// L2-resident and shared by every wave).  Lane l holds A[row l&31][k = 8kb + 4(l>>5) + s], s=0..3,
// and the packed W holds the matching k for the same lane, so MFMA s contracts k pairs
// {8kb+s, 8kb+4+s}; summation order over k does not matter.
// =====================================================================================================
template <int NT>
__global__ __launch_bounds__(64) void k_gemm(rd_gemm_args a)
{
    const int lane = threadIdx.x;
    const int rows = a.B * a.T;
    const int r0 = blockIdx.x * 32;
    const int ntt = (a.N + 31) >> 5;
    const int nt0 = blockIdx.y * NT;
    int r = r0 + (lane & 31);
    if (r >= rows) r = rows - 1;
    const int b = r / a.T, t = r - b * a.T;
    const int half = lane >> 5;
    const float *p1 = a.a1 + b * a.a1_sb + t * a.a1_st + 4 * half;
    const float *p0 = nullptr;
    if (a.K0) {
        const bool rst = a.reset && a.reset[b * a.reset_sb + t];
        p0 = (rst ? g_zero_row : a.a0 + b * a.a0_sb + t * a.a0_st) + 4 * half;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;

    const float *wp = a.Wp + ((size_t)nt0 * 64 + lane) * 4;
    const size_t wstep = (size_t)ntt * 256;
#pragma unroll 1
    for (int seg = 0; seg < 2; seg++) {
        const float *p = seg == 0 ? p0 : p1;
        const int nkb = (seg == 0 ? a.K0 : a.K1) >> 3;
        if (nkb == 0) continue;
        f32x4 av = *(const f32x4 *)p;
        f32x4 bv[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) bv[i] = *(const f32x4 *)(wp + i * 256);
        for (int kb = 0; kb < nkb; kb++) {
            f32x4 an = av; f32x4 bn[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) bn[i] = bv[i];
            if (kb + 1 < nkb) {          // prefetch next k-block while the MFMAs of this one run
                an = *(const f32x4 *)(p + (kb + 1) * 8);
#pragma unroll
                for (int i = 0; i < NT; i++) bn[i] = *(const f32x4 *)(wp + wstep + i * 256);
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < NT; i++)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[i][s], acc[i], 0, 0, 0);
            av = an;
#pragma unroll
            for (int i = 0; i < NT; i++) bv[i] = bn[i];
            wp += wstep;
        }
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (j&3) + 8*(j>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int col = (nt0 + i) * 32 + (lane & 31);
        if (col >= a.N) continue;
        const float bias = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int rr = r0 + (j & 3) + 8 * (j >> 2) + 4 * half;
            if (rr >= rows) continue;
            const int bb = rr / a.T, tt = rr - bb * a.T;
            if (a.n_rows && tt >= a.n_rows[bb]) continue;
            float v = acc[i][j] + bias;
            if (a.act == 1) v = clamp1(tanhf(v));
            else if (a.act == 2) v = clamp1(a.a1[bb * a.a1_sb + tt * a.a1_st + col] * sigmoid_f(v));
            a.y[bb * a.y_sb + tt * a.y_st + col] = v;
        }
    }
}

// The same GEMM on the f16 matrix cores, operands split in two binary16 planes (see ds_gemm16 below for the
// arithmetic): activations are split on the fly, W comes from rd_pack_weights_f16x2.  K segments are multiples of 16.
template <int NT, int RT>
__global__ __launch_bounds__(64) void k_gemm16(rd_gemm_args a)
{   // one wavefront = RT row tiles of 32 rows x NT column tiles: every W fragment is applied to RT row tiles, so the L2 traffic
    // for the weights (the whole matrix per workgroup) drops RT-fold
    const int lane = threadIdx.x;
    const int rows = a.B * a.T;
    const int r0 = blockIdx.x * 32 * RT;
    const int ntt = (a.N + 31) >> 5;
    const int nt0 = blockIdx.y * NT;
    const int half = lane >> 5;
    const float *p1[RT], *p0[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) {
        int r = r0 + 32 * q + (lane & 31);
        if (r >= rows) r = rows - 1;
        const int b = r / a.T, t = r - b * a.T;
        p1[q] = a.a1 + b * a.a1_sb + t * a.a1_st + 8 * half;
        p0[q] = nullptr;
        if (a.K0) {
            const bool rst = a.reset && a.reset[b * a.reset_sb + t];
            p0[q] = (rst ? g_zero_row : a.a0 + b * a.a0_sb + t * a.a0_st) + 8 * half;
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
    const bool single = a.Wscale != nullptr;          // int8-exact layer: the weights are ONE plane of integers (exact in binary16), two products per k-block
    const int planes = single ? 1 : 2;
    const unsigned short *wbase = a.Wp16 + ((size_t)nt0 * planes * 64 + lane) * 8;
    const size_t wstep = (size_t)ntt * planes * 64 * 8;
    f32x4 a4[RT][2]; f16x8 bh[NT], bl[NT];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int q = 0; q < RT; q++) {
            const float *p = kb < nkb0 ? p0[q] + kb * 16 : p1[q] + (kb - nkb0) * 16;
            a4[q][0] = *(const f32x4 *)p; a4[q][1] = *(const f32x4 *)(p + 4);
        }
#pragma unroll
        for (int i = 0; i < NT; i++) { bh[i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * planes * 64 * 8); if (!single) bl[i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * 2 * 64 * 8 + 64 * 8); }
    };
    fetch(0);
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb++) {
        f16x8 ah[RT], al[RT], ch[NT], cl[NT];
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = 256.0f * a4[q][j >> 2][j & 3];     // 2^8 (activations) x 2^10 (packed W): low planes stay normal binary16
                const _Float16 hi = (_Float16)x;
                ah[q][j] = hi; al[q][j] = (_Float16)(x - (float)hi);
            }
#pragma unroll
        for (int i = 0; i < NT; i++) { ch[i] = bh[i]; cl[i] = bl[i]; }
        if (kb + 1 < nkb) fetch(kb + 1);                    // next k-block's loads fly during the matrix instructions
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) {
                acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q], ch[i], acc[q][i], 0, 0, 0);
                if (!single) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], cl[i], acc[q][i], 0, 0, 0);
                acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], ch[i], acc[q][i], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int col = (nt0 + i) * 32 + (lane & 31);
            if (col >= a.N) continue;
            const float bias = a.bias ? a.bias[col] : 0.0f;
            const float scl = single ? a.Wscale[col] * 0x1p-8f : 0x1p-18f;       // integers x column scale (rows carry 2^8), or two planes of 2^10 w
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int rr = r0 + 32 * q + (j & 3) + 8 * (j >> 2) + 4 * half;
                if (rr >= rows) continue;
                const int bb = rr / a.T, tt = rr - bb * a.T;
                if (a.n_rows && tt >= a.n_rows[bb]) continue;
                float v = acc[q][i][j] * scl + bias;
                if (a.act == 1) v = clamp1(gate_tanh(v));                 // hardware exp2 / rcp, as in the recurrences
                else if (a.act == 2) v = clamp1(a.a1[bb * a.a1_sb + tt * a.a1_st + col] * gate_sigmoid(v));
                a.y[bb * a.y_sb + tt * a.y_st + col] = v;
            }
        }
}

// The same kernel with two k-blocks of operands in flight and no branch inside the k loop: "one or two weight planes" is a template
// parameter and the k-blocks come in pairs (every layer of the model has an even number), the fetches past the end re-read the last
// block.  With the plane count a run-time flag every matrix instruction sat behind a uniform branch and the loads of the next
// k-block could only be waited for all at once; here the compiler counts them (partial vmcnt waits) and a fetch has two k-blocks of
// matrix work to complete.
// GEMM16P_WPB wavefronts per workgroup = that many ADJACENT row tiles against the same column tiles (developer switch, 1 shipped): no LDS, no barrier, nothing shared
// in the source -- the wavefronts of a workgroup start together on one CU and walk the same weight fragments within a few k-blocks of each other, so all but the first
// find them in that CU's vector L1.  Measured (round 5, profiles/r05_ab_notes.txt): 2 / 4 / 8 per workgroup change nothing alone (+2.3 / +0.3 / +7.3 % GEMM time) and nothing
// decidable in the pipeline (-0.2 / +1.3 / -10.5 % frames/s): the weight traffic (3.5 GB of the 5.9 GB a pass moves from L2 to L1) is not what these launches wait for.
#ifndef GEMM16P_WPB
#define GEMM16P_WPB 1
#endif
template <int NT, int RT, bool SINGLE>
__global__ __launch_bounds__(64 * GEMM16P_WPB) void k_gemm16p(rd_gemm_args a)
{
    const int lane = threadIdx.x & 63;
    const int rows = a.B * a.T;
    const int r0 = (blockIdx.x * GEMM16P_WPB + (int)(threadIdx.x >> 6)) * 32 * RT;
    if (r0 >= rows) return;
    const int ntt = (a.N + 31) >> 5;
    const int nt0 = blockIdx.y * NT;
    const int half = lane >> 5;
    const float *p1[RT], *p0[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) {
        int r = r0 + 32 * q + (lane & 31);
        if (r >= rows) r = rows - 1;
        const int b = r / a.T, t = r - b * a.T;
        p1[q] = a.a1 + b * a.a1_sb + t * a.a1_st + 8 * half;
        p0[q] = p1[q];
        if (a.K0) {
            const bool rst = a.reset && a.reset[b * a.reset_sb + t];
            p0[q] = (rst ? g_zero_row : a.a0 + b * a.a0_sb + t * a.a0_st) + 8 * half;
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
    constexpr int planes = SINGLE ? 1 : 2;
    const unsigned short *wbase = a.Wp16 + ((size_t)nt0 * planes * 64 + lane) * 8;
    const size_t wstep = (size_t)ntt * planes * 64 * 8;
    f32x4 a4[2][RT][2]; f16x8 bh[2][NT], bl[2][NT];
    auto fetch = [&](int st, int kb_) {
        const int kb = min(kb_, nkb - 1);
#pragma unroll
        for (int q = 0; q < RT; q++) {
            const float *p = kb < nkb0 ? p0[q] + kb * 16 : p1[q] + (kb - nkb0) * 16;
            a4[st][q][0] = *(const f32x4 *)p; a4[st][q][1] = *(const f32x4 *)(p + 4);
        }
#pragma unroll
        for (int i = 0; i < NT; i++) {
            bh[st][i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * planes * 64 * 8);
            if (!SINGLE) bl[st][i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * 2 * 64 * 8 + 64 * 8);
        }
    };
    auto block = [&](int st, int kb_next) {
        f16x8 ah[RT], al[RT], ch[NT], cl[NT];
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = 256.0f * a4[st][q][j >> 2][j & 3];
                const _Float16 hi = (_Float16)x;
                ah[q][j] = hi; al[q][j] = (_Float16)(x - (float)hi);
            }
#pragma unroll
        for (int i = 0; i < NT; i++) { ch[i] = bh[st][i]; cl[i] = bl[st][i]; }
        fetch(st, kb_next);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) {
                acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q], ch[i], acc[q][i], 0, 0, 0);
                if (!SINGLE) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], cl[i], acc[q][i], 0, 0, 0);
                acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], ch[i], acc[q][i], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(0, 0); fetch(1, 1);
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += 2) { block(0, kb + 2); block(1, kb + 3); }
    // epilogue: a lane's rows once per row tile (tile row -> (stream, step) by ONE division per wavefront and carries, the row's output pointer), then the
    // column tiles.  (Rounds 1-4 divided by T for every one of the 48 elements of a lane: a third of a short layer's instructions.)
    const int cl0 = nt0 * 32 + (lane & 31);
    float bias[NT], scl[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int col = min(cl0 + 32 * i, a.N - 1);
        bias[i] = a.bias ? a.bias[col] : 0.0f;
        scl[i] = SINGLE ? a.Wscale[col] * 0x1p-8f : 0x1p-18f;
    }
    const int b0 = r0 / a.T, t0 = r0 - b0 * a.T;
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int off = 32 * q + (j & 3) + 8 * (j >> 2) + 4 * half;
            int bb = b0, tt = t0 + off;
            while (tt >= a.T) { tt -= a.T; bb++; }               // (a tile spans more than two streams only when T < 32 RT)
            if (r0 + off >= rows || (a.n_rows && tt >= a.n_rows[bb])) continue;
            float *yr = a.y + bb * a.y_sb + tt * a.y_st;
            const float *gr = a.a1 + bb * a.a1_sb + tt * a.a1_st;
#pragma unroll
            for (int i = 0; i < NT; i++) {
                const int col = cl0 + 32 * i;
                if (col >= a.N) continue;
                float v = acc[q][i][j] * scl[i] + bias[i];
                if (a.act == 1) v = clamp1(gate_tanh(v));
                else if (a.act == 2) v = clamp1(gr[col] * gate_sigmoid(v));
                yr[col] = v;
            }
        }
}

// Small-M variant (decoder rounds, single-stream API): the K loop is the latency, so 8 wavefronts of one
// workgroup split it (k-blocks interleaved), partial accumulators meet in LDS, and each wave finishes two of the
// sixteen accumulator registers of every tile (bias / activation / store).
#define SK_WAVES 8
template <int NT>
__global__ __launch_bounds__(64 * SK_WAVES) void k_gemm_splitk(rd_gemm_args a)
{
    extern __shared__ __attribute__((aligned(16))) float sk_red[];       // [SK_WAVES][NT][16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = a.B * a.T;
    const int r0 = blockIdx.x * 32;
    const int ntt = (a.N + 31) >> 5;
    const int nt0 = blockIdx.y * NT;
    if (a.n_rows) {                              // decoder rounds: skip tiles whose rows all lie beyond their stream's count
        const int rl = min(r0 + 31, rows - 1);
        bool any = false;
        for (int bb = r0 / a.T; bb <= rl / a.T; bb++) { const int tlo = max(r0 - bb * a.T, 0); any = any || (tlo < a.n_rows[bb]); }
        if (!any) return;
    }
    int r = r0 + (lane & 31);
    if (r >= rows) r = rows - 1;
    const int b = r / a.T, t = r - b * a.T;
    const int half = lane >> 5;
    const float *p1 = a.a1 + b * a.a1_sb + t * a.a1_st + 4 * half;
    const float *p0 = nullptr;
    if (a.K0) {
        const bool rst = a.reset && a.reset[b * a.reset_sb + t];
        p0 = (rst ? g_zero_row : a.a0 + b * a.a0_sb + t * a.a0_st) + 4 * half;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    const int nkb0 = a.K0 >> 3, nkb = nkb0 + (a.K1 >> 3);
    const float *wbase = a.Wp + ((size_t)nt0 * 64 + lane) * 4;
    const size_t wstep = (size_t)ntt * 256;
#pragma unroll 2
    for (int kb = wave; kb < nkb; kb += SK_WAVES) {
        const float *p = kb < nkb0 ? p0 + kb * 8 : p1 + (kb - nkb0) * 8;
        const f32x4 av = *(const f32x4 *)p;
        f32x4 bv[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) bv[i] = *(const f32x4 *)(wbase + kb * wstep + i * 256);
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < NT; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[i][s], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) sk_red[((wave * NT + i) * 16 + j) * 64 + lane] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int col = (nt0 + i) * 32 + (lane & 31);
        if (col >= a.N) continue;
        const float bias = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int j = wave * 2 + jj;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < SK_WAVES; w++) v += sk_red[((w * NT + i) * 16 + j) * 64 + lane];
            const int rr = r0 + (j & 3) + 8 * (j >> 2) + 4 * half;
            if (rr >= rows) continue;
            const int bb = rr / a.T, tt = rr - bb * a.T;
            if (a.n_rows && tt >= a.n_rows[bb]) continue;
            v += bias;
            if (a.act == 1) v = clamp1(tanhf(v));
            else if (a.act == 2) v = clamp1(a.a1[bb * a.a1_sb + tt * a.a1_st + col] * sigmoid_f(v));
            a.y[bb * a.y_sb + tt * a.y_st + col] = v;
        }
    }
}

extern "C" int rd_launch_gemm(const rd_gemm_args *a, rd_stream_t s)
{
    const int rows = a->B * a->T;
    if (rows <= 0) return 0;
    const int ntt = (a->N + 31) >> 5;
    hipStream_t st = (hipStream_t)s;
    const int gx = (rows + 31) / 32;
    if (rows <= 16384) {                       // too few row tiles to fill the chip: split K inside the workgroup
        dim3 block(64 * SK_WAVES);
        static int attr_done_dev[64];
        int dev_ = 0; (void)hipGetDevice(&dev_);
        int &attr_done = attr_done_dev[dev_ & 63];
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void *)k_gemm_splitk<3>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_WAVES * 3 * 16 * 64 * 4);
            (void)hipFuncSetAttribute((const void *)k_gemm_splitk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_WAVES * 2 * 16 * 64 * 4);
            (void)hipFuncSetAttribute((const void *)k_gemm_splitk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_WAVES * 1 * 16 * 64 * 4);
            attr_done = 1;
        }
        if (ntt % 3 == 0) hipLaunchKernelGGL(k_gemm_splitk<3>, dim3(gx, ntt / 3), block, SK_WAVES * 3 * 16 * 64 * 4, st, *a);
        else if (ntt % 2 == 0) hipLaunchKernelGGL(k_gemm_splitk<2>, dim3(gx, ntt / 2), block, SK_WAVES * 2 * 16 * 64 * 4, st, *a);
        else hipLaunchKernelGGL(k_gemm_splitk<1>, dim3(gx, ntt), block, SK_WAVES * 1 * 16 * 64 * 4, st, *a);
        return (int)hipGetLastError();
    }
    dim3 block(64);
    if (a->Wp16 && (a->K0 & 15) == 0 && (a->K1 & 15) == 0) {          // f16 matrix cores, two-plane operands
        const int gx2 = (rows + 63) / 64;
        static int pair_ok = -1; if (pair_ok < 0) pair_ok = getenv("RADE_GEMM_NO_PAIRS") ? 0 : 1;
        if (pair_ok && (((a->K0 + a->K1) >> 4) & 1) == 0 && ((a->K0 >> 4) & 1) == 0 && ntt % 3 == 0) {      // k-blocks in pairs (and the tap boundary on a pair)
            static int rt1 = -1; if (rt1 < 0) rt1 = getenv("RADE_GEMM_RT1") ? 1 : 0;
            // one 32-row tile per wavefront for the one-plane layers: 96 accumulator registers less, a third wavefront per SIMD
            // (0.711 -> 0.667 ms per step over the encoder's GEMMs); six column tiles per wavefront (activations read once) changed nothing
            dim3 blockp(64 * GEMM16P_WPB);
            if (a->Wscale) { dim3 g1((gx + GEMM16P_WPB - 1) / GEMM16P_WPB, ntt / 3); hipLaunchKernelGGL((k_gemm16p<3, 1, true>), g1, blockp, 0, st, *a); return (int)hipGetLastError(); }
            dim3 grid((gx2 + GEMM16P_WPB - 1) / GEMM16P_WPB, ntt / 3);
            hipLaunchKernelGGL((k_gemm16p<3, 2, false>), grid, blockp, 0, st, *a);
            return (int)hipGetLastError();
        }
        if (ntt % 3 == 0) { dim3 grid(gx2, ntt / 3); hipLaunchKernelGGL((k_gemm16<3, 2>), grid, block, 0, st, *a); }
        else if (ntt % 2 == 0) { dim3 grid(gx2, ntt / 2); hipLaunchKernelGGL((k_gemm16<2, 2>), grid, block, 0, st, *a); }
        else { dim3 grid(gx2, ntt); hipLaunchKernelGGL((k_gemm16<1, 2>), grid, block, 0, st, *a); }
        return (int)hipGetLastError();
    }
    if (ntt % 3 == 0) { dim3 grid(gx, ntt / 3); hipLaunchKernelGGL(k_gemm<3>, grid, block, 0, st, *a); }
    else if (ntt % 2 == 0) { dim3 grid(gx, ntt / 2); hipLaunchKernelGGL(k_gemm<2>, grid, block, 0, st, *a); }
    else { dim3 grid(gx, ntt); hipLaunchKernelGGL(k_gemm<1>, grid, block, 0, st, *a); }
    return (int)hipGetLastError();
}

// =====================================================================================================
// GRU recurrence (the only serial part of a layer): h_t = f(gi_t, W_hh h_{t-1}), one workgroup per stream.
// Four adjacent lanes own hidden unit j; lane part p holds the r/z/n rows of W_hh for k in [p*H/4, (p+1)*H/4)
// in VGPRs, partial dot products meet through quad shuffles, every lane of the quad evaluates the gates
// (no divergence) and part 0 publishes h_j to LDS: one barrier per time step.
// =====================================================================================================
template <int H>
__global__ __launch_bounds__(4 * H) void k_gru_scan(rd_scan_args a)
{
    // a latency chain (one barrier per step, a handful of instructions between two of them) that shares its SIMDs with receiver wavefronts of other batches:
    // at the default priority every one of its instructions queues behind theirs (72 us per launch alone, 200 us in the pipelined bench); raised, the
    // recurrence runs close to its own latency and takes few issue slots from anybody (same-box A/B: +1.6 .. +3.3 % frames/s; the GEMM / modulator / channel
    // kernels, which are throughput-bound, gained nothing from the same treatment)
    __builtin_amdgcn_s_setprio(3);
    constexpr int KP = H / 4;                       // k range per lane
    __shared__ __attribute__((aligned(16))) float hs[2][H];   // double-buffered so one barrier per step suffices
    __shared__ int rst[RD_DEC_ROWS_MAX];            // reset flags are only used by the decoder rounds (T <= 384)
    const int b = blockIdx.x, tid = threadIdx.x, j = tid >> 2, p = tid & 3;
    float wr[KP], wz[KP], wn[KP];
    {
        const float *w0 = a.Whh + (size_t)j * H + p * KP;
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
            const f32x4 v0 = *(const f32x4 *)(w0 + k), v1 = *(const f32x4 *)(w0 + (size_t)H * H + k), v2 = *(const f32x4 *)(w0 + (size_t)2 * H * H + k);
#pragma unroll
            for (int u = 0; u < 4; u++) { wr[k + u] = v0[u]; wz[k + u] = v1[u]; wn[k + u] = v2[u]; }
        }
    }
    const float br = a.bhh[j], bz = a.bhh[H + j], bn = a.bhh[2 * H + j];
    const int Tb = a.n_rows ? a.n_rows[b] : a.T;
    if (a.reset) for (int i = tid; i < a.T && i < RD_DEC_ROWS_MAX; i += blockDim.x) rst[i] = a.reset[b * a.reset_sb + i];
    float hj = a.h[(size_t)b * H + j];
    if (p == 0) hs[0][j] = hj;
    _Float16 *of = nullptr;                         // the batched encoder's fragment buffer (rade_enc.hip): unit j's slot in row 0 of the stream's history tile
    if (a.outf) { const int col = a.outf_col + j; of = (_Float16 *)a.outf + (size_t)b * a.outf_NQ * RD_EF_TILE + (col >> 4) * 1024 + ((col >> 3) & 1) * 256 + (col & 7); }
    const float *gi = a.gi + (size_t)b * a.gi_sb + (p < 3 ? p * H + j : j);   // lane part p < 3 fetches gate p of unit j
    // gi is fetched four steps at a time into TWO register sets that take turns: a set is refilled right after its block of steps and consumed a whole
    // block later, so the loads land in the registers they are used from.  (Rounds 1-3 had one set and a copy "next -> current" at the end of a block: the
    // compiler put the fresh loads and `s_waitcnt vmcnt(0)` in front of that copy -- a memory round trip exposed every four steps, a quarter of the step.)
    // Rows beyond Tb re-read the last row (never used): no branch around a load.
    float gA[4], gB[4];
    auto fetch = [&](float (&dst)[4], int t0) {
#pragma unroll
        for (int u = 0; u < 4; u++) dst[u] = gi[(size_t)min(t0 + u, max(Tb - 1, 0)) * a.gi_st];
    };
    // every load issued so far (W_hh rows, biases, state) completes here: left pending into the loop, the wait-count bookkeeping (one state per loop header,
    // merged over entry and back edge) makes the first step of every block wait for ALL outstanding loads
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    fetch(gA, 0); fetch(gB, 4);
    __syncthreads();
    int cur = 0;
    // fragment output (the batched encoder): a step's h leaves one step later -- the conversion to two binary16 planes and the two 2-byte stores are not on the
    // path between two barriers (on it they cost 40 ns per step: 0.39 -> 0.44 ms over the five scans of an encoder pass)
    float pend = 0.0f; int tpend = -1;
    auto store_frag = [&](int tp) {
        if (p == 0) {
            _Float16 *o = of + (size_t)(1 + (tp >> 5)) * RD_EF_TILE + (tp & 31) * 8;
            const float x = 256.0f * clamp1(pend); const _Float16 hi = (_Float16)x;
            o[0] = hi; o[512] = (_Float16)(x - (float)hi);
        }
    };
    auto block = [&](const float (&gq)[4], int t0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u;
            if (t >= Tb) break;
            if (a.reset && rst[t]) {                       // uniform over the workgroup
                hj = 0.0f;
                __syncthreads();
                if (p == 0) hs[cur][j] = 0.0f;
                __syncthreads();
            }
            // this lane's quarter of the three dot products as packed FMAs (even / odd k in the two halves of an accumulator pair): 3 KP / 2 instructions
            // instead of 3 KP on the step's serial path
            // (the three dot products one after the other, each gate's exp2 / rcp started under the next product's multiply-adds: 0.472 against 0.414 ms over the five
            // scans of a pass -- two dependent accumulator chains per product instead of six independent ones; profiles/r05_ab_notes.txt)
            f32x2 ar = { 0.0f, 0.0f }, az = ar, an = ar;
            const float *hp = hs[cur] + p * KP;
#pragma unroll
            for (int k = 0; k < KP; k += 4) {
                const f32x4 hv = *(const f32x4 *)(hp + k);
                const f32x2 h0 = { hv[0], hv[1] }, h1 = { hv[2], hv[3] };
                ar = __builtin_elementwise_fma((f32x2){ wr[k], wr[k + 1] }, h0, ar); az = __builtin_elementwise_fma((f32x2){ wz[k], wz[k + 1] }, h0, az); an = __builtin_elementwise_fma((f32x2){ wn[k], wn[k + 1] }, h0, an);
                ar = __builtin_elementwise_fma((f32x2){ wr[k + 2], wr[k + 3] }, h1, ar); az = __builtin_elementwise_fma((f32x2){ wz[k + 2], wz[k + 3] }, h1, az); an = __builtin_elementwise_fma((f32x2){ wn[k + 2], wn[k + 3] }, h1, an);
            }
            if (of && tpend >= 0) store_frag(tpend);       // the previous step's output: independent of this step's chain, issued behind its multiply-adds
            float sr = ar[0] + ar[1], sz = az[0] + az[1], sn = an[0] + an[1];
            sr += quad_dpp<QUAD_XOR1>(sr); sz += quad_dpp<QUAD_XOR1>(sz); sn += quad_dpp<QUAD_XOR1>(sn);
            sr += quad_dpp<QUAD_XOR2>(sr); sz += quad_dpp<QUAD_XOR2>(sz); sn += quad_dpp<QUAD_XOR2>(sn);
            const float g0 = gq[u];
            const float gr = quad_dpp<QUAD_BC0>(g0), gz = quad_dpp<QUAD_BC1>(g0), gn = quad_dpp<QUAD_BC2>(g0);
            const float r = gate_sigmoid((sr + br) + gr);
            const float z = gate_sigmoid((sz + bz) + gz);
            const float n = gate_tanh(gn + (sn + bn) * r);
            hj = (hj - n) * z + n;
            if (p == 0) {
                hs[cur ^ 1][j] = hj;
                if (!of) a.out[(size_t)b * a.out_sb + (size_t)t * a.out_st + j] = clamp1(hj);
            }
            pend = hj; tpend = t;
            cur ^= 1;
            __syncthreads();
        }
    };
    for (int t0 = 0; t0 < Tb; t0 += 8) {
        block(gA, t0);
        fetch(gA, t0 + 8);
        block(gB, t0 + 4);
        fetch(gB, t0 + 12);
    }
    if (of && tpend >= 0) store_frag(tpend);
    if (p == 0) a.h[(size_t)b * H + j] = hj;
}

extern "C" int rd_launch_gru_scan(const rd_scan_args *a, rd_stream_t s)
{
    if (a->B <= 0) return 0;
    hipStream_t st = (hipStream_t)s;
    if (a->H == 64) hipLaunchKernelGGL(k_gru_scan<64>, dim3(a->B), dim3(256), 0, st, *a);
    else if (a->H == 96) hipLaunchKernelGGL(k_gru_scan<96>, dim3(a->B), dim3(384), 0, st, *a);
    else return -1;
    return (int)hipGetLastError();
}

// =====================================================================================================
// small data-movement kernels
// =====================================================================================================
__global__ void k_enc_pack(const float *features, float *xin, int B, int T)
{   // model19 only: 4 x (20 features + aux symbol -1) padded 84 -> 88
    __builtin_amdgcn_s_setprio(3);
    const long n = (long)B * T * RD_ENC_IN;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % RD_ENC_IN); const long bt = i / RD_ENC_IN;
        float v = 0.0f;
        if (c < 84) { const int fr = c / 21, j = c - fr * 21; v = j < 20 ? features[(bt * 4 + fr) * 36 + j] : -1.0f; }
        xin[i] = v;
    }
}
extern "C" int rd_launch_enc_pack(const float *features, float *xin, int B, int T, rd_stream_t s)
{
    const long n = (long)B * T * RD_ENC_IN; if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_enc_pack, dim3(grid), dim3(256), 0, (hipStream_t)s, features, xin, B, T);
    return (int)hipGetLastError();
}

// dense rows [R][K] -> [R][Kpad] with zero fill (GEMM K must be a multiple of 8)
__global__ void k_pad_rows(const float *src, float *dst, long R, int K, int Kpad)
{
    const long n = R * Kpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Kpad); const long r = i / Kpad;
        dst[i] = c < K ? src[r * K + c] : 0.0f;
    }
}
extern "C" int rd_launch_pad_rows(const float *src, float *dst, long R, int K, int Kpad, rd_stream_t s)
{
    const long n = R * Kpad; if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_pad_rows, dim3(grid), dim3(256), 0, (hipStream_t)s, src, dst, R, K, Kpad);
    return (int)hipGetLastError();
}

// x is [B][nhist+Tcap][W]; copy rows [Tb, Tb+nhist) -> [0, nhist)  (Tb = n_rows[b] or T).  Source and
// destination overlap when Tb < nhist, so every thread reads all its elements before any write.
__global__ __launch_bounds__(256) void k_carry_rows(float *x, int Tcap, int W, int nhist, int T, const int *n_rows)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    const int Tb = n_rows ? n_rows[b] : T;
    if (Tb <= 0) return;
    float *base = x + (size_t)b * (nhist + Tcap) * W;
    const int n = nhist * W;       // <= 2048
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int i = threadIdx.x + q * 256; v[q] = i < n ? base[(size_t)Tb * W + i] : 0.0f; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++) { const int i = threadIdx.x + q * 256; if (i < n) base[i] = v[q]; }
}
extern "C" int rd_launch_carry_rows(float *x, int B, int Tcap, int W, int nhist, int T, const int *n_rows, rd_stream_t s)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_carry_rows, dim3(B), dim3(256), 0, (hipStream_t)s, x, Tcap, W, nhist, T, n_rows);
    return (int)hipGetLastError();
}

// tanh(|x|) * exp(j*angle(x))   (radae.py:218, dsp.py:377)

// one workgroup per (modem frame, stream): 5 symbols x 160 samples, 30-term IDFT per sample
__global__ __launch_bounds__(192) void k_ofdm_mod(const rd_tables *tab, const float *z, float2 *tx, long tx_stride, int n_mf)
{
    __shared__ float2 sym[RD_NS + 1][RD_NC];
    const int mf = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *zf = z + ((size_t)b * n_mf + mf) * RD_ZMF;
    if (tid < RD_NC) sym[0][tid] = make_float2(tab->P[tid] * tab->pilot_gain, 0.0f * tab->pilot_gain);
    if (tid < 120) sym[1 + tid / RD_NC][tid % RD_NC] = make_float2(zf[2 * tid], zf[2 * tid + 1]);
    __syncthreads();
    float2 *out = tx + (size_t)b * tx_stride + (size_t)mf * RD_NMF;
    if (tid < RD_M) {
        f32x2 acc[RD_NS + 1];
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) acc[s] = (f32x2){ 0.0f, 0.0f };
#pragma unroll 6
        for (int c = 0; c < RD_NC; c++) {                 // one Winv load feeds the five symbols of the frame
            const float2 w = ld2(tab->Winv[c], tid);
#pragma unroll
            for (int s = 0; s <= RD_NS; s++) acc[s] = idft_term(acc[s], sym[s][c], w);
        }
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) {
            const float2 v = pa_limit(make_float2(acc[s][0], acc[s][1]));
            out[s * RD_SYM + RD_NCP + tid] = v;
            if (tid >= RD_M - RD_NCP) out[s * RD_SYM + tid - (RD_M - RD_NCP)] = v;
        }
    }
}
extern "C" int rd_launch_ofdm_mod(const rd_tables *tab, const float *z, void *tx, long tx_stride, int B, int n_mf, rd_stream_t s)
{
    if (B <= 0 || n_mf <= 0) return 0;
    hipLaunchKernelGGL(k_ofdm_mod, dim3(n_mf, B), dim3(192), 0, (hipStream_t)s, tab, z, (float2 *)tx, tx_stride, n_mf);
    return (int)hipGetLastError();
}

// The modulator with the first half of the channel simulator folded in (rade_batch_tx_channel: RADAE.forward goes from latents to received
// samples in one pass too, radae.py:529-589): the workgroup keeps its modem frame's 960 samples in LDS, applies the two-path
// multipath model mp[i] = tx[i] G1[i] + tx[i-16] G2[i-16] while they are there (the 16 samples it needs from the frame before are
// re-synthesised: 16 x 30 terms) and leaves per-frame sums of |tx|^2 and |mp|^2 for the power normalisation.  tx never makes a round
// trip through HBM, k_chan_power disappears, and k_chan_apply reads 8 bytes per sample (mp) instead of 24 (tx + G).
__global__ __launch_bounds__(192) void k_ofdm_mod_mp(const rd_tables *tab, const float *z, float2 *tx, long tx_stride, int n_mf, const float2 *G, float2 *mp, double *part)
{
    __shared__ float2 sym[RD_NS + 1][RD_NC];
    __shared__ float2 prevsym[RD_NC];
    __shared__ float2 fr[16 + RD_NMF];                    // [0, 16): tail of the previous frame, then this frame
    __shared__ double red[2][4];
    const int mf = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *zf = z + ((size_t)b * n_mf + mf) * RD_ZMF;
    if (tid < RD_NC) sym[0][tid] = make_float2(tab->P[tid] * tab->pilot_gain, 0.0f * tab->pilot_gain);
    if (tid < 120) sym[1 + tid / RD_NC][tid % RD_NC] = make_float2(zf[2 * tid], zf[2 * tid + 1]);
    if (tid >= 128 && tid < 128 + RD_NC && mf > 0) { const int c = tid - 128; prevsym[c] = make_float2(zf[-RD_ZMF + 2 * (90 + c)], zf[-RD_ZMF + 2 * (90 + c) + 1]); }   // last data symbol of frame mf - 1
    __syncthreads();
    if (tid < RD_M) {
        f32x2 acc[RD_NS + 1];
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) acc[s] = (f32x2){ 0.0f, 0.0f };
#pragma unroll 6
        for (int c = 0; c < RD_NC; c++) {
            const float2 w = ld2(tab->Winv[c], tid);
#pragma unroll
            for (int s = 0; s <= RD_NS; s++) acc[s] = idft_term(acc[s], sym[s][c], w);
        }
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) {
            const float2 v = pa_limit(make_float2(acc[s][0], acc[s][1]));
            fr[16 + s * RD_SYM + RD_NCP + tid] = v;
            if (tid >= RD_M - RD_NCP) fr[16 + s * RD_SYM + tid - (RD_M - RD_NCP)] = v;
        }
    } else if (tid < RD_M + 16) {                          // samples 944..959 of the previous frame = the last 16 of its last symbol
        const int n = RD_M - 16 + (tid - RD_M);
        float2 a = make_float2(0.0f, 0.0f);
        if (mf > 0) { f32x2 ac = { 0.0f, 0.0f }; for (int c = 0; c < RD_NC; c++) ac = idft_term(ac, prevsym[c], ld2(tab->Winv[c], n)); a = pa_limit(make_float2(ac[0], ac[1])); }
        fr[tid - RD_M] = a;                                // frame 0: the signal starts here, nothing before it (chan_mp: i >= 16)
    }
    __syncthreads();
    const size_t base = (size_t)mf * RD_NMF;
    const f32x4 *Gb = (const f32x4 *)G + (size_t)b * n_mf * RD_NMF;      // (G1[i], G2[i]) as one 16-byte load per sample
    // (requesting these before the IDFT instead -- 20 more registers live across it -- made the kernel 5 % slower: 187 -> 196 us; the IDFT and the
    // limiter, not the round trip, are what a workgroup spends its time on)
    float2 *mpo = mp + (size_t)b * n_mf * RD_NMF + base;
    float2 *txo = tx ? tx + (size_t)b * tx_stride + base : nullptr;
    // second path: c2[i + 16] = tx[i] G2[i], written to LDS by the thread that holds G2[i]; the first 16 slots of the frame come from the
    // previous frame's tail (fr[0..16)) and its G2
    __shared__ float2 c2[16 + RD_NMF];
    float2 a1[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int i = tid + 192 * q;                       // 960 = 5 x 192
        const f32x4 g = Gb[base + i];
        const float2 x = fr[16 + i];
        a1[q] = cmul(x, make_float2(g[0], g[1]));
        c2[16 + i] = cmul(x, make_float2(g[2], g[3]));
    }
    if (tid < 16) { float2 v = make_float2(0.0f, 0.0f); if (mf > 0) { const f32x4 g = Gb[base - 16 + tid]; v = cmul(fr[tid], make_float2(g[2], g[3])); } c2[tid] = v; }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int i = tid + 192 * q;
        const float2 x = fr[16 + i];
        const float2 m = cadd(a1[q], c2[i]);               // c2[i] = tx[base + i - 16] G2[base + i - 16]; zero for the first 16 samples of the signal
        mpo[i] = m;
        if (txo) txo[i] = x;
        const float ax = hypotf(x.x, x.y), am = hypotf(m.x, m.y);
        s0 += (double)(ax * ax); s1 += (double)(am * am);
    }
    // frame sums: inside a wavefront by DPP, the three wavefronts' results through LDS (one barrier; the LDS tree this replaces had eight)
    s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s0; red[1][tid >> 6] = s1; }
    __syncthreads();
    if (tid == 0) { part[((size_t)b * n_mf + mf) * 2] = (red[0][0] + red[0][1]) + red[0][2]; part[((size_t)b * n_mf + mf) * 2 + 1] = (red[1][0] + red[1][1]) + red[1][2]; }
}
extern "C" int rd_launch_ofdm_mod_mp(const rd_tables *tab, const float *z, void *tx, long tx_stride, int B, int n_mf, const void *G, void *mp, double *part, rd_stream_t s)
{
    if (B <= 0 || n_mf <= 0) return 0;
    hipLaunchKernelGGL(k_ofdm_mod_mp, dim3(n_mf, B), dim3(192), 0, (hipStream_t)s, tab, z, (float2 *)tx, tx_stride, n_mf, (const float2 *)G, (float2 *)mp, part);
    return (int)hipGetLastError();
}

// EOO frame per stream: default table copy, optionally with 3 data symbols (90 QPSK) inserted
__global__ __launch_bounds__(192) void k_eoo_build(const rd_tables *tab, const float *bits, float2 *eoo)
{
    __shared__ float2 sym[RD_NS - 1][RD_NC];
    const int b = blockIdx.x, tid = threadIdx.x;
    float2 *out = eoo + (size_t)b * RD_NEOO;
    for (int i = tid; i < RD_NEOO; i += blockDim.x) out[i] = ld2(tab->eoo, i);
    if (!bits) return;
    if (tid < 90) sym[tid / RD_NC][tid % RD_NC] = make_float2(bits[b * RD_NEOOBITS + 2 * tid], bits[b * RD_NEOOBITS + 2 * tid + 1]);
    __syncthreads();
    if (tid < RD_M) {
        for (int s = 0; s < RD_NS - 1; s++) {
            float2 acc = make_float2(0.0f, 0.0f);
            for (int c = 0; c < RD_NC; c++) acc = cadd(acc, cmul(sym[s][c], ld2(tab->Winv[c], tid)));
            const float2 v = pa_limit(make_float2(acc.x * tab->pilot_gain, acc.y * tab->pilot_gain));
            out[(2 + s) * RD_SYM + RD_NCP + tid] = v;
            if (tid >= RD_M - RD_NCP) out[(2 + s) * RD_SYM + tid - (RD_M - RD_NCP)] = v;
        }
    }
}
extern "C" int rd_launch_eoo_build(const rd_tables *tab, const float *bits, float *eoo, int B, rd_stream_t s)
{
    hipLaunchKernelGGL(k_eoo_build, dim3(B), dim3(192), 0, (hipStream_t)s, tab, bits, (float2 *)eoo);
    return (int)hipGetLastError();
}
__global__ void k_copy_eoo(const float2 *eoo, float2 *out, long stride)
{
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < RD_NEOO; i += blockDim.x) out[(size_t)b * stride + i] = eoo[(size_t)b * RD_NEOO + i];
}
extern "C" int rd_launch_copy_eoo(const float *eoo, void *out, long stride, int B, rd_stream_t s)
{
    hipLaunchKernelGGL(k_copy_eoo, dim3(B), dim3(256), 0, (hipStream_t)s, (const float2 *)eoo, (float2 *)out, stride);
    return (int)hipGetLastError();
}

// =====================================================================================================
// channel simulator
// =====================================================================================================
#define CH_NCH 64   // partial-sum chunks per stream (fixed => deterministic reduction order)

__device__ __forceinline__ float2 chan_mp(const float2 *tx, const float2 *G, int i)
{
    if (!G) return tx[i];
    float2 v = cmul(tx[i], G[2 * i]);
    if (i >= 16) v = cadd(v, cmul(tx[i - 16], G[2 * (i - 16) + 1]));
    return v;
}

__global__ __launch_bounds__(256) void k_chan_power(rd_chan_args a, double *part)
{
    __shared__ double red[2][256];
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const float2 *tx = (const float2 *)a.tx + (size_t)b * a.tx_stride;
    const float2 *G = a.G ? (const float2 *)a.G + (size_t)b * a.n_sig * 2 : nullptr;
    const int per = (a.n_sig + CH_NCH - 1) / CH_NCH;
    const int lo = ch * per, hi = min(a.n_sig, lo + per);
    double s0 = 0.0, s1 = 0.0;
    for (int i = lo + tid; i < hi; i += 256) {
        const float2 x = tx[i], m = chan_mp(tx, G, i);
        const float ax = hypotf(x.x, x.y), am = hypotf(m.x, m.y);
        s0 += (double)(ax * ax); s1 += (double)(am * am);
    }
    red[0][tid] = s0; red[1][tid] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) { red[0][tid] += red[0][tid + w]; red[1][tid] += red[1][tid + w]; } __syncthreads(); }
    if (tid == 0) { part[((size_t)b * CH_NCH + ch) * 2] = red[0][0]; part[((size_t)b * CH_NCH + ch) * 2 + 1] = red[1][0]; }
}

// Philox4x32-10 counter-based generator (Salmon et al., SC'11) -> two complex N(0,1/2)+jN(0,1/2) samples
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float2 gauss_pair(uint32_t u0, uint32_t u1)
{   // Box-Muller, unit variance per component, on the hardware log2 / sqrt / sin / cos units (v_sin_f32 and v_cos_f32 take their argument in
    // revolutions: exactly what the second uniform is): this generator only feeds the device-noise path (seed != 0: benchmark and
    // statistics runs; parity tests pass an explicit noise tensor), where the libm versions were most of k_chan_apply's instructions
    const float a = ((float)u0 + 0.5f) * (1.0f / 4294967296.0f), bq = ((float)u1 + 0.5f) * (1.0f / 4294967296.0f);
    const float rad = __builtin_amdgcn_sqrtf(-1.38629436112f * __builtin_amdgcn_logf(a));       // -2 ln a = -2 ln 2 log2 a
    return make_float2(rad * __builtin_amdgcn_cosf(bq), rad * __builtin_amdgcn_sinf(bq));
}

__device__ __forceinline__ double chan_phase_acc(int i, float f0, float df_dt)
{   // sum_{k<=i} omega_k, omega_k = float32(freq_k*2*pi/Fs) summed in double (torch.cumsum on CPU)
    if (df_dt == 0.0f) { const float om = ((f0 * 2.0f) * (float)PI_D) / 8000.0f; return (double)(i + 1) * (double)om; }
    const double n = (double)(i + 1);
    return (2.0 * PI_D / 8000.0) * (n * (double)f0 + ((double)df_dt / 8000.0) * 0.5 * (double)i * n);
}

// per stream, ahead of k_chan_apply: the power-normalising gain and the phase the frequency offset has reached at the end of the signal, from the
// stream's partial power sums (added in their fixed order).  As a prologue of every k_chan_apply workgroup -- one thread, a hundred dependent
// additions, powf and a double-precision sincos while 255 threads wait -- this was a third of that kernel's time.
__global__ __launch_bounds__(64) void k_chan_gain(rd_chan_args a, const double *part, int n_part, float *gf)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    double p0 = 0.0, p1 = 0.0;
    for (int c = 0; c < n_part; c++) { p0 += part[((size_t)b * n_part + c) * 2]; p1 += part[((size_t)b * n_part + c) * 2 + 1]; }
    const float tx_power = (float)(p0 / a.n_sig), mp_power = (float)(p1 / a.n_sig);
    float2 fin = make_float2(1.0f, 0.0f);
    if (a.freq_offset != 0.0f && a.n_sig > 0) { float sn, cs; sincosf((float)chan_phase_acc(a.n_sig - 1, a.freq_offset, a.df_dt), &sn, &cs); fin = make_float2(cs, sn); }
    gf[4 * b] = a.G ? powf(tx_power / mp_power, 0.5f) : 1.0f; gf[4 * b + 1] = fin.x; gf[4 * b + 2] = fin.y;
}

__global__ __launch_bounds__(256) void k_chan_apply(rd_chan_args a, const float *gf)
{
    const int b = blockIdx.y;
    const int n_eoo = a.with_eoo ? RD_NEOO : 0;
    const int n_total = a.n_pre + a.n_sig + n_eoo + a.n_post;
    const float gain = gf[4 * b]; const float2 fin = make_float2(gf[4 * b + 1], gf[4 * b + 2]);
    const float2 *tx = (const float2 *)a.tx + (size_t)b * a.tx_stride;
    const float2 *G = a.G ? (const float2 *)a.G + (size_t)b * a.n_sig * 2 : nullptr;
    const float2 *noise = a.noise ? (const float2 *)a.noise + (size_t)b * n_total : nullptr;
    const float2 *eoo = (const float2 *)a.eoo + (size_t)b * RD_NEOO;
    float2 *rx = (float2 *)a.rx + (size_t)b * a.rx_stride;
    // two consecutive samples per thread: one Philox4x32 call yields the four uniforms of both (the generator and the Box-Muller
    // transcendentals, not the bytes, are what this kernel's time is made of), and a thread's store is 16 bytes
    auto sample = [&](int j, uint32_t u0, uint32_t u1) -> float2 {
        float2 v = make_float2(0.0f, 0.0f);
        bool real_noise = true;
        const int i = j - a.n_pre;
        if (i >= 0 && i < a.n_sig) {
            real_noise = false;
            const float2 m = a.mp ? ((const float2 *)a.mp)[(size_t)b * a.n_sig + i] : chan_mp(tx, G, i);
            v = make_float2(m.x * gain, m.y * gain);
            if (a.freq_offset != 0.0f) { float sn, cs; sincosf((float)chan_phase_acc(i, a.freq_offset, a.df_dt), &sn, &cs); v = cmul(v, make_float2(cs, sn)); }
        } else if (i >= a.n_sig && i < a.n_sig + n_eoo) {
            real_noise = false;
            const int e = i - a.n_sig;
            float sn, cs; sincosf((float)chan_phase_acc(e, a.freq_offset, a.df_dt), &sn, &cs);
            v = cmul(cmul(eoo[e], make_float2(cs, sn)), fin);
        }
        if (noise) { v.x += a.sigma * noise[j].x; v.y += a.sigma * noise[j].y; }
        else if (a.seed) {
            const float2 g = gauss_pair(u0, u1);
            if (real_noise) v.x += a.sigma * g.x;                                  // inference.py:277-284: real-valued randn
            else { v.x += a.sigma * 0.70710678f * g.x; v.y += a.sigma * 0.70710678f * g.y; }   // complex randn: 1/2 per component
        }
        if (a.sine_amp != 0.0f) {                                                  // inference.py:285-288, phase taken mod 1 cycle in double
            const double cyc = (double)j * (double)a.sine_freq / 8000.0;
            float sn, cs; sincosf((float)(6.283185307179586 * (cyc - floor(cyc))), &sn, &cs);
            v.x += a.sine_amp * cs; v.y += a.sine_amp * sn;
        }
        return make_float2(v.x * a.rx_gain, v.y * a.rx_gain);
    };
    const int n_pairs = (n_total + 1) >> 1;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += gridDim.x * 256) {
        uint32_t r[4] = { 0u, 0u, 0u, 0u };
        if (!noise && a.seed) philox4x32((uint32_t)p, (uint32_t)b, 0u, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
        const int j = 2 * p;
        const float2 v0 = sample(j, r[0], r[1]);
        if (j + 1 < n_total) {
            const float2 v1 = sample(j + 1, r[2], r[3]);
            if (((uintptr_t)rx & 15) == 0) *(f32x4 *)&rx[j] = (f32x4){ v0.x, v0.y, v1.x, v1.y };
            else { rx[j] = v0; rx[j + 1] = v1; }
        } else rx[j] = v0;
    }
}

// Watterson / Doppler-spread samples (doppler_spread.m:7-50, multipath_samples.m:25-31): one workgroup per stream.
// Low-rate noise -> FIR (double) into LDS, then two sweeps over the Fs-rate interpolation: variance, scaled write.
#define DG_MAXLOW 2048
// ybuf: [B][2][n_low] double2 in HBM for sequences of more than DG_MAXLOW low-rate points (lmr60: 500 low-rate points per second), else NULL (LDS)
__global__ __launch_bounds__(256) void k_multipath_gen(const float *taps, int n_taps, int low_ratio, int n_out, const float2 *noise_low,
                                                       unsigned long long seed, float2 *G, double2 *ybuf)
{
    __shared__ double2 ylds[2][DG_MAXLOW];
    __shared__ double red[256][6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n_low = max((n_out + low_ratio - 1) / low_ratio, 2), n_x = n_low + n_taps;
    double2 *y = ybuf ? ybuf + (size_t)b * 2 * n_low : &ylds[0][0];
    const int ys = ybuf ? n_low : DG_MAXLOW;
    for (int idx = tid; idx < 2 * n_low; idx += 256) {
        const int p = idx / n_low, i = idx - p * n_low;
        double ar = 0.0, ai = 0.0;
        for (int k = 0; k < n_taps; k++) {                         // np.convolve(x, b)[ntaps:][i] = sum_k b[k] x[i + ntaps - k]
            const int xi = i + n_taps - k;
            float2 x;
            if (noise_low) x = noise_low[((size_t)b * 2 + p) * n_x + xi];
            else { uint32_t r[4]; philox4x32((uint32_t)xi, (uint32_t)(b * 2 + p), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r); x = gauss_pair(r[0], r[1]); }
            ar += (double)taps[k] * x.x; ai += (double)taps[k] * x.y;
        }
        y[p * ys + i] = make_double2(ar, ai);
    }
    __threadfence_block();
    __syncthreads();
    auto interp = [&](int p, int n) {                               // linear interpolation, extrapolating past the last low-rate point
        const double pos = (double)n / (double)low_ratio;
        const int i0 = min((int)pos, n_low - 2);
        const double fr = pos - (double)i0;
        const double2 a0 = y[p * ys + i0], a1 = y[p * ys + i0 + 1];
        return make_double2(a0.x + (a1.x - a0.x) * fr, a0.y + (a1.y - a0.y) * fr);
    };
    double s[6] = { 0, 0, 0, 0, 0, 0 };                             // per path: sum re, sum im, sum |g|^2
    for (int n = tid; n < n_out; n += 256)
        for (int p = 0; p < 2; p++) { const double2 g = interp(p, n); s[3 * p] += g.x; s[3 * p + 1] += g.y; s[3 * p + 2] += g.x * g.x + g.y * g.y; }
    for (int k = 0; k < 6; k++) red[tid][k] = s[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) for (int k = 0; k < 6; k++) red[tid][k] += red[tid + off][k]; __syncthreads(); }
    double var = 0.0;
    for (int p = 0; p < 2; p++) { const double mr = red[0][3 * p] / n_out, mi = red[0][3 * p + 1] / n_out; var += red[0][3 * p + 2] / n_out - (mr * mr + mi * mi); }
    const double hf_gain = 1.0 / sqrt(var);                         // np.var: population variance of the complex samples
    float2 *Gb = G + (size_t)b * n_out * 2;
    for (int n = tid; n < n_out; n += 256) {
        const double2 g1 = interp(0, n), g2 = interp(1, n);
        Gb[2 * n] = make_float2((float)(hf_gain * g1.x), (float)(hf_gain * g1.y));
        Gb[2 * n + 1] = make_float2((float)(hf_gain * g2.x), (float)(hf_gain * g2.y));
    }
}
extern "C" int rd_multipath_gen_needs_scratch(int low_ratio, int n_out) { return low_ratio >= 1 && (n_out + low_ratio - 1) / low_ratio > DG_MAXLOW; }
extern "C" int rd_launch_multipath_gen(const float *taps_dev, int n_taps, int low_ratio, int n_out, const void *noise_low, unsigned long long seed, void *G, void *ybuf, int B, rd_stream_t s)
{
    if (B <= 0 || n_out <= 0) return 0;
    if (low_ratio < 1 || n_taps < 1 || (!ybuf && (n_out + low_ratio - 1) / low_ratio > DG_MAXLOW)) return -1;
    hipLaunchKernelGGL(k_multipath_gen, dim3(B), dim3(256), 0, (hipStream_t)s, taps_dev, n_taps, low_ratio, n_out, (const float2 *)noise_low, seed, (float2 *)G, (double2 *)ybuf);
    return (int)hipGetLastError();
}

// Rate-Rs channel matrix from the rate-Fs Doppler samples (multipath_samples.m:33-40, :73-80): H[t][c] = G1[t M] + G2[t M] exp(-j 2 pi c d Rs), M = Fs / Rs
// (hf_gain is already in G); magnitudes (the default `.f32` form, what BBFM.forward and the rate-Rs model take) or complex.
__global__ void k_multipath_h(const float2 *G, int n_g, int M, int n_sym, int Nc, float dRs, int want_complex, float *H)
{
    const int b = blockIdx.y;
    const float2 *Gb = G + (size_t)b * n_g * 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)n_sym * Nc; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / Nc), c = (int)(i - (long)t * Nc);
        const float2 g1 = Gb[2 * (size_t)t * M], g2 = Gb[2 * (size_t)t * M + 1];
        float sn, cs;
        sincosf(-6.283185307179586f * (float)c * dRs, &sn, &cs);
        const float hr = g1.x + g2.x * cs - g2.y * sn, hi = g1.y + g2.x * sn + g2.y * cs;
        if (want_complex) { H[2 * ((size_t)b * n_sym * Nc + i)] = hr; H[2 * ((size_t)b * n_sym * Nc + i) + 1] = hi; }
        else H[(size_t)b * n_sym * Nc + i] = sqrtf(hr * hr + hi * hi);
    }
}
extern "C" int rd_launch_multipath_h(const void *G, int n_g, int M, int n_sym, int Nc, float dRs, int want_complex, float *H, int B, rd_stream_t s)
{
    if (B <= 0 || n_sym <= 0) return 0;
    if (M < 1 || Nc < 1 || (long)(n_sym - 1) * M >= n_g) return -1;
    int gx = (int)(((long)n_sym * Nc + 255) / 256); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_multipath_h, dim3(gx, B), dim3(256), 0, (hipStream_t)s, (const float2 *)G, n_g, M, n_sym, Nc, dRs, want_complex, H);
    return (int)hipGetLastError();
}

// Symbol-domain channels of the non-OFDM configurations.
//  mode 0 (rate-Rs, radae.py:604-634, bottleneck 1): QPSK symbol k = (z[2k], z[2k+1]) * H[k] + sigma * CN(0,1)
//  mode 1 (BBFM, bbfm.py:157-197): per real symbol, FM demodulator SNR from the carrier-to-noise ratio:
//          CNRdB = 20log10(H)+CNR ; SNRdB = relu(CNR-12)+12+Gfm - relu(12-CNR)(1+Gfm/3) ; z_hat = clamp(z + N(0,1)/sqrt(SNR))
__global__ void k_chan_symbol(const float *z, const float *H, const float *noise, float *out, long n_real, int mode, float p0, float p1, unsigned long long seed)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_real; i += (long)gridDim.x * blockDim.x) {
        float nz;
        if (noise) nz = noise[i];
        else if (seed) { uint32_t r[4]; philox4x32((uint32_t)(i >> 1), (uint32_t)((i >> 1) >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r); const float2 g = gauss_pair(r[0], r[1]); nz = (i & 1) ? g.y : g.x; if (mode == 0) nz *= 0.70710678f; }
        else nz = 0.0f;
        float v;
        if (mode == 0) { const float h = H ? H[i >> 1] : 1.0f; v = z[i] * h + p0 * nz; }            // p0 = sigma; complex noise: 1/2 per component (explicit tensors already are)
        else {
            const float h = H ? H[i] : 1.0f;
            const float cnr = 20.0f * log10f(h) + p0;                                                 // p0 = CNRdB, p1 = Gfm
            float snr = fmaxf(cnr - 12.0f, 0.0f) + 12.0f + p1;
            snr += -fmaxf(-(cnr - 12.0f), 0.0f) * (1.0f + p1 / 3.0f);
            const float sigma = 1.0f / powf(powf(10.0f, snr / 10.0f), 0.5f);
            v = fminf(fmaxf(z[i] + sigma * nz, -1.0f), 1.0f);
        }
        out[i] = v;
    }
}
extern "C" int rd_launch_chan_symbol(const float *z, const float *H, const float *noise, float *out, long n_real, int mode, float p0, float p1, unsigned long long seed, rd_stream_t s)
{
    if (n_real <= 0) return 0;
    int grid = (int)((n_real + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_chan_symbol, dim3(grid), dim3(256), 0, (hipStream_t)s, z, H, noise, out, n_real, mode, p0, p1, seed);
    return (int)hipGetLastError();
}

extern "C" int rd_launch_channel(const rd_chan_args *a, rd_stream_t s)
{
    if (a->B <= 0) return 0;
    hipStream_t st = (hipStream_t)s;
    float *gf = (float *)a->scratch;                        // scratch: [B][4] floats (gain, final phase), then the partial power sums
    double *part = (double *)a->scratch + 2 * (size_t)a->B;
    const int n_part = a->mp ? a->n_sig / RD_NMF : CH_NCH;
    if (!a->mp) hipLaunchKernelGGL(k_chan_power, dim3(CH_NCH, a->B), dim3(256), 0, st, *a, part);      // a->mp: the modulator left mp and its per-frame power sums (k_ofdm_mod_mp)
    hipLaunchKernelGGL(k_chan_gain, dim3((a->B + 63) / 64), dim3(64), 0, st, *a, (const double *)part, n_part, gf);
    const int n_total = a->n_pre + a->n_sig + (a->with_eoo ? RD_NEOO : 0) + a->n_post;
    int gx = (n_total + 255) / 256; if (gx > 32) gx = 32;          // (16..32 workgroups per stream measure the same; 64: +4 %, 8: +13 %)
    hipLaunchKernelGGL(k_chan_apply, dim3(gx, a->B), dim3(256), 0, st, *a, (const float *)gf);
    return (int)hipGetLastError();
}

