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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "rade_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// two IEEE fused multiply-adds per lane in one instruction (v_pk_fma_f32: the full-rate f32 path of the vector ALU)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PI_D 3.14159265358979323846

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// e^{-j angle(c)} = conj(c)/|c| (np.exp(-1j*np.angle(c)) without atan2 / sincos); angle(0) = 0
__device__ __forceinline__ float2 unit_conj(float2 c)
{
#ifdef RD_NO_UNITCONJ
    { const float ang = atan2f(c.y, c.x); float sn, cs; sincosf(-ang, &sn, &cs); return make_float2(cs, sn); }
#endif
    const float n2 = c.x * c.x + c.y * c.y;
    if (n2 == 0.0f) return make_float2(1.0f, 0.0f);
    const float inv = 1.0f / sqrtf(n2);
    return make_float2(c.x * inv, -c.y * inv);
}
// (cos, sin) of a double angle: reduced to [-pi, pi] in double, evaluated in float (the results are used as float32)
__device__ __forceinline__ float2 cis_reduced(double ang)
{
#ifdef RD_NO_CIS
    double sd, cd; sincos(ang, &sd, &cd); return make_float2((float)cd, (float)sd);
#else
    const double r = ang - 6.283185307179586476925 * rint(ang * 0.15915494309189533577);
    float sn, cs; sincosf((float)r, &sn, &cs);
    return make_float2(cs, sn);
#endif
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// thread index through an opaque asm: inside the receiver's per-call loop this keeps the compiler from hoisting every
// thread-derived address computation of every phase out of the loop (hundreds of registers live across all phases)
__device__ __forceinline__ int rx_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
// lane exchange inside a quad on the DPP path (v_mov_b32_dpp quad_perm): __shfl / __shfl_xor go through ds_bpermute, an
// LDS-pipe round trip on the serial chain of the recurrences
template <int CTRL> __device__ __forceinline__ float quad_dpp(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
#define QUAD_XOR1 0xB1   /* [1,0,3,2] */
#define QUAD_XOR2 0x4E   /* [2,3,0,1] */
#define QUAD_BC0  0x00   /* [0,0,0,0] */
#define QUAD_BC1  0x55
#define QUAD_BC2  0xAA
#define ROW_ROR4  0x124  /* rotate right by 4 inside each row of 16 lanes */
#define ROW_ROR8  0x128
template <int CTRL> __device__ __forceinline__ int quad_dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)quad_dpp_i<CTRL>((int)(unsigned)b), hi = (unsigned)quad_dpp_i<CTRL>((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the wavefront, the same value in every lane: quad and row steps on DPP, the four row totals through readlane
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<QUAD_XOR1>(v); v += dpp_f64<QUAD_XOR2>(v); v += dpp_f64<ROW_ROR4>(v); v += dpp_f64<ROW_ROR8>(v);
    const long long b = __double_as_longlong(v);
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 16 * r), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 16 * r);
        t += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return t;
}
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += quad_dpp<QUAD_XOR1>(v); v += quad_dpp<QUAD_XOR2>(v); v += quad_dpp<ROW_ROR4>(v); v += quad_dpp<ROW_ROR8>(v);
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * r));
    return t;
}
// max over the wavefront of non-negative values (lanes DPP cannot reach read 0), the same value in every lane
__device__ __forceinline__ float wave_max_f32(float v)
{
    v = fmaxf(v, quad_dpp<QUAD_XOR1>(v)); v = fmaxf(v, quad_dpp<QUAD_XOR2>(v)); v = fmaxf(v, quad_dpp<ROW_ROR4>(v)); v = fmaxf(v, quad_dpp<ROW_ROR8>(v));
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) t = fmaxf(t, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * r)));
    return t;
}
__device__ __forceinline__ float clamp1(float x) { return fminf(fmaxf(x, -1.0f), 1.0f); }
// sqrt(-ln(P / 5)) of the Rayleigh thresholds (dsp.py:221, 318-320), P = 1e-4 / 1e-5: correctly rounded doubles, i.e. what the
// reference's (and the oracle's) libm returns; the device log / sqrt on the single thread that sets the thresholds were a few hundred
// f64 instructions on the serial path of every call
#define RD_SQRT_NLOG_1EM4_5 3.2893431387452243
#define RD_SQRT_NLOG_1EM5_5 3.622480279781289
// gate activations of the GRU recurrences on the hardware exp2 / rcp units (about 1 ulp each): the recurrence is a
// serial chain, so the libm-grade expf / tanhf / IEEE division sequences would dominate every time step
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float gate_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }
__device__ __forceinline__ float2 ld2(const float (*p)[2], int i) { return make_float2(p[i][0], p[i][1]); }

#if defined(RD_PHASE_TIMING) && !defined(RADE_RX2_TU)
__device__ long long g_phase_cycles[32];
#define PH_T0() long long ph_t_ = clock64()
#define PH(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { long long n_ = clock64(); atomicAdd((unsigned long long *)&g_phase_cycles[i], (unsigned long long)(n_ - ph_t_)); ph_t_ = n_; } } while (0)
extern "C" void rd_debug_phase_cycles(long long *out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), sizeof(long long) * 32); long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z); }
#else
#define PH_T0() do { } while (0)
#define PH(i) do { } while (0)
#endif

#ifndef RADE_RX2_TU   // rade_rx2.hip includes this file for its device helpers only: kernels and launch shims stay in this translation unit
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
template <int NT, int RT, bool SINGLE>
__global__ __launch_bounds__(64) void k_gemm16p(rd_gemm_args a)
{
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
#pragma unroll
    for (int q = 0; q < RT; q++)
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int col = (nt0 + i) * 32 + (lane & 31);
            if (col >= a.N) continue;
            const float bias = a.bias ? a.bias[col] : 0.0f;
            const float scl = SINGLE ? a.Wscale[col] * 0x1p-8f : 0x1p-18f;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int rr = r0 + 32 * q + (j & 3) + 8 * (j >> 2) + 4 * half;
                if (rr >= rows) continue;
                const int bb = rr / a.T, tt = rr - bb * a.T;
                if (a.n_rows && tt >= a.n_rows[bb]) continue;
                float v = acc[q][i][j] * scl + bias;
                if (a.act == 1) v = clamp1(gate_tanh(v));
                else if (a.act == 2) v = clamp1(a.a1[bb * a.a1_sb + tt * a.a1_st + col] * gate_sigmoid(v));
                a.y[bb * a.y_sb + tt * a.y_st + col] = v;
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
            if (a->Wscale) { dim3 g1(gx, ntt / 3); hipLaunchKernelGGL((k_gemm16p<3, 1, true>), g1, block, 0, st, *a); return (int)hipGetLastError(); }
            dim3 grid(gx2, ntt / 3);
            hipLaunchKernelGGL((k_gemm16p<3, 2, false>), grid, block, 0, st, *a);
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
    const float *gi = a.gi + (size_t)b * a.gi_sb + (p < 3 ? p * H + j : j);   // lane part p < 3 fetches gate p of unit j
    // gi is fetched four steps at a time, one block ahead: a load issued inside the step that consumes an older one makes
    // the compiler wait for all of them (vmcnt(0)), i.e. one L2 round trip per step on the serial chain
    float gcur[4], gnxt[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { gcur[u] = u < Tb ? gi[(size_t)u * a.gi_st] : 0.0f; gnxt[u] = 4 + u < Tb ? gi[(size_t)(4 + u) * a.gi_st] : 0.0f; }
    __syncthreads();
    int cur = 0;
    for (int t0 = 0; t0 < Tb; t0 += 4) {
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
            float sr = 0.0f, sz = 0.0f, sn = 0.0f;
            const float *hp = hs[cur] + p * KP;
#pragma unroll
            for (int k = 0; k < KP; k += 4) {
                const f32x4 hv = *(const f32x4 *)(hp + k);
#pragma unroll
                for (int q = 0; q < 4; q++) { sr += wr[k + q] * hv[q]; sz += wz[k + q] * hv[q]; sn += wn[k + q] * hv[q]; }
            }
            sr += quad_dpp<QUAD_XOR1>(sr); sz += quad_dpp<QUAD_XOR1>(sz); sn += quad_dpp<QUAD_XOR1>(sn);
            sr += quad_dpp<QUAD_XOR2>(sr); sz += quad_dpp<QUAD_XOR2>(sz); sn += quad_dpp<QUAD_XOR2>(sn);
            const float g0 = gcur[u];
            const float gr = quad_dpp<QUAD_BC0>(g0), gz = quad_dpp<QUAD_BC1>(g0), gn = quad_dpp<QUAD_BC2>(g0);
            const float r = gate_sigmoid((sr + br) + gr);
            const float z = gate_sigmoid((sz + bz) + gz);
            const float n = gate_tanh(gn + (sn + bn) * r);
            hj = (hj - n) * z + n;
            if (p == 0) {
                hs[cur ^ 1][j] = hj;
                a.out[(size_t)b * a.out_sb + (size_t)t * a.out_st + j] = clamp1(hj);
            }
            cur ^= 1;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 4; u++) gcur[u] = gnxt[u];         // loaded a whole block ago: no stall
#pragma unroll
        for (int u = 0; u < 4; u++) gnxt[u] = t0 + 8 + u < Tb ? gi[(size_t)(t0 + 8 + u) * a.gi_st] : 0.0f;
    }
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

#endif  // !RADE_RX2_TU
// =====================================================================================================
// Per-stream decoder stage of the receiver kernel: the whole DenseNet stack for one stream's pending rows, run by the
// stream's own workgroup of eight wavefronts with every activation RESIDENT IN LDS.  Only the weights stream in (from L2);
// nothing the stage produces goes through HBM except the 84-float output rows and the one conv-history row.
//
//  * rows are processed in chunks of DQ_ROWS = 24 (the eight modem frames between two unique-word checks);
//  * the DenseNet rows x[t][0..735] live as two binary16 planes (x * 2^8 = hi + lo, 22 bits), written once by each layer's
//    epilogue, so the matrix-core operands are read from LDS ready-made (no conversion in the product loops).  A row is 96
//    blocks of 8 halfs (92 used, 1536 B = a multiple of the 256-byte bank width) and block c of logical row t sits at
//    c ^ (t & 15): the 16-byte operand reads of v_mfma_f32_16x16x32_f16 (lane = row t, k-slice c) are bank-conflict free;
//  * products run as v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand (16 output columns) and the rows as the B
//    operand (two 16-row tiles share every weight fragment), three products per k-step (lo*hi + hi*lo + hi*hi).  One
//    wavefront owns a column tile over the whole K: no split-K, no reduction scratch, no barrier inside a product;
//  * narrow layers would leave wavefronts idle (conv: 32 columns = 2 tiles), so each conv runs in the same phase as the
//    part of the NEXT layer's input projection that does not need its output (K = the columns already final); a one-k-step
//    fix-up product then adds the 32 new columns.  The 84-float output layer is treated the same way behind the last conv.
// =====================================================================================================
#define NT_RX 512                // threads of the receiver workgroup (k_rx_sync)
#define DQ_ROWS 24
#define DQ_XB 96                 // 8-half blocks per x row
#define DQ_HB 16                 // blocks per GRU-output row (12 used)
#define DQ_PEND_MAX 64           // pending rows a stream can hold (engine: dec_rows <= 63)
struct DecShared {
    __attribute__((aligned(16))) _Float16 xh[DQ_ROWS + 2][DQ_XB * 8];    // physical row 0: conv history (the row before this chunk), 1..24: the chunk, 25: zeros
    __attribute__((aligned(16))) _Float16 xl[DQ_ROWS + 2][DQ_XB * 8];
    __attribute__((aligned(16))) float gi[DQ_ROWS][288];                 // GRU input projections of the layer being scanned; staging of the output layer
    __attribute__((aligned(16))) _Float16 hbh[DQ_ROWS][DQ_HB * 8], hbl[DQ_ROWS][DQ_HB * 8];   // clamp(h_t) of the layer just scanned
    __attribute__((aligned(16))) float hs[2][96];
    int rst[DQ_PEND_MAX];
    int err[DQ_PEND_MAX];                    // receiver: aux-bit (UW) decisions of the decoded rows
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// half index of x[logical row t][col] inside a plane; the history row is logical -1 (swizzle key 15), the zero row needs no key
__device__ __forceinline__ int dq_xoff(int t, int col) { return (t + 1) * (DQ_XB * 8) + ((((col >> 3) ^ (t & 15)) << 3) | (col & 7)); }
__device__ __forceinline__ int dq_hoff(int t, int col) { return t * (DQ_HB * 8) + ((((col >> 3) ^ (t & 15)) << 3) | (col & 7)); }
// v in [-1, 1] -> the two planes of 2^8 v
__device__ __forceinline__ void dq_split(float v, _Float16 &hi, _Float16 &lo) { const float x = 256.0f * v; hi = (_Float16)x; lo = (_Float16)(x - (float)hi); }

enum { DQ_OUT_X = 0, DQ_OUT_GI = 1, DQ_OUT_GLOBAL = 2 };
struct DqGemm {
    const unsigned short *wa; int nct;     // rd_pack_weights_f16x2_a16: [K/32][nct][2 planes][64 lanes][8]
    const float *bias; int N;              // bias may be null; N = valid output columns
    const float *wscale;                   // non-null: int8-exact layer, ONE plane of integers, wscale[n] = the column's scale; null: two planes of 2^10 w
    int from_hb;                           // B operand: 0 = the x planes, 1 = the GRU-output planes
    int ktap;                              // k-steps [0, ktap) read the PREVIOUS row (conv tap 0), the rest the row itself
    int ks0, nks;                          // k-steps of the weight's K axis this product covers
    int init_gi;                           // accumulators start from gi[t][n] (fix-up products) instead of zero
    int out, ocol, act;                    // DQ_OUT_*; first x column (DQ_OUT_X); act 0 none, 1 tanh+clamp, 2 GLU
    float *gout; int gstride;              // DQ_OUT_GLOBAL
};

// one column tile (16 outputs) of a product for rows [0, Tb): see the notes above.  D k-steps of weights are in flight.
// A real function (one copy, its own register allocation, five call sites): arguments arrive in vector registers and as
// generic pointers, so everything wave-uniform is moved to scalar registers first and the pointers get their address spaces
// back -- otherwise the weight loads become flat_load, which count on BOTH wait counters: every LDS wait would then also
// wait for the weight prefetch it is supposed to run under.
typedef __attribute__((address_space(3))) _Float16 lds_half;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(1))) const unsigned short glb_u16;
typedef __attribute__((address_space(1))) const float glb_cf32;
typedef __attribute__((address_space(1))) float glb_f32;
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ T *uni_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    return (T *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
}
#ifndef DQ_TILE_INLINE
#define DQ_TILE_INLINE __forceinline__
#endif
// NT adjacent column tiles (16 outputs each) of a product for rows [0, Tb), one wavefront, the whole K: every row fragment read
// from LDS feeds NT weight fragments, D k-steps of weights (NT KB per plane each) are in flight, and the pipeline fills once per
// call -- with one tile per call the first round trip to L2 of every tile was most of its time.
template <int NT, bool SINGLE>
__device__ DQ_TILE_INLINE void dq_gemm_tiles_(DecShared *sh_, const DqGemm g_, int ct_, int Tb_, unsigned rstmask_)
{   // SINGLE (int8-exact layer: one plane of integers) is a template parameter and both 16-row tiles are always computed: with
    // either as a run-time flag every matrix instruction of the k loop sat behind a (uniform) branch, and the compiler, unable to
    // count outstanding loads across branches, waited for ALL weight fragments in flight before each k-step (vmcnt(0)): the
    // weight pipeline was one k-step deep whatever D said (296 cycles per k-step for 64 cycles of matrix work, tools/ubench/dq_gemm_bench)
    constexpr int D = NT == 1 ? 12 : 4;                                // k-steps of weights in flight
    const int ct = uni(ct_), Tb = uni(Tb_); const unsigned rstmask = (unsigned)uni((int)rstmask_);
    const int nct = uni(g_.nct), N = uni(g_.N), from_hb = uni(g_.from_hb), ktap = uni(g_.ktap), ks0 = uni(g_.ks0), nks = uni(g_.nks), init_gi = uni(g_.init_gi),
              outk = uni(g_.out), ocol = uni(g_.ocol), act = uni(g_.act), gstride = uni(g_.gstride);
    DecShared *sh = uni_ptr(sh_);
    glb_u16 *wbase = (glb_u16 *)uni_ptr(g_.wa); glb_cf32 *biasp = (glb_cf32 *)uni_ptr(g_.bias); glb_f32 *gout = (glb_f32 *)uni_ptr(g_.gout);
    glb_cf32 *wscale = (glb_cf32 *)uni_ptr(g_.wscale);
    constexpr bool single = SINGLE;
    const int lane = threadIdx.x & 63, t = lane & 15, gq = lane >> 4;
    constexpr bool two = true;                                         // Tb <= 16: the second tile repeats the last row, results dropped
    const int r0 = min(t, Tb - 1), r1 = min(16 + t, Tb - 1);          // rows beyond Tb repeat the last one (results dropped)
    const lds_half *bh = (const lds_half *)(from_hb ? &sh->hbh[0][0] : &sh->xh[0][0]), *bl = (const lds_half *)(from_hb ? &sh->hbl[0][0] : &sh->xl[0][0]);
    const int stride = from_hb ? DQ_HB * 8 : DQ_XB * 8;
    // tap 1: physical row 1 + r (x) / r (hb), key r & 15;  tap 0: physical row r = logical r - 1, key (r - 1) & 15, or the zero row
    const int p1a = (from_hb ? r0 : r0 + 1) * stride, p1b = (from_hb ? r1 : r1 + 1) * stride, k1a = r0 & 15, k1b = r1 & 15;
    const int p0a = ((rstmask >> r0) & 1u) ? (DQ_ROWS + 1) * stride : r0 * stride, p0b = ((rstmask >> r1) & 1u) ? (DQ_ROWS + 1) * stride : r1 * stride;
    const int k0a = (r0 - 1) & 15, k0b = (r1 - 1) & 15;
    const int planes = single ? 1 : 2;
    glb_u16 *wa = wbase + (((size_t)ks0 * nct + ct) * planes * 64 + lane) * 8;
    const size_t wstep = (size_t)nct * planes * 64 * 8, tstep = (size_t)planes * 64 * 8;
    lds_f32 *gi = (lds_f32 *)&sh->gi[0][0];
    f32x4 acc0[NT], acc1[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) { acc0[i] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; acc1[i] = acc0[i]; }
    const int n0 = 16 * ct + 4 * gq;                                   // this lane's four output columns of tile 0 (tile i: + 16 i)
    // bias and column scales: fetched now, used after the k loop (a load issued in the epilogue is one more exposed round trip per call)
    f32x4 bias[NT], scl[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        bias[i] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; scl[i] = (f32x4){ 0x1p-18f, 0x1p-18f, 0x1p-18f, 0x1p-18f };      // two planes: 2^8 (rows) x 2^10 (weights)
        if (biasp && !init_gi) {
#pragma unroll
            for (int r = 0; r < 4; r++) bias[i][r] = n0 + 16 * i + r < N ? biasp[n0 + 16 * i + r] : 0.0f;
        }
        if (single) scl[i] = *(const __attribute__((address_space(1))) f32x4 *)(wscale + n0 + 16 * i) * 0x1p-8f;    // integers x column scale, rows carry 2^8
    }
    typedef const __attribute__((address_space(1))) f16x8 glb_f16x8;
    typedef const __attribute__((address_space(3))) f16x8 lds_f16x8;
    f16x8 wh[D][NT], wl[D][NT];
    auto fetch = [&](int d, int ks) {                                  // k-steps past the end re-read the last one; their products are skipped
        const int kq = min(ks, nks - 1);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            wh[d][i] = *(glb_f16x8 *)(wa + kq * wstep + i * tstep);
            if (!single) wl[d][i] = *(glb_f16x8 *)(wa + kq * wstep + i * tstep + 64 * 8);
        }
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, d);
    // row fragments (LDS) one k-step ahead of their products
    f16x8 nha, nla, nhb, nlb;
    auto rows = [&](int kidx) {                                        // kidx past the end re-reads the last k-step
        const int kk = ks0 + min(kidx, nks - 1);
        const bool tap0 = kk < ktap;
        const int cb = 4 * (tap0 ? kk : kk - ktap) + gq;
        const int oa = (tap0 ? p0a : p1a) + ((cb ^ (tap0 ? k0a : k1a)) << 3), ob = (tap0 ? p0b : p1b) + ((cb ^ (tap0 ? k0b : k1b)) << 3);
        nha = *(lds_f16x8 *)(bh + oa); nla = *(lds_f16x8 *)(bl + oa);
        nhb = *(lds_f16x8 *)(bh + ob); nlb = *(lds_f16x8 *)(bl + ob);
    };
    rows(0);
    auto step = [&](int d, int kidx, bool refill) {
        const f16x8 xha = nha, xla = nla, xhb = nhb, xlb = nlb;
        rows(kidx + 1);
        __builtin_amdgcn_sched_barrier(0);          // the next k-step's four row fragments in flight before this one's first product (left
                                                    // alone, the scheduler reads them one at a time into the same registers: an LDS round trip per product)
#pragma unroll
        for (int i = 0; i < NT; i++) {
            if (!single) {
                acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[d][i], xha, acc0[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[d][i], xhb, acc1[i], 0, 0, 0);
            }
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[d][i], xla, acc0[i], 0, 0, 0);
            acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[d][i], xlb, acc1[i], 0, 0, 0);
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[d][i], xha, acc0[i], 0, 0, 0);
            acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[d][i], xhb, acc1[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (refill) fetch(d, kidx + D);
        __builtin_amdgcn_sched_barrier(0);
    };
    int ks = 0;
#pragma unroll 1
    for (; ks + D <= nks; ks += D) {                                   // whole groups: straight-line, every load counted
#pragma unroll
        for (int d = 0; d < D; d++) step(d, ks + d, true);
    }
#pragma unroll
    for (int d = 0; d < D; d++) if (ks + d < nks) step(d, ks + d, false);     // the last nks % D k-steps (their fragments are already here)
    // C layout: column = lane & 15 (row t), registers r = outputs n0 + r
    lds_half *xh = (lds_half *)&sh->xh[0][0], *xl = (lds_half *)&sh->xl[0][0];
    const lds_half *hbh = (const lds_half *)&sh->hbh[0][0], *hbl = (const lds_half *)&sh->hbl[0][0];
    typedef __attribute__((address_space(3))) f16x4 lds_f16x4;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int n = n0 + 16 * i;
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            const int tt = 16 * rt + t;
            if (tt >= Tb) continue;
            f32x4 v = (rt ? acc1[i] : acc0[i]) * scl[i] + bias[i];
            if (init_gi) v += *(const __attribute__((address_space(3))) f32x4 *)(gi + tt * 288 + n);          // fix-up product: onto the staged sums
            if (outk == DQ_OUT_GI) { *(__attribute__((address_space(3))) f32x4 *)(gi + tt * 288 + n) = v; continue; }
            if (outk == DQ_OUT_GLOBAL) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (n + r < N) gout[(size_t)tt * gstride + n + r] = v[r];
                continue;
            }
            if (act == 2) {                                             // GLU: x * sigmoid(W x), x = the GRU output of the same row / column
                const f16x4 hh = *(const lds_f16x4 *)(hbh + dq_hoff(tt, n)), hl = *(const lds_f16x4 *)(hbl + dq_hoff(tt, n));
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = clamp1(((float)hh[r] + (float)hl[r]) * 0x1p-8f * gate_sigmoid(v[r]));
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = clamp1(gate_tanh(v[r]));
            }
            f16x4 oh, ol;
#pragma unroll
            for (int r = 0; r < 4; r++) { _Float16 a, b; dq_split(v[r], a, b); oh[r] = a; ol[r] = b; }
            *(lds_f16x4 *)(xh + dq_xoff(tt, ocol + n)) = oh; *(lds_f16x4 *)(xl + dq_xoff(tt, ocol + n)) = ol;
        }
    }
}

template <int NT>
__device__ DQ_TILE_INLINE void dq_gemm_tiles(DecShared *sh, const DqGemm g, int ct, int Tb, unsigned rstmask)
{
    if (uni_ptr(g.wscale) != nullptr) dq_gemm_tiles_<NT, true>(sh, g, ct, Tb, rstmask);
    else dq_gemm_tiles_<NT, false>(sh, g, ct, Tb, rstmask);
}

// dense1 (K = 80 -> 96 columns) on the f32 matrix cores (v_mfma_f32_32x32x2_f32, weights from rd_pack_weights): its input
// z_hat = symbol / pilot magnitude is the one operand of the stack that is not bounded -- a deep fade or a false sync can push
// it past the +-255.9 the 2^8-scaled binary16 planes hold, an overflow there turns into inf - inf = NaN in the accumulators
// and poisons the GRU state until the next reset, where the reference computes a finite value that tanh squashes.
__device__ void dq_dense1(DecShared *sh, const float *z, const rd_lin w, int Tb)
{
    constexpr int NKB = RD_LATENT / 8;
    const int lane = threadIdx.x & 63, nt = threadIdx.x >> 6, half = lane >> 5;
    if (nt >= 3) return;
    const float *wp = w.wp + ((size_t)nt * 64 + lane) * 4;
    const size_t wstep = (size_t)3 * 256;
    const int col = nt * 32 + (lane & 31);
    const float bias = w.bias[col];
    const int t = min(lane & 31, Tb - 1);
    const float *p1 = z + (size_t)t * RD_LATENT + 4 * half;
    f32x4 av[NKB], bv[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) { av[kb] = *(const f32x4 *)(p1 + kb * 8); bv[kb] = *(const f32x4 *)(wp + kb * wstep); }
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NKB; kb++)
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb][s], bv[kb][s], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int tt = (j & 3) + 8 * (j >> 2) + 4 * half;
        if (tt >= Tb) continue;
        _Float16 a, b; dq_split(clamp1(gate_tanh(acc[j] + bias)), a, b);
        sh->xh[0][dq_xoff(tt, col)] = a; sh->xl[0][dq_xoff(tt, col)] = b;
    }
}

// GRU recurrence over Tb steps: four lanes per hidden unit (threads >= 384 only keep the barriers), gi and the outputs in LDS
__device__ void dq_scan(DecShared *sh, const float *Whh, const float *bhh, float *hstate, int Tb, unsigned rstmask)
{
    constexpr int H = 96, KP = H / 4;
    const int tid = rx_tid();
    const bool on = tid < 4 * H;
    const int j = on ? tid >> 2 : 0, p = tid & 3;
    f32x2 wr[KP / 2], wz[KP / 2], wn[KP / 2];            // weight pairs (k, k+1): the products run as v_pk_fma_f32, two per lane and instruction
    {
        const float *w0 = Whh + (size_t)j * H + p * KP;
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
            const f32x4 v0 = *(const f32x4 *)(w0 + k), v1 = *(const f32x4 *)(w0 + (size_t)H * H + k), v2 = *(const f32x4 *)(w0 + (size_t)2 * H * H + k);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                wr[k / 2 + u] = (f32x2){ v0[2 * u], v0[2 * u + 1] }; wz[k / 2 + u] = (f32x2){ v1[2 * u], v1[2 * u + 1] }; wn[k / 2 + u] = (f32x2){ v2[2 * u], v2[2 * u + 1] };
            }
        }
    }
    const float br = bhh[j], bz = bhh[H + j], bn = bhh[2 * H + j];
    float hj = hstate[j];
    if (on && p == 0) sh->hs[0][j] = hj;
    const float *gi = &sh->gi[0][0] + (p < 3 ? p * H + j : j);
    float g0 = gi[0];
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < Tb; t++) {
        if ((rstmask >> t) & 1u) {                     // uniform over the workgroup
            hj = 0.0f;
            __syncthreads();
            if (on && p == 0) sh->hs[cur][j] = 0.0f;
            __syncthreads();
        }
        const float g1 = gi[(size_t)min(t + 1, Tb - 1) * 288];           // next step's input: its LDS latency hides under this step
        // six independent accumulation chains (two per gate): the step is a chain of dependent instructions on a wavefront
        // that shares its SIMD with at most one other, so the length of the longest chain is the step's time
        f32x2 ar = { 0.0f, 0.0f }, az = { 0.0f, 0.0f }, an = { 0.0f, 0.0f }, ar2 = { 0.0f, 0.0f }, az2 = { 0.0f, 0.0f }, an2 = { 0.0f, 0.0f };
        const float *hp = sh->hs[cur] + p * KP;
#pragma unroll
        for (int k = 0; k < KP; k += 8) {
            const f32x4 hv = *(const f32x4 *)(hp + k), hw = *(const f32x4 *)(hp + k + 4);
            const f32x2 h0 = { hv[0], hv[1] }, h1 = { hv[2], hv[3] }, h2 = { hw[0], hw[1] }, h3 = { hw[2], hw[3] };
            ar = pk_fma(wr[k / 2], h0, ar); az = pk_fma(wz[k / 2], h0, az); an = pk_fma(wn[k / 2], h0, an);
            ar2 = pk_fma(wr[k / 2 + 1], h1, ar2); az2 = pk_fma(wz[k / 2 + 1], h1, az2); an2 = pk_fma(wn[k / 2 + 1], h1, an2);
            ar = pk_fma(wr[k / 2 + 2], h2, ar); az = pk_fma(wz[k / 2 + 2], h2, az); an = pk_fma(wn[k / 2 + 2], h2, an);
            ar2 = pk_fma(wr[k / 2 + 3], h3, ar2); az2 = pk_fma(wz[k / 2 + 3], h3, az2); an2 = pk_fma(wn[k / 2 + 3], h3, an2);
        }
        ar += ar2; az += az2; an += an2;
        float sr = ar[0] + ar[1], sz = az[0] + az[1], sn = an[0] + an[1];
        sr += quad_dpp<QUAD_XOR1>(sr); sz += quad_dpp<QUAD_XOR1>(sz); sn += quad_dpp<QUAD_XOR1>(sn);
        sr += quad_dpp<QUAD_XOR2>(sr); sz += quad_dpp<QUAD_XOR2>(sz); sn += quad_dpp<QUAD_XOR2>(sn);
        const float gr = quad_dpp<QUAD_BC0>(g0), gz = quad_dpp<QUAD_BC1>(g0), gn = quad_dpp<QUAD_BC2>(g0);
        const float r = gate_sigmoid((sr + br) + gr);
        const float z = gate_sigmoid((sz + bz) + gz);
        const float n = gate_tanh(gn + (sn + bn) * r);
        hj = (hj - n) * z + n;
        if (on && p == 0) {
            sh->hs[cur ^ 1][j] = hj;
            _Float16 a, b; dq_split(clamp1(hj), a, b);
            sh->hbh[0][dq_hoff(t, j)] = a; sh->hbl[0][dq_hoff(t, j)] = b;
        }
        g0 = g1;
        cur ^= 1;
        __syncthreads();
    }
    if (on && p == 0) hstate[j] = hj;
}

// all decoder layers for rows [0, Tb) (Tb <= DQ_ROWS) of stream b: z rows in, 84-float rows out; rstmask bit t = state reset before row t
__device__ void dq_layers(DecShared *sh, const rd_decs_args &a, int b, const float *z, float *out, int Tb, unsigned rstmask)
{
    const int tid = rx_tid(), wave = tid >> 6;
    PH_T0();
    {   // conv history (the row before this chunk) from HBM into physical row 0 (key 15); the zero row
        // (the slot holds the two planes as they were, one dword per column: a value re-split from their float sum could round to a
        // different (hi, lo) pair, and the dropped lo x lo term would then depend on how the rows were cut into chunks)
        const unsigned *hist = (const unsigned *)(a.x + (size_t)b * a.x_sb - RD_DEC_W);
        for (int c = tid; c < RD_DEC_W; c += NT_RX) {
            const unsigned u = hist[c];
            sh->xh[0][dq_xoff(-1, c)] = __builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); sh->xl[0][dq_xoff(-1, c)] = __builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
        }
        for (int c = tid; c < DQ_XB * 8; c += NT_RX) { sh->xh[DQ_ROWS + 1][c] = (_Float16)0.0f; sh->xl[DQ_ROWS + 1][c] = (_Float16)0.0f; }
    }
    dq_dense1(sh, z, a.dense1, Tb);
    __syncthreads();
    DqGemm g;
    // layer 0's input projection: K = 96
    g = (DqGemm){ a.gin[0].wa16, 18, a.gin[0].bias, 288, a.gin[0].wscale, 0, 0, 0, 3, 0, DQ_OUT_GI, 0, 0, nullptr, 0 };
    if (wave >= 2) dq_gemm_tiles<3>(sh, g, 3 * (wave - 2), Tb, rstmask);          // 18 column tiles = six wavefronts x three
    __syncthreads();
    PH(24);
#pragma unroll 1
    for (int l = 0; l < 5; l++) {
        const int in = 96 + 128 * l, cin = in + 96;      // radae_base.py:378-386
        dq_scan(sh, a.whh[l], a.bhh[l], a.h[l] + (size_t)b * 96, Tb, rstmask);
        PH(23);
        // GLU gates: K = 96 from the GRU-output planes, 6 column tiles
        g = (DqGemm){ a.glu[l].wa16, 6, nullptr, 96, a.glu[l].wscale, 1, 0, 0, 3, 0, DQ_OUT_X, in, 2, nullptr, 0 };
        if (wave < 6) dq_gemm_tiles<1>(sh, g, wave, Tb, rstmask);
        __syncthreads();
        PH(25);
        // conv (2 tiles, K = 2 cin, both taps) beside the next product's columns that are already final (K = cin):
        // the next layer's input projection (18 tiles) or, behind the last conv, the output layer (6 tiles, staged in gi)
        const DqGemm gc = (DqGemm){ a.conv[l].wa16, 2, a.conv[l].bias, 32, a.conv[l].wscale, 0, cin / 32, 0, 2 * cin / 32, 0, DQ_OUT_X, cin, 1, nullptr, 0 };
        const bool last = l == 4;
        const rd_lin &nx = last ? a.output : a.gin[l + 1];
        const int nct = last ? 6 : 18;
        const DqGemm gm = (DqGemm){ nx.wa16, nct, nx.bias, last ? a.out_w : 288, nx.wscale, 0, 0, 0, cin / 32, 0, DQ_OUT_GI, 0, 0, nullptr, 0 };
        // wavefronts 0 and 1 take the two conv tiles (twice the K), the other six three projection tiles each (or one output tile)
        if (wave < 2) dq_gemm_tiles<1>(sh, gc, wave, Tb, rstmask);
        else if (last) dq_gemm_tiles<1>(sh, gm, wave - 2, Tb, rstmask);
        else dq_gemm_tiles<3>(sh, gm, 3 * (wave - 2), Tb, rstmask);
        __syncthreads();
        PH(26);
        // fix-up: the conv's 32 new columns (one k-step) added onto the staged sums
        const DqGemm gf = (DqGemm){ nx.wa16, nct, nullptr, last ? a.out_w : 288, nx.wscale, 0, 0, cin / 32, 1, 1, last ? DQ_OUT_GLOBAL : DQ_OUT_GI, 0, 0, out, a.out_w };
        if (wave >= 2) { if (last) dq_gemm_tiles<1>(sh, gf, wave - 2, Tb, rstmask); else dq_gemm_tiles<3>(sh, gf, 3 * (wave - 2), Tb, rstmask); }
        __syncthreads();
        PH(27);
    }
    {   // conv history of the next chunk = this chunk's last row, back to HBM as float32
        unsigned *hist = (unsigned *)(a.x + (size_t)b * a.x_sb - RD_DEC_W);
        for (int c = tid; c < RD_DEC_W; c += NT_RX)
            hist[c] = (unsigned)__builtin_bit_cast(unsigned short, sh->xh[0][dq_xoff(Tb - 1, c)]) | ((unsigned)__builtin_bit_cast(unsigned short, sh->xl[0][dq_xoff(Tb - 1, c)]) << 16);
    }
    __syncthreads();
}

#ifndef RADE_RX2_TU
// =====================================================================================================
// small data-movement kernels
// =====================================================================================================
__global__ void k_enc_pack(const float *features, float *xin, int B, int T)
{   // model19 only: 4 x (20 features + aux symbol -1) padded 84 -> 88
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
__device__ __forceinline__ float2 pa_limit(float2 x)
{
    const float mag = hypotf(x.x, x.y);
    if (mag == 0.0f) return make_float2(0.0f, 0.0f);
    const float g = tanhf(mag) / mag;                     // tanh(|x|) e^{j angle(x)} = x tanh(|x|)/|x|
    return make_float2(x.x * g, x.y * g);
}

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
        float2 acc[RD_NS + 1];
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) acc[s] = make_float2(0.0f, 0.0f);
#pragma unroll 6
        for (int c = 0; c < RD_NC; c++) {                 // one Winv load feeds the five symbols of the frame
            const float2 w = ld2(tab->Winv[c], tid);
#pragma unroll
            for (int s = 0; s <= RD_NS; s++) acc[s] = cadd(acc[s], cmul(sym[s][c], w));
        }
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) {
            const float2 v = pa_limit(acc[s]);
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
    __shared__ double red[2][192];
    const int mf = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *zf = z + ((size_t)b * n_mf + mf) * RD_ZMF;
    if (tid < RD_NC) sym[0][tid] = make_float2(tab->P[tid] * tab->pilot_gain, 0.0f * tab->pilot_gain);
    if (tid < 120) sym[1 + tid / RD_NC][tid % RD_NC] = make_float2(zf[2 * tid], zf[2 * tid + 1]);
    if (tid >= 128 && tid < 128 + RD_NC && mf > 0) { const int c = tid - 128; prevsym[c] = make_float2(zf[-RD_ZMF + 2 * (90 + c)], zf[-RD_ZMF + 2 * (90 + c) + 1]); }   // last data symbol of frame mf - 1
    __syncthreads();
    if (tid < RD_M) {
        float2 acc[RD_NS + 1];
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) acc[s] = make_float2(0.0f, 0.0f);
#pragma unroll 6
        for (int c = 0; c < RD_NC; c++) {
            const float2 w = ld2(tab->Winv[c], tid);
#pragma unroll
            for (int s = 0; s <= RD_NS; s++) acc[s] = cadd(acc[s], cmul(sym[s][c], w));
        }
#pragma unroll
        for (int s = 0; s <= RD_NS; s++) {
            const float2 v = pa_limit(acc[s]);
            fr[16 + s * RD_SYM + RD_NCP + tid] = v;
            if (tid >= RD_M - RD_NCP) fr[16 + s * RD_SYM + tid - (RD_M - RD_NCP)] = v;
        }
    } else if (tid < RD_M + 16) {                          // samples 944..959 of the previous frame = the last 16 of its last symbol
        const int n = RD_M - 16 + (tid - RD_M);
        float2 a = make_float2(0.0f, 0.0f);
        if (mf > 0) { for (int c = 0; c < RD_NC; c++) a = cadd(a, cmul(prevsym[c], ld2(tab->Winv[c], n))); a = pa_limit(a); }
        fr[tid - RD_M] = a;                                // frame 0: the signal starts here, nothing before it (chan_mp: i >= 16)
    }
    __syncthreads();
    const size_t base = (size_t)mf * RD_NMF;
    const f32x4 *Gb = (const f32x4 *)G + (size_t)b * n_mf * RD_NMF;      // (G1[i], G2[i]) as one 16-byte load per sample
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
    red[0][tid] = s0; red[1][tid] = s1;
    __syncthreads();
    if (tid < 64) { red[0][tid] += red[0][tid + 64] + red[0][tid + 128]; red[1][tid] += red[1][tid + 64] + red[1][tid + 128]; }
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) { if (tid < w) { red[0][tid] += red[0][tid + w]; red[1][tid] += red[1][tid + w]; } __syncthreads(); }
    if (tid == 0) { part[((size_t)b * n_mf + mf) * 2] = red[0][0]; part[((size_t)b * n_mf + mf) * 2 + 1] = red[1][0]; }
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
{   // Box-Muller, unit variance per component
    const float a = ((float)u0 + 0.5f) * (1.0f / 4294967296.0f), bq = ((float)u1 + 0.5f) * (1.0f / 4294967296.0f);
    const float rad = sqrtf(-2.0f * logf(a));
    float sn, cs; sincosf(6.28318530718f * bq, &sn, &cs);
    return make_float2(rad * cs, rad * sn);
}

__device__ __forceinline__ double chan_phase_acc(int i, float f0, float df_dt)
{   // sum_{k<=i} omega_k, omega_k = float32(freq_k*2*pi/Fs) summed in double (torch.cumsum on CPU)
    if (df_dt == 0.0f) { const float om = ((f0 * 2.0f) * (float)PI_D) / 8000.0f; return (double)(i + 1) * (double)om; }
    const double n = (double)(i + 1);
    return (2.0 * PI_D / 8000.0) * (n * (double)f0 + ((double)df_dt / 8000.0) * 0.5 * (double)i * n);
}

__global__ __launch_bounds__(256) void k_chan_apply(rd_chan_args a, const double *part, int n_part)
{
    const int b = blockIdx.y;
    const int n_eoo = a.with_eoo ? RD_NEOO : 0;
    const int n_total = a.n_pre + a.n_sig + n_eoo + a.n_post;
    __shared__ float s_gain; __shared__ float2 s_fin;
    if (threadIdx.x == 0) {
        double p0 = 0.0, p1 = 0.0;
        for (int c = 0; c < n_part; c++) { p0 += part[((size_t)b * n_part + c) * 2]; p1 += part[((size_t)b * n_part + c) * 2 + 1]; }
        const float tx_power = (float)(p0 / a.n_sig), mp_power = (float)(p1 / a.n_sig);
        s_gain = a.G ? powf(tx_power / mp_power, 0.5f) : 1.0f;
        float2 fin = make_float2(1.0f, 0.0f);
        if (a.freq_offset != 0.0f && a.n_sig > 0) { float sn, cs; sincosf((float)chan_phase_acc(a.n_sig - 1, a.freq_offset, a.df_dt), &sn, &cs); fin = make_float2(cs, sn); }
        s_fin = fin;
    }
    __syncthreads();
    const float gain = s_gain; const float2 fin = s_fin;
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
__global__ __launch_bounds__(256) void k_multipath_gen(const float *taps, int n_taps, int low_ratio, int n_out, const float2 *noise_low,
                                                       unsigned long long seed, float2 *G)
{
    __shared__ double2 y[2][DG_MAXLOW];
    __shared__ double red[256][6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n_low = max((n_out + low_ratio - 1) / low_ratio, 2), n_x = n_low + n_taps;
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
        y[p][i] = make_double2(ar, ai);
    }
    __syncthreads();
    auto interp = [&](int p, int n) {                               // linear interpolation, extrapolating past the last low-rate point
        const double pos = (double)n / (double)low_ratio;
        const int i0 = min((int)pos, n_low - 2);
        const double fr = pos - (double)i0;
        const double2 a0 = y[p][i0], a1 = y[p][i0 + 1];
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
extern "C" int rd_launch_multipath_gen(const float *taps_dev, int n_taps, int low_ratio, int n_out, const void *noise_low, unsigned long long seed, void *G, int B, rd_stream_t s)
{
    if (B <= 0 || n_out <= 0) return 0;
    if (low_ratio < 1 || n_taps < 1 || (n_out + low_ratio - 1) / low_ratio > DG_MAXLOW) return -1;
    hipLaunchKernelGGL(k_multipath_gen, dim3(B), dim3(256), 0, (hipStream_t)s, taps_dev, n_taps, low_ratio, n_out, (const float2 *)noise_low, seed, (float2 *)G);
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
    double *part = (double *)a->scratch;
    if (!a->mp) hipLaunchKernelGGL(k_chan_power, dim3(CH_NCH, a->B), dim3(256), 0, st, *a, part);      // a->mp: the modulator left mp and its per-frame power sums (k_ofdm_mod_mp)
    const int n_total = a->n_pre + a->n_sig + (a->with_eoo ? RD_NEOO : 0) + a->n_post;
    int gx = (n_total + 255) / 256; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_chan_apply, dim3(gx, a->B), dim3(256), 0, st, *a, (const double *)part, a->mp ? a->n_sig / RD_NMF : CH_NCH);
    return (int)hipGetLastError();
}

#endif  // !RADE_RX2_TU
// =====================================================================================================
// receiver: one workgroup (512 threads) per stream, up to round_calls do_radae_rx calls per launch
// =====================================================================================================
enum { ST_SEARCH = 0, ST_CANDIDATE = 1, ST_SYNC = 2 };



#define FFT_N 2048
#define FFT_SCR (32 * 66)               // floats per wave of the FFT transpose scratch: [q][l + (l >> 5)], row stride 66: both
                                        // halves of the wave hit 32 distinct banks on the write (fixed q) and on the read (fixed l)

struct RxScalars {
    int state, nin, tmax, tmax_candidate, valid_count, uw_errors, synced_count, mf, f_ind_max, dec_reset_pending, bpf_mem_len, has_eoo;
    uint32_t lcg;
    unsigned rxmax_cur, rxmax_h0, rxmax_h1;   // float bits of max |re|,|im| of the filtered samples of this call / the two calls before (check_pilots operand scale)
    int consumed_inv, calls_inv, valid_inv, eoo_inv, n_calls, n_rows, uw_from_row, consumed_round, pending_valid, out_base;
    int pf_n;                 // samples of the NEXT call already mixed down into xm[102..] by the end of this (synchronised) call; 0 = none
    int entry;                // this candidate call enters sync (decided by thread 0 before a barrier: see do_entry)
    int go, need_decode, batch_call0, state_before, nin_before, valid_output, endofover, uw_fail, candidate, dt_valid, dt_new, lds_sync;
    float snr_est, mag; float2 bpf_phase;
    double fmax, foff_err, rph_r, rph_i, Dthresh, Dtmax12, Dtmax12_eoo;
    double rph_th;                        // k_rx_sync2: the phase accumulator as an angle in [-pi, pi] (rph_r + j rph_i = e^{j rph_th})
};

struct RxShared {
    RxScalars S;
    float2 bmem[102];                     // BPF memory (dsp.py:55,96)
    double2 pd[RD_M], pendd[RD_M];        // pilot / end-of-over replicas as doubles (refine, check_pilots)
    float2 rxb[RD_RXBUF];                 // rx_buf (radae_rxe.py:141)
    float2 sym[6][RD_NC];
    float2 rp[2][RD_NC];
    __attribute__((aligned(16))) float bpf_h[RD_NTAP + 3];
    float eqP[RD_NC]; float2 eqPmat[RD_NC][2][3], eqrot[RD_NC];   // est_pilots' constants (dsp.py:400-433): an L2 round trip per call if read from the table in HBM
    float eq_pg, eq_snrc1, eq_snrc2;                                // pilot_gain and the SNR estimator's constants
    union {
      struct {
        __attribute__((aligned(16))) float2 xm[1408];   // BPF [mem | mixed-down new] (1224); refine(): rx window as doubles (11264 B); rx1[1152] for the demod
        union {
        struct {                          // synchronised state (S.lds_sync != 0)
            float2 wfwd[RD_M][RD_NC];     // forward DFT matrix (dsp.py:501)
            union {
                float absd[96][RD_NFC + 1];   // check_pilots scratch |Dt| rows
                float2 dtr[2 * 80 * 16];      // refine(): complex64 Dt1 / Dt2 at [(frame * nf + f) * 16 + t]; does not overlap the FFT area
                struct { char rpart_pad[8192]; double rpart[6][64][4]; };   // refine(), in-sync grid (dtr uses 5 KB): second-half partial tiles
            };
            unsigned rxh[RD_RXBUF], rxl[RD_RXBUF];   // check_pilots: rx_buf x 2^(7-E) split in two binary16 planes, one dword = (re, im) of a sample
            double vm[8][RD_M];               // refine(), in-sync grid: ((n - 79.5) / 80)^m, m = 0..7 (rebuilt with wfwd)
            double rmom[4][2][64][4];         // refine(), in-sync grid: partial moment tiles [quarter of the samples][frame]
        };
        struct {                          // search / candidate state: FFT pilot correlator (-DRX1_SEARCH_FFT)
            float2 fftX[FFT_N];           // spectrum of the rx_buf window being correlated
            float fftscr[NT_RX / 64][FFT_SCR];
        };
        struct {                          // search / candidate state: pilot search on the matrix cores (rx_detect_mfma)
            __attribute__((aligned(16))) _Float16 sA[2][5 * 2 * 64 * 8];   // the correlation table's A operands of one k-step, double-buffered
            unsigned srxh[RD_RXBUF], srxl[RD_RXBUF];                        // rx_buf in two binary16 planes
        };
        };
      };
      __attribute__((aligned(16))) unsigned char dec_raw[sizeof(DecShared)];   // decoder stage (rx_decode_pending): runs between calls, when xm and the tables are dead
    };
    double2 rtw[80], rrot[80], rt80[80];  // refine(): e^{-jw_f}, e^{-jw_f Nmf}, e^{-jw_f 80} per candidate frequency
    double2 rq[4], rzc, rph[24]; double ral[24];   // in-sync grid (moments about the centre frequency w_c): e^{-jw_c 40 q}, e^{-jw_c}, e^{-j(w_k - w_c) 79.5}, (w_k - w_c) 80
    float rowsum1[RD_NMF], rowsum2[RD_NMF]; // sum_f |Dt1[t,f]|, |Dt2[t,f]|
    int rows48[48];
    double redd[(NT_RX / 64 + 1) * 10];   // block reductions (double): per-wave partials + totals
    float redf[16]; int redi[16]; int redj[16];   // arg-max reduction: slot 0 result, 1.. per-wave partials
    double corrp[2][8];                   // pilot / end-of-over correlations at (tmax, fmax): partial sums of the two side wavefronts
};

// ---------------------------------------------------------------------------------------------------------
// FFT pilot correlator (search / candidate state).  One wavefront transforms 2048 points held 32 per lane:
//   pass 1: lane l owns x[l + 64 k2]; in-lane radix-2 DIF DFT over k2 -> A_l[q], times e^{-j2pi l q/2048}
//   transpose through a per-wave LDS scratch so lane (q, h) owns A_l[q] for l in [32h, 32h+32)
//   pass 2: one DIF radix-2 stage across the lane pair, then the in-lane 32-point DFT over l
// Lane (q, h) ends with X[q + 32 (2 p + h)] in v[brev5(p)].  Twiddles of the in-lane DFTs are literals.
// ---------------------------------------------------------------------------------------------------------
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
typedef __attribute__((address_space(3))) float lds_float;
#ifdef RD_FFT_PAD65
#define FFT_WR(q, lane) ((q) * 65 + (lane))
#define FFT_RD(q2, h, l) ((q2) * 65 + 32 * (h) + (l))
#else
#define FFT_WR(q, lane) ((q) * 66 + (lane) + ((lane) >> 5))
#define FFT_RD(q2, h, l) ((q2) * 66 + 33 * (h) + (l))
#endif
typedef __attribute__((address_space(1))) float glb_float;
__device__ __forceinline__ void fft2048_wave(float2 (&v)[32], lds_float *scr, const glb_float *__restrict__ tw, int lane)
{
    // the inter-pass twiddles e^{-j2pi lane q/2048} are fetched before the first in-lane DFT, which hides their L2 round trip
    float2 w1[32];
#pragma unroll
    for (int q = 1; q < 32; q++) w1[q] = make_float2(tw[2 * (q * 64 + lane)], tw[2 * (q * 64 + lane) + 1]);
    __builtin_amdgcn_sched_barrier(0);
    dft32_inlane(v);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 1; q < 32; q++) v[brev5(q)] = cmul(v[brev5(q)], w1[q]);
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
    // radix-2 DIF stage over l <-> l + 32 (the partner lane): even outputs on h = 0, odd outputs (twiddled) on h = 1
    const float sg = h ? -1.0f : 1.0f;
    // h = 0: u + o;  h = 1: (o - u) e^{-j2pi l/64}, the twiddle as a literal (no table fetch inside the transform)
#pragma unroll
    for (int l = 0; l < 32; l++) {
        const float sr = fmaf(ur[l], sg, lane_swap1(ur[l])), si = fmaf(ui[l], sg, lane_swap1(ui[l]));
        if (l == 0) v[l] = make_float2(sr, si);
        else {
            const float2 t = cmul(make_float2(sr, si), make_float2(C64[l], -S64[l]));
            v[l] = h ? t : make_float2(sr, si);
        }
    }
    dft32_inlane(v);
}

// |Dt| surfaces by FFT convolution (see the FFT correlator notes above); runs with its own register allocation.
// pass 0: Dt1 (rx_buf[t + m]) -> buffer oldb, pass 1: Dt2 (rx_buf[Nmf + t + m]) -> buffer newb; cached skips pass 0.
#ifndef RD_DETECT_INLINE
#define RD_DETECT_INLINE __forceinline__
#endif
__device__ RD_DETECT_INLINE void rx_detect_fft(RxShared *sh, const float *G_, const float *tw_, float *cache_, int cached, int oldb, int newb,
                                              float &best, int &bt, int &bfi)
{
    const glb_float *G = (const glb_float *)G_, *tw = (const glb_float *)tw_; glb_float *cache = (glb_float *)cache_;
    const int tid = rx_tid(), wave = tid >> 6, lane = tid & 63, q2 = lane >> 1, h = lane & 1;
    lds_float *scr = (lds_float *)&sh->fftscr[wave][0];
    lds_float *rxf = (lds_float *)&sh->rxb[0], *Xf = (lds_float *)&sh->fftX[0];
    constexpr int NFW = RD_NFC / (NT_RX / 64);                                // frequencies per wave
#pragma unroll 1
    for (int pass = cached ? 1 : 0; pass < 2; pass++) {
        lds_float *x = rxf + 2 * pass * RD_NMF;
        glb_float *dst = cache + (size_t)(pass ? newb : oldb) * RD_NFC * RD_NMF;
        const glb_float *prev = cache + (size_t)oldb * RD_NFC * RD_NMF;       // |Dt1| while pass 1 produces |Dt2|
        float2 v[32];
        float rs[15], mx[15]; unsigned long long argw = 0ull;                 // this wave's partial row sums / max_f(|Dt1|+|Dt2|) of its 15 t per lane
#pragma unroll
        for (int p = 0; p < 15; p++) { rs[p] = 0.0f; mx[p] = -1.0f; }
        // fi = -1: forward transform of the window (every wave repeats it: identical values, no workgroup barrier);
        // fi >= 0: inverse transform of X . G_f as the forward transform of the re/im-swapped product
#pragma unroll 1
        for (int fi = -1; fi < NFW; fi++) {
            const int f = wave * NFW + fi;
            if (fi < 0) {
#pragma unroll
                for (int k2 = 0; k2 < 32; k2++) {                             // Nmf + M - 1 = 1119 samples, zero padded
                    const int n = lane + 64 * k2;
                    v[k2] = (k2 < 18 && n < RD_NMF + RD_M - 1) ? make_float2(x[2 * n], x[2 * n + 1]) : make_float2(0.0f, 0.0f);
                }
            } else {
                const glb_float *Gf = G + (size_t)f * FFT_N * 2;
                float2 g[32];                                                 // one L2 round trip for the whole spectrum row, not four
#pragma unroll
                for (int u = 0; u < 32; u++) { const int k = lane + 64 * u; g[u] = make_float2(Gf[2 * k], Gf[2 * k + 1]); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 32; u++) { const int k = lane + 64 * u; const float2 y = cmul(make_float2(Xf[2 * k], Xf[2 * k + 1]), g[u]); v[u] = make_float2(y.y, y.x); }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int t0 = q2 + 32 * h;                                       // this lane's outputs: t = t0 + 64 p < Nmf
            float d1[15];                                                     // |Dt1| of the same (f, t): fetched now, used after the transform
#pragma unroll
            for (int p = 0; p < 15; p++) d1[p] = (pass && fi >= 0) ? prev[(size_t)f * RD_NMF + t0 + 64 * p] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);
            fft2048_wave(v, scr, tw, lane);
            if (fi < 0) {
#pragma unroll
                for (int p = 0; p < 32; p++) { const int k = q2 + 64 * p + 32 * h; Xf[2 * k] = v[brev5(p)].x; Xf[2 * k + 1] = v[brev5(p)].y; }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int p = 0; p < 15; p++) {
                    const float2 c = v[brev5(p)];
                    const float d = __builtin_amdgcn_sqrtf(fmaf(c.x, c.x, c.y * c.y));
                    dst[(size_t)f * RD_NMF + t0 + 64 * p] = d;      // (plain accesses on purpose: with nt stores + nt loads a stream now and then read back a stale surface)
                    rs[p] += d;
                    if (pass) { const float s12 = d1[p] + d; if (s12 > mx[p]) { mx[p] = s12; argw = (argw & ~(15ull << (4 * p))) | ((unsigned long long)fi << (4 * p)); } }
                }
            }
        }
        // per-wave partials -> own scratch region (free now), combined by the whole workgroup in wave (= f) order
        {
            const int t0 = q2 + 32 * h;
#pragma unroll
            for (int p = 0; p < 15; p++) { scr[t0 + 64 * p] = rs[p]; scr[RD_NMF + t0 + 64 * p] = mx[p]; }
            scr[2 * RD_NMF + 2 * lane] = __uint_as_float((unsigned)argw); scr[2 * RD_NMF + 2 * lane + 1] = __uint_as_float((unsigned)(argw >> 32));
        }
        __syncthreads();
        for (int t = tid; t < RD_NMF; t += NT_RX) {
            const int ln = 2 * (t & 31) + ((t >> 5) & 1), sh4 = 4 * (t >> 6);
            float sum = 0.0f, lmax = -1.0f; int larg = 0;
#pragma unroll
            for (int w = 0; w < NT_RX / 64; w++) {
                lds_float *sw = (lds_float *)&sh->fftscr[w][0];
                sum += sw[t];
                const float m = sw[RD_NMF + t];
                if (m > lmax) {                                               // strict: the first maximum in f order is kept
                    const unsigned lo = __float_as_uint(sw[2 * RD_NMF + 2 * ln]), hi = __float_as_uint(sw[2 * RD_NMF + 2 * ln + 1]);
                    const unsigned long long aw = ((unsigned long long)hi << 32) | lo;
                    lmax = m; larg = w * NFW + (int)((aw >> sh4) & 15);
                }
            }
            if (pass) { sh->rowsum2[t] = sum; if (lmax > best) { best = lmax; bt = t; bfi = larg; } }   // t ascends per thread: first maximum kept
            else sh->rowsum1[t] = sum;
        }
        __syncthreads();                                                      // scratch and fftX are rewritten by the next pass / the caller
    }
}

// max reduction with lexicographic tie-break (smaller k0, then smaller k1 wins); every thread returns with the winner in (v, k0, k1).
// One barrier: the per-wave winners go to LDS and EVERY thread combines the eight of them (no single-thread pass, no second and
// third barrier).  The caller guarantees a workgroup barrier between the previous reduction's reads and this call (both call sites
// come straight after one), which is what keeps the partial slots safe to overwrite.
__device__ void block_argmax(RxShared *sh, float &v, int &k0, int &k1)
{
    const int tid = rx_tid(), lane = tid & 63, wave = tid >> 6;
    // max under a total order: any combination order gives the same winner; quad / row steps on DPP, rows through readlane
#define ARGMAX_STEP(CTRL) do { const float ov = quad_dpp<CTRL>(v); const int o0 = quad_dpp_i<CTRL>(k0), o1 = quad_dpp_i<CTRL>(k1); \
        if (ov > v || (ov == v && (o0 < k0 || (o0 == k0 && o1 < k1)))) { v = ov; k0 = o0; k1 = o1; } } while (0)
    ARGMAX_STEP(QUAD_XOR1); ARGMAX_STEP(QUAD_XOR2); ARGMAX_STEP(ROW_ROR4); ARGMAX_STEP(ROW_ROR8);
#undef ARGMAX_STEP
    {
        float bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)); int b0 = __builtin_amdgcn_readlane(k0, 0), b1 = __builtin_amdgcn_readlane(k1, 0);
#pragma unroll
        for (int r = 1; r < 4; r++) {
            const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * r)); const int o0 = __builtin_amdgcn_readlane(k0, 16 * r), o1 = __builtin_amdgcn_readlane(k1, 16 * r);
            if (ov > bv || (ov == bv && (o0 < b0 || (o0 == b0 && o1 < b1)))) { bv = ov; b0 = o0; b1 = o1; }
        }
        v = bv; k0 = b0; k1 = b1;
    }
    if (lane == 0) { sh->redf[1 + wave] = v; sh->redi[1 + wave] = k0; sh->redj[1 + wave] = k1; }
    __syncthreads();
    float bv = sh->redf[1]; int b0 = sh->redi[1], b1 = sh->redj[1];
#pragma unroll
    for (int w = 1; w < NT_RX / 64; w++) {
        const float ov = sh->redf[1 + w]; const int o0 = sh->redi[1 + w], o1 = sh->redj[1 + w];
        if (ov > bv || (ov == bv && (o0 < b0 || (o0 == b0 && o1 < b1)))) { bv = ov; b0 = o0; b1 = o1; }
    }
    v = bv; k0 = b0; k1 = b1;
}

// sum NV doubles per thread over the workgroup: wave shuffles, the per-wave sums to LDS, one barrier, then every thread adds the
// eight partials in wave order (same order, hence same rounding, in every thread).  Same contract as block_argmax: a workgroup
// barrier lies between the previous call's reads of redd and this call.
template <int NV>
__device__ void block_sum_multi(RxShared *sh, double (&v)[NV])
{
    const int tid = rx_tid(), lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = wave_sum_f64(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) sh->redd[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NT_RX / 64; w++) t += sh->redd[w * NV + k];
        v[k] = t;
    }
}

__device__ __forceinline__ float sigma_r_from_sums(double t1, double t2)
{   // dsp.py:218-220: (mean|Dt1| + mean|Dt2|)/sqrt(pi/2)/2 in float32
    const float k = (float)sqrt(PI_D / 2.0);
    const float m1 = (float)(t1 / (RD_NMF * RD_NFC)) / k, m2 = (float)(t2 / (RD_NMF * RD_NFC)) / k;
    return (m1 + m2) / 2.0f;
}
__device__ float sigma_r_from_rowsums(RxShared *sh)
{
    double v[2] = { 0.0, 0.0 };
    for (int t = rx_tid(); t < RD_NMF; t += NT_RX) { v[0] += (double)sh->rowsum1[t]; v[1] += (double)sh->rowsum2[t]; }
    block_sum_multi<2>(sh, v);
    return sigma_r_from_sums(v[0], v[1]);
}

// refine(): fine timing/frequency search maximising |Dt1+Dt2| (dsp.py:233-270).  NumPy evaluates the dot products in
// complex128 and stores them as complex64, so this runs on the f64 matrix cores (v_mfma_f64_16x16x4_f64):
//   C[(f,c'), t] = sum_{(n,c)} A[(f,c'),(n,c)] B[(n,c), t],  A = realified e^{-jw_f n},  B = conj(p[n]) rx[t+n] (re | im)
// one 16x16 tile per (8 frequencies, modem frame).  Each lane's A entry is cos(w n) or +-sin(w n) for n = 2s + n0 and
// follows the three-term recurrence x[s+1] = 2 cos(2w) x[s] - x[s-1]: one FMA per MFMA.  The window is converted to
// double once ((xr, xi, xi, -xr) per sample so a lane reads the pair its component needs).
__device__ __forceinline__ f64x4 refine_tile(const RxShared *sh, int mt, int frame, int s0, int ns, int nf, int nt, int lane)
{
    const int i = lane & 15, kk = lane >> 4, c = kk & 1, n0 = kk >> 1;
    const int row = 16 * mt + i, fi = row >> 1, cp = row & 1;
    const bool rv = fi < nf;
    const double2 z1 = sh->rtw[rv ? fi : 0];                                  // e^{-jw}
    double2 cur = s0 ? sh->rt80[rv ? fi : 0] : make_double2(1.0, 0.0);        // e^{-jw 2 s0} (s0 = 0 or 40)
    if (n0) cur = make_double2(cur.x * z1.x - cur.y * z1.y, cur.x * z1.y + cur.y * z1.x);
    const double c2r = z1.x * z1.x - z1.y * z1.y, c2i = 2.0 * z1.x * z1.y;   // e^{-2jw}
    const double2 prv = make_double2(cur.x * c2r + cur.y * c2i, cur.y * c2r - cur.x * c2i);   // cur * e^{+2jw}
    // realified rotation: (c',c) = (0,0) cos, (0,1) sin, (1,0) -sin, (1,1) cos;  cos = Re e^{-jwn}, sin = -Im e^{-jwn}
    double xc = cp == c ? cur.x : (cp == 0 ? -cur.y : cur.y), xp = cp == c ? prv.x : (cp == 0 ? -prv.y : prv.y);
    if (!rv) { xc = 0.0; xp = 0.0; }
    const double k2 = 2.0 * c2r;
    const double *xw = (const double *)&sh->xm[0] + 4 * (frame * 176 + (i < nt ? i : 0) + 2 * s0 + n0) + 2 * c;
    const double2 *pp = &sh->pd[2 * s0 + n0];
    f64x4 acc0 = { 0.0, 0.0, 0.0, 0.0 }, acc1 = acc0;
    // software pipeline over batches of 4 samples: the next batch's LDS reads are in flight while this batch's products
    // and matrix instructions issue (two waves per SIMD cannot hide the 40-cycle f64 latency by themselves)
    double2 pn[4]; double a1[4], a2[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { pn[u] = pp[2 * u]; a1[u] = xw[8 * u]; a2[u] = xw[8 * u + 1]; }
#pragma unroll 1
    for (int s = 0; s < ns; s += 4) {
        double b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = pn[u].x * a1[u];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = fma(pn[u].y, a2[u], b[u]);         // component c of conj(p[n]) rx[t+n]
        __builtin_amdgcn_sched_barrier(0);
        const int sn = s + 4 < ns ? s + 4 : s;                                // last batch re-reads itself: branch-free
#pragma unroll
        for (int u = 0; u < 4; u++) { pn[u] = pp[2 * (sn + u)]; a1[u] = xw[8 * (sn + u)]; a2[u] = xw[8 * (sn + u) + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        const double x1 = fma(k2, xc, -xp), x2 = fma(k2, x1, -xc), x3 = fma(k2, x2, -x1);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xc, b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b[3], acc1, 0, 0, 0);
        xp = x3; xc = fma(k2, x3, -x2);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc0 + acc1;
}

// refine()'s three phasors per candidate frequency (e^{-jw}, e^{-jw Nmf}, e^{-jw 80}), one thread each: k in [0, 3 nf)
__device__ __forceinline__ void refine_tables(RxShared *sh, int k, double fstart, double fstop, double fstep)
{
    const int nf = (int)ceil((fstop - fstart) / fstep);                 // np.arange length
    const double delta = (fstart + fstep) - fstart;                       // np.arange fill rule
    if (k < 0 || k >= 3 * nf) return;
    const int which = k / nf, fi = k - which * nf;
    const double w = 2.0 * PI_D * (fstart + fi * delta) / 8000.0;
    const double arg = which == 0 ? -w : (which == 1 ? -w * RD_NMF : -w * 80.0);
    double sn, cs; sincos(arg, &sn, &cs);
    double2 *dstp = which == 0 ? sh->rtw : (which == 1 ? sh->rrot : sh->rt80);
    dstp[fi] = make_double2(cs, sn);
}

// In-sync grid (np.arange(fmax - 1, fmax + 1, 0.1): 20 or 21 frequencies, 16 timings): the frequencies lie within +-1.05 Hz of their
// centre w_c, i.e. within 0.066 rad over the 160-sample window measured from its middle, so
//   Dt(t, f_k) = sum_n y_t[n] e^{-jw_k n} = e^{-j dw_k 79.5} sum_m (-j dw_k 80)^m / m! * M_m(t),
//   M_m(t) = sum_n ((n - 79.5) / 80)^m e^{-jw_c n} y_t[n],   y_t[n] = conj(p[n]) rx[t + n],   dw_k = w_k - w_c,
// and eight moments M_0..M_7 (remainder <= 0.066^8 / 8! = 9e-15 of sum|y|, the size of the rounding error of the 160-term complex128
// sum itself: 6000 random cases, worst 3.7e-16 sum|y| against a long-double evaluation, every complex64-rounded value equal to the
// direct sum's) replace the twenty per-frequency sums: ONE 16x16 tile (8 moments x re/im) per modem frame instead of three -- the
// f64 matrix instructions are what this phase is made of.  The moments are realified like the frequency rows were; their extra real
// factor comes from a small table (vm).  refine() on sync entry (+-10 Hz) keeps the direct sums.
__device__ __forceinline__ void refine_tables_sync(RxShared *sh, int k, double fstart, double fstop, double fstep)
{   // one lane, ONE sincos each (the wavefront that runs this during the FIR must not outlast it): k < 24 e^{-jw_k Nmf}, 24..47 the
    // constants of frequency k - 24, 48..51 the quarter starts of the sample range, 52 e^{-jw_c}.  np.arange(fmax - 1, fmax + 1, 0.1)
    // has 20 OR 21 entries depending on how fmax rounds, so nf is computed, not assumed (w_c = the middle of the grid either way).
    const int nf = (int)ceil((fstop - fstart) / fstep);                 // np.arange length (<= 24 here)
    const double delta = (fstart + fstep) - fstart;                       // np.arange fill rule
    const double wc = 0.5 * (2.0 * PI_D * fstart / 8000.0 + 2.0 * PI_D * (fstart + (nf - 1) * delta) / 8000.0);
    if (k < 0 || k > 52) return;
    const int kf = k < 24 ? k : k - 24;
    if (k < 48 && kf >= nf) return;
    const double w = 2.0 * PI_D * (fstart + kf * delta) / 8000.0, dw = w - wc;
    const double arg = k < 24 ? -w * RD_NMF : (k < 48 ? -dw * 79.5 : (k < 52 ? -wc * 40.0 * (k - 48) : -wc));
    double sn, cs; sincos(arg, &sn, &cs);
    const double2 v = make_double2(cs, sn);
    if (k < 24) sh->rrot[k] = v;
    else if (k < 48) { sh->rph[kf] = v; sh->ral[kf] = dw * 80.0; }
    else if (k < 52) sh->rq[k - 48] = v;
    else sh->rzc = v;
}
__device__ __forceinline__ f64x4 refine_moments(const RxShared *sh, int frame, int q, int nt, int lane)
{
    const int i = lane & 15, kk = lane >> 4, c = kk & 1, n0 = kk >> 1;
    const int m = i >> 1, cp = i & 1, s0 = 20 * q;
    const double2 z1 = sh->rzc;                                               // e^{-jw_c}
    double2 cur = sh->rq[q];                                                  // e^{-jw_c 2 s0}
    if (n0) cur = make_double2(cur.x * z1.x - cur.y * z1.y, cur.x * z1.y + cur.y * z1.x);
    const double c2r = z1.x * z1.x - z1.y * z1.y, c2i = 2.0 * z1.x * z1.y;   // e^{-2jw_c}
    const double2 prv = make_double2(cur.x * c2r + cur.y * c2i, cur.y * c2r - cur.x * c2i);   // cur * e^{+2jw_c}
    // realified rotation: (c',c) = (0,0) cos, (0,1) sin, (1,0) -sin, (1,1) cos;  cos = Re e^{-jwn}, sin = -Im e^{-jwn}
    double xc = cp == c ? cur.x : (cp == 0 ? -cur.y : cur.y), xp = cp == c ? prv.x : (cp == 0 ? -prv.y : prv.y);
    const double k2 = 2.0 * c2r;
    const double *xw = (const double *)&sh->xm[0] + 4 * (frame * 176 + (i < nt ? i : 0) + 2 * s0 + n0) + 2 * c;
    const double2 *pp = &sh->pd[2 * s0 + n0];
    const double *vp = &sh->vm[m][2 * s0 + n0];
    f64x4 acc0 = { 0.0, 0.0, 0.0, 0.0 }, acc1 = acc0;
    double2 pn[4]; double a1[4], a2[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { pn[u] = pp[2 * u]; a1[u] = xw[8 * u]; a2[u] = xw[8 * u + 1]; vv[u] = vp[2 * u]; }
#pragma unroll 1
    for (int s = 0; s < 20; s += 4) {
        double b[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { b[u] = pn[u].x * a1[u]; v[u] = vv[u]; }
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = fma(pn[u].y, a2[u], b[u]);         // component c of conj(p[n]) rx[t+n]
        __builtin_amdgcn_sched_barrier(0);
        const int sn = s + 4 < 20 ? s + 4 : s;                                // last batch re-reads itself: branch-free
#pragma unroll
        for (int u = 0; u < 4; u++) { pn[u] = pp[2 * (sn + u)]; a1[u] = xw[8 * (sn + u)]; a2[u] = xw[8 * (sn + u) + 1]; vv[u] = vp[2 * (sn + u)]; }
        __builtin_amdgcn_sched_barrier(0);
        const double x1 = fma(k2, xc, -xp), x2 = fma(k2, x1, -xc), x3 = fma(k2, x2, -x1);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xc * v[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1 * v[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2 * v[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3 * v[3], b[3], acc1, 0, 0, 0);
        xp = x3; xc = fma(k2, x3, -x2);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc0 + acc1;
}

__device__ void rx_refine(RxShared *sh, int *tmax, double *fmax, int t0, int nt, double fstart, double fstop, double fstep, bool have_tables)
{
    const int tid = rx_tid(), lane = tid & 63, wave = tid >> 6;
    const int nf = (int)ceil((fstop - fstart) / fstep);                 // np.arange length
    const double delta = (fstart + fstep) - fstart;                       // np.arange fill rule
    const int ntasks = ((2 * nf + 15) >> 4) * 2;
    const int i = lane & 15, kk = lane >> 4;
    PH_T0();
    if (!have_tables) refine_tables(sh, tid, fstart, fstop, fstep);
    if (tid >= 64 && tid < 64 + 2 * 176) {                                // the two windows as doubles: (xr, xi, xi, -xr)
        const int j = tid - 64, frame = j / 176, k = j - frame * 176;
        const float2 x = sh->rxb[min(t0 + frame * RD_NMF + k, RD_RXBUF - 1)];
        double *d = (double *)&sh->xm[0] + 4 * j;
        d[0] = (double)x.x; d[1] = (double)x.y; d[2] = (double)x.y; d[3] = -(double)x.x;
    }
    __syncthreads();
    PH(12);
    // C layout (f64 16x16x4): col = lane&15 (t), row = (lane>>4) + 4*reg.  Rows alternate re/im, so the lane 16
    // positions away holds the other component of the same (f, t).
    auto finish = [&](const f64x4 &acc, int mt, int frame) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double mine = acc[r], other = __shfl_xor(mine, 16);
            if ((kk & 1) == 0) {
                const int fo = 8 * mt + (kk >> 1) + 2 * r;
                double re = mine, im = other;
                if (frame == 1 && fo < nf) {                              // w_vec2 = w_vec1 * exp(-1j*w*Nmf)
                    const double2 rt = sh->rrot[fo];
                    const double tr = re * rt.x - im * rt.y; im = re * rt.y + im * rt.x; re = tr;
                }
                if (fo < nf && i < nt) sh->dtr[(frame * nf + fo) * 16 + i] = make_float2((float)re, (float)im);
            }
        }
    };
    float best = -1.0f; int bf = 0x7fffffff, bt = 0x7fffffff;
    if (ntasks <= 6) {
        // in-sync grid: eight moments per (frame, t) instead of twenty frequencies (refine_moments): wavefront w = (quarter w >> 1 of
        // the samples, frame w & 1), 20 matrix instructions each; the quarters meet in LDS and are added in a fixed order
        {
            const f64x4 part = refine_moments(sh, wave & 1, wave >> 1, nt, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) sh->rmom[wave >> 1][wave & 1][lane][r] = part[r];
        }
        __syncthreads();
        double (*mtot)[16][16] = (double (*)[16][16])&sh->rpart[0][0][0];     // [frame][2 m + (re | im)][t]
        {   // C layout (f64 16x16x4): col = lane & 15 (t), row = (lane >> 4) + 4 * reg
            const int frame = tid >> 8, l = (tid >> 2) & 63, r = tid & 3;
            mtot[frame][(l >> 4) + 4 * r][l & 15] = ((sh->rmom[0][frame][l][r] + sh->rmom[1][frame][l][r]) + sh->rmom[2][frame][l][r]) + sh->rmom[3][frame][l][r];
        }
        __syncthreads();
        // one thread per (frequency, timing): both frames' polynomials, the complex64 roundings NumPy makes, and the metric
        // |Dt1 + Dt2| straight away (no pass through LDS, no separate scan)
        for (int o = tid; o < nf * 16; o += NT_RX) {
            const int fo = o >> 4, t = o & 15;
            if (t >= nt) continue;
            const double al = sh->ral[fo];
            const double2 ph = sh->rph[fo], rt = sh->rrot[fo];               // e^{-j dw_k 79.5}, e^{-jw_k Nmf}
            float2 d12[2];
#pragma unroll
            for (int frame = 0; frame < 2; frame++) {
                // sum_m (-j al)^m / m! M_m:  (-j)^m = 1, -j, -1, j
                double re = 0.0, im = 0.0, cm = 1.0;
#pragma unroll
                for (int mq = 0; mq < 8; mq++) {
                    const double mr = mtot[frame][2 * mq][t], mi = mtot[frame][2 * mq + 1][t];
                    if ((mq & 3) == 0) { re = fma(cm, mr, re); im = fma(cm, mi, im); }
                    else if ((mq & 3) == 1) { re = fma(cm, mi, re); im = fma(-cm, mr, im); }
                    else if ((mq & 3) == 2) { re = fma(-cm, mr, re); im = fma(-cm, mi, im); }
                    else { re = fma(-cm, mi, re); im = fma(cm, mr, im); }
                    cm = cm * al * (1.0 / (double)(mq + 1));
                }
                double xr = re * ph.x - im * ph.y, xi = re * ph.y + im * ph.x;
                if (frame == 1) { const double tr = xr * rt.x - xi * rt.y; xi = xr * rt.y + xi * rt.x; xr = tr; }   // w_vec2 = w_vec1 * exp(-1j*w*Nmf)
                d12[frame] = make_float2((float)xr, (float)xi);
            }
            const float v = hypotf(d12[0].x + d12[1].x, d12[0].y + d12[1].y);  // |Dt1 + Dt2| in complex64
            if (v > best || (v == best && (fo < bf || (fo == bf && t < bt)))) { best = v; bf = fo; bt = t; }
        }
        PH(13);
    } else {
        for (int task = wave; task < ntasks; task += NT_RX / 64) finish(refine_tile(sh, task >> 1, task & 1, 0, 80, nf, nt, lane), task >> 1, task & 1);
        __syncthreads();
        PH(13);
        for (int task = tid; task < nf * nt; task += NT_RX) {
            const int fi = task / nt, ti = task - fi * nt;
            const float2 a = sh->dtr[fi * 16 + ti], b = sh->dtr[(nf + fi) * 16 + ti];
            const float v = hypotf(a.x + b.x, a.y + b.y);                     // |Dt1 + Dt2| in complex64
            if (v > best || (v == best && (fi < bf || (fi == bf && ti < bt)))) { best = v; bf = fi; bt = ti; }
        }
    }
    PH(14);
    block_argmax(sh, best, bf, bt);                                       // dtr and the window are free from its barrier on
    if (best > 0.0f) { *tmax = t0 + bt; *fmax = fstart + bf * delta; }
    PH(15);
}

// one term of dot(conj(w_vec*rx[t0..]), ref) in complex128 (dsp.py:307-313); thread n < 160 owns sample n
__device__ __forceinline__ void rx_corr_term(const RxShared *sh, int t0, double s, double c, const double2 *ref, double &ar, double &ai)
{
    const int tid = rx_tid();
    const float2 x = sh->rxb[t0 + tid];
    const double qr = c * x.x - s * x.y, qi = -(c * x.y + s * x.x);    // conj(w_vec*rx)
    const double2 r = ref[tid];
    ar = qr * r.x - qi * r.y; ai = qr * r.y + qi * r.x;
}

// Scalar receiver state lives in LDS (sh->S): thread 0 is the only writer, everybody reads it after a
// barrier.  (Keeping ~40 loop-carried "uniform" scalars in every thread's registers cost 256 VGPRs
// and proved fragile under -O3.)
static_assert(sizeof(RxShared) <= 160 * 1024, "k_rx_sync working set must fit the 160 KiB LDS of a CU");

__device__ static constexpr uint32_t LCG_A[48] = { 1664525u, 389569705u, 2940799637u, 158984081u, 2862450781u, 3211393721u, 1851289957u, 3934847009u, 2184914861u, 246739401u, 1948736821u, 2941245873u, 4195587069u, 4088025561u, 980655621u, 2001863745u, 657792333u, 65284841u, 1282409429u, 3808694225u, 2968195997u, 2417331449u, 2878627493u, 307989601u, 504219373u, 1897564169u, 2574089845u, 3294562801u, 3478292285u, 2651335705u, 2523738949u, 666245249u, 4137395341u, 2604435753u, 1706708245u, 3963176977u, 3678957277u, 3530469177u, 3858799589u, 629287073u, 3146069549u, 3820924489u, 2403397557u, 2390444593u, 2593868413u, 4291139161u, 1705056389u, 3186638017u };
__device__ static constexpr uint32_t LCG_C[48] = { 1013904223u, 1196435762u, 3519870697u, 2868466484u, 1649599747u, 2670642822u, 1476291629u, 2748932008u, 2180890343u, 2498801434u, 3421909937u, 3167820124u, 2636375307u, 3801544430u, 28987765u, 2210837584u, 3039689583u, 1338634754u, 1649346937u, 2768872580u, 2254235155u, 2326606934u, 1719328701u, 1061592568u, 53332215u, 1140036074u, 4224358465u, 2629538988u, 1946028059u, 573775550u, 1473591045u, 95141024u, 1592739711u, 1618554578u, 4257218569u, 2685635028u, 2617994019u, 740185638u, 4194465613u, 2426187848u, 967350023u, 366635194u, 2557108433u, 3503432700u, 353185579u, 706247310u, 408928405u, 1855199472u };

// Decoder + output stage for the rows a stream has pending, run by the stream's own workgroup: CoreDecoder over the
// rows (ds_layers), rows -> 36-float feature frames (rade_api.c:488-513), aux-bit (UW) error accounting
// (radae_rxe.py:300-319) and the per-call trace.  The decoder's LDS scratch overlays the demod / correlator tables,
// which are reloaded on the next synchronised call.
__device__ __forceinline__ void rx_decode_pending(RxShared *sh, const rd_sync_args &a, int b)
{
    RxScalars *S = &sh->S;
    DecShared *ds = (DecShared *)&sh->dec_raw[0];
    rd_rx_round *rnd = a.round + b;
    const int tid = rx_tid();
    const int Tb = S->n_rows;
    for (int i = tid; i < Tb; i += NT_RX) ds->rst[i] = rnd->row_reset[i];
    if (tid == 0) S->lds_sync = 0;
    __syncthreads();
    for (int c0 = 0; c0 < Tb; c0 += DQ_ROWS) {                       // chunks of 24 rows: GRU state and conv history carry across
        const int n = min(DQ_ROWS, Tb - c0);
        unsigned rstmask = 0u;
        for (int t = 0; t < n; t++) rstmask |= ds->rst[c0 + t] ? (1u << t) : 0u;
        dq_layers(ds, a.dec, b, a.dec.z + (size_t)b * a.dec.z_sb + (size_t)c0 * RD_LATENT, a.dec.out + (size_t)b * a.dec.out_sb + (size_t)c0 * a.dec.out_w, n, rstmask);
    }
    const float *f84 = a.dec.out + (size_t)b * a.dec.out_sb;
    for (int r = tid; r < Tb; r += NT_RX) ds->err[r] = f84[r * 84 + 20] > 0.0f ? 1 : 0;      // first aux symbol of each group of 4
    // valid frame v (3 rows) -> 12 feature frames x 36 floats, 20 used + 16 zeros
    float *out = a.features_out + (size_t)b * a.feat_stride + (size_t)S->out_base * RD_FEAT_MF;
    for (int i = tid; i < (Tb / 3) * RD_FEAT_MF; i += NT_RX) {
        const int fr = i / 36, j = i - fr * 36;          // fr = 10 ms frame index within this batch
        const int row = fr >> 2, sub = fr & 3;
        out[i] = j < 20 ? f84[row * 84 + sub * 21 + j] : 0.0f;
    }
    __syncthreads();
    if (a.trace) {
        for (int c = S->batch_call0 + tid; c < S->n_calls; c += NT_RX) {
            const int idx = rnd->call_trace_idx[c];
            if (idx >= a.trace_cap) continue;
            int e = 0;
            for (int r = rnd->call_row_lo[c]; r < rnd->call_row_hi[c]; r++) e += ds->err[r];
            a.trace[(size_t)b * a.trace_cap + idx].uw_errors += e;
        }
    }
    if (tid == 0) {
        int add = 0;
        for (int r = S->uw_from_row; r < Tb; r++) add += ds->err[r];
        S->uw_errors += add;
        S->out_base += Tb / 3; S->n_rows = 0; S->uw_from_row = 0; S->pending_valid = 0; S->batch_call0 = S->n_calls; S->need_decode = 0;
    }
    __syncthreads();
}

// check_pilots' row refresh for one (modem frame, group of NTN frequency tiles): three row tiles x NTN f-tiles of 16x16 outputs,
// K = 160 samples x (re, im) in ten k-steps of three binary16-plane products each.  Row operands: the pre-split rx_buf planes in LDS
// (4-byte aligned windows: dword reads); pilot operands: a.corr16 fragments from L2 in two register sets, each refilled for k-step
// s + 2 as soon as the products of k-step s have issued, so an L2 round trip has two k-steps to complete.
template <int NTN>
__device__ __forceinline__ void check_rows_tiles(RxShared *sh, const unsigned short *corr16, int frame, int nt_base, int lane, float rx_unsc)
{
    const int i = lane & 15, g = lane >> 4;
    const unsigned *xh[3], *xl[3];
#pragma unroll
    for (int rt = 0; rt < 3; rt++) { const int o = sh->rows48[rt * 16 + i] + frame * RD_NMF + 4 * g; xh[rt] = sh->rxh + o; xl[rt] = sh->rxl + o; }
    const unsigned short *pt = corr16 + ((size_t)nt_base * 10 * 2 * 64 + lane) * 8;
    f32x4 acc[3][NTN];
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int q = 0; q < NTN; q++) acc[rt][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    f16x8 p0h[NTN], p0l[NTN], p1h[NTN], p1l[NTN];
    auto fetch = [&](f16x8 (&ph)[NTN], f16x8 (&pl)[NTN], int sidx) {
#pragma unroll
        for (int q = 0; q < NTN; q++) {
            ph[q] = *(const f16x8 *)(pt + ((size_t)(q * 10 + sidx) * 2) * 64 * 8);
            pl[q] = *(const f16x8 *)(pt + ((size_t)(q * 10 + sidx) * 2 + 1) * 64 * 8);
        }
    };
    auto kstep = [&](const f16x8 (&ch)[NTN], const f16x8 (&cl)[NTN], int sidx) {
        f16x8 ah[3], al[3];
#pragma unroll
        for (int rt = 0; rt < 3; rt++) {
            u32x4 vh, vl;
#pragma unroll
            for (int j = 0; j < 4; j++) { vh[j] = xh[rt][16 * sidx + j]; vl[j] = xl[rt][16 * sidx + j]; }
            ah[rt] = __builtin_bit_cast(f16x8, vh); al[rt] = __builtin_bit_cast(f16x8, vl);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NTN; q++) {
#pragma unroll
            for (int rt = 0; rt < 3; rt++) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[q], ah[rt], acc[rt][q], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 3; rt++) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[q], al[rt], acc[rt][q], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 3; rt++) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[q], ah[rt], acc[rt][q], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(p0h, p0l, 0); fetch(p1h, p1l, 1);
#pragma unroll
    for (int sidx = 0; sidx < 10; sidx += 2) {
        kstep(p0h, p0l, sidx);
        if (sidx + 2 < 10) fetch(p0h, p0l, sidx + 2);
        kstep(p1h, p1l, sidx + 1);
        if (sidx + 3 < 10) fetch(p1h, p1l, sidx + 3);
    }
    // C layout: column = lane & 15 (row draw), rows 4 (lane >> 4) + r = (re, im) of f = 8 nt + 2 g and f + 1
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int q = 0; q < NTN; q++) {
            const int f = 8 * (nt_base + q) + 2 * g, r2 = 2 * (rt * 16 + i) + frame;
            sh->absd[r2][f] = rx_unsc * hypotf(acc[rt][q][0], acc[rt][q][1]); sh->absd[r2][f + 1] = rx_unsc * hypotf(acc[rt][q][2], acc[rt][q][3]);
        }
}

#ifndef RADE_RX2_TU
// ---- |Dt| surfaces on the matrix cores (the pilot search of k_rx_sync2, rade_rx2.inc: rx2_detect_mfma -- see there and DESIGN.md 3.9 for the method) in
// this kernel's shape: wavefronts 0..3 (one per SIMD) own the 60 timing tiles exactly as there, wavefronts 4..7 only help staging the table fragments
// and keep the barriers.  Same outputs as rx_detect_fft; the surfaces in the stream's HBM cache are in the writer lane's order.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rx_detect_mfma(RxShared *sh, const unsigned short *corr16_, float *cache_, int cached, int oldb, int newb, float rx_unsc,
                                                float &best, int &bt, int &bfi)
{
    constexpr int RT = 5, NTF = 5, TPW = 15;
    static_assert(TPW * 4 * 16 == RD_NMF && TPW % RT == 0, "timing tiles per wavefront");
    typedef const __attribute__((address_space(1))) f16x8 glb_f16x8_t;
    const int tid = rx_tid(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, g = lane >> 4;
    const bool worker = wave < 4;                      // four wavefronts (one per SIMD) own the timing tiles; the other four only help staging the table
    // chunk c = 128 nt + 64 plane + lane of a k-step (640 x 16 B): thread tid takes c = tid and (wavefronts 0 / 1) tid + 512
    // Buffer loads (uniform descriptor + 32-bit lane offset + scalar offset): with plain pointers the compiler keeps one 64-bit address
    // per (base, k-step) in VGPRs, spills them, and every k-step starts with scratch reloads under s_waitcnt vmcnt(0).
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(corr16_), 0, 5 * 10 * 2048, 0x00020000);
    const int vo = (((tid >> 7) * 10) * 128 + (tid & 127)) * 16;
    const bool third = wave < 2;                       // wave-uniform
    u32x4 stg[2];
    auto stage_load = [&](int sidx) {
        stg[0] = __builtin_amdgcn_raw_buffer_load_b128(crs, vo, sidx * 2048, 0);
        if (third) stg[1] = __builtin_amdgcn_raw_buffer_load_b128(crs, vo, sidx * 2048 + 4 * 10 * 2048, 0);
    };
    auto stage_store = [&](int buf) {
        _Float16 *d = &sh->sA[buf][tid * 8];
        *(u32x4 *)d = stg[0];
        if (third) *(u32x4 *)(d + 512 * 8) = stg[1];
    };
    float lbest = best; int lkey = 0x7fffffff;                      // (t << 6) | f of the best; callers start from best = -1, which the first sum replaces
    stage_load(0); stage_store(0);
    __syncthreads();
#pragma unroll 1
    for (int pass = cached ? 1 : 0; pass < 2; pass++) {
        const unsigned *ph = sh->srxh + pass * RD_NMF + i + 4 * g, *pl = sh->srxl + pass * RD_NMF + i + 4 * g;
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(cache_ + (size_t)(pass ? newb : oldb) * RD_NFC * RD_NMF), 0, RD_NFC * RD_NMF * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(cache_ + (size_t)oldb * RD_NFC * RD_NMF), 0, RD_NFC * RD_NMF * 4, 0x00020000);
        float *rowsum = pass ? sh->rowsum2 : sh->rowsum1;
#pragma unroll 1
        for (int grp = 0; grp < TPW / RT; grp++) {
            const int T0 = wave * TPW + grp * RT;
            f32x4 acc[RT][NTF];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int q = 0; q < NTF; q++) acc[rt][q] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
            u32x4 wh[RT + 1], wl[RT + 1];
            if (worker) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int j = 0; j < 4; j++) { wh[rt][j] = ph[16 * (T0 + rt) + j]; wl[rt][j] = pl[16 * (T0 + rt) + j]; }
            }
#pragma unroll
            for (int sidx = 0; sidx < 10; sidx++) {
                stage_load(sidx == 9 ? 0 : sidx + 1);
                if (worker) {
                if (sidx < 9) {
#pragma unroll
                    for (int j = 0; j < 4; j++) { wh[RT][j] = ph[16 * (T0 + RT + sidx) + j]; wl[RT][j] = pl[16 * (T0 + RT + sidx) + j]; }
                }
                const _Float16 *Ab = &sh->sA[sidx & 1][lane * 8];
                // the next frequency tile's fragments are in flight while this one's 15 instructions issue (left to itself the compiler
                // reads each fragment right before its use and waits for it: ten exposed LDS latencies per k-step, half the phase)
                f16x8 ch[2], cl[2];
                ch[0] = *(const f16x8 *)(Ab); cl[0] = *(const f16x8 *)(Ab + 512);
#pragma unroll
                for (int q = 0; q < NTF; q++) {
                    if (q + 1 < NTF) { ch[(q + 1) & 1] = *(const f16x8 *)(Ab + ((q + 1) * 2) * 512); cl[(q + 1) & 1] = *(const f16x8 *)(Ab + ((q + 1) * 2 + 1) * 512); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[q & 1], __builtin_bit_cast(f16x8, wh[rt]), acc[rt][q], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[q & 1], __builtin_bit_cast(f16x8, wl[rt]), acc[rt][q], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc[rt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[q & 1], __builtin_bit_cast(f16x8, wh[rt]), acc[rt][q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rt = 0; rt < RT; rt++) { wh[rt] = wh[rt + 1]; wl[rt] = wl[rt + 1]; }
                }
                stage_store((sidx & 1) ^ 1);
                __syncthreads();
            }
            if (!worker) continue;                      // (wave-uniform; the helpers meet the workers again at the next k-step's barrier)
            // C layout: column = lane & 15 (timing), rows 4 g + r = (re, im) of f = 8 q + 2 g and f + 1
            const int tb = 16 * T0 + i;
            // The surfaces in the stream's HBM cache are only ever read back by the lane that wrote them (|Dt2| of this call is |Dt1| of the
            // next), so their layout is the lane's: per (wavefront, group) 64 lanes x 50 values (rt-major, then frequency tile, then the two
            // frequencies) as twelve 16-byte vectors [k][lane] + one 8-byte vector [lane].  13 fully coalesced instructions per group and
            // direction instead of 50 single dwords (store ISSUE was the epilogue: ~10 k cycles per group).
            const int gb = (wave * (TPW / RT) + grp) * 64 * 2 * RT * NTF * 4;     // byte offset of the group's block
            // all of the group's |Dt1| (HBM latency) are requested before any arithmetic; the accumulators turn into |Dt2| in place (4 -> 2
            // registers per tile), which is what makes room for them
            float pv[RT * 2 * NTF];
            if (pass) {
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(prs, lane * 16, gb + k * 1024, 0);
                    pv[4 * k] = __uint_as_float(v[0]); pv[4 * k + 1] = __uint_as_float(v[1]); pv[4 * k + 2] = __uint_as_float(v[2]); pv[4 * k + 3] = __uint_as_float(v[3]);
                }
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(prs, lane * 8, gb + 12 * 1024, 0);
                pv[48] = __uint_as_float(v[0]); pv[49] = __uint_as_float(v[1]);
            }
            float dd[RT * 2 * NTF], rsum[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                float rs = 0.0f;
#pragma unroll
                for (int q = 0; q < NTF; q++) {
                    const f32x4 c = acc[rt][q];
                    const float d0 = rx_unsc * __builtin_amdgcn_sqrtf(fmaf(c[0], c[0], c[1] * c[1])), d1 = rx_unsc * __builtin_amdgcn_sqrtf(fmaf(c[2], c[2], c[3] * c[3]));
                    rs += d0; rs += d1;
                    dd[rt * 2 * NTF + 2 * q] = d0; dd[rt * 2 * NTF + 2 * q + 1] = d1;
                }
                {   // the other three lane groups hold the row's other frequencies: v_permlane16/32_swap (vector ALU, no LDS round trip)
                    const auto p16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                    rs = __uint_as_float(p16[0]) + __uint_as_float(p16[1]);
                    const auto p32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                    rs = __uint_as_float(p32[0]) + __uint_as_float(p32[1]);
                }
                rsum[rt] = rs;
            }
#pragma unroll
            for (int k = 0; k < 12; k++)
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){ __float_as_uint(dd[4 * k]), __float_as_uint(dd[4 * k + 1]), __float_as_uint(dd[4 * k + 2]), __float_as_uint(dd[4 * k + 3]) }, drs, lane * 16, gb + k * 1024, 0);
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){ __float_as_uint(dd[48]), __float_as_uint(dd[49]) }, drs, lane * 8, gb + 12 * 1024, 0);
            // every lane keeps its own best (t ascending, then f ascending, strict >: the earliest wins); block_argmax orders the lanes the same
            // way.  Branch-free, (t, f) packed in one register: as conditional blocks the compiler kept the three in scratch memory and every
            // one of the 50 updates was a store + load under s_waitcnt vmcnt(0).
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int t = tb + 16 * rt;
                if (pass) {
#pragma unroll
                    for (int q = 0; q < NTF; q++) {
                        const int k0 = (t << 6) | (8 * q + 2 * g);
                        const float s0 = pv[rt * 2 * NTF + 2 * q] + dd[rt * 2 * NTF + 2 * q], s1 = pv[rt * 2 * NTF + 2 * q + 1] + dd[rt * 2 * NTF + 2 * q + 1];
                        const bool c0 = s0 > lbest; lbest = c0 ? s0 : lbest; lkey = c0 ? k0 : lkey;
                        const bool c1 = s1 > lbest; lbest = c1 ? s1 : lbest; lkey = c1 ? k0 + 1 : lkey;
                    }
                }
                if (g == 0) rowsum[t] = rsum[rt];
            }
        }
        __syncthreads();
    }
    best = lbest; bt = lkey >> 6; bfi = lkey & 63;
}


__global__ __launch_bounds__(NT_RX) void k_rx_sync(rd_sync_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    RxShared *sh = (RxShared *)smem_raw;
    RxScalars *S = &sh->S;
    const rd_tables *tab = a.tab;
    const int b = blockIdx.x, tid = threadIdx.x;
    rd_rx_stream *st = a.st + b;
    rd_rx_round *rnd = a.round + b;
    const float2 *rxin = (const float2 *)a.rx + (size_t)b * a.rx_stride;

    // ---- load the stream's working set into LDS
    for (int i = tid; i < RD_RXBUF; i += NT_RX) sh->rxb[i] = make_float2(st->rx_buf[i][0], st->rx_buf[i][1]);
    for (int i = tid; i < RD_NTAP; i += NT_RX) sh->bpf_h[i] = tab->bpf_h[i];
    if (tid == 0) { sh->eq_pg = tab->pilot_gain; sh->eq_snrc1 = tab->snr_c1; sh->eq_snrc2 = tab->snr_c2; }
    if (tid < RD_NC) { sh->eqP[tid] = tab->P[tid]; sh->eqrot[tid] = make_float2(tab->eq_rot[tid][0], tab->eq_rot[tid][1]); }
    if (tid < RD_NC * 6) { const int c = tid / 6, r = tid - 6 * c; sh->eqPmat[c][r / 3][r % 3] = make_float2(tab->Pmat[c][r / 3][r % 3][0], tab->Pmat[c][r / 3][r % 3][1]); }
    for (int i = tid; i < RD_M; i += NT_RX) { sh->pd[i] = make_double2(tab->p[i][0], tab->p[i][1]); sh->pendd[i] = make_double2(tab->pend[i][0], tab->pend[i][1]); }
    for (int i = tid; i < RD_NMF; i += NT_RX) { sh->rowsum1[i] = st->rowsum1[i]; sh->rowsum2[i] = st->rowsum2[i]; }
    for (int i = tid; i < 102; i += NT_RX) sh->bmem[i] = make_float2(st->bpf_mem[i][0], st->bpf_mem[i][1]);
    if (tid == 0) {
        S->state = st->state; S->nin = st->nin; S->tmax = st->tmax; S->tmax_candidate = st->tmax_candidate; S->valid_count = st->valid_count;
        S->uw_errors = st->uw_errors; S->synced_count = st->synced_count; S->mf = st->mf; S->f_ind_max = st->f_ind_max;
        S->dec_reset_pending = st->dec_reset_pending; S->bpf_mem_len = st->bpf_mem_len; S->has_eoo = st->has_eoo; S->lcg = st->lcg;
        S->rxmax_cur = st->rxmax[0]; S->rxmax_h0 = st->rxmax[1]; S->rxmax_h1 = st->rxmax[2];
        S->fmax = st->fmax; S->foff_err = st->foff_err; S->rph_r = st->rx_phase[0]; S->rph_i = st->rx_phase[1];
        S->Dthresh = st->Dthresh; S->Dtmax12 = st->Dtmax12; S->Dtmax12_eoo = st->Dtmax12_eoo; S->snr_est = st->snr_est;
        S->bpf_phase = make_float2(st->bpf_phase[0], st->bpf_phase[1]);
        S->consumed_inv = a.acc[b * 4 + 0]; S->calls_inv = a.acc[b * 4 + 1]; S->valid_inv = a.acc[b * 4 + 2]; S->eoo_inv = a.acc[b * 4 + 3];
        S->n_calls = 0; S->n_rows = 0; S->uw_from_row = 0; S->consumed_round = 0; S->pending_valid = 0; S->out_base = S->valid_inv;
        S->go = 0; S->dt_valid = st->dt_valid; S->dt_new = 0; S->lds_sync = 0; S->need_decode = 0; S->batch_call0 = 0; S->pf_n = 0;
    }
    const int avail = a.avail[b];
    const long long wg_t0 = clock64();                 // per-stream duration of this launch (tail analysis: the launch lasts as long as its slowest stream)
    __syncthreads();
    PH_T0(); PH(0);

    for (int it = 0; it <= a.round_calls; it++) {  // <= round_calls calls (sizes of the per-launch arrays); the extra pass decodes what is pending
        // opaque per-iteration copy: keeps the compiler from hoisting every thread-index address computation of the
        // loop body into registers that stay live across the whole call (they starve the FFT correlator of registers)
        int tid = threadIdx.x; asm volatile("" : "+v"(tid));
        // ---- can this stream make another call right now?  Decided once, by thread 0: for the first call here, for every later
        // one at the end of the call before it (same thread, just ahead of that call's closing barrier: no barrier of its own).
        // Every operand is read up front and combined without branches -- a short-circuit chain is a chain of LDS round trips.
        auto prepare_next = [&]() {
            const int calls_inv = S->calls_inv, n_calls = S->n_calls, valid_inv = S->valid_inv, consumed = S->consumed_inv, nin_n = S->nin;
            const int n_rows = S->n_rows, st_n = S->state, sc = S->synced_count;
            const unsigned m0 = S->rxmax_h0, m1 = S->rxmax_cur, m2 = S->rxmax_h1;
            const int go = (int)(calls_inv < a.max_calls) & (int)(n_calls < a.round_calls) & (int)(valid_inv < a.feat_cap) & (int)(consumed + nin_n <= avail);   // feat_cap: room in features_out
            // the decoder runs right here, in this workgroup, when its output is needed: before a unique-word check
            // (radae_rxe.py:220-224 looks at the aux bits of the 8 frames before this one) or when the row buffer is full
            // ... or when this launch ends for the stream (out of samples, call limit)
            const int need = (int)(n_rows > 0) & ((go ^ 1) | ((int)(st_n == ST_SYNC) & (int)(((sc + 1) % 8) == 0)) | (int)(n_rows + 3 > a.dec_rows));
            S->need_decode = need; S->go = go;
            if ((go ^ 1) | need) S->pf_n = 0;                    // the decoder stage overlays xm
            S->rxmax_h1 = go ? m0 : m2; S->rxmax_h0 = go ? m1 : m0; S->rxmax_cur = go ? 0u : m1;   // rx_buf holds this call's samples and (parts of) the two calls' before
            S->state_before = st_n; S->nin_before = nin_n;
            S->valid_output = 0; S->endofover = 0; S->uw_fail = 0; S->candidate = 0;
        };
        if (it == 0) {
            if (tid == 0) prepare_next();
            __syncthreads();
        }
        if (S->need_decode) { PH(22); rx_decode_pending(sh, a, b); PH(20); }
        if (!S->go) break;
        const int nin = S->nin, state = S->state, ml = S->bpf_mem_len;
        const int mf0 = S->mf, n_rows0 = S->n_rows;          // stable until this call's state update
        // state machine (radae_rxe.py:248-297) and per-call bookkeeping, run by ONE thread: thread 0 after a search / candidate call,
        // the first lane of an idle wavefront during the demodulator of a synchronised call (nothing it writes is read before the
        // barrier that ends the call: the phases in between use mf0 / n_rows0 and their own locals)
        auto state_update = [&](int entry, int valid_out, int eoo) {
            int next_state = state;
            if (state == ST_SEARCH) {
                if (S->candidate) { next_state = ST_CANDIDATE; S->tmax_candidate = S->tmax; S->valid_count = 1; }
            } else if (state == ST_CANDIDATE) {
                if (entry) {
                    next_state = ST_SYNC;
                    S->dec_reset_pending = 1; S->synced_count = 0; S->uw_fail = 0; S->uw_errors = 0; S->uw_from_row = S->n_rows; S->valid_count = 25;
                } else if (S->candidate && abs(S->tmax - S->tmax_candidate) < RD_NCP) S->valid_count++;
                else next_state = ST_SEARCH;
            } else {
                const bool unsync_enable = !(a.unsync_off_after >= 0 && S->synced_count > a.unsync_off_after);     // radae_rxe.py:277-281
                if (S->candidate) S->valid_count = 25;
                else { S->valid_count--; if (unsync_enable && S->valid_count == 0) next_state = ST_SEARCH; }
                if (unsync_enable && (eoo || S->uw_fail)) next_state = ST_SEARCH;
            }
            S->dt_valid = (state != ST_SYNC && next_state != ST_SYNC) ? S->dt_new + 1 : 0;   // next call's Dt1 == this call's Dt2 (buffer dt_new)
            S->state = next_state;
            if (next_state == ST_SEARCH) S->nin = RD_NMF;
            S->mf++;
            const int ret = valid_out | (eoo << 1);
            const int call_idx = S->mf - 2;                   // 0-based index of this call since reset
            if (valid_out) {
                for (int k = 0; k < 3; k++) { const int rf = (k == 0) ? S->dec_reset_pending : 0; rnd->row_reset[S->n_rows + k] = rf; }
                S->dec_reset_pending = 0; S->n_rows += 3; S->pending_valid++; S->valid_inv++;
            }
            if (eoo) { S->has_eoo = 1; S->eoo_inv++; }
            const int nc = S->n_calls;
            rnd->call_ret[nc] = ret; rnd->call_row_lo[nc] = S->uw_from_row; rnd->call_row_hi[nc] = S->n_rows; rnd->call_trace_idx[nc] = call_idx;
            S->n_calls = nc + 1; S->calls_inv++;
        };
        const float2 bpf_phase = S->bpf_phase;
        if (state == ST_SYNC && !S->lds_sync) {      // the demod / check_pilots tables share LDS with the FFT correlator
            for (int i = tid; i < RD_M * RD_NC; i += NT_RX) sh->wfwd[i / RD_NC][i % RD_NC] = make_float2(tab->Wfwd[i / RD_NC][i % RD_NC][0], tab->Wfwd[i / RD_NC][i % RD_NC][1]);
            if (tid < RD_M) { const double nu = ((double)tid - 79.5) / 80.0; double v = 1.0; for (int m = 0; m < 8; m++) { sh->vm[m][tid] = v; v *= nu; } }
            __syncthreads();
            if (tid == 0) S->lds_sync = 1;
        }

        // ---- complex_bpf.bpf (dsp.py:63-102)
        const int cons0 = S->consumed_inv;
        const float2 *xin = rxin + cons0;
        const bool staged = S->pf_n == nin && ml == 102;      // the previous call fetched and mixed these samples while its equaliser ran
        for (int i = tid; i < ml; i += NT_RX) sh->xm[i] = sh->bmem[i];
        if (!staged) for (int i = tid; i < nin; i += NT_RX) { sh->xm[ml + i] = cmul(xin[i], cmul(bpf_phase, ld2(tab->bpf_E, i))); }
        __syncthreads();
        PH(18);
        // 101-tap FIR, three consecutive outputs per thread over a sliding register window (one LDS read per tap and
        // thread instead of one per tap and output); taps accumulate in ascending order
        const float2 e_last = ld2(tab->bpf_E, nin - 1);    // next call's starting phase (thread 0, below): fetched ahead of the FIR
        // in sync, refine() of this call searches fmax +-1 Hz: its f64 sincos tables only depend on last call's fmax, so the last
        // wavefront (idle during the FIR) prepares them now instead of everybody waiting for them later
        if (state == ST_SYNC && tid >= NT_RX - 64) { const double fm = S->fmax; refine_tables_sync(sh, tid - (NT_RX - 64), fm - 1.0, fm + 1.0, 0.1); }
        if (state == ST_SYNC && tid >= NT_RX - 128 && tid < NT_RX - 128 + 48) {      // check_pilots' 48 row draws of this call, likewise
            const int k = tid - (NT_RX - 128);
            const uint32_t x = LCG_A[k] * S->lcg + LCG_C[k];
            sh->rows48[k] = (int)((x >> 8) % RD_NMF);
            if (k == 47) sh->redi[15] = (int)x;                                        // the new LCG state, committed after the barrier below
        }
        if (tid >= NT_RX - 128 && tid < NT_RX - 64) {
            // the call after this one reads the next 800..1120 samples of the stream: their first touch costs an HBM access and,
            // with 256 streams spread over as many separate regions, an address translation (about 5,000 cycles together).  This
            // wavefront has nothing else to do during the FIR, so it takes that miss now (one load per 128-byte line, values
            // dropped); the real fetch -- under the equaliser of a synchronised call, at the start of the next call otherwise --
            // then finds the lines in the L2.
            const int l = tid - (NT_RX - 128), rem = min(avail - cons0 - nin, RD_NINMAX);
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < 2; q++) { const int i = (l + 64 * q) * 16; if (i < rem) t += xin[nin + i].x; }
            asm volatile("" :: "v"(t));
        }
        float2 filt[3];
        {
            const int i0 = 3 * tid;
            float2 eup[3];                                  // mix-up phasors of this thread's outputs: fetched ahead of the FIR
#pragma unroll
            for (int j = 0; j < 3; j++) eup[j] = ld2(tab->bpf_E, min(i0 + j, RD_NINMAX - 1));
            f32x2 acc[3] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };      // (re, im) of three outputs: one v_pk_fma_f32 per tap and output
            if (i0 < nin) {
                const f32x2 *xm2 = (const f32x2 *)sh->xm;
#pragma unroll 2
                for (int kb = 0; kb < 96; kb += 8) {
                    f32x2 x[10]; float h[8];
#pragma unroll
                    for (int u = 0; u < 10; u++) x[u] = xm2[i0 + kb + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) h[u] = sh->bpf_h[kb + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const f32x2 hh = { h[u], h[u] };
#pragma unroll
                        for (int j = 0; j < 3; j++) acc[j] = pk_fma(x[u + j], hh, acc[j]);
                    }
                }
                {
                    f32x2 x[7]; float h[5];
#pragma unroll
                    for (int u = 0; u < 7; u++) x[u] = xm2[i0 + 96 + u];
#pragma unroll
                    for (int u = 0; u < 5; u++) h[u] = sh->bpf_h[96 + u];
#pragma unroll
                    for (int u = 0; u < 5; u++) {
                        const f32x2 hh = { h[u], h[u] };
#pragma unroll
                        for (int j = 0; j < 3; j++) acc[j] = pk_fma(x[u + j], hh, acc[j]);
                    }
                }
            }
            float ar[3], ai[3];
#pragma unroll
            for (int j = 0; j < 3; j++) { ar[j] = acc[j][0]; ai[j] = acc[j][1]; }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int i = i0 + j;
                filt[j] = make_float2(0.0f, 0.0f);
                if (i < nin) filt[j] = cmul(make_float2(ar[j], ai[j]), cconj(cmul(bpf_phase, eup[j])));   // mix back up
            }
            // largest component of the new samples: sets the power-of-two scale of check_pilots' binary16 operand planes, so
            // that no input level (int16-scaled samples, a strong interferer) can overflow them
           
            float mloc = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; j++) mloc = fmaxf(mloc, fmaxf(fabsf(filt[j].x), fabsf(filt[j].y)));
            mloc = wave_max_f32(mloc);
            if ((tid & 63) == 0) atomicMax(&S->rxmax_cur, __float_as_uint(mloc));       // non-negative floats order like their bit patterns
        }
        PH(19);
        // new BPF memory = last 102 of [mem | new]; rx_buf shift (radae_rxe.py:196-197)
        float2 keep[(RD_RXBUF + NT_RX - 1) / NT_RX];
#pragma unroll
        for (int q = 0; q < (RD_RXBUF + NT_RX - 1) / NT_RX; q++) { const int i = tid + q * NT_RX; keep[q] = (i + nin < RD_RXBUF) ? sh->rxb[i + nin] : make_float2(0.0f, 0.0f); }
        float2 memv = make_float2(0.0f, 0.0f);
        if (tid < 102) memv = sh->xm[ml + nin - 102 + tid];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (RD_RXBUF + NT_RX - 1) / NT_RX; q++) { const int i = tid + q * NT_RX; if (i + nin < RD_RXBUF) sh->rxb[i] = keep[q]; }
#pragma unroll
        for (int j = 0; j < 3; j++) { const int i = 3 * tid + j; if (i < nin) sh->rxb[RD_RXBUF - nin + i] = filt[j]; }
        if (tid < 102) sh->bmem[tid] = memv;
        if (tid == 0) {
            S->bpf_phase = cmul(bpf_phase, e_last); S->pf_n = 0;
            if (state == ST_SYNC) S->lcg = (uint32_t)sh->redi[15];
            S->bpf_mem_len = 102; S->consumed_inv += nin; S->consumed_round += nin;
        }
        __syncthreads();

        PH(1);
        if (state == ST_SEARCH || state == ST_CANDIDATE) {
            // ---- acquisition.detect_pilots (dsp.py:178-231).  While searching nin == Nmf, so this call's Dt1 surface is
            // the previous call's Dt2 surface: |Dt2| is cached in HBM ([960][40] f32 per stream) and only the new
            // Dt2 is correlated (two adjacent t tiles per MFMA pairing).  First call after (re)entering search: both.
            // Correlation along t runs as FFT convolution: 1 forward + 40 inverse 2048-point transforms per surface, one
            // wavefront per transform (every wave repeats the forward one: same values, no workgroup barrier needed).
            float best = -1.0f; int bt = 0x7fffffff, bfi = 0;
            const bool cached = S->dt_valid != 0;
            const int oldb = cached ? S->dt_valid - 1 : 0, newb = 1 - oldb;
            float *cache = a.dtcache + (size_t)b * 2 * RD_NFC * RD_NMF;      // [2][f][t] |Dt| surfaces
            if (tid == 0) { S->lds_sync = 0; S->dt_new = newb; }
            {   // rowsum1 <- rowsum2 when the previous |Dt2| surface is reused as |Dt1|; branch-free (two slots per thread, the
                // clamped duplicates write equal values) so that no divergent join sits in front of the correlator
                const int t2 = min(tid + NT_RX, RD_NMF - 1);
                const float r1a = sh->rowsum1[tid], r1b = sh->rowsum1[t2], r2a = sh->rowsum2[tid], r2b = sh->rowsum2[t2];
                __syncthreads();
                sh->rowsum1[tid] = cached ? r2a : r1a; sh->rowsum1[t2] = cached ? r2b : r1b;
            }
            __syncthreads();
            PH(16);
#ifdef RX1_SEARCH_FFT
            rx_detect_fft(sh, a.fftG, a.ffttw, cache, cached ? 1 : 0, oldb, newb, best, bt, bfi);
#else
            {   // operand planes of the whole rx_buf (as for check_pilots in the synchronised state: one power-of-two scale from the running maximum)
                const unsigned mb = max(max(S->rxmax_cur, S->rxmax_h0), S->rxmax_h1);
                const int eb = min(max((int)((mb >> 23) & 0xffu), 32), 222);
                const float rx_sc = __uint_as_float((unsigned)(127 + 7 - (eb - 127)) << 23);
                const float rx_unsc = __uint_as_float((unsigned)(127 - 12 - 7 + (eb - 127)) << 23);
                for (int i = tid; i < RD_RXBUF; i += NT_RX) {
                    float2 v = sh->rxb[i]; v.x *= rx_sc; v.y *= rx_sc;
                    const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y;
                    const _Float16 l0 = (_Float16)(v.x - (float)h0), l1 = (_Float16)(v.y - (float)h1);
                    sh->srxh[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    sh->srxl[i] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
                rx_detect_mfma(sh, a.corr16, cache, cached ? 1 : 0, oldb, newb, rx_unsc, best, bt, bfi);
            }
#endif
            PH(17);
            PH(2);
            block_argmax(sh, best, bt, bfi);
            const float Dmax = best; const int tbest = bt, fbest = bfi;
            const float sr = sigma_r_from_rowsums(sh);
            if (tid == 0) {
                S->Dthresh = (double)(2.0f * sr) * RD_SQRT_NLOG_1EM5_5;
                if (Dmax > 0.0f) { S->tmax = tbest; S->f_ind_max = fbest; S->fmax = -50.0 + 2.5 * fbest;   /* = tab->fcoarse[fbest] (dsp.py:163), without the dependent global load in the serial section */ S->Dtmax12 = (double)Dmax; }
                else { S->tmax = 0; S->f_ind_max = 0; S->fmax = 0.0; S->Dtmax12 = 0.0; }
                S->candidate = S->Dtmax12 > S->Dthresh;
                // radae_rxe.py:256-260, decided HERE, by the one thread that also runs the state machine: evaluated by every thread after
                // the barrier it raced with thread 0's state update (valid_count++ a few lines further down), and a wavefront that read
                // the incremented count walked into refine() alone, one call early -- its barriers then paired with the wrong ones
                S->entry = S->candidate && (abs(S->tmax - S->tmax_candidate) < RD_NCP) && (S->valid_count + 1 > 3);
            }
            __syncthreads();
            PH(3);
        } else {
            // ---- in sync: refine, check_pilots, slips, UW, frequency correction, demod
            int tm_ref; double fm_ref;                  // refine()'s result, known to every thread (S->tmax / S->fmax become visible at the next barrier)
            {
                const int tm = S->tmax; const double fm = S->fmax;
                const int t0 = max(0, tm - 8);
                int tnew = tm; double fhat = fm;
                {   // check_pilots' operand planes of the whole rx_buf (read after refine(), whose barriers order these stores):
                    // operand scale 2^(7 - E), E = exponent of the largest component in rx_buf: samples x scale stay below 256 (the
                    // pilot planes carry 2^12); undone exactly in the |Dt| epilogue
                    const unsigned mb = max(max(S->rxmax_cur, S->rxmax_h0), S->rxmax_h1);
                    const int eb = min(max((int)((mb >> 23) & 0xffu), 32), 222);              // biased exponent, clamped so both factors stay normal
                    const float rx_sc = __uint_as_float((unsigned)(127 + 7 - (eb - 127)) << 23);
                    if (tid == 0) sh->redf[12] = __uint_as_float((unsigned)(127 - 12 - 7 + (eb - 127)) << 23);
                    for (int i = tid; i < RD_RXBUF; i += NT_RX) {
                        float2 v = sh->rxb[i]; v.x *= rx_sc; v.y *= rx_sc;
                        const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y;
                        const _Float16 l0 = (_Float16)(v.x - (float)h0), l1 = (_Float16)(v.y - (float)h1);
                        sh->rxh[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                        sh->rxl[i] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                    }
                }
                rx_refine(sh, &tnew, &fhat, t0, tm + 8 - t0, fm - 1.0, fm + 1.0, 0.1, true);      // tables: see the BPF stage
                tm_ref = tnew; fm_ref = 0.9 * fm + 0.1 * fhat;
                if (tid == 0) { S->tmax = tm_ref; S->fmax = fm_ref; }
            }
            PH(4);
            // check_pilots (dsp.py:273-320): refresh 48 pseudo-random rows
            // x_{i+1} = 1664525 x_i + 1013904223 (mod 2^32), 48 draws: thread i jumps straight to draw i (x_i = A^i x_0 + C_i)
            // (the draws were made during the BPF stage by an idle wavefront: rows48)
            // 96 rows (48 draws x {Dt1, Dt2}) x 40 frequencies on the f16 matrix cores, operands split in two binary16 planes.
            // The rx_buf planes were split ONCE for this call (rxh / rxl, before refine()): every sample is an operand of up to
            // 48 x 2 windows, and converting it inside the k loop (as this block did) was ~200 vector instructions per k-step
            // and wavefront -- the pace of the whole phase.  Four wavefronts, one per SIMD: (frame, f-tile group) x all three row
            // tiles, so each pilot plane fragment (L2, a.corr16) is applied to three row tiles.  The other four wavefronts do the
            // scalar-ish f64 work that only depends on refine()'s result (below).
            {
                const int wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
                const float rx_unsc = sh->redf[12];              // 2^(E - 19): undoes the operand scales exactly (set with the planes)
#ifdef RD_PHASE_TIMING
                const long long cr_t0 = clock64();
#endif
                if (wave < 4) {
                    // straight-line code per tile count (a tile count known only at run time puts the fragment loads behind
                    // branches, and the compiler then waits for ALL outstanding loads at the first use)
                    if (wave < 2) check_rows_tiles<3>(sh, a.corr16, wave & 1, 0, lane, rx_unsc);
                    else check_rows_tiles<2>(sh, a.corr16, wave & 1, 3, lane, rx_unsc);
                } else {
                    // ---- wavefronts 4..7 have no matrix work here.  refine() has fixed (tmax, fmax), so they prepare what the
                    // phases after this one used to compute with everybody waiting: the four correlations of check_pilots
                    // (dsp.py:307-313; wavefronts 4 and 5, two samples per lane) and the frequency-corrected window the
                    // demodulator reads (radae_rxe.py:209-218, :227-233; samples [0, 704) on wavefronts 6 and 7, the rest on 4 and 5)
                    const int k = tid & 127, hi = wave >= 6;
                    const int tm = tm_ref; const double w = 2.0 * PI_D * fm_ref / 8000.0;
                    if (!hi) {
                        double cr[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
                        for (int n = k; n < RD_M; n += 128) {
                            const float2 cf = cis_reduced(-w * n);
                            const double sn = cf.y, cs = cf.x;
                            const double2 rp = sh->pd[n], re = sh->pendd[n];
                            const int t0s[4] = { tm, tm + RD_NMF, tm + RD_M + RD_NCP, tm + RD_NMF };
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const float2 x = sh->rxb[t0s[q] + n];
                                const double qr = cs * x.x - sn * x.y, qi = -(cs * x.y + sn * x.x);    // conj(w_vec*rx)
                                const double2 r = q < 2 ? rp : re;
                                cr[2 * q] += qr * r.x - qi * r.y; cr[2 * q + 1] += qr * r.y + qi * r.x;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 8; q++) cr[q] = wave_sum_f64(cr[q]);
                        if (lane == 0) {
#pragma unroll
                            for (int q = 0; q < 8; q++) sh->corrp[wave - 4][q] = cr[q];
                        }
                    }
                    int t2 = tm;                                                  // timing slip, as the state update below applies it
                    if (t2 >= RD_NMF - RD_M) t2 -= RD_M;
                    if (t2 < RD_M) t2 += RD_M;
                    const double rph_r = S->rph_r, rph_i = S->rph_i;
                    float2 *rx1 = sh->xm;                                         // free between refine() and the next call's BPF
                    const int n_lo = hi ? 0 : 704, n_hi = hi ? 704 : RD_NEOO;
                    for (int n = n_lo + k; n < n_hi; n += 128) {
                        const float2 cs = cis_reduced(-w * (double)(n + 1));
                        const double c = cs.x, s_ = cs.y;
                        const float pr = (float)(rph_r * c - rph_i * s_), pi = (float)(rph_r * s_ + rph_i * c);
                        rx1[n] = cmul(sh->rxb[t2 - RD_NCP + n], make_float2(pr, pi));
                    }
                }
#ifdef RD_PHASE_TIMING
                if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 2 || wave == 4 || wave == 6)) atomicAdd((unsigned long long *)&g_phase_cycles[wave == 0 ? 21 : (wave == 2 ? 28 : (wave == 4 ? 29 : 30))], (unsigned long long)(clock64() - cr_t0));
#endif
            }
            __syncthreads();
            // duplicates in rows48 are harmless: every copy writes the same value
            if (tid < 96) {
                float s = 0.0f;
                for (int f = 0; f < RD_NFC; f++) s += sh->absd[tid][f];
                const int t = sh->rows48[tid >> 1];
                if (tid & 1) sh->rowsum2[t] = s; else sh->rowsum1[t] = s;
            }
            __syncthreads();
            PH(5);
            // From here to the equaliser ONE phase: wavefronts 0..5 run the demodulator DFT below; wavefront 6 meanwhile reduces the row
            // sums to the Rayleigh thresholds, decides candidate / end-of-over / slip and runs the state machine (none of it feeds the DFT:
            // the corrected window was cut with the slip-adjusted timing by the side wavefronts above); wavefront 7 advances the phase
            // accumulator.  As a phase of its own (workgroup-wide f64 reduction, a barrier, one thread's serial work) this was ~3 k cycles.
            const double w = 2.0 * PI_D * S->fmax / 8000.0;
            const double rph_r = S->rph_r, rph_i = S->rph_i;
            float2 *rx1 = sh->xm;
            if (tid >= NT_RX - 128 && tid < NT_RX - 64) {
                const int l = tid - (NT_RX - 128);
                double r0 = 0.0, r1 = 0.0;
                for (int t = l; t < RD_NMF; t += 64) { r0 += (double)sh->rowsum1[t]; r1 += (double)sh->rowsum2[t]; }
                r0 = wave_sum_f64(r0); r1 = wave_sum_f64(r1);
                if (l == 0) {
                    const int tm = S->tmax;
                    double red[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) red[q] = sh->corrp[0][q] + sh->corrp[1][q];     // prepared during the matrix phase above
                    const float sr = sigma_r_from_sums(r0, r1);
                    const double D = hypot(red[0], red[1]) + hypot(red[2], red[3]);
                    const double De = hypot(red[4], red[5]) + hypot(red[6], red[7]);
                    S->Dthresh = (double)(2.0f * sr) * RD_SQRT_NLOG_1EM4_5;          // 2 sigma_r sqrt(-ln(P / 5)), P = 1e-4 (dsp.py:318-320)
                    const double Dthresh_eoo = (double)(2.0f * sr) * RD_SQRT_NLOG_1EM5_5;
                    S->Dtmax12 = D; S->Dtmax12_eoo = De;
                    const int eoo = De > Dthresh_eoo;
                    S->candidate = D > S->Dthresh; S->endofover = eoo;
                    int nn = RD_NMF, t2 = tm;                                       // radae_rxe.py:209-218
                    if (t2 >= RD_NMF - RD_M) { nn = RD_NMF + RD_M; t2 -= RD_M; }
                    if (t2 < RD_M) { nn = RD_NMF - RD_M; t2 += RD_M; }
                    S->nin = nn; S->tmax = t2;
                    S->synced_count++;                                              // :220-224
                    if (S->synced_count % 8 == 0) { if (S->uw_errors > 7) S->uw_fail = 1; S->uw_errors = 0; S->uw_from_row = S->n_rows; }
                    state_update(0, !eoo, eoo);                                     // valid_output of a synchronised call = !endofover (set by the EQ below)
                }
            }
            // frequency correction (:227-233): rx_phase advances e^{-jw} per sample in complex128; the corrected window rx1 was
            // written by the side wavefronts of the matrix phase
            // the phase accumulator advances on the last wavefront, which has no part in the DFT that follows (a f64 sincos on
            // thread 0 would hold back wavefront 0 and with it the barrier after the DFT)
            if (tid == NT_RX - 64) { double s, c; sincos(-w * (double)RD_NEOO, &s, &c); S->rph_r = rph_r * c - rph_i * s; S->rph_i = rph_r * s + rph_i * c; }
            PH(7);
            // The NEXT call's input, fetched under the demodulator and the equaliser (an HBM / L2 round trip of several thousand
            // cycles that would otherwise open the next call): up to nin_max samples and their mix-down phasors -- the next size
            // and whether there is a next call are settled by the state update running beside the DFT, so the decision what to
            // keep comes at the end of this call, where the samples are stored mixed down into xm (free from the DFT on).  Same
            // products as the BPF stage computes, so the result does not depend on which of the two ran.
            float2 pfx[3], pfe[3];
            {
                const int lim = min(avail - S->consumed_inv, RD_NINMAX);
                const float2 *xn = rxin + S->consumed_inv;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int i = tid + q * NT_RX;
                    pfx[q] = pfe[q] = make_float2(0.0f, 0.0f);
                    if (i < lim) { pfx[q] = xn[i]; pfe[q] = ld2(tab->bpf_E, i); }
                }
            }
            // receiver_one (dsp.py:487-526): window [16:176] of each 192-sample symbol, 160->30 DFT
            // two lanes per (symbol, carrier), 80 samples each in four independent chains; the halves meet through a lane swap
            if (tid < 2 * 6 * RD_NC) {
                const int o = tid >> 1, hf = tid & 1, s = o / RD_NC, c = o - s * RD_NC;
                const float2 *x = rx1 + s * RD_SYM + RD_NCP - 16 + 80 * hf;
                float2 acc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] = make_float2(0.0f, 0.0f);
#pragma unroll 5
                for (int n = 0; n < 80; n += 4)
#pragma unroll
                    for (int u = 0; u < 4; u++) acc[u] = cadd(acc[u], cmul(x[n + u], sh->wfwd[80 * hf + n + u][c]));
                float2 t = cadd(cadd(acc[0], acc[1]), cadd(acc[2], acc[3]));
                t.x += quad_dpp<QUAD_XOR1>(t.x); t.y += quad_dpp<QUAD_XOR1>(t.y);
                if (hf == 0) sh->sym[s][c] = t;
            }
            __syncthreads();
            PH(8);
            const int endofover = S->endofover, n_rows = n_rows0;
            int pf_n = 0;                                        // settled now (state update above): is there a next call, does the decoder stage run first
            {
                const int nn = S->nin;
                const bool go_n = !(S->calls_inv >= a.max_calls || S->n_calls >= a.round_calls || S->valid_inv >= a.feat_cap) && S->consumed_inv + nn <= avail;
                const bool dec_n = S->n_rows > 0 && ((S->state == ST_SYNC && ((S->synced_count + 1) % 8) == 0) || S->n_rows + 3 > a.dec_rows);
                if (go_n && !dec_n) pf_n = nn;
            }
            float *zrow = a.zrows + ((size_t)b * a.dec_rows + n_rows) * RD_LATENT;   // 3 rows = 240 contiguous floats
            float *eoo_dst = a.eoo_out ? a.eoo_out + (size_t)b * RD_NEOOBITS : nullptr;
            const int call_idx0 = mf0 - 1;
            if (!endofover) {
                // est_pilots (dsp.py:418-435) for pilot rows 0 and 5
                if (tid < 2 * RD_NC) {
                    const int i = tid / RD_NC, c = tid - i * RD_NC;
                    const int cm = c == 0 ? 1 : (c == RD_NC - 1 ? RD_NC - 2 : c);
                    const float2 *row = sh->sym[i ? 5 : 0];
                    float2 g0 = make_float2(0.0f, 0.0f), g1 = g0;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float pp = sh->eqP[cm - 1 + k];
                        const float2 h = make_float2(row[cm - 1 + k].x / pp, row[cm - 1 + k].y / pp);
                        g0 = cadd(g0, cmul(sh->eqPmat[c][0][k], h));
                        g1 = cadd(g1, cmul(sh->eqPmat[c][1][k], h));
                    }
                    sh->rp[i][c] = cadd(g0, cmul(g1, sh->eqrot[c]));
                }
                __syncthreads();
                // update_snr_est (dsp.py:438-456) + coarse magnitude (:477-482): per-carrier terms reduced by wave shuffles
                // two independent chains on two wavefronts: 0 = coarse magnitude (the EQ below waits for it), 1 = SNR estimate
                if (tid < 128) {
                    const int c = tid & 63;
                    if (tid < 64) {
                        float pm = 0.0f;
                        if (c < RD_NC) {
                            const float2 r0 = sh->rp[0][c], r1 = sh->rp[1][c];
                            const float a0 = hypotf(r0.x, r0.y), a1 = hypotf(r1.x, r1.y); pm = a0 * a0 + a1 * a1;
                        }
                        pm = wave_sum_f32(pm);                                                   // lanes >= Nc hold zeros
                        if (c == 0) {
                            float mag = powf(pm / 60.0f, 0.5f) + 1e-6f;
                            S->mag = (mag * fabsf(sh->eqP[0])) / sh->eq_pg;
                            S->valid_output = 1;
                        }
                    } else {
                        float s1 = 0.0f, s2 = 0.0f;
                        if (c < RD_NC) {
                            const float2 r0 = sh->rp[0][c], pc = sh->sym[0][c];
                            const float2 rc = cmul(pc, unit_conj(r0));
                            const float ap = hypotf(pc.x, pc.y); s1 = ap * ap; s2 = fabsf(rc.y) * fabsf(rc.y);
                        }
                        s1 = wave_sum_f32(s1); s2 = wave_sum_f32(s2);
                        if (c == 0) {
                            const float S1 = s1, S2 = s2 + 1e-12f;
                            float snr = S1 / (2.0f * S2) - 1.0f;
                            if (snr <= 0.0f) snr = 0.1f;
                            float snrdB = 10.0f * log10f(snr);
                            snrdB = (snrdB - 2.513f) / 0.8070f;
                            const float snr3k = snrdB + sh->eq_snrc1 + sh->eq_snrc2;
                            S->snr_est = 0.9f * S->snr_est + 0.1f * snr3k;
                        }
                    }
                }
                __syncthreads();
                const float mag = S->mag;
                // linear-interpolated phase EQ of the 4 data symbols (:468-474), demap to z_hat
                if (tid < RD_NS * RD_NC) {
                    const int k = 1 + tid / RD_NC, c = tid % RD_NC;
                    const float2 r0 = sh->rp[0][c], r1 = sh->rp[1][c];
                    const float2 slope = make_float2((r1.x - r0.x) / 5.0f, (r1.y - r0.y) / 5.0f);
                    const float2 ch = make_float2(slope.x * (float)k + r0.x, slope.y * (float)k + r0.y);
                    const float2 v = cmul(sh->sym[k][c], unit_conj(ch));
                    const float zr = v.x / mag, zi = v.y / mag;
                    zrow[2 * tid] = zr; zrow[2 * tid + 1] = zi;
                    if (a.trace_z && call_idx0 < a.trace_cap) { float *tz = a.trace_z + ((size_t)b * a.trace_cap + call_idx0) * RD_ZMF; tz[2 * tid] = zr; tz[2 * tid + 1] = zi; }
                }
            } else {
                // EOO branch (:513-524): mean of the three pilots per carrier, symbols 2..4 carry the 180 soft bits
                if (tid < 3 * RD_NC) {
                    const int k = 2 + tid / RD_NC, c = tid % RD_NC;
                    const float pp = tab->P[c], pe = tab->Pend[c];
                    const float2 s = make_float2(sh->sym[0][c].x / pp + sh->sym[1][c].x / pe + sh->sym[5][c].x / pe,
                                                 sh->sym[0][c].y / pp + sh->sym[1][c].y / pe + sh->sym[5][c].y / pe);
                    const float2 v = cmul(sh->sym[k][c], unit_conj(s));
                    if (eoo_dst) { eoo_dst[2 * tid] = v.x; eoo_dst[2 * tid + 1] = v.y; }
                    if (a.trace_z && call_idx0 < a.trace_cap) { float *tz = a.trace_z + ((size_t)b * a.trace_cap + call_idx0) * RD_ZMF; tz[2 * tid] = v.x; tz[2 * tid + 1] = v.y; }
                }
            }
            if (pf_n) {
                const float2 ph = S->bpf_phase;                  // already advanced to the next call's start by this call's BPF stage
#pragma unroll
                for (int q = 0; q < 3; q++) { const int i = tid + q * NT_RX; if (i < pf_n) sh->xm[102 + i] = cmul(pfx[q], cmul(ph, pfe[q])); }
                if (tid == 0) S->pf_n = pf_n;
            }
            __syncthreads();
        }

        PH(9);
        // ---- state machine (radae_rxe.py:248-297).  Sync entry needs the whole workgroup for refine().
        const int do_entry = (state == ST_CANDIDATE) && S->entry;      // S->entry is final since the barrier that ended the detect stage
        if (do_entry) {
            const int tm = S->tmax; const double fm = S->fmax;
            const int t0 = max(0, tm - 1);
            int tnew = tm; double fnew = fm;
            rx_refine(sh, &tnew, &fnew, t0, tm + 2 - t0, fm - 10.0, fm + 10.0, 0.25, false);
            if (tid == 0) { S->tmax = tnew; S->fmax = fnew + S->foff_err; S->foff_err = 0.0; }
            __syncthreads();
        }
        if (tid == 0 && state != ST_SYNC) state_update(do_entry, S->valid_output, S->endofover);   // (in sync: done during the demodulator, above)
        if (tid == 0 && a.trace) {                        // per-call trace record: after the EQ, it carries this call's SNR estimate
            const int call_idx = S->mf - 2;               // 0-based index of this call since reset
            if (call_idx < a.trace_cap) {
                rd_rx_trace *tr = a.trace + (size_t)b * a.trace_cap + call_idx;
                tr->state_before = S->state_before; tr->state_after = S->state; tr->nin_before = S->nin_before; tr->nin_after = S->nin; tr->ret = S->valid_output | (S->endofover << 1);
                tr->tmax = S->tmax; tr->f_ind_max = S->f_ind_max; tr->valid_count = S->valid_count; tr->uw_errors = S->uw_errors; tr->synced_count = S->synced_count;
                tr->snr_int = (int)S->snr_est; tr->fmax = S->fmax; tr->Dthresh = S->Dthresh; tr->Dtmax12 = S->Dtmax12; tr->Dtmax12_eoo = S->Dtmax12_eoo; tr->snrdB_3k_est = S->snr_est;
            }
        }
        if (tid == 0) prepare_next();                      // the next call's go / decode decision (see the loop top)
        __syncthreads();
        PH(10);
    }

    // ---- write the stream state back
    __syncthreads();
    PH(11);
    for (int i = tid; i < RD_RXBUF; i += NT_RX) { st->rx_buf[i][0] = sh->rxb[i].x; st->rx_buf[i][1] = sh->rxb[i].y; }
    for (int i = tid; i < RD_NMF; i += NT_RX) { st->rowsum1[i] = sh->rowsum1[i]; st->rowsum2[i] = sh->rowsum2[i]; }
    for (int i = tid; i < 102; i += NT_RX) { st->bpf_mem[i][0] = sh->bmem[i].x; st->bpf_mem[i][1] = sh->bmem[i].y; }
    if (tid == 0) {
        st->state = S->state; st->nin = S->nin; st->tmax = S->tmax; st->tmax_candidate = S->tmax_candidate; st->valid_count = S->valid_count;
        st->uw_errors = S->uw_errors; st->synced_count = S->synced_count; st->mf = S->mf; st->f_ind_max = S->f_ind_max;
        st->dec_reset_pending = S->dec_reset_pending; st->bpf_mem_len = S->bpf_mem_len; st->has_eoo = S->has_eoo; st->lcg = S->lcg; st->dt_valid = S->dt_valid;
        st->rxmax[0] = S->rxmax_cur; st->rxmax[1] = S->rxmax_h0; st->rxmax[2] = S->rxmax_h1;
        st->fmax = S->fmax; st->foff_err = S->foff_err; st->rx_phase[0] = S->rph_r; st->rx_phase[1] = S->rph_i;
        st->Dthresh = S->Dthresh; st->Dtmax12 = S->Dtmax12; st->Dtmax12_eoo = S->Dtmax12_eoo; st->snr_est = S->snr_est;
        st->bpf_phase[0] = S->bpf_phase.x; st->bpf_phase[1] = S->bpf_phase.y; st->consumed += S->consumed_round;
        rnd->n_calls = S->n_calls; rnd->n_rows = S->n_rows; rnd->uw_from_row = S->uw_from_row; rnd->consumed = S->consumed_round;
        rnd->out_base = S->out_base;
        a.acc[b * 4 + 0] = S->consumed_inv; a.acc[b * 4 + 1] = S->calls_inv; a.acc[b * 4 + 2] = S->valid_inv; a.acc[b * 4 + 3] = S->eoo_inv;
        a.status[b * 4 + 0] = S->nin; a.status[b * 4 + 1] = S->state == ST_SYNC; a.status[b * 4 + 2] = (int)S->snr_est; a.status[b * 4 + 3] = S->state;
        if (a.wg_cycles) a.wg_cycles[b] = clock64() - wg_t0;
        if (S->n_calls) atomicAdd(&a.progress[0], S->n_calls);
        if (S->calls_inv < a.max_calls && S->valid_inv < a.feat_cap && S->consumed_inv + S->nin <= avail) atomicAdd(&a.progress[1], 1);   // stopped at the per-launch limit
    }
}

extern "C" int rd_launch_rx_sync2(const rd_sync_args *a, rd_stream_t s);      /* rade_rx2.hip */

extern "C" int rd_launch_rx_sync(const rd_sync_args *a, rd_stream_t s)
{
    if (a->B <= 0) return 0;
    if ((a->variant & 0xff) == 2) return rd_launch_rx_sync2(a, s);
    static int attr_set_dev[64];                     // the attribute is per device (one engine per GPU in a multi-GPU host process)
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (!attr_set_dev[dev_ & 63]) { (void)hipFuncSetAttribute((const void *)k_rx_sync, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RxShared)); attr_set_dev[dev_ & 63] = 1; }
    hipLaunchKernelGGL(k_rx_sync, dim3(a->B), dim3(NT_RX), sizeof(RxShared), (hipStream_t)s, *a);
    return (int)hipGetLastError();
}

// (re)initialise every stream's receiver state on the device (radae_rxe.py:128-142)
__global__ __launch_bounds__(256) void k_rx_reset(rd_rx_stream *st, const unsigned *seeds, double foff_err)
{
    rd_rx_stream *s = st + blockIdx.x;
    float *raw = (float *)s;
    for (int i = threadIdx.x; i < (int)(sizeof(rd_rx_stream) / 4); i += blockDim.x) raw[i] = 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        s->state = ST_SEARCH; s->nin = RD_NMF; s->mf = 1; s->bpf_mem_len = 100; s->lcg = seeds ? seeds[blockIdx.x] : 1u;
        s->rx_phase[0] = 1.0; s->rx_theta = 0.0; s->bpf_phase[0] = 1.0f; s->foff_err = foff_err;
    }
}
extern "C" int rd_launch_rx_reset(rd_rx_stream *st, const unsigned *seeds, double foff_err, int B, rd_stream_t s)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_rx_reset, dim3(B), dim3(256), 0, (hipStream_t)s, st, seeds, foff_err);
    return (int)hipGetLastError();
}


#endif  // !RADE_RX2_TU