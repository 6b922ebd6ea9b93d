// rade_devutil.h -- device-side helpers shared by the HIP translation units (rade_kernels.hip: transmit side, channel, GEMMs, scans;
// rade_rx.hip: the receiver).  gfx950 only: 64-lane wavefronts, DPP lane exchange, hardware exp2 / rcp.
#ifndef RADE_DEVUTIL_H
#define RADE_DEVUTIL_H
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
// the same product with every operation rounded on its own (no fused multiply-add: contraction switched off for this function, whatever the
// command line says -- HIP's __fmul_rn / __fadd_rn are plain operators and contract like any other): what NumPy's complex64 multiply does, and,
// unlike cmul, whose contraction the compiler decides per call site, the same bits in every kernel that uses it
__device__ __forceinline__ float2 cmul_nc(float2 a, float2 b)
{
#pragma clang fp contract(off)
    const float rr = a.x * b.x, ii = a.y * b.y, ri = a.x * b.y, ir = a.y * b.x;
    return make_float2(rr - ii, ri + ir);
}

// Double-precision expressions whose LAST BIT reaches a decision: the receiver's frequency estimate fmax = 0.9 fmax + 0.1 fhat and the grid value
// fhat = start + i delta are doubles in the reference (radae_rxe.py:202-206), and the next call's grid np.arange(fmax - 1, fmax + 1, 0.1) has 20 or 21
// points depending on whether (fmax + 1) - (fmax - 1) rounds to 2 or to 2 + 4e-16.  Contracted into an FMA (one rounding instead of two) fmax differs
// from the reference's in the last bit now and then, the device searched 20 frequencies where the reference searches 21, and when the 21st won the two
// parted for good (two of 1248 random utterances, tools/parity_sweep.py seeds 99 / 12345).  These forms round like numpy and the C oracle.
__device__ __forceinline__ double dlin2_nc(double a, double x, double b, double y)
{
#pragma clang fp contract(off)
    const double p = a * x, q = b * y;
    return p + q;
}
__device__ __forceinline__ double dgrid_nc(double start, int i, double delta)
{
#pragma clang fp contract(off)
    const double p = (double)i * delta;
    return start + p;
}

// one term of the modulator's 30-term IDFT: acc += s.re (w.re, w.im); acc += s.im (-w.im, w.re), as four fused multiply-adds.  Every place that synthesises a
// transmit sample uses this one form, so the same sample comes out the same bits wherever it is computed.
// Round 4 issued the four as two v_pk_fma_f32 (both components of a sample in one instruction, the (s.re, s.re) / (-w.im, w.re) operands formed by the
// instruction's op_sel / neg_lo modifiers): 10 us faster per modulator launch, and NOT reproducible -- under load (three batches in flight) about one 16-sample
// block in 15,000 frames came out different from run to run, always lanes 48..63 of a wavefront, always a data symbol, the encoder's latents bit-identical
// (tools/tx_determinism.py: 125..283 differing blocks per 3.8 M frames with the packed form, 0 in 19 M frames with this one; profiles/r05_tx_determinism.txt).
// The plain packed FMAs of the encoder's GRU scan (no operand modifiers) do not show it (same runs: z identical).  It is the hazard round 3 met in the receiver's FIR
// (HISTORY.md 3.7: "a dense sequence of packed-f32 FMAs with op_sel operands misbehaves in the last quarter-wave when the SIMD's other wavefront belongs to a different
// workgroup running matrix instructions") -- which is exactly where a modulator wavefront runs in the pipelined bench.  Rule for this code base: NO v_pk_*_f32 with op_sel / neg
// modifiers in any kernel that can share a SIMD with another kernel's wavefronts; tests/test_hip_parity.py::test_transmit_side_is_bit_reproducible_under_load holds the line.
__device__ __forceinline__ f32x2 idft_term(f32x2 acc, float2 sy, float2 w)
{
    acc[0] = fmaf(sy.x, w.x, acc[0]); acc[1] = fmaf(sy.x, w.y, acc[1]);
    acc[0] = fmaf(sy.y, -w.y, acc[0]); acc[1] = fmaf(sy.y, w.x, acc[1]);
    return acc;
}

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
// the transmitter's limiter tanh(|x|) e^{j angle(x)} = x tanh(|x|) / |x| (radae.py:218, dsp.py:377) for every kernel that synthesises a transmit sample: the magnitude on the
// hardware square root, tanh on exp2 / rcp as the gates have it (absolute error of tanh ~1e-7, i.e. ~1e-7 |x| on the sample: the golden bar is 2e-5).  libm's hypotf + tanhf
// + a division were as many instructions per sample as the 30-term IDFT in front of them.
__device__ __forceinline__ float2 pa_limit(float2 x)
{
    const float mag = __builtin_amdgcn_sqrtf(fmaf(x.x, x.x, x.y * x.y));
    if (mag == 0.0f) return make_float2(0.0f, 0.0f);
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * mag);           // e^{-2 |x|}
    const float g = (1.0f - e) * __builtin_amdgcn_rcpf((1.0f + e) * mag);
    return make_float2(x.x * g, x.y * g);
}
__device__ __forceinline__ float gate_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }
__device__ __forceinline__ float2 ld2(const float (*p)[2], int i) { return make_float2(p[i][0], p[i][1]); }

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
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
#endif
