// rade_rx.hip -- THE receiver of the RADE hot path: do_radae_rx (radae_rxe.py:171-330) for one stream per workgroup, all of its calls in one launch.
//
//   k_rx_bpf      complex_bpf.bpf (dsp.py:63-102) for every sample of a rade_batch_rx invocation, ahead of the receiver kernel (round 4: the
//                 band-pass filter does not depend on any sync decision, so it left the per-stream serial chain)
//   k_rx_sync2    acquisition (detect_pilots / refine / check_pilots, dsp.py:178-320), sync state machine, frequency correction, OFDM demod +
//                 3-pilot LS EQ (dsp.py:418-526), the CoreDecoder stage (radae_base.py:358-430) and UW accounting (rade_api.c:480-513): 256 threads
//                 and at most 80 KB of LDS per stream, so that two streams share a CU
//   k_batch_reset radae_rxe.py:128-142 (and the encoder / decoder start-of-utterance state, one launch)
//
// Rounds 2-3 carried a second receiver kernel (k_rx_sync: 512 threads, one stream per CU) that this one was forked from; it is gone: one
// receiver, every fix lands once.
#include "rade_devutil.h"
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ---- decoder stage inside the receiver: LDS layout helpers and product descriptors ----
#define DQ_XB 96                 // 8-half blocks per x row
#define DQ_HB 16                 // blocks per GRU-output row (12 used)
#define DQ_PEND_MAX 64           // pending rows a stream can hold (engine: dec_rows <= 63)
// floats per row of the GRU-input sums in LDS: 288 + 4.  With 288 (= 9 x 32) every row started on the same bank, and the 16-byte accesses of a product's epilogue -- lanes = rows --
// were 8-way conflicts: two thirds of ALL bank-conflict cycles of the kernel (tools/rx2_lds_conflicts.sh, round 5); with 292 a row starts 16 bytes further on.
#ifndef DQ_GIS
#define DQ_GIS 292
#endif
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

enum { ST_SEARCH = 0, ST_CANDIDATE = 1, ST_SYNC = 2 };
struct RxScalars {
    int state, nin, tmax, tmax_candidate, valid_count, uw_errors, synced_count, mf, f_ind_max, dec_reset_pending, has_eoo;
    uint32_t lcg;
    unsigned rxmax_cur, rxmax_h0, rxmax_h1;   // float bits of max |re|,|im| of the filtered samples of this call / the two calls before (check_pilots operand scale)
    int consumed_inv, calls_inv, valid_inv, eoo_inv, n_calls, n_rows, uw_from_row, consumed_round, pending_valid, out_base;
    int tab_ok;               // refine()'s per-frequency constants for the CURRENT fmax are in LDS (left by the previous synchronised call's idle wavefront)
    int bpf_grid, nin0;       // the stream's calls still follow the block grid of the invocation's band-pass pre-pass (k_rx_bpf); the grid's first block length
    int entry;                // this candidate call enters sync (decided by thread 0 before a barrier: see do_entry)
    int go, need_decode, batch_call0, state_before, nin_before, valid_output, endofover, uw_fail, candidate, dt_valid, dt_new, lds_sync;
    float snr_est, mag; float2 bpf_phase;
    double fmax, foff_err, rph_r, rph_i, Dthresh, Dtmax12, Dtmax12_eoo;
    double rph_th;                        // k_rx_sync2: the phase accumulator as an angle in [-pi, pi] (rph_r + j rph_i = e^{j rph_th})
};

__device__ __forceinline__ float sigma_r_from_sums(double t1, double t2)
{   // dsp.py:218-220: (mean|Dt1| + mean|Dt2|)/sqrt(pi/2)/2 in float32
    const float k = (float)sqrt(PI_D / 2.0);
    const float m1 = (float)(t1 / (RD_NMF * RD_NFC)) / k, m2 = (float)(t2 / (RD_NMF * RD_NFC)) / k;
    return (m1 + m2) / 2.0f;
}

__device__ static constexpr uint32_t LCG_A[48] = { 1664525u, 389569705u, 2940799637u, 158984081u, 2862450781u, 3211393721u, 1851289957u, 3934847009u, 2184914861u, 246739401u, 1948736821u, 2941245873u, 4195587069u, 4088025561u, 980655621u, 2001863745u, 657792333u, 65284841u, 1282409429u, 3808694225u, 2968195997u, 2417331449u, 2878627493u, 307989601u, 504219373u, 1897564169u, 2574089845u, 3294562801u, 3478292285u, 2651335705u, 2523738949u, 666245249u, 4137395341u, 2604435753u, 1706708245u, 3963176977u, 3678957277u, 3530469177u, 3858799589u, 629287073u, 3146069549u, 3820924489u, 2403397557u, 2390444593u, 2593868413u, 4291139161u, 1705056389u, 3186638017u };
__device__ static constexpr uint32_t LCG_C[48] = { 1013904223u, 1196435762u, 3519870697u, 2868466484u, 1649599747u, 2670642822u, 1476291629u, 2748932008u, 2180890343u, 2498801434u, 3421909937u, 3167820124u, 2636375307u, 3801544430u, 28987765u, 2210837584u, 3039689583u, 1338634754u, 1649346937u, 2768872580u, 2254235155u, 2326606934u, 1719328701u, 1061592568u, 53332215u, 1140036074u, 4224358465u, 2629538988u, 1946028059u, 573775550u, 1473591045u, 95141024u, 1592739711u, 1618554578u, 4257218569u, 2685635028u, 2617994019u, 740185638u, 4194465613u, 2426187848u, 967350023u, 366635194u, 2557108433u, 3503432700u, 353185579u, 706247310u, 408928405u, 1855199472u };

#define NT2 256
// thread index rebuilt from the lane counter and the wavefront's index (held in a scalar register): three instructions wherever it
// is needed, instead of one value that stays live -- and gets spilled -- across the whole receive call (rx_tid() keeps threadIdx.x
// alive the same way)
// developer aid (-DRX2_CENSUS, tools/rx2_census.sh): a.variant >> 8 is a mask that skips the decoder stage / its recurrence or runs an idempotent
// phase twice, so that the per-phase share of the instruction counters (rocprofv3 --pmc SQ_INSTS_*) is the difference between two runs
#ifdef RX2_CENSUS
#define CENSUS(bit) ((a.variant >> 8) & (bit))
#define CENSUS_REPS(bit) (CENSUS(bit) ? 2 : 1)
#else
#define CENSUS(bit) 0
#define CENSUS_REPS(bit) 1
#endif
__device__ __forceinline__ int rx2_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int rx2_tid(int wv) { int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(l)); return (wv << 6) | l; }
#define NW2 (NT2 / 64)
#ifdef RD_PHASE_TIMING   // developer aid: per-phase shader-clock totals of workgroup 0 (tools/ab_build.sh timing -DRD_PHASE_TIMING; tools/phase_timing2.py)
__device__ long long g_phase_cycles2[32];
#define PH2_T0() long long ph2_t_ = clock64()
#define PH2(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { long long n_ = clock64(); atomicAdd((unsigned long long *)&g_phase_cycles2[i], (unsigned long long)(n_ - ph2_t_)); ph2_t_ = n_; } } while (0)
#define PH2_RESTART() do { ph2_t_ = clock64(); } while (0)
extern "C" void rd_debug_phase_cycles2(long long *out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles2), sizeof(long long) * 32); long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles2), z, sizeof z); }
#else
#define PH2_T0() do { } while (0)
#define PH2(i) do { } while (0)
#define PH2_RESTART() do { } while (0)
#endif
#define DQ2_ROWS 12
#ifndef RX2_DQ_D3
#define RX2_DQ_D3 4
#endif

struct DecShared2 {
    __attribute__((aligned(16))) _Float16 xh[DQ2_ROWS + 2][DQ_XB * 8];   // physical row 0: conv history, 1..12: the chunk, 13: zeros
    __attribute__((aligned(16))) _Float16 xl[DQ2_ROWS + 2][DQ_XB * 8];
    __attribute__((aligned(16))) float gi[DQ2_ROWS][DQ_GIS];
    __attribute__((aligned(16))) _Float16 hbh[DQ2_ROWS][DQ_HB * 8], hbl[DQ2_ROWS][DQ_HB * 8];
    __attribute__((aligned(16))) float hs[2][96];
    int rst[DQ_PEND_MAX];
    int err[DQ_PEND_MAX];
};

// NT adjacent column tiles of a product for rows [0, Tb), Tb <= 12: ONE 16-row tile (k_rx_sync's version carries two).
// sync_first: the barrier that separates this product from the phase before it is taken HERE, behind the first weight requests (which depend on nothing the phase before
// wrote): the round trip to L2 (~800 cycles, once per phase: five phases per layer and chunk) runs while the workgroup's other wavefronts arrive, instead of after them
template <int NT, bool SINGLE>
__device__ __forceinline__ void dq2_gemm_tiles_(DecShared2 *sh_, const DqGemm g_, int ct_, int Tb_, unsigned rstmask_, int sync_first)
{
    constexpr int D = NT == 1 ? 12 : ((NT == 3 && SINGLE) ? RX2_DQ_D3 : 4);   // k-steps of weights in flight (one plane of a three-tile product: 48 registers at 4, 96 at 8)
    const int ct = uni(ct_), Tb = uni(Tb_); const unsigned rstmask = (unsigned)uni((int)rstmask_);
    const int nct = uni(g_.nct), N = uni(g_.N), from_hb = uni(g_.from_hb), ktap = uni(g_.ktap), ks0 = uni(g_.ks0), nks = uni(g_.nks), init_gi = uni(g_.init_gi),
              outk = uni(g_.out), ocol = uni(g_.ocol), act = uni(g_.act), gstride = uni(g_.gstride);
    DecShared2 *sh = uni_ptr(sh_);
    glb_u16 *wbase = (glb_u16 *)uni_ptr(g_.wa); glb_cf32 *biasp = (glb_cf32 *)uni_ptr(g_.bias); glb_f32 *gout = (glb_f32 *)uni_ptr(g_.gout);
    glb_cf32 *wscale = (glb_cf32 *)uni_ptr(g_.wscale);
    constexpr bool single = SINGLE;
    const int lane = threadIdx.x & 63, t = lane & 15, gq = lane >> 4;
    const int r0 = min(t, Tb - 1);                                     // rows beyond Tb repeat the last one (results dropped)
    const lds_half *bh = (const lds_half *)(from_hb ? &sh->hbh[0][0] : &sh->xh[0][0]), *bl = (const lds_half *)(from_hb ? &sh->hbl[0][0] : &sh->xl[0][0]);
    const int stride = from_hb ? DQ_HB * 8 : DQ_XB * 8;
    const int p1a = (from_hb ? r0 : r0 + 1) * stride, k1a = r0 & 15;
    const int p0a = ((rstmask >> r0) & 1u) ? (DQ2_ROWS + 1) * stride : r0 * stride;
    const int k0a = (r0 - 1) & 15;
    const int planes = single ? 1 : 2;
    glb_u16 *wa = wbase + (((size_t)ks0 * nct + ct) * planes * 64 + lane) * 8;
    const size_t wstep = (size_t)nct * planes * 64 * 8, tstep = (size_t)planes * 64 * 8;
    lds_f32 *gi = (lds_f32 *)&sh->gi[0][0];
    f32x4 acc0[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) acc0[i] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
    const int n0 = 16 * ct + 4 * gq;
    typedef const __attribute__((address_space(1))) f16x8 glb_f16x8;
    typedef const __attribute__((address_space(3))) f16x8 lds_f16x8;
    f16x8 wh[D][NT], wl[D][NT];
    auto fetch = [&](int d, int ks) {
        const int kq = min(ks, nks - 1);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            wh[d][i] = *(glb_f16x8 *)(wa + kq * wstep + i * tstep);
            if (!single) wl[d][i] = *(glb_f16x8 *)(wa + kq * wstep + i * tstep + 64 * 8);
        }
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, d);
    if (sync_first) __syncthreads();
    f16x8 nha, nla;
    auto rows = [&](int kidx) {
        const int kk = ks0 + min(kidx, nks - 1);
        const bool tap0 = kk < ktap;
        const int cb = 4 * (tap0 ? kk : kk - ktap) + gq;
        const int oa = (tap0 ? p0a : p1a) + ((cb ^ (tap0 ? k0a : k1a)) << 3);
        nha = *(lds_f16x8 *)(bh + oa); nla = *(lds_f16x8 *)(bl + oa);
    };
    rows(0);
    auto step = [&](int d, int kidx, bool refill) {
        const f16x8 xha = nha, xla = nla;
        rows(kidx + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            if (!single) acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[d][i], xha, acc0[i], 0, 0, 0);
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[d][i], xla, acc0[i], 0, 0, 0);
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[d][i], xha, acc0[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (refill) fetch(d, kidx + D);
        __builtin_amdgcn_sched_barrier(0);
    };
    int ks = 0;
#pragma unroll 1
    for (; ks + D <= nks; ks += D) {
#pragma unroll
        for (int d = 0; d < D; d++) step(d, ks + d, true);
    }
    // bias and row scales are requested here, under the last k-steps: held from the top of the function they cost up to 48 registers
    // across the K loop, and came back from scratch one s_waitcnt vmcnt(0) at a time in the epilogue
    f32x4 bias[NT], scl[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        bias[i] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; scl[i] = (f32x4){ 0x1p-18f, 0x1p-18f, 0x1p-18f, 0x1p-18f };
        if (biasp && !init_gi) {        // (N is a multiple of 4 in every layer: one 16-byte load from a clamped address instead of four guarded dwords)
            const int nn = n0 + 16 * i;
            bias[i] = *(const __attribute__((address_space(1))) f32x4 *)(biasp + min(nn, N - 4));
            if (nn >= N) bias[i] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
        }
        if (single) scl[i] = *(const __attribute__((address_space(1))) f32x4 *)(wscale + n0 + 16 * i) * 0x1p-8f;
    }
#pragma unroll
    for (int d = 0; d < D; d++) if (ks + d < nks) step(d, ks + d, false);
    lds_half *xh = (lds_half *)&sh->xh[0][0], *xl = (lds_half *)&sh->xl[0][0];
    const lds_half *hbh = (const lds_half *)&sh->hbh[0][0], *hbl = (const lds_half *)&sh->hbl[0][0];
    typedef __attribute__((address_space(3))) f16x4 lds_f16x4;
    if (t < Tb) {
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int n = n0 + 16 * i, tt = t;
            f32x4 v = acc0[i] * scl[i] + bias[i];
            if (init_gi) v += *(const __attribute__((address_space(3))) f32x4 *)(gi + tt * DQ_GIS + n);
            if (outk == DQ_OUT_GI) { *(__attribute__((address_space(3))) f32x4 *)(gi + tt * DQ_GIS + n) = v; continue; }
            if (outk == DQ_OUT_GLOBAL) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (n + r < N) gout[(size_t)tt * gstride + n + r] = v[r];
                continue;
            }
            if (act == 2) {
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
__device__ __forceinline__ void dq2_gemm_tiles(DecShared2 *sh, const DqGemm g, int ct, int Tb, unsigned rstmask, int sync_first)
{
    if (uni_ptr(g.wscale) != nullptr) dq2_gemm_tiles_<NT, true>(sh, g, ct, Tb, rstmask, sync_first);
    else dq2_gemm_tiles_<NT, false>(sh, g, ct, Tb, rstmask, sync_first);
}

// dense1 on the f32 matrix cores (see dq_dense1): three wavefronts, 32 columns each
__device__ void dq2_dense1(DecShared2 *sh, const float *z, const rd_lin w, int Tb)
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

// two plain FMAs (see the FIR in k_rx_sync2 for why this kernel avoids v_pk_fma_f32)
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return (f32x2){ fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1]) }; }
// GRU recurrence over Tb steps: TWO lanes per hidden unit (192 threads; the fourth wavefront only keeps the barriers)
__device__ void dq2_scan(DecShared2 *sh, const float *Whh, const float *bhh, float *hstate, int Tb, unsigned rstmask)
{
    constexpr int H = 96, KP = H / 2;
    const int tid = rx_tid();
    const bool on = tid < 2 * H;
    const int j = on ? tid >> 1 : 0, p = tid & 1;
    f32x2 wr[KP / 2], wz[KP / 2], wn[KP / 2];
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
    __syncthreads();                                   // (the barrier behind the input projection / fix-up that wrote gi: taken behind this function's global loads)
    if (on && p == 0) sh->hs[0][j] = hj;
    const float *gi = &sh->gi[0][0] + j;
    float g0r = gi[0], g0z = gi[H], g0n = gi[2 * H];
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < Tb; t++) {
        if ((rstmask >> t) & 1u) {                     // uniform over the workgroup
            hj = 0.0f;
            __syncthreads();
            if (on && p == 0) sh->hs[cur][j] = 0.0f;
            __syncthreads();
        }
        const float *gn_ = gi + (size_t)min(t + 1, Tb - 1) * DQ_GIS;
        const float g1r = gn_[0], g1z = gn_[H], g1n = gn_[2 * H];          // next step's inputs: their LDS latency hides under this step
        f32x2 ar = { 0.0f, 0.0f }, az = { 0.0f, 0.0f }, an = { 0.0f, 0.0f }, ar2 = { 0.0f, 0.0f }, az2 = { 0.0f, 0.0f }, an2 = { 0.0f, 0.0f };
        const float *hp = sh->hs[cur] + p * KP;
#pragma unroll
        for (int k = 0; k < KP; k += 8) {
            const f32x4 hv = *(const f32x4 *)(hp + k), hw = *(const f32x4 *)(hp + k + 4);
            const f32x2 h0 = { hv[0], hv[1] }, h1 = { hv[2], hv[3] }, h2 = { hw[0], hw[1] }, h3 = { hw[2], hw[3] };
            ar = fma2(wr[k / 2], h0, ar); az = fma2(wz[k / 2], h0, az); an = fma2(wn[k / 2], h0, an);
            ar2 = fma2(wr[k / 2 + 1], h1, ar2); az2 = fma2(wz[k / 2 + 1], h1, az2); an2 = fma2(wn[k / 2 + 1], h1, an2);
            ar = fma2(wr[k / 2 + 2], h2, ar); az = fma2(wz[k / 2 + 2], h2, az); an = fma2(wn[k / 2 + 2], h2, an);
            ar2 = fma2(wr[k / 2 + 3], h3, ar2); az2 = fma2(wz[k / 2 + 3], h3, az2); an2 = fma2(wn[k / 2 + 3], h3, an2);
        }
        ar += ar2; az += az2; an += an2;
        float sr = ar[0] + ar[1], sz = az[0] + az[1], sn = an[0] + an[1];
        sr += quad_dpp<QUAD_XOR1>(sr); sz += quad_dpp<QUAD_XOR1>(sz); sn += quad_dpp<QUAD_XOR1>(sn);
        const float r = gate_sigmoid((sr + br) + g0r);
        const float z = gate_sigmoid((sz + bz) + g0z);
        const float n = gate_tanh(g0n + (sn + bn) * r);
        hj = (hj - n) * z + n;
        if (on && p == 0) {
            sh->hs[cur ^ 1][j] = hj;
            _Float16 a, b; dq_split(clamp1(hj), a, b);
            sh->hbh[0][dq_hoff(t, j)] = a; sh->hbl[0][dq_hoff(t, j)] = b;
        }
        g0r = g1r; g0z = g1z; g0n = g1n;
        cur ^= 1;
        if (t + 1 < Tb) __syncthreads();               // (the last step's barrier is the consumer's: dq2_gemm_tiles(..., sync_first))
    }
    if (on && p == 0) hstate[j] = hj;
}

// The recurrence with W_hh h on the matrix cores.  Beside another stream's workgroup on the same CU the vector ALU is what the two compete
// for, and dq2_scan spends 432 vector FMAs per step and hidden unit row on a product the matrix pipe does in a few instructions: W_hh is
// int8 in the blob, its integers sit in registers as A-operand fragments of v_mfma_f32_16x16x32_f16 (exact in binary16, row scales
// applied afterwards); h_{t-1} is the B operand, read from LDS as two binary16 planes (2^8 h = hi + lo): EVEN columns of B carry the
// high plane, ODD columns the low plane, so one instruction per (gate tile, k-step) yields both partial products and a DPP add of
// neighbouring lanes (quad_perm [1,0,3,2]) joins them -- 9 instructions per block of 16 hidden units instead of 18.  A wavefront
// owns "unit blocks": the three tiles r / z / n of its units, so the gates are evaluated in registers by the lanes that hold them
// (lane column c < 4 finalises row c of its lane group).  Four wavefronts: blocks {0,1} {2,3} {4} {5}; the two-block wavefronts issue
// both blocks' matrix instructions first and then evaluate both blocks' gates in one straight-line region (two independent chains).
// (In k_rx_sync, alone on its CU, the matrix form was no faster -- the step there is latency, not ALU -- and was not kept.)
template <int NB>
__device__ __forceinline__ void dq2_scan_mfma_body(DecShared2 *sh, const unsigned short *whq, const float *whs, const float *bhh, float *hstate, int Tb, unsigned rstmask, int ub0)
{
    constexpr int H = 96;
    const int tid = rx_tid(), lane = tid & 63, c = lane & 15, g = lane >> 4, cs = c & 3, par = c & 1;
    // every column of a tile's C holds the same sums once neighbouring lanes are added (the planes sit in even / odd columns), so in a two-block wavefront the lanes of
    // columns 8..15 take block 1 and those of columns 0..7 block 0: ONE pass over the gates (two sigmoids and a tanh: six transcendental instructions and their
    // latencies, the longest dependent chain of the step) serves both blocks instead of one pass per block
    const int blk = NB == 2 ? (c >> 3) & 1 : 0;
    typedef const __attribute__((address_space(1))) f16x8 glb_f16x8_t;
    f16x8 A[NB][3][3];
    float sc[3], bb[3];
    const bool finl = (c & 4) == 0 && (NB == 2 || c < 4);      // the lanes that publish: columns 0..3 (block 0) and, with two blocks, 8..11 (block 1)
    const int ju = 16 * (ub0 + blk) + 4 * g + cs;
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
        for (int gate = 0; gate < 3; gate++)
#pragma unroll
            for (int ks = 0; ks < 3; ks++) A[k][gate][ks] = *(glb_f16x8_t *)(whq + (((size_t)ks * 18 + gate * 6 + ub0 + k) * 64 + lane) * 8);
#pragma unroll
    for (int gate = 0; gate < 3; gate++) { sc[gate] = whs[gate * H + ju] * 0x1p-8f; bb[gate] = bhh[gate * H + ju]; }
    float hj = hstate[ju];
    __syncthreads();                                   // (the barrier behind the input projection / fix-up that wrote gi: taken behind the 9 / 18 weight fragments' round trip to L2)
    _Float16 (*hp)[2][H] = (_Float16 (*)[2][H])&sh->hs[0][0];           // [buffer][plane][k]: 2^8 h_{t-1} = hi + lo
    if (finl) { _Float16 a, b; dq_split(hj, a, b); hp[0][0][ju] = a; hp[0][1][ju] = b; }
    const float *gi = &sh->gi[0][0];
    float g0[3];
#pragma unroll
    for (int gate = 0; gate < 3; gate++) g0[gate] = gi[gate * H + ju];
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < Tb; t++) {
        if ((rstmask >> t) & 1u) {                     // uniform over the workgroup
            __syncthreads();
            hj = 0.0f; if (finl) { hp[cur][0][ju] = (_Float16)0.0f; hp[cur][1][ju] = (_Float16)0.0f; }
            __syncthreads();
        }
        const float *gn_ = gi + (size_t)min(t + 1, Tb - 1) * DQ_GIS;
        float g1[3];
#pragma unroll
        for (int gate = 0; gate < 3; gate++) g1[gate] = gn_[gate * H + ju];
        f16x8 bq[3];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) bq[ks] = *(const f16x8 *)&hp[cur][par][32 * ks + 8 * g];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NB][3];
#pragma unroll
        for (int k = 0; k < NB; k++)
#pragma unroll
            for (int gate = 0; gate < 3; gate++) acc[k][gate] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
#pragma unroll
            for (int k = 0; k < NB; k++)
#pragma unroll
                for (int gate = 0; gate < 3; gate++) acc[k][gate] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k][gate][ks], bq[ks], acc[k][gate], 0, 0, 0);
        // C layout: this lane holds rows 4 g + 0..3 of each tile for column c: high-plane product in even columns, low-plane product in odd ones
        float s3[3];
#pragma unroll
        for (int gate = 0; gate < 3; gate++) {
            float sr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = NB == 2 ? (blk ? acc[NB - 1][gate][r] : acc[0][gate][r]) : acc[0][gate][r];      // this lane's block
                sr[r] = v + quad_dpp<QUAD_XOR1>(v); asm volatile("" : "+v"(sr[r]));                               // (computed before the select: no branches)
            }
            const float s01 = (cs & 1) ? sr[1] : sr[0], s23 = (cs & 1) ? sr[3] : sr[2];
            s3[gate] = (cs & 2) ? s23 : s01;
        }
        {
            const float r = gate_sigmoid((s3[0] * sc[0] + bb[0]) + g0[0]);
            const float z = gate_sigmoid((s3[1] * sc[1] + bb[1]) + g0[1]);
            const float n = gate_tanh(g0[2] + (s3[2] * sc[2] + bb[2]) * r);
            hj = (hj - n) * z + n;
        }
        if (finl) {
            _Float16 a, b; dq_split(hj, a, b);
            hp[cur ^ 1][0][ju] = a; hp[cur ^ 1][1][ju] = b;
            dq_split(clamp1(hj), a, b);
            sh->hbh[0][dq_hoff(t, ju)] = a; sh->hbl[0][dq_hoff(t, ju)] = b;
        }
#pragma unroll
        for (int gate = 0; gate < 3; gate++) g0[gate] = g1[gate];
        cur ^= 1;
        if (t + 1 < Tb) __syncthreads();               // (the last step's barrier is the consumer's: dq2_gemm_tiles(..., sync_first))
    }
    if (finl) hstate[ju] = hj;
}
__device__ void dq2_scan_mfma(DecShared2 *sh, const unsigned short *whq, const float *whs, const float *bhh, float *hstate, int Tb, unsigned rstmask)
{
    const int wave = rx2_wave();
    if (wave < 2) dq2_scan_mfma_body<2>(sh, whq, whs, bhh, hstate, Tb, rstmask, 2 * wave);
    else dq2_scan_mfma_body<1>(sh, whq, whs, bhh, hstate, Tb, rstmask, 2 + wave);
}

// all decoder layers for rows [0, Tb) (Tb <= 12) of stream b on four wavefronts
__device__ void dq2_layers(DecShared2 *sh, const rd_decs_args &a, int b, const float *z, float *out, int Tb, unsigned rstmask, int census_noscan)
{
    const int tid = rx_tid(), wave = tid >> 6;
    PH2_T0();
    {
        const unsigned *hist = (const unsigned *)(a.x + (size_t)b * a.x_sb - RD_DEC_W);
        constexpr int NH_ = (RD_DEC_W + NT2 - 1) / NT2;
        unsigned hu[NH_];                                       // (all requested before the first LDS store: see rx2_load_rxbuf)
#pragma unroll
        for (int q = 0; q < NH_; q++) hu[q] = hist[min(tid + q * NT2, RD_DEC_W - 1)];
#pragma unroll
        for (int q = 0; q < NH_; q++) {
            const int c = tid + q * NT2; const unsigned u = hu[q];
            if (c < RD_DEC_W) { sh->xh[0][dq_xoff(-1, c)] = __builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); sh->xl[0][dq_xoff(-1, c)] = __builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
        }
        for (int c = tid; c < DQ_XB * 8; c += NT2) { sh->xh[DQ2_ROWS + 1][c] = (_Float16)0.0f; sh->xl[DQ2_ROWS + 1][c] = (_Float16)0.0f; }
    }
    dq2_dense1(sh, z, a.dense1, Tb);
    DqGemm g;
    // Barriers: every product phase takes the barrier that separates it from the phase before it INSIDE its first dq2_gemm_tiles call (sync_first), behind that call's
    // first weight requests; the recurrences do the same behind their weight fragments.  A wavefront that has no product in a phase takes the barrier bare.
    // 18 column tiles of an input projection over four wavefronts: 5 + 5 + 4 + 4
    const int ct18 = wave < 2 ? 5 * wave : 10 + 4 * (wave - 2);
    g = (DqGemm){ a.gin[0].wa16, 18, a.gin[0].bias, 288, a.gin[0].wscale, 0, 0, 0, 3, 0, DQ_OUT_GI, 0, 0, nullptr, 0 };
    if (wave < 2) dq2_gemm_tiles<5>(sh, g, ct18, Tb, rstmask, 1); else dq2_gemm_tiles<4>(sh, g, ct18, Tb, rstmask, 1);
    PH2(20);
#pragma unroll 1
    for (int l = 0; l < 5; l++) {
        const int in = 96 + 128 * l, cin = in + 96;
        if (census_noscan) __syncthreads();
        else if (a.whq[l]) dq2_scan_mfma(sh, a.whq[l], a.whs[l], a.bhh[l], a.h[l] + (size_t)b * 96, Tb, rstmask);
        else dq2_scan(sh, a.whh[l], a.bhh[l], a.h[l] + (size_t)b * 96, Tb, rstmask);
        PH2(21);
        // GLU gates: 6 column tiles, K = 96: 2 + 2 + 1 + 1
        g = (DqGemm){ a.glu[l].wa16, 6, nullptr, 96, a.glu[l].wscale, 1, 0, 0, 3, 0, DQ_OUT_X, in, 2, nullptr, 0 };
        if (wave < 2) dq2_gemm_tiles<2>(sh, g, 2 * wave, Tb, rstmask, 1); else dq2_gemm_tiles<1>(sh, g, 2 + wave, Tb, rstmask, 1);
        PH2(22);
        // conv (2 tiles, K = 2 cin) beside the columns of the next product that are already final (K = cin): in units of cin k-steps the
        // conv tiles weigh 2 each, a projection tile 1: wavefronts 0 / 1 = one conv tile + 3 projection tiles, 2 / 3 = 6 projection tiles;
        // behind the last conv the output layer (6 tiles): one conv tile each on 0 / 1, three output tiles each on 2 / 3
        const DqGemm gc = (DqGemm){ a.conv[l].wa16, 2, a.conv[l].bias, 32, a.conv[l].wscale, 0, cin / 32, 0, 2 * cin / 32, 0, DQ_OUT_X, cin, 1, nullptr, 0 };
        const bool last = l == 4;
        const rd_lin &nx = last ? a.output : a.gin[l + 1];
        const int nct = last ? 6 : 18;
        const DqGemm gm = (DqGemm){ nx.wa16, nct, nx.bias, last ? a.out_w : 288, nx.wscale, 0, 0, 0, cin / 32, 0, DQ_OUT_GI, 0, 0, nullptr, 0 };
        if (wave < 2) {
            dq2_gemm_tiles<1>(sh, gc, wave, Tb, rstmask, 1);
            if (!last) dq2_gemm_tiles<3>(sh, gm, 3 * wave, Tb, rstmask, 0);
        } else if (last) dq2_gemm_tiles<3>(sh, gm, 3 * (wave - 2), Tb, rstmask, 1);
        else {          // six projection tiles as two calls of three: with six tiles' fragments (4 k-steps x 6 x 4 registers) in flight the K loop spilled -- 20 scratch instructions per k-step
            dq2_gemm_tiles<3>(sh, gm, 6 + 6 * (wave - 2), Tb, rstmask, 1);
            dq2_gemm_tiles<3>(sh, gm, 9 + 6 * (wave - 2), Tb, rstmask, 0);
        }
        PH2(23);
        // fix-up: the conv's 32 new columns (one k-step) added onto the staged sums
        const DqGemm gf = (DqGemm){ nx.wa16, nct, nullptr, last ? a.out_w : 288, nx.wscale, 0, 0, cin / 32, 1, 1, last ? DQ_OUT_GLOBAL : DQ_OUT_GI, 0, 0, out, a.out_w };
        if (last) { if (wave >= 2) dq2_gemm_tiles<3>(sh, gf, 3 * (wave - 2), Tb, rstmask, 1); else __syncthreads(); }
        else if (wave < 2) dq2_gemm_tiles<5>(sh, gf, ct18, Tb, rstmask, 1);
        else dq2_gemm_tiles<4>(sh, gf, ct18, Tb, rstmask, 1);
        PH2(24);
    }
    __syncthreads();                                   // (behind the last fix-up: the history row below reads what the last conv wrote)
    {
        unsigned *hist = (unsigned *)(a.x + (size_t)b * a.x_sb - RD_DEC_W);
        for (int c = tid; c < RD_DEC_W; c += NT2)
            hist[c] = (unsigned)__builtin_bit_cast(unsigned short, sh->xh[0][dq_xoff(Tb - 1, c)]) | ((unsigned)__builtin_bit_cast(unsigned short, sh->xl[0][dq_xoff(Tb - 1, c)]) << 16);
    }
    __syncthreads();
}

// =====================================================================================================
// LDS of k_rx_sync2: at most 80 KB, so that two workgroups share a CU
// =====================================================================================================
struct RxShared2 {
    RxScalars S;
    double2 rq[4], rzc, rph[24], rrot[24]; double ral[24];   // refine(), in-sync grid: e^{-jw_c 40 q}, e^{-jw_c}, e^{-j(w_k - w_c) 79.5}, e^{-jw_k Nmf}, (w_k - w_c) 80
    int rows48[48];
    double redd[(NW2 + 1) * 10];
    float redf[16]; int redi[16]; int redj[16];
    double corrp[NW2][8];
    union {
      struct {
        float2 rxb[RD_RXBUF];                                   // rx_buf (radae_rxe.py:141): parked in the stream's HBM record while the decoder stage runs
        float rowsum1[RD_NMF], rowsum2[RD_NMF];                 // likewise
        union {
          struct {                                              // BPF / synchronised state
            __attribute__((aligned(16))) float2 xm[1408];       // BPF [mem | mixed-down new]; refine(): rx window as doubles; rx1[1152] for the demodulator
            double2 pd[RD_M], pendd[RD_M];                      // pilot / end-of-over replicas as doubles (reloaded with the state: S.lds_sync)
            float2 sym[6][RD_NC]; float2 rp[2][RD_NC];
            float eqP[RD_NC]; float2 eqPmat[RD_NC][2][3], eqrot[RD_NC]; float eq_pg, eq_snrc1, eq_snrc2, eq_pad;
            float2 cisA[2][18], cisB[2][64];                    // the frequency-corrected window's phasor e^{j(theta - w (n + 1))} = cisA[n >> 6] cisB[n & 63], one copy per wavefront that cuts the window
            union {
              struct {
                u32x2 rxhl[RD_RXBUF];                           // check_pilots: rx_buf in two binary16 planes, a sample's (high, low) words side by side: one 8-byte read brings both
                double rmom[2][2][64][4];                       // refine(), in-sync grid: partial moment tiles [half of the samples][frame]
                double mtot[2][16][16];                         // the moments [frame][2 m + (re | im)][t]
              };
              struct {                                          // refine() on sync entry (+-10 Hz): direct sums
                double2 rtw[80], rrot80[80], rt80[80];
                float2 dtr[2 * 80 * 16];
              };
            };
          };
          struct {                                              // search / candidate state: two-stage pilot correlator on the matrix cores (rx2_detect_q)
            __attribute__((aligned(16))) _Float16 sA[2][2 * 2 * 2 * 64 * 8];   // stage 1's A operands (moment table) of TWO k-steps, double-buffered (2 x 8 KB)
            __attribute__((aligned(16))) _Float16 sA2[5 * 2 * 64 * 8];         // stage 2's A operands (moments -> 40 frequencies), whole: 10 KB
            unsigned srxh[RD_RXBUF], srxl[RD_RXBUF];            // rx_buf in two binary16 planes
          };
        };
      };
      __attribute__((aligned(16))) unsigned char dec_raw[sizeof(DecShared2)];
    };
};
static_assert(sizeof(RxShared2) <= 80 * 1024, "two k_rx_sync2 workgroups must fit the 160 KiB LDS of a CU");

// ---- workgroup reductions on four wavefronts (same contracts as block_argmax / block_sum_multi) ------------------------------------
__device__ void block_argmax2(RxShared2 *sh, float &v, int &k0, int &k1)
{
    const int tid = rx_tid(), lane = tid & 63, wave = tid >> 6;
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
    for (int w = 1; w < NW2; w++) {
        const float ov = sh->redf[1 + w]; const int o0 = sh->redi[1 + w], o1 = sh->redj[1 + w];
        if (ov > bv || (ov == bv && (o0 < b0 || (o0 == b0 && o1 < b1)))) { bv = ov; b0 = o0; b1 = o1; }
    }
    v = bv; k0 = b0; k1 = b1;
}
template <int NV>
__device__ void block_sum_multi2(RxShared2 *sh, double (&v)[NV])
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
        for (int w = 0; w < NW2; w++) t += sh->redd[w * NV + k];
        v[k] = t;
    }
}
__device__ float sigma_r_from_rowsums2(RxShared2 *sh)
{
    double v[2] = { 0.0, 0.0 };
    for (int t = rx_tid(); t < RD_NMF; t += NT2) { v[0] += (double)sh->rowsum1[t]; v[1] += (double)sh->rowsum2[t]; }
    block_sum_multi2<2>(sh, v);
    return sigma_r_from_sums(v[0], v[1]);
}


// ---- |Dt| surfaces on the matrix cores ----------------------------------------------------------------------------------------------
// acquisition.detect_pilots (dsp.py:178-231): Dt[t, f] = sum_m conj(rx[t + m]) p_w[m, f] for all 960 timings x 40 frequencies of a frame: a real GEMM
// [(row, re | im)] x [320 = (m, re | im)] times the Toeplitz matrix rx[t + m] on v_mfma_f32_16x16x32_f16, both operands in two binary16 planes
// (hi hi + hi lo + lo hi: 22 bits).
//   * a wavefront owns 15 timing tiles (240 timings) and walks them in groups of RT = 5;
//   * the A operands (table) are staged through LDS by the whole workgroup, double-buffered, one barrier per stage;
//   * the B operand of (timing tile T, k-step s) is the fragment of (T + s, 0): the window slides by one tile per k-step, so a group reads RT + 9 fragments
//     from the planes instead of 10 RT.
// (Rounds 3-4 multiplied by the 80 rows of p_w itself: tools/experiments/rx2_search_one_stage.inc.)
// ---- the pilot correlator in two stages (round 5; the tables and the algebra: rade_host.c, rd_corrq16_table_fill) ----------------------------------
// Dt[t][f] = sum_r alpha[r][f] Mom_r[t]: stage 1 is the product above with the 32 rows (r, re | im) of the moment table instead of the 80 rows (f, re | im)
// of p_w -- two row tiles instead of five, 60 matrix instructions per tile of 16 timings instead of 150 --, stage 2 expands the 16 complex moments of a timing
// tile to the 40 frequencies with ONE k-step (K = 32) per frequency tile.  Stage 1's accumulators ARE stage 2's B operand: the C layout of two 16-row tiles
// gives lane group g rows 4 g .. 4 g + 3 of either tile, and the host orders stage 2's K axis exactly so (rd_corra16_table_fill) -- the moments are scaled,
// split into three binary16 planes (33 bits: stage 2 adds nothing to stage 1's rounding) and fed back without leaving the lane.
struct MomPlanes { f16x8 h, m, l; };
__device__ __forceinline__ MomPlanes mom_split(const f32x4 a0, const f32x4 a1)
{
    MomPlanes p;
#pragma unroll
    for (int j = 0; j < 8; j++) {      // |moment| <= 2^8 sqrt(2) x the table row's L1 norm (3545 x 2^... : rade_host.c) -> 2^-7 of it is below 7100, far inside binary16
        const float v = 0x1p-7f * (j < 4 ? a0[j & 3] : a1[j & 3]);
        const _Float16 h = (_Float16)v; const float r1 = v - (float)h;
        const _Float16 m = (_Float16)r1; const float r2 = r1 - (float)m;
        p.h[j] = h; p.m[j] = m; p.l[j] = (_Float16)r2;
    }
    return p;
}
// one frequency tile (8 frequencies x (re, im)) of Dt for the 16 timings whose moments are in p: smallest partial products first
__device__ __forceinline__ f32x4 mom_expand(const f16x8 ah, const f16x8 al, const MomPlanes &p)
{
    f32x4 c = { 0.0f, 0.0f, 0.0f, 0.0f };
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, p.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, p.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, p.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, p.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, p.h, c, 0, 0, 0);
    return c;
}
// acquisition.detect_pilots' surfaces by the two-stage correlator: same contract as rx2_detect_mfma (outputs: |Dt2| -- and |Dt1| when not cached -- to the stream's
// cache in HBM, the row sums to rowsum1 / rowsum2, the lane's best (Dt1 + Dt2, t, f)).  rx_unsc undoes the scales of the rx planes and the stage-1 table; the
// 2^-7 of the moments and the 2^10 of the stage-2 table are undone here.
__device__ __forceinline__ void rx2_detect_q(RxShared2 *sh, const unsigned short *corrq16_, const unsigned short *corra16_, float *cache_, int cached, int oldb, int newb,
                                             float rx_unsc_, float &best, int &bt, int &bfi)
{
    constexpr int RT = 5, NTF = 5, TPW = 15;
    static_assert(TPW * NW2 * 16 == RD_NMF && TPW % RT == 0, "timing tiles per wavefront");
    const int tid = rx_tid(), wave = rx2_wave(), lane = tid & 63, i = lane & 15, g = lane >> 4;
    const float rx_unsc = rx_unsc_ * 0x1p-3f;
    // stage 1's table [tile][k-step][plane][lane] (16 B per lane): a stage buffer holds two k-steps of both tiles, chunk u = 256 tile + 128 (k-step & 1) + 64 plane + lane,
    // i.e. thread tid brings chunks tid and 256 + tid: one lane offset under two uniform bases
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(corrq16_), 0, 2 * 10 * 2048, 0x00020000);
    const int vo = tid * 16;
    u32x4 stg[2];
    auto stage_load = [&](int sb) {
        stg[0] = __builtin_amdgcn_raw_buffer_load_b128(qrs, vo, sb * 4096, 0); stg[1] = __builtin_amdgcn_raw_buffer_load_b128(qrs, vo, sb * 4096 + 10 * 2048, 0);
    };
    auto stage_store = [&](int buf) { _Float16 *d = &sh->sA[buf][tid * 8]; *(u32x4 *)d = stg[0]; *(u32x4 *)(d + 256 * 8) = stg[1]; };
    PH2_T0();
    {   // stage 2's table, whole (640 x 16 B)
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(corra16_), 0, 5 * 2048, 0x00020000);
        const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, 0, 0), t1 = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, 4096, 0);
        u32x4 t2 = { 0u, 0u, 0u, 0u };
        if (wave < 2) t2 = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, 8192, 0);
        stage_load(0);
        _Float16 *d = &sh->sA2[tid * 8];
        *(u32x4 *)d = t0; *(u32x4 *)(d + 256 * 8) = t1;
        if (wave < 2) *(u32x4 *)(d + 512 * 8) = t2;
        stage_store(0);
    }
    float lbest = best; int lkey = 0x7fffffff;
    int pb = 0;                                                 // the stage buffer being read: flips every stage (five stages per group: the parity runs on across groups)
    __syncthreads();
    PH2(15);
#pragma unroll 1
    for (int pass = cached ? 1 : 0; pass < 2; pass++) {
        const unsigned *ph = sh->srxh + pass * RD_NMF + i + 4 * g, *pl = sh->srxl + pass * RD_NMF + i + 4 * g;
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(cache_ + (size_t)(pass ? newb : oldb) * RD_NFC * RD_NMF), 0, RD_NFC * RD_NMF * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(cache_ + (size_t)oldb * RD_NFC * RD_NMF), 0, RD_NFC * RD_NMF * 4, 0x00020000);
        float *rowsum = pass ? sh->rowsum2 : sh->rowsum1;
#pragma unroll 1
        for (int grp = 0; grp < TPW / RT; grp++) {
            const int T0 = wave * TPW + grp * RT;
            f32x4 acc1[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) { acc1[rt][0] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; acc1[rt][1] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; }
            u32x4 wh[RT + 1], wl[RT + 1];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int j = 0; j < 4; j++) { wh[rt][j] = ph[16 * (T0 + rt) + j]; wl[rt][j] = pl[16 * (T0 + rt) + j]; }
#pragma unroll
            for (int sb = 0; sb < 5; sb++) {
                stage_load(sb == 4 ? 0 : sb + 1);
#pragma unroll
                for (int ds = 0; ds < 2; ds++) {
                    const int sidx = 2 * sb + ds;
                    // fragment F of the sliding window lives in slot (F - T0) mod (RT + 1): tile rt reads slot (rt + sidx) mod (RT + 1), the fragment the NEXT k-step adds goes
                    // into the slot the first tile just left (all indices are compile-time: the loops are unrolled).  Rounds 3-4 shifted the window through the registers
                    // instead: 40 moves per k-step behind the matrix instructions
                    if (sidx < 9) {
#pragma unroll
                        for (int j = 0; j < 4; j++) { wh[(RT + sidx) % (RT + 1)][j] = ph[16 * (T0 + RT + sidx) + j]; wl[(RT + sidx) % (RT + 1)][j] = pl[16 * (T0 + RT + sidx) + j]; }
                    }
                    const _Float16 *Ab = &sh->sA[pb][lane * 8];
                    const f16x8 a0h = *(const f16x8 *)(Ab + ((0 * 2 + ds) * 2 + 0) * 512), a0l = *(const f16x8 *)(Ab + ((0 * 2 + ds) * 2 + 1) * 512);
                    const f16x8 a1h = *(const f16x8 *)(Ab + ((1 * 2 + ds) * 2 + 0) * 512), a1l = *(const f16x8 *)(Ab + ((1 * 2 + ds) * 2 + 1) * 512);
                    __builtin_amdgcn_sched_barrier(0);
#define WSL(rt) (((rt) + sidx) % (RT + 1))
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc1[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l, __builtin_bit_cast(f16x8, wh[WSL(rt)]), acc1[rt][0], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc1[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, __builtin_bit_cast(f16x8, wh[WSL(rt)]), acc1[rt][1], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc1[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, __builtin_bit_cast(f16x8, wl[WSL(rt)]), acc1[rt][0], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc1[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, __builtin_bit_cast(f16x8, wl[WSL(rt)]), acc1[rt][1], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc1[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, __builtin_bit_cast(f16x8, wh[WSL(rt)]), acc1[rt][0], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc1[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, __builtin_bit_cast(f16x8, wh[WSL(rt)]), acc1[rt][1], 0, 0, 0);
#undef WSL
                    __builtin_amdgcn_sched_barrier(0);
                }
                stage_store(pb ^ 1);
                pb ^= 1;
                __syncthreads();
            }
            PH2(16);
            // ---- stage 2 and the epilogue, one timing tile at a time.  The surfaces in the stream's HBM cache are only ever read back by the lane that wrote them
            // (|Dt2| of this call is |Dt1| of the next), so their layout is the lane's: per (wavefront, group, timing tile) 64 lanes x 10 values (frequency tile,
            // then the two frequencies) as two 16-byte vectors [lane] + one 8-byte vector [lane]: fully coalesced.
            f16x8 A2h[NTF], A2l[NTF];
#pragma unroll
            for (int q = 0; q < NTF; q++) { A2h[q] = *(const f16x8 *)&sh->sA2[((q * 2) * 64 + lane) * 8]; A2l[q] = *(const f16x8 *)&sh->sA2[((q * 2 + 1) * 64 + lane) * 8]; }
            const int gb = (wave * (TPW / RT) + grp) * RT * 2560;         // byte offset of the group's block; 2560 B per timing tile
            float pv[2][2 * NTF];
            auto pv_load = [&](int slot, int rt) {
                const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(prs, lane * 16, gb + rt * 2560, 0), v1 = __builtin_amdgcn_raw_buffer_load_b128(prs, lane * 16, gb + rt * 2560 + 1024, 0);
                const u32x2 v2 = __builtin_amdgcn_raw_buffer_load_b64(prs, lane * 8, gb + rt * 2560 + 2048, 0);
#pragma unroll
                for (int k = 0; k < 4; k++) { pv[slot][k] = __uint_as_float(v0[k]); pv[slot][4 + k] = __uint_as_float(v1[k]); }
                pv[slot][8] = __uint_as_float(v2[0]); pv[slot][9] = __uint_as_float(v2[1]);
            };
            if (pass) pv_load(0, 0);
            const int tb = 16 * T0 + i;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if (pass && rt + 1 < RT) pv_load((rt + 1) & 1, rt + 1);
                const MomPlanes mp = mom_split(acc1[rt][0], acc1[rt][1]);
                float dd[2 * NTF], rs = 0.0f;
#pragma unroll
                for (int q = 0; q < NTF; q++) {
                    // C layout: column = lane & 15 (timing), rows 4 g + r = (re, im) of f = 8 q + 2 g and f + 1
                    const f32x4 c = mom_expand(A2h[q], A2l[q], mp);
                    const float d0 = rx_unsc * __builtin_amdgcn_sqrtf(fmaf(c[0], c[0], c[1] * c[1])), d1 = rx_unsc * __builtin_amdgcn_sqrtf(fmaf(c[2], c[2], c[3] * c[3]));
                    rs += d0; rs += d1;
                    dd[2 * q] = d0; dd[2 * q + 1] = d1;
                }
                {   // the other three lane groups hold the row's other frequencies: v_permlane16/32_swap (vector ALU, no LDS round trip)
                    const auto p16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                    rs = __uint_as_float(p16[0]) + __uint_as_float(p16[1]);
                    const auto p32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                    rs = __uint_as_float(p32[0]) + __uint_as_float(p32[1]);
                }
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){ __float_as_uint(dd[0]), __float_as_uint(dd[1]), __float_as_uint(dd[2]), __float_as_uint(dd[3]) }, drs, lane * 16, gb + rt * 2560, 0);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){ __float_as_uint(dd[4]), __float_as_uint(dd[5]), __float_as_uint(dd[6]), __float_as_uint(dd[7]) }, drs, lane * 16, gb + rt * 2560 + 1024, 0);
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){ __float_as_uint(dd[8]), __float_as_uint(dd[9]) }, drs, lane * 8, gb + rt * 2560 + 2048, 0);
                // every lane keeps its own best (t ascending, then f ascending, strict >: the earliest wins); block_argmax2 orders the lanes the same way.  Branch-free,
                // (t, f) packed in one register
                const int t = tb + 16 * rt;
                if (pass) {
#pragma unroll
                    for (int q = 0; q < NTF; q++) {
                        const int k0 = (t << 6) | (8 * q + 2 * g);
                        const float s0 = pv[rt & 1][2 * q] + dd[2 * q], s1 = pv[rt & 1][2 * q + 1] + dd[2 * q + 1];
                        const bool c0 = s0 > lbest; lbest = c0 ? s0 : lbest; lkey = c0 ? k0 : lkey;
                        const bool c1 = s1 > lbest; lbest = c1 ? s1 : lbest; lkey = c1 ? k0 + 1 : lkey;
                    }
                }
                if (g == 0) rowsum[t] = rs;
            }
            PH2(17);
        }
        __syncthreads();
        PH2(18);
    }
    best = lbest; bt = lkey >> 6; bfi = lkey & 63;
}

// ---- refine() (dsp.py:233-270): see refine_tile / refine_moments above for the arithmetic ------------------------------------------
__device__ __forceinline__ f64x4 refine2_tile(const RxShared2 *sh, int mt, int frame, int s0, int ns, int nf, int nt, int lane)
{
    const int i = lane & 15, kk = lane >> 4, c = kk & 1, n0 = kk >> 1;
    const int row = 16 * mt + i, fi = row >> 1, cp = row & 1;
    const bool rv = fi < nf;
    const double2 z1 = sh->rtw[rv ? fi : 0];
    double2 cur = s0 ? sh->rt80[rv ? fi : 0] : make_double2(1.0, 0.0);
    if (n0) cur = make_double2(cur.x * z1.x - cur.y * z1.y, cur.x * z1.y + cur.y * z1.x);
    const double c2r = z1.x * z1.x - z1.y * z1.y, c2i = 2.0 * z1.x * z1.y;
    const double2 prv = make_double2(cur.x * c2r + cur.y * c2i, cur.y * c2r - cur.x * c2i);
    double xc = cp == c ? cur.x : (cp == 0 ? -cur.y : cur.y), xp = cp == c ? prv.x : (cp == 0 ? -prv.y : prv.y);
    if (!rv) { xc = 0.0; xp = 0.0; }
    const double k2 = 2.0 * c2r;
    const double *xw = (const double *)&sh->xm[0] + 4 * (frame * 176 + (i < nt ? i : 0) + 2 * s0 + n0) + 2 * c;
    const double2 *pp = &sh->pd[2 * s0 + n0];
    f64x4 acc0 = { 0.0, 0.0, 0.0, 0.0 }, acc1 = acc0;
    double2 pn[4]; double a1[4], a2[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { pn[u] = pp[2 * u]; a1[u] = xw[8 * u]; a2[u] = xw[8 * u + 1]; }
#pragma unroll 1
    for (int s = 0; s < ns; s += 4) {
        double b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = pn[u].x * a1[u];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = fma(pn[u].y, a2[u], b[u]);
        __builtin_amdgcn_sched_barrier(0);
        const int sn = s + 4 < ns ? s + 4 : s;
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
__device__ __forceinline__ void refine2_tables(RxShared2 *sh, int k, double fstart, double fstop, double fstep)
{
    const int nf = (int)ceil((fstop - fstart) / fstep);
    const double delta = (fstart + fstep) - fstart;
    if (k < 0 || k >= 3 * nf) return;
    const int which = k / nf, fi = k - which * nf;
    const double w = 2.0 * PI_D * dgrid_nc(fstart, fi, delta) / 8000.0;
    const double arg = which == 0 ? -w : (which == 1 ? -w * RD_NMF : -w * 80.0);
    double sn, cs; sincos(arg, &sn, &cs);
    double2 *dstp = which == 0 ? sh->rtw : (which == 1 ? sh->rrot80 : sh->rt80);
    dstp[fi] = make_double2(cs, sn);
}
__device__ __forceinline__ void refine2_tables_sync(RxShared2 *sh, int k, double fstart, double fstop, double fstep)
{
    const int nf = (int)ceil((fstop - fstart) / fstep);
    const double delta = (fstart + fstep) - fstart;
    const double wc = 0.5 * (2.0 * PI_D * fstart / 8000.0 + 2.0 * PI_D * dgrid_nc(fstart, nf - 1, delta) / 8000.0);
    if (k < 0 || k > 52) return;
    const int kf = k < 24 ? k : k - 24;
    if (k < 48 && kf >= nf) return;
    const double w = 2.0 * PI_D * dgrid_nc(fstart, kf, delta) / 8000.0, dw = w - wc;
    const double arg = k < 24 ? -w * RD_NMF : (k < 48 ? -dw * 79.5 : (k < 52 ? -wc * 40.0 * (k - 48) : -wc));
    double sn, cs; sincos(arg, &sn, &cs);
    const double2 v = make_double2(cs, sn);
    if (k < 24) sh->rrot[k] = v;
    else if (k < 48) { sh->rph[kf] = v; sh->ral[kf] = dw * 80.0; }
    else if (k < 52) sh->rq[k - 48] = v;
    else sh->rzc = v;
}
// moments of one quarter q of the samples (20 matrix instructions), added onto acc0 / acc1; the powers ((n - 79.5) / 80)^m come from L2
__device__ __forceinline__ void refine2_moments(const RxShared2 *sh, const double *vmg, int frame, int q, int nt, int lane, f64x4 &acc0, f64x4 &acc1)
{
    const int i = lane & 15, kk = lane >> 4, c = kk & 1, n0 = kk >> 1;
    const int m = i >> 1, cp = i & 1, s0 = 20 * q;
    const double2 z1 = sh->rzc;
    double2 cur = sh->rq[q];
    if (n0) cur = make_double2(cur.x * z1.x - cur.y * z1.y, cur.x * z1.y + cur.y * z1.x);
    const double c2r = z1.x * z1.x - z1.y * z1.y, c2i = 2.0 * z1.x * z1.y;
    const double2 prv = make_double2(cur.x * c2r + cur.y * c2i, cur.y * c2r - cur.x * c2i);
    double xc = cp == c ? cur.x : (cp == 0 ? -cur.y : cur.y), xp = cp == c ? prv.x : (cp == 0 ? -prv.y : prv.y);
    const double k2 = 2.0 * c2r;
    const double *xw = (const double *)&sh->xm[0] + 4 * (frame * 176 + (i < nt ? i : 0) + 2 * s0 + n0) + 2 * c;
    const double2 *pp = &sh->pd[2 * s0 + n0];
    // (an explicit global pointer: rx2_refine is a real function, its `vmg` a generic pointer, and the 20 loads below were flat loads, which count on the
    // LDS wait counter too; as global loads they do not -- measured: no change in the cycles per call, the batch is issued far enough ahead either way)
    const __attribute__((address_space(1))) double *vp = (const __attribute__((address_space(1))) double *)vmg + m * RD_M + 2 * s0 + n0;
    double vall[20];                                            // this lane's 20 powers of the quarter: one batch of loads ahead of the loop
#pragma unroll
    for (int u = 0; u < 20; u++) vall[u] = vp[2 * u];
    double2 pn[4]; double a1[4], a2[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { pn[u] = pp[2 * u]; a1[u] = xw[8 * u]; a2[u] = xw[8 * u + 1]; }
#pragma unroll
    for (int s = 0; s < 20; s += 4) {
        double b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = pn[u].x * a1[u];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = fma(pn[u].y, a2[u], b[u]);
        __builtin_amdgcn_sched_barrier(0);
        const int sn = s + 4 < 20 ? s + 4 : s;
#pragma unroll
        for (int u = 0; u < 4; u++) { pn[u] = pp[2 * (sn + u)]; a1[u] = xw[8 * (sn + u)]; a2[u] = xw[8 * (sn + u) + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        const double x1 = fma(k2, xc, -xp), x2 = fma(k2, x1, -xc), x3 = fma(k2, x2, -x1);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xc * vall[s], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1 * vall[s + 1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2 * vall[s + 2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3 * vall[s + 3], b[3], acc1, 0, 0, 0);
        xp = x3; xc = fma(k2, x3, -x2);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ void rx2_refine(RxShared2 *sh, const double *vmg, int *tmax, double *fmax, int t0, int nt, double fstart, double fstop, double fstep, bool in_sync)
{
    const int tid = rx_tid(), lane = tid & 63, wave = tid >> 6;
    const int nf = (int)ceil((fstop - fstart) / fstep);
    const double delta = (fstart + fstep) - fstart;
    const int ntasks = ((2 * nf + 15) >> 4) * 2;
    const int i = lane & 15, kk = lane >> 4;
    PH2_T0();
    if (!in_sync) refine2_tables(sh, tid, fstart, fstop, fstep);
    for (int j = tid; j < 2 * 176; j += NT2) {                            // the two windows as doubles: (xr, xi, xi, -xr)
        const int frame = j / 176, k = j - frame * 176;
        const float2 x = sh->rxb[min(t0 + frame * RD_NMF + k, RD_RXBUF - 1)];
        double *d = (double *)&sh->xm[0] + 4 * j;
        d[0] = (double)x.x; d[1] = (double)x.y; d[2] = (double)x.y; d[3] = -(double)x.x;
    }
    __syncthreads();
    PH2(27);
    float best = -1.0f; int bf = 0x7fffffff, bt = 0x7fffffff;
    if (in_sync) {
        // wavefront w = (half w >> 1 of the samples, frame w & 1): the two quarters of its half accumulate into the same tile
        {
            f64x4 a0 = { 0.0, 0.0, 0.0, 0.0 }, a1 = a0;
            refine2_moments(sh, vmg, wave & 1, 2 * (wave >> 1), nt, lane, a0, a1);
            refine2_moments(sh, vmg, wave & 1, 2 * (wave >> 1) + 1, nt, lane, a0, a1);
            const f64x4 part = a0 + a1;
#pragma unroll
            for (int r = 0; r < 4; r++) sh->rmom[wave >> 1][wave & 1][lane][r] = part[r];
        }
        __syncthreads();
        PH2(28);
        for (int o = tid; o < 2 * 64 * 4; o += NT2) {                     // C layout (f64 16x16x4): col = lane & 15 (t), row = (lane >> 4) + 4 * reg
            const int frame = o >> 8, l = (o >> 2) & 63, r = o & 3;
            sh->mtot[frame][(l >> 4) + 4 * r][l & 15] = sh->rmom[0][frame][l][r] + sh->rmom[1][frame][l][r];
        }
        __syncthreads();
        PH2(29);
        for (int o = tid; o < nf * 16; o += NT2) {
            const int fo = o >> 4, t = o & 15;
            if (t >= nt) continue;
            const double al = sh->ral[fo];
            const double2 ph = sh->rph[fo], rt = sh->rrot[fo];
            float2 d12[2];
#pragma unroll
            for (int frame = 0; frame < 2; frame++) {
                double re = 0.0, im = 0.0, cm = 1.0;
#pragma unroll
                for (int mq = 0; mq < 8; mq++) {
                    const double mr = sh->mtot[frame][2 * mq][t], mi = sh->mtot[frame][2 * mq + 1][t];
                    if ((mq & 3) == 0) { re = fma(cm, mr, re); im = fma(cm, mi, im); }
                    else if ((mq & 3) == 1) { re = fma(cm, mi, re); im = fma(-cm, mr, im); }
                    else if ((mq & 3) == 2) { re = fma(-cm, mr, re); im = fma(-cm, mi, im); }
                    else { re = fma(-cm, mi, re); im = fma(cm, mr, im); }
                    cm = cm * al * (1.0 / (double)(mq + 1));
                }
                double xr = re * ph.x - im * ph.y, xi = re * ph.y + im * ph.x;
                if (frame == 1) { const double tr = xr * rt.x - xi * rt.y; xi = xr * rt.y + xi * rt.x; xr = tr; }
                d12[frame] = make_float2((float)xr, (float)xi);
            }
            const float v = hypotf(d12[0].x + d12[1].x, d12[0].y + d12[1].y);
            if (v > best || (v == best && (fo < bf || (fo == bf && t < bt)))) { best = v; bf = fo; bt = t; }
        }
    } else {
        auto finish = [&](const f64x4 &acc, int mt, int frame) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double mine = acc[r], other = __shfl_xor(mine, 16);
                if ((kk & 1) == 0) {
                    const int fo = 8 * mt + (kk >> 1) + 2 * r;
                    double re = mine, im = other;
                    if (frame == 1 && fo < nf) {
                        const double2 rt = sh->rrot80[fo];
                        const double tr = re * rt.x - im * rt.y; im = re * rt.y + im * rt.x; re = tr;
                    }
                    if (fo < nf && i < nt) sh->dtr[(frame * nf + fo) * 16 + i] = make_float2((float)re, (float)im);
                }
            }
        };
        for (int task = wave; task < ntasks; task += NW2) finish(refine2_tile(sh, task >> 1, task & 1, 0, 80, nf, nt, lane), task >> 1, task & 1);
        __syncthreads();
        for (int task = tid; task < nf * nt; task += NT2) {
            const int fi = task / nt, ti = task - fi * nt;
            const float2 a = sh->dtr[fi * 16 + ti], b = sh->dtr[(nf + fi) * 16 + ti];
            const float v = hypotf(a.x + b.x, a.y + b.y);
            if (v > best || (v == best && (fi < bf || (fi == bf && ti < bt)))) { best = v; bf = fi; bt = ti; }
        }
    }
    PH2(30);
    block_argmax2(sh, best, bf, bt);
    PH2(31);
    if (best > 0.0f) { *tmax = t0 + bt; *fmax = dgrid_nc(fstart, bf, delta); }      // (two roundings, like np.arange's elements: rade_devutil.h)
}

// rx_buf and the |Dt| row sums from the stream's HBM record into LDS: every load of a thread requested before the first store (as a plain loop
// the compiler cannot tell that the LDS stores do not alias the record and waits for each load before the next: 13 round trips in a row)
__device__ __forceinline__ void rx2_load_rxbuf(RxShared2 *sh, const rd_rx_stream *st, int tid)
{
    constexpr int NB_ = (RD_RXBUF + NT2 - 1) / NT2, NR_ = (RD_NMF + NT2 - 1) / NT2;
    float2 v[NB_]; float r1[NR_], r2[NR_];
#pragma unroll
    for (int q = 0; q < NB_; q++) { const int i = min(tid + q * NT2, RD_RXBUF - 1); v[q] = make_float2(st->rx_buf[i][0], st->rx_buf[i][1]); }
#pragma unroll
    for (int q = 0; q < NR_; q++) { const int i = min(tid + q * NT2, RD_NMF - 1); r1[q] = st->rowsum1[i]; r2[q] = st->rowsum2[i]; }
#pragma unroll
    for (int q = 0; q < NB_; q++) { const int i = tid + q * NT2; if (i < RD_RXBUF) sh->rxb[i] = v[q]; }
#pragma unroll
    for (int q = 0; q < NR_; q++) { const int i = tid + q * NT2; if (i < RD_NMF) { sh->rowsum1[i] = r1[q]; sh->rowsum2[i] = r2[q]; } }
}

// ---- decoder + output stage for the rows a stream has pending (rx_decode_pending on four wavefronts, chunks of 12 rows) -----------
// rx_buf and the row sums leave LDS for the duration (the stage's 64 KB overlay them): out to the stream's HBM record, back afterwards
__device__ __forceinline__ void rx2_decode_pending(RxShared2 *sh, const rd_sync_args &a, int b)
{
    RxScalars *S = &sh->S;
    DecShared2 *ds = (DecShared2 *)&sh->dec_raw[0];
    rd_rx_round *rnd = a.round + b;
    rd_rx_stream *st = a.st + b;
    const int tid = rx_tid();
    const int Tb = S->n_rows;
    PH2_T0();
    for (int i = tid; i < RD_RXBUF; i += NT2) { st->rx_buf[i][0] = sh->rxb[i].x; st->rx_buf[i][1] = sh->rxb[i].y; }
    for (int i = tid; i < RD_NMF; i += NT2) { st->rowsum1[i] = sh->rowsum1[i]; st->rowsum2[i] = sh->rowsum2[i]; }
    __syncthreads();
    PH2(19);
    for (int i = tid; i < Tb; i += NT2) ds->rst[i] = rnd->row_reset[i];
    if (tid == 0) S->lds_sync = 0;
    __syncthreads();
    for (int c0 = 0; c0 < Tb; c0 += DQ2_ROWS) {
        const int n = min(DQ2_ROWS, Tb - c0);
        unsigned rstmask = 0u;
        for (int t = 0; t < n; t++) rstmask |= ds->rst[c0 + t] ? (1u << t) : 0u;
        if (!CENSUS(1)) dq2_layers(ds, a.dec, b, a.dec.z + (size_t)b * a.dec.z_sb + (size_t)c0 * RD_LATENT, a.dec.out + (size_t)b * a.dec.out_sb + (size_t)c0 * a.dec.out_w, n, rstmask, CENSUS(2));
    }
    const float *f84 = a.dec.out + (size_t)b * a.dec.out_sb;
    for (int r = tid; r < Tb; r += NT2) ds->err[r] = f84[r * 84 + 20] > 0.0f ? 1 : 0;
    float *out = a.features_out + (size_t)b * a.feat_stride + (size_t)S->out_base * RD_FEAT_MF;
    for (int i0 = tid; i0 < (Tb / 3) * RD_FEAT_MF; i0 += 4 * NT2) {      // four loads in flight per thread (a plain loop waits for each load before its store: the stores may alias)
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = min(i0 + q * NT2, (Tb / 3) * RD_FEAT_MF - 1), fr = i / 36, j = i - fr * 36;
            const int row = fr >> 2, sub = fr & 3;
            v[q] = f84[row * 84 + sub * 21 + min(j, 19)];
            if (j >= 20) v[q] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = i0 + q * NT2; if (i < (Tb / 3) * RD_FEAT_MF) out[i] = v[q]; }
    }
    __syncthreads();
    if (a.trace) {
        for (int c = S->batch_call0 + tid; c < S->n_calls; c += NT2) {
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
    PH2_RESTART();
    rx2_load_rxbuf(sh, st, tid);
    __syncthreads();
    PH2(19);
}

// radae_rxe.py --bypass_dec (:300-302, :315; the mode rade_api.c drives with the external C decoder): the pending z_hat rows leave as they are -- 240 floats per
// valid modem frame -- and the aux-bit (UW) errors are never summed, so uw_fail cannot occur (sum_uw_errors is only called behind the decoder, :303-312)
__device__ __forceinline__ void rx2_bypass_pending(RxShared2 *sh, const rd_sync_args &a, int b)
{
    RxScalars *S = &sh->S;
    const int tid = rx_tid();
    const int Tb = S->n_rows;
    const float *z = a.dec.z + (size_t)b * a.dec.z_sb;
    float *out = a.features_out + (size_t)b * a.feat_stride + (size_t)S->out_base * RD_ZMF;
    for (int i = tid; i < Tb * RD_LATENT; i += NT2) out[i] = z[i];
    __syncthreads();
    if (tid == 0) { S->out_base += Tb / 3; S->n_rows = 0; S->uw_from_row = 0; S->pending_valid = 0; S->batch_call0 = S->n_calls; S->need_decode = 0; }
    __syncthreads();
}

// check_pilots' row refreshes of one modem frame, NRT tiles of 16 row draws per wavefront, by the two-stage correlator (rx2_detect_q's algebra): the three tiles of 16 row draws against the
// moment table (A fragments straight from L2, two k-steps ahead: 40 KB per call and wavefront instead of 60 + 40 over the two wavefronts a frame had), the moments
// expanded to the 40 frequencies, |Dt| summed over them in the wavefront -> rowsum1 / rowsum2 directly (rounds 3-4: partial sums of two wavefronts through a table and
// a second barrier).  The gathers: lane (row draw i, group g) needs samples 16 s + 4 g .. + 3 of its window in k-step s, and the windows start at random offsets, so the
// lanes of a read hit banks at random whatever the layout.  What the layout decides is how many LDS cycles that costs: with a sample's two plane words side by side
// one ds_read_b64 (64 banks, 2 cycles conflict-free) brings what took two ds_read_b32 (32 banks, 2 cycles each) -- measured by bank model over random draws: 6.6 against
// 13.1 LDS cycles per sample and wave-instruction -- and every window is read by ONE wavefront, not two.
typedef __attribute__((address_space(3))) const u32x2 lds_cu32x2;
// pre() runs once behind the first table requests, side(s) behind the matrix instructions of k-step s: work of the caller's that does not depend on this
// function's results, placed where the wavefront would otherwise wait for table fragments from L2 (6 NRT matrix instructions per k-step cover 100-200 cycles of
// a round trip of 800)
template <int NRT, int DA, class Pre, class Side>
__device__ __forceinline__ void check2_rows_q(RxShared2 *sh, const unsigned short *corrq16_, const unsigned short *corra16_, int frame, int rt0, int lane, float rx_unsc_, Pre pre, Side side)
{
    const int i = lane & 15, g = lane >> 4;
    const float rx_unsc = rx_unsc_ * 0x1p-3f;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(corrq16_), 0, 2 * 10 * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)uni_ptr(corra16_), 0, 5 * 2048, 0x00020000);
    // four address registers per row tile that the compiler cannot see through: merged into ds_read2_b64 (or an unaligned ds_read_b128) the reads would run at half rate
    lds_cu32x2 *px[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        lds_cu32x2 *b0 = (lds_cu32x2 *)&sh->rxhl[0] + (sh->rows48[(rt0 + rt) * 16 + i] + frame * RD_NMF + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; j++) { px[rt][j] = b0 + j; asm volatile("" : "+v"(px[rt][j])); }
    }
    f32x4 acc1[NRT][2];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) { acc1[rt][0] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; acc1[rt][1] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f }; }
    u32x4 A[DA][4];                                            // [k-step mod DA][2 tile + plane]: DA k-steps of table fragments in flight
    auto fetchA = [&](int slot, int s) {
#pragma unroll
        for (int u = 0; u < 4; u++) A[slot][u] = __builtin_amdgcn_raw_buffer_load_b128(qrs, lane * 16, (((u >> 1) * 10 + s) * 2 + (u & 1)) * 1024, 0);
    };
    u32x4 bh[2][NRT], bl[2][NRT];
    auto rows = [&](int slot, int s) {
#pragma unroll
        for (int rt = 0; rt < NRT; rt++)
#pragma unroll
            for (int j = 0; j < 4; j++) { const u32x2 v = px[rt][j][16 * s]; bh[slot][rt][j] = v[0]; bl[slot][rt][j] = v[1]; }
    };
#pragma unroll
    for (int d = 0; d < DA; d++) fetchA(d, d);
    pre();
    rows(0, 0);
    u32x4 A2[2 * 5];                                           // stage 2's fragments [2 q + plane]: requested under the last k-steps
#pragma unroll
    for (int s = 0; s < 10; s++) {
        if (s + 1 < 10) rows((s + 1) & 1, s + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const f16x8 ah = __builtin_bit_cast(f16x8, A[s % DA][2 * nt]), al = __builtin_bit_cast(f16x8, A[s % DA][2 * nt + 1]);
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) acc1[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, __builtin_bit_cast(f16x8, bh[s & 1][rt]), acc1[rt][nt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) acc1[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, bl[s & 1][rt]), acc1[rt][nt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) acc1[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, bh[s & 1][rt]), acc1[rt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + DA < 10) fetchA(s % DA, s + DA);
        if (s == 10 - DA) {
#pragma unroll
            for (int u = 0; u < 10; u++) A2[u] = __builtin_amdgcn_raw_buffer_load_b128(ars, lane * 16, u * 1024, 0);
        }
        side(s);
        __builtin_amdgcn_sched_barrier(0);
    }
    float *rowsum = frame ? sh->rowsum2 : sh->rowsum1;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        const MomPlanes mp = mom_split(acc1[rt][0], acc1[rt][1]);
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 5; q++) {      // C layout: column = lane & 15 (row draw), rows 4 g + r = (re, im) of f = 8 q + 2 g and f + 1
            const f32x4 c = mom_expand(__builtin_bit_cast(f16x8, A2[2 * q]), __builtin_bit_cast(f16x8, A2[2 * q + 1]), mp);
            s += rx_unsc * __builtin_amdgcn_sqrtf(fmaf(c[0], c[0], c[1] * c[1])) + rx_unsc * __builtin_amdgcn_sqrtf(fmaf(c[2], c[2], c[3] * c[3]));
        }
        {   // the other three lane groups hold the row's other frequencies
            const auto p16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
            s = __uint_as_float(p16[0]) + __uint_as_float(p16[1]);
            const auto p32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
            s = __uint_as_float(p32[0]) + __uint_as_float(p32[1]);
        }
        // (two draws of the same row compute the same sum from the same samples: whichever store lands last, the value is the same)
        if (g == 0) rowsum[sh->rows48[(rt0 + rt) * 16 + i]] = s;
    }
}

// ---- band-pass filter (complex_bpf, dsp.py:39-102) ahead of the receiver -----------------------------------------------------------------
// The filter is a stateful streaming FIR of the input: mix down with a running phase, 101 real taps, mix up.  Nothing in it depends on the sync
// state machine, so it runs as a bulk pre-pass over all the samples of a rade_batch_rx invocation (k_rx_bpf_chain + k_rx_bpf) and the receiver
// kernel reads filtered samples.  What the reference's arithmetic does depend on is how the stream is cut into calls: the phase is a complex64
// carried from call to call (phase_vec = phase * phase_vec_exp[0:n], self.phase = phase_vec[-1]: dsp.py:70-71, :99).  The pre-pass therefore
// follows the reference's own partition: block 0 = the nin the stream's next call will consume (from its state record), every later block Nmf
// samples -- exact unless a timing slip changes nin INSIDE an invocation; from that call on the stream filters its own samples (rx2_bpf_own)
// until the invocation ends, and the next invocation's pre-pass starts from the state it left.  Either way every output sample is what
// complex_bpf.bpf would have produced for the stream's actual sequence of calls, from ONE arithmetic (bpf_stage_planes + bpf_fir_tile, shared by
// the pre-pass kernel and rx2_bpf_own; mixers without fused multiply-adds: cmul_nc), so cutting a stream into invocations differently cannot move a bit.
// The same kernels filter the transmit side (radae_txe.py:74-83, :130-132: the optional Tx band-pass filter, there followed by the magnitude clip).
//   chain[b] (float2 [chain_stride]): entry 0 = (nin0, mem_len0) as integer bits, entry 1 + k = the phase block k starts from.
// Memory quirk kept (dsp.py:55 against :96): the memory holds Ntap - 1 = 100 samples before the first call and Ntap + 1 = 102 after it while the
// strided window always starts at index 0, so outputs are delayed by two more samples from the second call on.

// ---- the 101-tap FIR on the matrix cores (shared by the pre-pass kernel and the receiver's own off-grid filtering: one arithmetic) -----------
// y[i] = sum_t h[t] w[i + t] over a window w of baseband samples is the product of the Toeplitz matrix T[r][m] = h[m - r] (16 x 128, taps padded
// with zeros) with the Hankel matrix U[m][q] = w[16 q + m]: Y[r][q] = y[16 q + r], one 16 x 16 tile = 256 consecutive outputs.  On
// v_mfma_f32_16x16x32_f16: T is the A operand (two binary16 planes of 2^10 h, a constant table: rd_bpf16_table_fill), U the B operand -- its
// fragment for (column q, k-group g, k-step ks) is EIGHT CONSECUTIVE window samples starting at 16 q + 32 ks + 8 g, one aligned 16-byte LDS read
// from a plane -- real and imaginary parts as separate planes, each split hi + lo (22 bits, one power-of-two scale per block from its largest
// component): hi hi + hi lo + lo hi in f32, 24 matrix instructions per tile against 51,712 vector FMAs.  The vector FIR was bound by LDS reads
// (every thread re-read its sliding window); this form reads each plane entry 8 times instead of 101.
#define BPF_NPL 1408                   /* halfs per plane: five tiles of 256 outputs + the 127 samples the last rows reach beyond */
#define BPF_TILES(n) (((n) + 255) >> 8)
struct BpfLds {                        // 11,264 B: exactly the receiver's xm work area
    __attribute__((aligned(16))) _Float16 rh[BPF_NPL], rl[BPF_NPL], ih[BPF_NPL], il[BPF_NPL];
};
// The window [memory (102) | block (n)] into the four planes, entry i of it at plane index i - o (o = 2 before a stream's first call, whose memory is two
// samples shorter: dsp.py:55; entries below o are dropped): thread tid brings head = entry tid (tid < 102) and body[q] = entry 102 + tid + 256 q, ZERO
// beyond the block -- all loaded by the caller in one batch, so that a workgroup pays one memory round trip for its window and not one per entry.
// Returns the factor that undoes the operand scales.  Every thread of the 256-thread workgroup calls this (two barriers inside); maxw is an LDS word.
#define BPF_NQ 5                       /* body entries per thread: 5 x 256 >= the longest block (1152, the end-of-over frame on the transmit side) */
__device__ __forceinline__ float bpf_stage_planes(BpfLds *pl, unsigned *maxw, int tid, float2 head, const float2 (&body)[BPF_NQ], int o)
{
    float m = tid < 102 ? fmaxf(fabsf(head.x), fabsf(head.y)) : 0.0f;
#pragma unroll
    for (int q = 0; q < BPF_NQ; q++) m = fmaxf(m, fmaxf(fabsf(body[q].x), fabsf(body[q].y)));
    if (tid == 0) *maxw = 0u;
    __syncthreads();
    m = wave_max_f32(m);
    if ((tid & 63) == 0) atomicMax(maxw, __float_as_uint(m));
    __syncthreads();
    const int eb = min(max((int)((*maxw >> 23) & 0xffu), 32), 222);
    const float sc = __uint_as_float((unsigned)(127 + 7 - (eb - 127)) << 23);          // the largest component lands in [2^7, 2^8)
    auto put = [&](int w, float2 v) {
        const float xr = v.x * sc, xi = v.y * sc;
        const _Float16 h0 = (_Float16)xr, h1 = (_Float16)xi;
        pl->rh[w] = h0; pl->rl[w] = (_Float16)(xr - (float)h0); pl->ih[w] = h1; pl->il[w] = (_Float16)(xi - (float)h1);
    };
    if (tid < 102 && tid >= o) put(tid - o, head);
#pragma unroll
    for (int q = 0; q < BPF_NQ; q++) put(102 - o + tid + 256 * q, body[q]);
    if (tid < BPF_NPL - (102 + 256 * BPF_NQ)) put(102 + 256 * BPF_NQ + tid, make_float2(0.0f, 0.0f));      // the tail the last tile's rows reach into
    if (tid < o) put(BPF_NPL - o + tid, make_float2(0.0f, 0.0f));
    __syncthreads();
    return __uint_as_float((unsigned)(eb - 7 - 10) << 23);                            // 2^(E - 7) from the samples, 2^-10 from the taps
}
// outputs 256 tile + 16 (lane & 15) + 4 (lane >> 4) + r, r = 0..3, of the staged window: (re[r], im[r]), still in operand scale
struct BpfTaps { f16x8 h[4], l[4]; };          // the lane's A fragments (rd_bpf16_table_fill): loaded once, ahead of the staging
__device__ __forceinline__ void bpf_load_taps(BpfTaps &t, const unsigned short *tab16, int lane)
{
    typedef const __attribute__((address_space(1))) f16x8 glb_f16x8_t;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) { t.h[ks] = *(glb_f16x8_t *)(tab16 + (((size_t)ks * 2) * 64 + lane) * 8); t.l[ks] = *(glb_f16x8_t *)(tab16 + (((size_t)ks * 2 + 1) * 64 + lane) * 8); }
}
__device__ __forceinline__ void bpf_fir_tile(const BpfLds *pl, const BpfTaps &t, int tile, int lane, f32x4 &re, f32x4 &im)
{
    const f16x8 (&Ah)[4] = t.h, (&Al)[4] = t.l;
    const int w0 = 256 * tile + 16 * (lane & 15) + 8 * (lane >> 4);
    f32x4 a[6];
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const f16x8 brh = *(const f16x8 *)&pl->rh[w0 + 32 * ks], brl = *(const f16x8 *)&pl->rl[w0 + 32 * ks];
        const f16x8 bih = *(const f16x8 *)&pl->ih[w0 + 32 * ks], bil = *(const f16x8 *)&pl->il[w0 + 32 * ks];
        a[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[ks], brh, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ks], brl, a[1], 0, 0, 0);
        a[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ks], brh, a[2], 0, 0, 0);
        a[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[ks], bih, a[3], 0, 0, 0);
        a[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ks], bil, a[4], 0, 0, 0);
        a[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ks], bih, a[5], 0, 0, 0);
    }
    re = (a[0] + a[1]) + a[2]; im = (a[3] + a[4]) + a[5];
}

// baseband sample q of the invocation (x[q] times the phase of its block), q >= -102: the filter memory a stream needs when it leaves the grid or
// the launch ends; negative q reaches into the memory the invocation started with
__device__ __forceinline__ float2 rx2_bpf_mem(const rd_sync_args &a, int b, int nin0, int q)
{
    const rd_rx_stream *st = a.st + b;
    if (q < 0) return make_float2(st->bpf.mem[102 + q][0], st->bpf.mem[102 + q][1]);
    const float2 *x = (const float2 *)a.rx + (size_t)b * a.rx_stride;
    const float2 *chain = (const float2 *)a.bpf_chain + (size_t)b * a.chain_stride;
    const int k = q < nin0 ? 0 : 1 + (q - nin0) / RD_NMF, sk = k ? nin0 + (k - 1) * RD_NMF : 0;
    return cmul_nc(x[q], cmul_nc(chain[1 + k], ld2(a.tab->bpf_E, q - sk)));
}

// One call's filtering by the stream's own workgroup (off the grid): the pre-pass's arithmetic on the stream's actual call.  Cold path; the filter
// memory lives in the stream record between calls.
__device__ __forceinline__ void rx2_bpf_own(RxShared2 *sh, const rd_sync_args &a, int b, float2 *rxf, int cons0, int nin, int calls0)
{
    static_assert(sizeof(BpfLds) <= sizeof(sh->xm), "the planes overlay the xm work area");
    RxScalars *S = &sh->S;
    rd_rx_stream *st = a.st + b;
    const rd_tables *tab = a.tab;
    const float2 *x = (const float2 *)a.rx + (size_t)b * a.rx_stride;
    const int tid = rx_tid();
    const bool leaving = S->bpf_grid != 0;     // leaving the grid at this call: memory = the 102 baseband samples before it, phase = the chain's value at this block boundary
    const float2 ph = leaving ? ((const float2 *)a.bpf_chain)[(size_t)b * a.chain_stride + 1 + calls0] : S->bpf_phase;
    const int nin0 = S->nin0;
    BpfLds *pl = (BpfLds *)&sh->xm[0];
    const int wave = tid >> 6, lane = tid & 63;
    BpfTaps taps; bpf_load_taps(taps, a.bpf16, lane);
    auto mixed = [&](int j) { return cmul_nc(x[cons0 + j], cmul_nc(ph, ld2(tab->bpf_E, j))); };       // baseband sample j of this call
    float2 head = make_float2(0.0f, 0.0f), body[BPF_NQ];
    if (tid < 102) head = leaving ? rx2_bpf_mem(a, b, nin0, cons0 - 102 + tid) : make_float2(st->bpf.mem[tid][0], st->bpf.mem[tid][1]);
#pragma unroll
    for (int q = 0; q < BPF_NQ; q++) { const int j = tid + 256 * q; body[q] = mixed(min(j, nin - 1)); if (j >= nin) body[q] = make_float2(0.0f, 0.0f); }
    const float2 memv = mixed(nin - 102 + min(tid, 101));      // new memory = the last 102 of [memory | new] (nin >= 800: all of them new samples)
    const float unsc = bpf_stage_planes(pl, (unsigned *)&sh->redi[14], tid, head, body, 0);
    // The outputs go to the stream's slice of the pre-pass buffer (what rade_batch_rx_filtered shows) AND, through LDS, to the caller: the caller's threads
    // read samples other lanes produced, and a plain global load may hit the vector L1 line the previous call's first-touch loads left there (stale
    // pre-pass values) -- stores go through to L2 without refreshing it.  xm is free once every wavefront is done with the planes.
    f32x4 re[2], im[2];
    for (int u = 0; u < 2; u++) { const int tile = wave + (NT2 / 64) * u; if (tile < BPF_TILES(nin)) bpf_fir_tile(pl, taps, tile, lane, re[u], im[u]); }
    __syncthreads();
    for (int u = 0; u < 2; u++) {
        const int tile = wave + (NT2 / 64) * u;
        if (tile >= BPF_TILES(nin)) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 256 * tile + 16 * (lane & 15) + 4 * (lane >> 4) + r;
            if (i < nin) { const float2 y = cmul_nc(make_float2(re[u][r] * unsc, im[u][r] * unsc), cconj(cmul_nc(ph, ld2(tab->bpf_E, i)))); rxf[cons0 + i] = y; sh->xm[i] = y; }
        }
    }
    __syncthreads();
    if (tid < 102) { st->bpf.mem[tid][0] = memv.x; st->bpf.mem[tid][1] = memv.y; }
    if (tid == 0) { S->bpf_phase = cmul_nc(ph, ld2(tab->bpf_E, nin - 1)); S->bpf_grid = 0; }
    __syncthreads();
}

template <int CTRL> __device__ __forceinline__ float dpp_f32(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
#define DPP_ROW_HALF_MIRROR 0x141   /* lane i <-> 7 - i inside every group of eight lanes */

#ifndef RX2_WG_PER_CU
#define RX2_WG_PER_CU 2          /* developer switch: the register budget of a build that would hold three workgroups per CU (168 VGPRs) */
#endif
__global__ __launch_bounds__(NT2, RX2_WG_PER_CU) void k_rx_sync2(rd_sync_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    RxShared2 *sh = (RxShared2 *)smem_raw;
    RxScalars *S = &sh->S;
    const rd_tables *tab = a.tab;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wv = rx2_wave();
    rd_rx_stream *st = a.st + b;
    rd_rx_round *rnd = a.round + b;
    const float2 *rxin = (const float2 *)a.rx + (size_t)b * a.rx_stride;

    // ---- load the stream's working set into LDS
    rx2_load_rxbuf(sh, st, tid);
    if (tid == 0) {
        S->state = st->state; S->nin = st->nin; S->tmax = st->tmax; S->tmax_candidate = st->tmax_candidate; S->valid_count = st->valid_count;
        S->uw_errors = st->uw_errors; S->synced_count = st->synced_count; S->mf = st->mf; S->f_ind_max = st->f_ind_max;
        S->dec_reset_pending = st->dec_reset_pending; S->has_eoo = st->has_eoo; S->lcg = st->lcg;
        S->rxmax_cur = st->rxmax[0]; S->rxmax_h0 = st->rxmax[1]; S->rxmax_h1 = st->rxmax[2];
        S->fmax = st->fmax; S->foff_err = st->foff_err; S->rph_th = st->rx_theta;     // rx_phase (radae_rxe.py:227-233) kept as its angle: see the corrected window
        S->Dthresh = st->Dthresh; S->Dtmax12 = st->Dtmax12; S->Dtmax12_eoo = st->Dtmax12_eoo; S->snr_est = st->snr_est;
        S->bpf_phase = make_float2(st->bpf.phase[0], st->bpf.phase[1]);      // only used off the grid (rx2_bpf_own)
        S->consumed_inv = a.acc[b * 4 + 0]; S->calls_inv = a.acc[b * 4 + 1]; S->valid_inv = a.acc[b * 4 + 2]; S->eoo_inv = a.acc[b * 4 + 3];
        S->n_calls = 0; S->n_rows = 0; S->uw_from_row = 0; S->consumed_round = 0; S->pending_valid = 0; S->out_base = S->valid_inv;
        S->go = 0; S->dt_valid = st->dt_valid; S->dt_new = 0; S->lds_sync = 0; S->need_decode = 0; S->batch_call0 = 0; S->tab_ok = 0;
        S->bpf_grid = st->bpf.grid_off == 0; S->nin0 = __float_as_int(((const float *)a.bpf_chain)[(size_t)b * a.chain_stride * 2]);
    }
    const int avail = a.avail[b];
    const long long wg_t0 = clock64();
    __syncthreads();
    PH2_T0(); PH2(0);

    for (int it = 0; it <= a.round_calls; it++) {
        int tid = rx2_tid(wv);
        auto prepare_next = [&]() {
            const int calls_inv = S->calls_inv, n_calls = S->n_calls, valid_inv = S->valid_inv, consumed = S->consumed_inv, nin_n = S->nin;
            const int n_rows = S->n_rows, st_n = S->state, sc = S->synced_count;
            const unsigned m0 = S->rxmax_h0, m1 = S->rxmax_cur, m2 = S->rxmax_h1;
            const int go = (int)(calls_inv < a.max_calls) & (int)(n_calls < a.round_calls) & (int)(valid_inv < a.feat_cap) & (int)(consumed + nin_n <= avail);
            const int need = (int)(n_rows > 0) & ((go ^ 1) | ((int)(st_n == ST_SYNC) & (int)(((sc + 1) % 8) == 0)) | (int)(n_rows + 3 > a.dec_rows));
            S->need_decode = need; S->go = go;
            S->rxmax_h1 = go ? m0 : m2; S->rxmax_h0 = go ? m1 : m0; S->rxmax_cur = go ? 0u : m1;
            S->state_before = st_n; S->nin_before = nin_n;
            S->valid_output = 0; S->endofover = 0; S->uw_fail = 0; S->candidate = 0;
        };
        if (it == 0) {
            if (tid == 0) prepare_next();
            __syncthreads();
            tid = rx2_tid(wv);
        }
        PH2(1);
        if (S->need_decode) { if (a.bypass_dec) rx2_bypass_pending(sh, a, b); else rx2_decode_pending(sh, a, b); PH2(2); }
        if (!S->go) break;
        const int nin = S->nin, state = S->state;
        if (tid == 0 && state != ST_SYNC) S->tab_ok = 0;
        const int mf0 = S->mf, n_rows0 = S->n_rows;
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
                const bool unsync_enable = !(a.unsync_off_after >= 0 && S->synced_count > a.unsync_off_after);
                if (S->candidate) S->valid_count = 25;
                else { S->valid_count--; if (unsync_enable && S->valid_count == 0) next_state = ST_SEARCH; }
                if (unsync_enable && (eoo || S->uw_fail)) next_state = ST_SEARCH;
            }
            S->dt_valid = (state != ST_SYNC && next_state != ST_SYNC) ? S->dt_new + 1 : 0;
            S->state = next_state;
            if (next_state == ST_SEARCH) S->nin = RD_NMF;
            S->mf++;
            const int ret = valid_out | (eoo << 1);
            const int call_idx = S->mf - 2;
            if (valid_out) {
                for (int k = 0; k < 3; k++) { const int rf = (k == 0) ? S->dec_reset_pending : 0; rnd->row_reset[S->n_rows + k] = rf; }
                S->dec_reset_pending = 0; S->n_rows += 3; S->pending_valid++; S->valid_inv++;
            }
            if (eoo) { S->has_eoo = 1; S->eoo_inv++; }
            const int nc = S->n_calls;
            rnd->call_ret[nc] = ret; rnd->call_row_lo[nc] = S->uw_from_row; rnd->call_row_hi[nc] = S->n_rows; rnd->call_trace_idx[nc] = call_idx;
            S->n_calls = nc + 1; S->calls_inv++;
        };
        const int cons0 = S->consumed_inv, calls0 = S->calls_inv;
        // ---- complex_bpf.bpf (dsp.py:63-102) ran ahead of this kernel for every sample of the invocation (k_rx_bpf): the filter does not depend on
        // any sync decision.  This call's nin filtered samples are rxf[cons0 ..) as long as the stream's calls follow the pre-pass's block grid (first
        // block = the nin the invocation started with, then Nmf each: the reference's own call partition unless nin changes inside an invocation);
        // after such a timing slip the stream filters the rest of the invocation's samples itself (rx2_bpf_own: same arithmetic, cold path).
        float2 *rxf = (float2 *)a.rxf + (size_t)b * a.rxf_stride;
        const bool on_grid = S->bpf_grid && nin == (calls0 ? RD_NMF : S->nin0);
        if (!on_grid) { rx2_bpf_own(sh, a, b, rxf, cons0, nin, calls0); tid = rx2_tid(wv); }
        constexpr int NNEW = (RD_NINMAX + NT2 - 1) / NT2;
        float2 nv[NNEW];
#pragma unroll
        for (int q = 0; q < NNEW; q++) { const int i = tid + q * NT2; nv[q] = i < nin ? (on_grid ? rxf[cons0 + i] : sh->xm[i]) : make_float2(0.0f, 0.0f); }
        if (state == ST_SYNC && !S->lds_sync) {      // pilot replicas and equaliser constants share LDS with the pilot search and the decoder stage
            // (after every decoder stage: all of a thread's table loads requested before its first LDS store -- as one statement after the other each load was waited for)
            const int i0 = min(tid, RD_M - 1), cc = min(tid, RD_NC - 1), c6 = min(tid, RD_NC * 6 - 1) / 6, r6 = min(tid, RD_NC * 6 - 1) - 6 * c6;
            const float2 vp = ld2(tab->p, i0), ve = ld2(tab->pend, i0);
            const float vP = tab->P[cc]; const float2 vr = ld2(tab->eq_rot, cc);
            const float2 vm_ = make_float2(tab->Pmat[c6][r6 / 3][r6 % 3][0], tab->Pmat[c6][r6 / 3][r6 % 3][1]);
            const float g0_ = tab->pilot_gain, g1_ = tab->snr_c1, g2_ = tab->snr_c2;
            if (tid < RD_M) { sh->pd[tid] = make_double2(vp.x, vp.y); sh->pendd[tid] = make_double2(ve.x, ve.y); }
            if (tid < RD_NC) { sh->eqP[tid] = vP; sh->eqrot[tid] = vr; }
            if (tid < RD_NC * 6) sh->eqPmat[c6][r6 / 3][r6 % 3] = vm_;
            if (tid == 0) { sh->eq_pg = g0_; sh->eq_snrc1 = g1_; sh->eq_snrc2 = g2_; }
        }
        PH2(3);
        // rx_buf shift and append (radae_rxe.py:196-197); the largest component of the new samples sets the operand scale of the binary16 planes
        constexpr int NK = (RD_RXBUF + NT2 - 1) / NT2;
        float2 keep[NK];
#pragma unroll
        for (int q = 0; q < NK; q++) { const int i = tid + q * NT2; keep[q] = (i + nin < RD_RXBUF) ? sh->rxb[i + nin] : make_float2(0.0f, 0.0f); }
        float mloc = 0.0f;
#pragma unroll
        for (int q = 0; q < NNEW; q++) mloc = fmaxf(mloc, fmaxf(fabsf(nv[q].x), fabsf(nv[q].y)));
        mloc = wave_max_f32(mloc);
        __syncthreads();
        tid = rx2_tid(wv);
#pragma unroll
        for (int q = 0; q < NK; q++) { const int i = tid + q * NT2; if (i + nin < RD_RXBUF) sh->rxb[i] = keep[q]; }
#pragma unroll
        for (int q = 0; q < NNEW; q++) { const int i = tid + q * NT2; if (i < nin) sh->rxb[RD_RXBUF - nin + i] = nv[q]; }
        if ((tid & 63) == 0) atomicMax(&S->rxmax_cur, __float_as_uint(mloc));
        if (tid == 0) {
            if (state == ST_SYNC) S->lds_sync = 1;
            S->consumed_inv += nin; S->consumed_round += nin;
        }
        __syncthreads();
        tid = rx2_tid(wv);

        PH2(5);
        if (state == ST_SEARCH || state == ST_CANDIDATE) {
            // ---- acquisition.detect_pilots (dsp.py:178-231) by FFT convolution, |Dt2| of the previous call reused as |Dt1|
            float best = -1.0f; int bt = 0x7fffffff, bfi = 0;
            const bool cached = S->dt_valid != 0;
            const int oldb = cached ? S->dt_valid - 1 : 0, newb = 1 - oldb;
            float *cache = a.dtcache + (size_t)b * 2 * RD_NFC * RD_NMF;
            if (tid == 0) { S->lds_sync = 0; S->dt_new = newb; }
            if (cached) {
                constexpr int NR = (RD_NMF + NT2 - 1) / NT2;
                float r2[NR];
#pragma unroll
                for (int q = 0; q < NR; q++) r2[q] = sh->rowsum2[min(tid + q * NT2, RD_NMF - 1)];
#pragma unroll
                for (int q = 0; q < NR; q++) sh->rowsum1[min(tid + q * NT2, RD_NMF - 1)] = r2[q];   // each thread moves its own slots: no barrier in between
            }
            __syncthreads();
            tid = rx2_tid(wv);
            PH2(6);
            float rx_unsc;
            {   // operand planes of the whole rx_buf (as for check_pilots in the synchronised state: one power-of-two scale from the running maximum)
                const unsigned mb = max(max(S->rxmax_cur, S->rxmax_h0), S->rxmax_h1);
                const int eb = min(max((int)((mb >> 23) & 0xffu), 32), 222);
                const float rx_sc = __uint_as_float((unsigned)(127 + 7 - (eb - 127)) << 23);
                rx_unsc = __uint_as_float((unsigned)(127 - 12 - 7 + (eb - 127)) << 23);
                for (int i = tid; i < RD_RXBUF; i += NT2) {
                    float2 v = sh->rxb[i]; v.x *= rx_sc; v.y *= rx_sc;
                    const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y;
                    const _Float16 l0 = (_Float16)(v.x - (float)h0), l1 = (_Float16)(v.y - (float)h1);
                    sh->srxh[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    sh->srxl[i] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
            }
            if (CENSUS(512)) { float b_ = -1.0f; int t_ = 0x7fffffff, f_ = 0; rx2_detect_q(sh, a.corrq16, a.corra16, cache, cached ? 1 : 0, oldb, newb, rx_unsc, b_, t_, f_); asm volatile("" :: "v"(b_), "v"(t_), "v"(f_)); }
            rx2_detect_q(sh, a.corrq16, a.corra16, cache, cached ? 1 : 0, oldb, newb, rx_unsc, best, bt, bfi);
            PH2(7);
            block_argmax2(sh, best, bt, bfi);
            const float Dmax = best; const int tbest = bt, fbest = bfi;
            const float sr = sigma_r_from_rowsums2(sh);
            if (tid == 0) {
                S->Dthresh = (double)(2.0f * sr) * RD_SQRT_NLOG_1EM5_5;
                if (Dmax > 0.0f) { S->tmax = tbest; S->f_ind_max = fbest; S->fmax = -50.0 + 2.5 * fbest; S->Dtmax12 = (double)Dmax; }
                else { S->tmax = 0; S->f_ind_max = 0; S->fmax = 0.0; S->Dtmax12 = 0.0; }
                S->candidate = S->Dtmax12 > S->Dthresh;
                S->entry = S->candidate && (abs(S->tmax - S->tmax_candidate) < RD_NCP) && (S->valid_count + 1 > 3);
            }
            __syncthreads();
            tid = rx2_tid(wv);
        } else {
            // ---- in sync: refine, check_pilots, slips, UW, frequency correction, demod
            int tm_ref; double fm_ref;
            {
                const int tm = S->tmax; const double fm = S->fmax;
                const int t0 = max(0, tm - 8);
                int tnew = tm; double fhat = fm;
                {   // check_pilots' operand planes of the whole rx_buf (see k_rx_sync)
                    const unsigned mb = max(max(S->rxmax_cur, S->rxmax_h0), S->rxmax_h1);
                    const int eb = min(max((int)((mb >> 23) & 0xffu), 32), 222);
                    const float rx_sc = __uint_as_float((unsigned)(127 + 7 - (eb - 127)) << 23);
                    if (tid == 0) sh->redf[12] = __uint_as_float((unsigned)(127 - 12 - 7 + (eb - 127)) << 23);
                    if (tid >= NT2 - 64) {
                        // side jobs of the call on the fourth wavefront (they only depend on last call's results): the per-frequency constants of
                        // refine()'s grid, check_pilots' 48 row draws, and the first touch of the NEXT call's filtered samples (HBM + address translation:
                        // 256 streams read 256 separate regions; one load per 128-byte line, values dropped)
                        const int l = tid - (NT2 - 64);
                        if (!S->tab_ok) refine2_tables_sync(sh, l, fm - 1.0, fm + 1.0, 0.1);      // (first synchronised call after sync entry: nobody prepared them)
                        const int k = min(l, 47);
                        const uint32_t x = LCG_A[k] * S->lcg + LCG_C[k];
                        if (l < 48) sh->rows48[k] = (int)((x >> 8) % RD_NMF);
                        if (l == 47) sh->redi[15] = (int)x;
                        const int rem = min(avail - cons0 - nin, RD_NINMAX);
                        float t = 0.0f;
#pragma unroll
                        for (int q = 0; q < 2; q++) { const int i = (l + 64 * q) * 16; if (i < rem) t += rxf[cons0 + nin + i].x; }
                        asm volatile("" :: "v"(t));
                    } else
                    for (int rep = CENSUS_REPS(8); rep > 0; rep--)
                    for (int i = tid; i < RD_RXBUF; i += NT2 - 64) {
                        asm volatile("" ::: "memory");
                        float2 v = sh->rxb[i]; v.x *= rx_sc; v.y *= rx_sc;
                        const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y;
                        const _Float16 l0 = (_Float16)(v.x - (float)h0), l1 = (_Float16)(v.y - (float)h1);
                        sh->rxhl[i] = (u32x2){ (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16),
                                               (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16) };
                    }
                }
                if (CENSUS(16)) { int t_ = tm; double f_ = fm; rx2_refine(sh, a.vm, &t_, &f_, t0, tm + 8 - t0, fm - 1.0, fm + 1.0, 0.1, true); __syncthreads(); }
                rx2_refine(sh, a.vm, &tnew, &fhat, t0, tm + 8 - t0, fm - 1.0, fm + 1.0, 0.1, true);
                tm_ref = tnew; fm_ref = dlin2_nc(0.9, fm, 0.1, fhat);                  // radae_rxe.py:206, rounded like the reference's doubles (rade_devutil.h)
                if (tid == 0) { S->tmax = tm_ref; S->fmax = fm_ref; S->lcg = (uint32_t)sh->redi[15]; }
            }
            PH2(9);
            // check_pilots (dsp.py:273-320): 48 pseudo-random rows x {Dt1, Dt2} x 40 frequencies on the f16 matrix cores, one
            // wavefront per (frame, frequency-tile group); then what only depends on refine()'s result: the four correlations at
            // (tmax, fmax) and the frequency-corrected window the demodulator reads
            {
                const int wave = tid >> 6, lane = tid & 63;
                const float rx_unsc = sh->redf[12];
                // wavefronts 0 / 1: frame 0 / 1, row tiles 0 and 1; wavefronts 2 / 3: frame 0 / 1, row tile 2, and everything else of the phase -- the four correlations at
                // (tmax, fmax) behind the first table requests, one slice of the frequency-corrected window behind the matrix instructions of every k-step
                const int tm = tm_ref; const double w = 2.0 * PI_D * fm_ref / 8000.0;
                int t2 = tm;
                if (t2 >= RD_NMF - RD_M) t2 -= RD_M;
                if (t2 < RD_M) t2 += RD_M;
                const double rph_th = S->rph_th;
                float2 *cA = sh->cisA[wave & 1], *cB = sh->cisB[wave & 1];
                unsigned *dxh = (unsigned *)sh->xm, *dxl = dxh + 1200;
                const float dx_sc = 0x1p-12f / rx_unsc;           // = the planes' 2^(7 - E): a power of two, the division is exact
                auto pre23 = [=]() {      // (captures by VALUE: by reference the captured locals -- tid among them, reassigned all over the call loop -- stay in scratch memory for the whole kernel: 1700 spills)
                    // rx_phase e^{-jw(n+1)} (complex128 in the reference) = e^{j(theta - w(n+1))} for the 1152 window samples as the product of two small tables --
                    // cisA[a] = e^{j(theta - w(64 a + 1))}, cisB[b] = e^{-j w b}, 82 reduced-angle evaluations per wavefront (each wavefront builds its own copy: no
                    // barrier) -- instead of one evaluation per sample (rounds 2-4: sincosf and the double-precision angle were 1.2 k cycles per 128 samples, more
                    // than half of this phase's wave-cycles)
                    cB[lane] = cis_reduced(-w * (double)lane);
                    if (lane < 18) cA[lane] = cis_reduced(rph_th - w * (double)(64 * lane + 1));
                    // the four correlations at (tmax, fmax): 160 samples over the two wavefronts
                    double cr[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
                    for (int rep = CENSUS_REPS(64); rep > 0; rep--)
                    for (int n = tid - 128; n >= 0 && n < RD_M; n += 128) {
                        asm volatile("" ::: "memory");
                        if (rep == 1 && n < 128) { for (int q = 0; q < 8; q++) cr[q] = 0.0; }
                        const float2 cf = cis_reduced(-w * n);
                        const double sn = cf.y, cs = cf.x;
                        const double2 rp = sh->pd[n], re = sh->pendd[n];
                        const int t0s[4] = { tm, tm + RD_NMF, tm + RD_M + RD_NCP, tm + RD_NMF };
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float2 x = sh->rxb[t0s[q] + n];
                            const double qr = cs * x.x - sn * x.y, qi = -(cs * x.y + sn * x.x);
                            const double2 r = q < 2 ? rp : re;
                            cr[2 * q] += qr * r.x - qi * r.y; cr[2 * q + 1] += qr * r.y + qi * r.x;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++) cr[q] = wave_sum_f64(cr[q]);
                    if (lane == 0) {
#pragma unroll
                        for (int q = 0; q < 8; q++) sh->corrp[wave][q] = cr[q];
                    }
                };
                // the corrected window goes straight into two binary16 planes (same power-of-two scale as the rx_buf planes): the demodulator DFT reads them as
                // matrix-core operands.  Sample n sits at n + (n >> 5): the six symbols' fragments then start in different banks.  Slice s = samples
                // 128 s + (tid - 128), s = 0..8 (9 x 128 = 1152)
                auto side23 = [=](int sl) {
                    if (sl >= RD_NEOO / 128) return;
                    const int n = 128 * sl + (tid - 128);
                    float2 v = cmul(sh->rxb[t2 - RD_NCP + n], cmul(cA[n >> 6], cB[n & 63]));
                    v.x *= dx_sc; v.y *= dx_sc;
                    const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y;
                    const _Float16 l0 = (_Float16)(v.x - (float)h0), l1 = (_Float16)(v.y - (float)h1);
                    dxh[n + (n >> 5)] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    dxl[n + (n >> 5)] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                };
                for (int rep = CENSUS_REPS(32); rep > 0; rep--) {
                    asm volatile("" ::: "memory");
                    if (wave < 2) check2_rows_q<2, 5>(sh, a.corrq16, a.corra16, wave, 0, lane, rx_unsc, []() {}, [](int) {});
                    else check2_rows_q<1, 4>(sh, a.corrq16, a.corra16, wave - 2, 2, lane, rx_unsc, pre23, side23);
                }
                PH2(8);
            }
            PH2(11);
            __syncthreads();
            tid = rx2_tid(wv);
            PH2(10);
            // From here to the equaliser ONE phase: the fourth wavefront reduces the row sums to the Rayleigh thresholds, decides candidate /
            // end-of-over / slip, runs the state machine and advances the phase accumulator; the other three run the demodulator DFT
            // (none of the decisions feeds it: the corrected window was cut with the slip-adjusted timing above)
            const double w = 2.0 * PI_D * S->fmax / 8000.0;
            if (tid >= NT2 - 64) {
                const int l = tid - (NT2 - 64);
                double r0 = 0.0, r1 = 0.0;
                for (int t = l; t < RD_NMF; t += 64) { r0 += (double)sh->rowsum1[t]; r1 += (double)sh->rowsum2[t]; }
                r0 = wave_sum_f64(r0); r1 = wave_sum_f64(r1);
                if (l == 0) {
                    const int tm = S->tmax;
                    double red[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) red[q] = sh->corrp[2][q] + sh->corrp[3][q];      // (the two wavefronts that cut the correlations: rx2 check phase)
                    const float sr = sigma_r_from_sums(r0, r1);
                    const double D = sqrt(red[0] * red[0] + red[1] * red[1]) + sqrt(red[2] * red[2] + red[3] * red[3]);     // (well inside double's range: no hypot() scaling)
                    const double De = sqrt(red[4] * red[4] + red[5] * red[5]) + sqrt(red[6] * red[6] + red[7] * red[7]);
                    S->Dthresh = (double)(2.0f * sr) * RD_SQRT_NLOG_1EM4_5;
                    const double Dthresh_eoo = (double)(2.0f * sr) * RD_SQRT_NLOG_1EM5_5;
                    S->Dtmax12 = D; S->Dtmax12_eoo = De;
                    const int eoo = De > Dthresh_eoo;
                    S->candidate = D > S->Dthresh; S->endofover = eoo;
                    int nn = RD_NMF, t2 = tm;
                    if (t2 >= RD_NMF - RD_M) { nn = RD_NMF + RD_M; t2 -= RD_M; }
                    if (t2 < RD_M) { nn = RD_NMF - RD_M; t2 += RD_M; }
                    S->nin = nn; S->tmax = t2;
                    S->synced_count++;
                    if (S->synced_count % 8 == 0) { if (S->uw_errors > 7) S->uw_fail = 1; S->uw_errors = 0; S->uw_from_row = S->n_rows; }
                    state_update(0, !eoo, eoo);
                }
                if (l == 1) { const double th = S->rph_th - w * (double)RD_NEOO; S->rph_th = th - 6.283185307179586476925 * rint(th * 0.15915494309189533577); }
            } else {
                // receiver_one (dsp.py:487-526): 160 -> 30 DFT of the six symbols on the f16 matrix cores.  sym[s][c] = sum_n x_s[n] Wfwd[n][c]: with the row
                // R[c] = (wr, -wi interleaved over n) the real part is R . (xr, xi) and the imaginary part R . (xi, -xr), so the A operand is 32 rows (30 carriers, two
                // planes, a.wfwd16 from L2) and the B operand twelve columns: the six symbols' window planes as they are, and with every (re, im) pair turned into
                // (im, -re) in registers (exact).  hi hi + hi lo + lo hi as in check_pilots: 30 matrix instructions per 16-row tile in three independent chains,
                // one tile each on wavefronts 0 and 1.  (As vector FMAs -- five lanes per carrier, 32 samples each -- this phase was 9 k cycles, and beside a
                // second workgroup the vector ALU is what the two compete for: HISTORY.md 3.9.)
                typedef const __attribute__((address_space(1))) f16x8 glb_f16x8_t;
                const int wave = tid >> 6, lane = tid & 63, col = lane & 15, g = lane >> 4;
                const int sc_ = col < 6 ? col : min(col - 6, 5);
                const unsigned im_var = col >= 6 ? 0xffffffffu : 0u;       // columns 6..11 (and the unused 12..15): the (im, -re) variant
                const unsigned *dxh = (const unsigned *)sh->xm, *dxl = dxh + 1200;
                const float unsc = sh->redf[12];
                static_assert(RD_NCP - 16 == 16 && RD_SYM == 192, "padded window index");
                for (int rep = CENSUS_REPS(256); rep > 0; rep--)
                if (wave < 2) {
                    asm volatile("" ::: "memory");
                    const int tile = wave;
                    const unsigned short *pt = a.wfwd16 + ((size_t)tile * 10 * 2 * 64 + lane) * 8;
                    f16x8 Ah[10], Al[10];
#pragma unroll
                    for (int ks = 0; ks < 10; ks++) { Ah[ks] = *(glb_f16x8_t *)(pt + (size_t)(ks * 2) * 512); Al[ks] = *(glb_f16x8_t *)(pt + (size_t)(ks * 2 + 1) * 512); }
                    f32x4 a0 = { 0.0f, 0.0f, 0.0f, 0.0f }, a1 = a0, a2 = a0;
                    u32x4 bh[2], bl[2];
                    auto turn = [&](unsigned v) { const unsigned t = (v >> 16) | (((v & 0xffffu) ^ 0x8000u) << 16); return (t & im_var) | (v & ~im_var); };   // (re, im) -> (im, -re)
                    auto rows = [&](int slot, int ks) {
                        const int idx = 16 + RD_SYM * sc_ + 16 * ks + 4 * g, pp = idx + (idx >> 5);
#pragma unroll
                        for (int j = 0; j < 4; j++) { bh[slot][j] = turn(dxh[pp + j]); bl[slot][j] = turn(dxl[pp + j]); }
                    };
                    rows(0, 0);
#pragma unroll
                    for (int ks = 0; ks < 10; ks++) {
                        if (ks + 1 < 10) rows((ks + 1) & 1, ks + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[ks], __builtin_bit_cast(f16x8, bh[ks & 1]), a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ks], __builtin_bit_cast(f16x8, bl[ks & 1]), a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ks], __builtin_bit_cast(f16x8, bh[ks & 1]), a2, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f32x4 r = ((a0 + a1) + a2) * unsc;
                    // C layout: column = lane & 15 (symbol, variant), rows 4 g + (0..3) = carriers 16 tile + 4 g + (0..3)
                    if (col < 12) {
                        float *dst = (float *)&sh->sym[sc_][0] + (col >= 6);
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) { const int c = 16 * tile + 4 * g + rr; if (c < RD_NC) dst[2 * c] = r[rr]; }
                    }
                } else if (wave == 2) {
                    // the third wavefront has no part in this phase: it prepares the NEXT call's refine() constants (one double-precision sincos per lane: 4 k cycles
                    // that every wavefront waited for at refine()'s first barrier when they were computed at the top of the call).  fmax is final for this call
                    // (refine() is done, the state machine does not touch it); a call that is not synchronised, or a sync entry, clears the flag.
                    const double fm = S->fmax;
                    refine2_tables_sync(sh, lane, fm - 1.0, fm + 1.0, 0.1);
                    if (lane == 0) S->tab_ok = 1;
                }
            }
            PH2(25);
            __syncthreads();
            tid = rx2_tid(wv);
            PH2(12);
            const int endofover = S->endofover, n_rows = n_rows0;
            float *zrow = a.zrows + ((size_t)b * a.dec_rows + n_rows) * RD_LATENT;
            float *eoo_dst = a.eoo_out ? a.eoo_out + (size_t)b * RD_NEOOBITS : nullptr;
            const int call_idx0 = mf0 - 1;
            if (!endofover) {
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
                tid = rx2_tid(wv);
                if (tid < 128) {
                    const int c = tid & 63;
                    if (tid < 64) {
                        float pm = 0.0f;
                        if (c < RD_NC) {
                            const float2 r0 = sh->rp[0][c], r1 = sh->rp[1][c];
                            const float a0 = hypotf(r0.x, r0.y), a1 = hypotf(r1.x, r1.y); pm = a0 * a0 + a1 * a1;
                        }
                        pm = wave_sum_f32(pm);
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
                tid = rx2_tid(wv);
                const float mag = S->mag;
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
            __syncthreads();
            tid = rx2_tid(wv);
        }

        PH2(13);
        // ---- state machine (radae_rxe.py:248-297).  Sync entry needs the whole workgroup for refine().
        const int do_entry = (state == ST_CANDIDATE) && S->entry;
        if (do_entry) {
            const int tm = S->tmax; const double fm = S->fmax;
            const int t0 = max(0, tm - 1);
            int tnew = tm; double fnew = fm;
            for (int i = tid; i < RD_M; i += NT2) sh->pd[i] = make_double2(tab->p[i][0], tab->p[i][1]);    // the FFT area is dead: pilot replica for the direct sums
            rx2_refine(sh, a.vm, &tnew, &fnew, t0, tm + 2 - t0, fm - 10.0, fm + 10.0, 0.25, false);
            if (tid == 0) { S->tmax = tnew; S->fmax = fnew + S->foff_err; S->foff_err = 0.0; }
            __syncthreads();
            tid = rx2_tid(wv);
        }
        if (tid == 0 && state != ST_SYNC) state_update(do_entry, S->valid_output, S->endofover);
        if (tid == 0 && a.trace) {
            const int call_idx = S->mf - 2;
            if (call_idx < a.trace_cap) {
                rd_rx_trace *tr = a.trace + (size_t)b * a.trace_cap + call_idx;
                tr->state_before = S->state_before; tr->state_after = S->state; tr->nin_before = S->nin_before; tr->nin_after = S->nin; tr->ret = S->valid_output | (S->endofover << 1);
                tr->tmax = S->tmax; tr->f_ind_max = S->f_ind_max; tr->valid_count = S->valid_count; tr->uw_errors = S->uw_errors; tr->synced_count = S->synced_count;
                tr->snr_int = (int)S->snr_est; tr->fmax = S->fmax; tr->Dthresh = S->Dthresh; tr->Dtmax12 = S->Dtmax12; tr->Dtmax12_eoo = S->Dtmax12_eoo; tr->snrdB_3k_est = S->snr_est;
            }
        }
        if (tid == 0) prepare_next();
        __syncthreads();
        tid = rx2_tid(wv);
        PH2(14);
    }

    // ---- write the stream state back
    __syncthreads();
    for (int i = tid; i < RD_RXBUF; i += NT2) { st->rx_buf[i][0] = sh->rxb[i].x; st->rx_buf[i][1] = sh->rxb[i].y; }
    for (int i = tid; i < RD_NMF; i += NT2) { st->rowsum1[i] = sh->rowsum1[i]; st->rowsum2[i] = sh->rowsum2[i]; }
    // band-pass state for the next invocation's pre-pass: on the grid the 102 baseband samples before the consumption point are rebuilt from the
    // input and the block phases (what complex_bpf keeps: dsp.py:96-99); off the grid rx2_bpf_own has kept them in the stream record call by call
    if (S->bpf_grid && S->consumed_inv > 0)
        for (int i = tid; i < 102; i += NT2) { const float2 m = rx2_bpf_mem(a, b, S->nin0, S->consumed_inv - 102 + i); st->bpf.mem[i][0] = m.x; st->bpf.mem[i][1] = m.y; }
    if (tid == 0) {
        st->state = S->state; st->nin = S->nin; st->tmax = S->tmax; st->tmax_candidate = S->tmax_candidate; st->valid_count = S->valid_count;
        st->uw_errors = S->uw_errors; st->synced_count = S->synced_count; st->mf = S->mf; st->f_ind_max = S->f_ind_max;
        st->dec_reset_pending = S->dec_reset_pending; st->has_eoo = S->has_eoo; st->lcg = S->lcg; st->dt_valid = S->dt_valid;
        st->rxmax[0] = S->rxmax_cur; st->rxmax[1] = S->rxmax_h0; st->rxmax[2] = S->rxmax_h1;
        st->fmax = S->fmax; st->foff_err = S->foff_err; st->rx_theta = S->rph_th; { double s_, c_; sincos(S->rph_th, &s_, &c_); st->rx_phase[0] = c_; st->rx_phase[1] = s_; }
        st->Dthresh = S->Dthresh; st->Dtmax12 = S->Dtmax12; st->Dtmax12_eoo = S->Dtmax12_eoo; st->snr_est = S->snr_est;
        if (S->consumed_inv > 0) {
            const float2 ph = S->bpf_grid ? ((const float2 *)a.bpf_chain)[(size_t)b * a.chain_stride + 1 + S->calls_inv] : S->bpf_phase;
            st->bpf.phase[0] = ph.x; st->bpf.phase[1] = ph.y; st->bpf.mem_len = 102; st->bpf.grid_off = S->bpf_grid ? 0 : 1;
        }
        st->consumed += S->consumed_round;
        rnd->n_calls = S->n_calls; rnd->n_rows = S->n_rows; rnd->uw_from_row = S->uw_from_row; rnd->consumed = S->consumed_round;
        rnd->out_base = S->out_base;
        a.acc[b * 4 + 0] = S->consumed_inv; a.acc[b * 4 + 1] = S->calls_inv; a.acc[b * 4 + 2] = S->valid_inv; a.acc[b * 4 + 3] = S->eoo_inv;
        a.status[b * 4 + 0] = S->nin; a.status[b * 4 + 1] = S->state == ST_SYNC; a.status[b * 4 + 2] = (int)S->snr_est; a.status[b * 4 + 3] = S->state;
        if (a.wg_cycles) a.wg_cycles[b] = clock64() - wg_t0;
        if (S->n_calls) atomicAdd(&a.progress[0], S->n_calls);
        if (S->calls_inv < a.max_calls && S->valid_inv < a.feat_cap && S->consumed_inv + S->nin <= avail) atomicAdd(&a.progress[1], 1);
    }
}


// start-of-utterance state of stream blockIdx.x in one launch: the receiver record (radae_rxe.py:128-142), the decoder's / encoder's GRU states and conv
// history rows -- whichever are given.  (Rounds 1-4 issued a kernel + two memsets + two 2-D memsets per reset: five launches of a batch's stream, each of which
// waits for a free slot beside the other batches' receiver workgroups.)
__global__ __launch_bounds__(256) void k_batch_reset(rd_reset_args a)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.st) {
        rd_rx_stream *s = a.st + b;
        float2 *raw = (float2 *)s;
        static_assert(sizeof(rd_rx_stream) % 8 == 0, "rd_rx_stream is cleared in 8-byte pieces");
        for (int i = tid; i < (int)(sizeof(rd_rx_stream) / 8); i += 256) raw[i] = make_float2(0.0f, 0.0f);
    }
    if (a.dec_h) for (int i = tid; i < 5 * 96; i += 256) a.dec_h[((size_t)(i / 96) * a.B + b) * 96 + i % 96] = 0.0f;
    if (a.dec_x) for (int i = tid; i < RD_DEC_W; i += 256) a.dec_x[(size_t)b * a.dec_x_sb + i] = 0.0f;
    if (a.enc_h) for (int i = tid; i < 5 * 64; i += 256) a.enc_h[((size_t)(i / 64) * a.B + b) * 64 + i % 64] = 0.0f;
    if (a.enc_x) for (int i = tid; i < 2 * RD_ENC_W; i += 256) a.enc_x[(size_t)b * a.enc_x_sb + i] = 0.0f;
    if (!a.st) return;
    __syncthreads();
    if (tid == 0) {
        rd_rx_stream *s = a.st + b;
        s->state = ST_SEARCH; s->nin = RD_NMF; s->mf = 1; s->bpf.mem_len = 100; s->lcg = a.seeds ? a.seeds[b] : 1u;
        s->rx_phase[0] = 1.0; s->rx_theta = 0.0; s->bpf.phase[0] = 1.0f; s->foff_err = a.foff_err;
    }
}
extern "C" int rd_launch_reset(const rd_reset_args *a, rd_stream_t s)
{
    if (a->B <= 0) return 0;
    hipLaunchKernelGGL(k_batch_reset, dim3(a->B), dim3(256), 0, (hipStream_t)s, *a);
    return (int)hipGetLastError();
}


__device__ __forceinline__ rd_bpf_state *bpf_state_of(const rd_bpf_args &a, int b) { return (rd_bpf_state *)((char *)a.state + (size_t)b * a.state_stride); }
__device__ __forceinline__ int bpf_len0_of(const rd_bpf_args &a, int b) { return a.len0 ? *(const int *)((const char *)a.len0 + (size_t)b * a.len0_stride) : a.len0_const; }
__device__ __forceinline__ int bpf_avail_of(const rd_bpf_args &a, int b) { return a.avail ? a.avail[b] : a.avail_const; }
// the block phases of an invocation, one thread per stream: P[0] = the phase the stream's last call left, P[k + 1] = P[k] E[len_k - 1] in complex64
// as complex_bpf does from call to call; also resets the stream's off-grid flag (a new invocation starts on the grid)
__global__ __launch_bounds__(64) void k_bpf_chain(rd_bpf_args a)
{
    __builtin_amdgcn_s_setprio(3);                         // a serial chain of ~100 steps on one thread per stream: latency, not throughput
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    if (a.zero_acc) {                                      // the receiver's per-invocation counters (rade_batch_rx)
        ((int4 *)a.zero_acc)[b] = make_int4(0, 0, 0, 0);
        if (b == 0) *(int4 *)a.zero_progress = make_int4(0, 0, 0, 0);
    }
    rd_bpf_state *s = bpf_state_of(a, b);
    float2 *c = (float2 *)a.chain + (size_t)b * a.chain_stride;
    const int nin0 = bpf_len0_of(a, b), av = bpf_avail_of(a, b);
    c[0] = make_float2(__int_as_float(nin0), __int_as_float(s->mem_len));
    s->grid_off = 0;
    float2 P = make_float2(s->phase[0], s->phase[1]);
    // the two step factors in registers: a load inside the loop waits for the store before it as well (one counter, in order), i.e. a full
    // memory round trip per block on a serial chain of ~100
    const float2 e0 = ld2(a.tab->bpf_E, min(max(nin0, 1), RD_NEOO) - 1), e1 = ld2(a.tab->bpf_E, RD_NMF - 1);
    int start = 0;
    for (int k = 0; k + 1 < a.chain_stride; k++) {
        c[1 + k] = P;
        if (start >= av) break;                            // P of the first block beyond the input: the phase after the last whole call
        P = cmul_nc(P, k ? e1 : e0);
        start += k ? RD_NMF : nin0;
    }
}
// baseband sample q >= 0 of an invocation: x[q] times the phase of its block
__device__ __forceinline__ float2 bpf_baseband(const float2 *x, const float2 *chain, const rd_tables *tab, int nin0, int q)
{
    const int k = q < nin0 ? 0 : 1 + (q - nin0) / RD_NMF, sk = k ? nin0 + (k - 1) * RD_NMF : 0;
    return cmul_nc(x[q], cmul_nc(chain[1 + k], ld2(tab->bpf_E, q - sk)));
}
// after a whole invocation was filtered and consumed (the transmit side: every sample is): the state complex_bpf would hold now (dsp.py:96-99)
__global__ __launch_bounds__(128) void k_bpf_advance(rd_bpf_args a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    rd_bpf_state *s = bpf_state_of(a, b);
    const float2 *c = (const float2 *)a.chain + (size_t)b * a.chain_stride;
    const int nin0 = __float_as_int(c[0].x), n = bpf_avail_of(a, b);
    if (n < 102 || nin0 <= 0) return;                    // (blocks are whole modem / end-of-over frames)
    const float2 *x = (const float2 *)a.x + (size_t)b * a.x_stride;
    if (tid < 102) { const float2 m = bpf_baseband(x, c, a.tab, nin0, n - 102 + tid); s->mem[tid][0] = m.x; s->mem[tid][1] = m.y; }
    if (tid == 0) { const int nb = n <= nin0 ? 1 : 1 + (n - nin0 + RD_NMF - 1) / RD_NMF; s->phase[0] = c[1 + nb].x; s->phase[1] = c[1 + nb].y; s->mem_len = 102; }
}

// complex_bpf.bpf for BPF_BPW consecutive blocks of stream blockIdx.y: the window [102 earlier baseband samples | the block mixed down] staged as binary16
// planes in LDS, one 256-output tile per wavefront on the matrix cores (bpf_fir_tile), mix up, store.  HBM-bound by design (8 bytes in, 8 out per
// sample), so the loop is built around the memory system: the NEXT block's samples are requested before this block is staged (the round trip hides under
// the staging, the matrix instructions and the stores), the block's last 102 baseband samples are handed to the next block through LDS (they are its
// filter memory: nothing is read twice), and the phase-table entries a thread needs are the same for every block (loaded once).
#define BPF_NT 256
#define BPF_BPW 8
__global__ __launch_bounds__(BPF_NT, 4) void k_bpf_fir(rd_bpf_args a)      // (at most 128 registers: two of its wavefronts fit on a SIMD beside a receiver wavefront)
{
    const rd_tables *tab = a.tab; const unsigned short *tab16 = a.bpf16;
    const float2 *rx = (const float2 *)a.x; const long rx_stride = a.x_stride; float2 *rxf = (float2 *)a.y; const long rxf_stride = a.y_stride;
    const float2 *chain = (const float2 *)a.chain; const int chain_stride = a.chain_stride;
    __shared__ BpfLds pl;
    __shared__ unsigned maxw;
    __shared__ float2 tailbuf[102];
    const int k0 = blockIdx.x * BPF_BPW, b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float2 *c = chain + (size_t)b * chain_stride;
    // first round trip: the block grid's header, the first block's phases (their addresses do not depend on the header), the taps, the phase-table entries
    const float2 hdr = c[0];
    float2 Pk = c[1 + k0];
    const float2 Pp = c[k0 ? k0 : 1];
    const int av = bpf_avail_of(a, b);
    BpfTaps taps; bpf_load_taps(taps, tab16, lane);
    const int i0 = 16 * (lane & 15) + 4 * (lane >> 4);    // this lane's four outputs inside a tile
    float2 eb[BPF_NQ], eo[4];
#pragma unroll
    for (int q = 0; q < BPF_NQ; q++) eb[q] = ld2(tab->bpf_E, min(tid + 256 * q, RD_NEOO - 1));
#pragma unroll
    for (int r = 0; r < 4; r++) eo[r] = ld2(tab->bpf_E, 256 * wave + i0 + r);
    const int nin0 = __float_as_int(hdr.x), ml0 = __float_as_int(hdr.y);
    if (nin0 <= 0) return;
    int sk = k0 ? nin0 + (k0 - 1) * RD_NMF : 0;
    if (sk >= av) return;
    const float2 *x = rx + (size_t)b * rx_stride;
    const rd_bpf_state *s = bpf_state_of(a, b);
    // second round trip: the first block's samples in one batch (clamped addresses, values dropped afterwards: no branch around a load), and its filter memory
    int n = min(k0 ? RD_NMF : nin0, av - sk);              // (a partial last block is filtered too; no call will consume it)
    float2 xb[BPF_NQ], head = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int q = 0; q < BPF_NQ; q++) xb[q] = x[sk + min(tid + 256 * q, n - 1)];
    if (tid < 102) {
        if (k0) { const int sp = k0 > 1 ? sk - RD_NMF : 0, q = sk - 102 + tid; head = cmul_nc(x[q], cmul_nc(Pp, ld2(tab->bpf_E, q - sp))); }
        else { const int mi = ml0 - 102 + tid; if (mi >= 0) head = make_float2(s->mem[mi][0], s->mem[mi][1]); }
    }
    for (int kk = 0; kk < BPF_BPW; kk++) {
        const int k = k0 + kk;
        const int o = (k == 0 && ml0 == 100) ? 2 : 0;      // before the first call the memory is two samples shorter (dsp.py:55)
        float2 body[BPF_NQ];
#pragma unroll
        for (int q = 0; q < BPF_NQ; q++) { body[q] = cmul_nc(xb[q], cmul_nc(Pk, eb[q])); if (tid + 256 * q >= n) body[q] = make_float2(0.0f, 0.0f); }
        // the next block's samples and phase: in flight from here on
        const int sk2 = sk + (k ? RD_NMF : nin0);
        const bool more = kk + 1 < BPF_BPW && sk2 < av;    // uniform
        const int n2 = more ? min(RD_NMF, av - sk2) : 1;
        const float2 Pn = c[1 + k + (more ? 1 : 0)];
        if (more) {
#pragma unroll
            for (int q = 0; q < BPF_NQ; q++) xb[q] = x[sk2 + min(tid + 256 * q, n2 - 1)];
        }
        const float unsc = bpf_stage_planes(&pl, &maxw, tid, head, body, o);
        // this block's last 102 baseband samples are the next block's filter memory
#pragma unroll
        for (int q = 0; q < BPF_NQ; q++) { const int ti = tid + 256 * q - (n - 102); if (ti >= 0 && ti < 102) tailbuf[ti] = body[q]; }
        float2 *dst = rxf + (size_t)b * rxf_stride + sk;
        for (int tile = wave; tile < BPF_TILES(n); tile += BPF_NT / 64) {
            f32x4 re, im;
            bpf_fir_tile(&pl, taps, tile, lane, re, im);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 256 * tile + i0 + r;
                const float2 e = tile == wave ? eo[r] : ld2(tab->bpf_E, min(i, RD_NEOO - 1));     // (a fifth tile only exists for blocks longer than 1024 samples)
                if (i < n) {
                    float2 y = cmul_nc(make_float2(re[r] * unsc, im[r] * unsc), cconj(cmul_nc(Pk, e)));
                    if (a.clip) {                           // np.clip(abs(tx), 0, 1) * exp(1j * angle(tx)) (radae_txe.py:132): the phase kept, the magnitude limited to 1
                        const float m2 = y.x * y.x + y.y * y.y;
                        if (m2 > 1.0f) { const float inv = 1.0f / sqrtf(m2); y.x *= inv; y.y *= inv; }
                    }
                    dst[i] = y;
                }
            }
        }
        if (!more) break;
        __syncthreads();
        if (tid < 102) head = tailbuf[tid];
        sk = sk2; n = n2; Pk = Pn;
    }
}
// the launches of a filtering pass: block phases, the FIR over every block, and (transmit side) the state the filter is left in
extern "C" int rd_launch_bpf(const rd_bpf_args *a, rd_stream_t s)
{
    if (a->B <= 0 || a->n_blocks <= 0) return 0;
    hipLaunchKernelGGL(k_bpf_chain, dim3((a->B + 63) / 64), dim3(64), 0, (hipStream_t)s, *a);
    hipLaunchKernelGGL(k_bpf_fir, dim3((a->n_blocks + BPF_BPW - 1) / BPF_BPW, a->B), dim3(BPF_NT), 0, (hipStream_t)s, *a);
    if (a->advance) hipLaunchKernelGGL(k_bpf_advance, dim3(a->B), dim3(128), 0, (hipStream_t)s, *a);
    return (int)hipGetLastError();
}

extern "C" int rd_launch_rx_sync(const rd_sync_args *a, rd_stream_t s)
{
    if (a->B <= 0) return 0;
    hipLaunchKernelGGL(k_rx_sync2, dim3(a->B), dim3(NT2), a->lds_bytes, (hipStream_t)s, *a);
    return (int)hipGetLastError();
}
// dynamic LDS of the receiver kernel (above the 64 KB default): set once per device by rade_batch_open, before any launch
extern "C" int rd_rx_sync_prepare(int solo)
{
    const int l = solo ? 100 * 1024 : (int)sizeof(RxShared2);          // solo (developer switch): more than half the LDS = one workgroup per CU
    if (hipFuncSetAttribute((const void *)k_rx_sync2, hipFuncAttributeMaxDynamicSharedMemorySize, l) != hipSuccess) return -1;
    return l;
}
