// rade_enc.hip -- the batched core encoder's feed-forward layers (CoreEncoderStatefull.forward, radae_base.py:260-286) on activations that live in HBM
// as the matrix cores' operand fragments.
//
// The DenseNet concat buffer x[b][t][864] is read by every later layer (7,600 column-reads per row and pass against 864 column-writes).  k_gemm16p
// (rade_kernels.hip) keeps it as float32 rows: every consumer then (i) reads 64 separate 16-byte pieces in 32 cache lines per wave-load (lane = row, rows
// 3456 bytes apart) and (ii) splits every element into its two binary16 planes again -- 40 vector instructions per k-block beside 6 matrix instructions.
// Here the PRODUCER does both once: an activation is stored as the two binary16 planes 2^8 x = hi + lo, in 32-row time tiles laid out
//     xf[b][tile][kb = col / 16][plane][k-half = (col % 16) / 8][row = t % 32][col % 8]         (2 KB per k-block and tile, 108 KB per tile)
// which is exactly what one v_mfma_f32_32x32x16_f16 operand load of a wavefront wants: 64 lanes x 16 bytes = 1 KB contiguous per plane, no conversion.
// Tile 0 of a stream is its history tile (rows 30, 31 = steps -2, -1: the conv taps before the first step of a call), tile 1 + t / 32 holds step t.
//
// The products are computed TRANSPOSED: the packed weights (rd_pack_weights_q16 / _f16x2: lane = output column, 8 k per lane -- the same bytes k_gemm16p
// streams as B operands) are the A operand, the activation fragments the B operand, so a lane ends up with 4 x 4 consecutive output channels of ONE row:
// 16-byte stores for float32 outputs (gi, z), 8-byte stores per plane into the fragment layout for the concat buffer.  Same products, same order of the
// partial products per k-block (lo x W, [hi x Wlo,] hi x W), same epilogue arithmetic as k_gemm16p: with the conv taps summed in its order ($RADE_ENCF_SEQ_TAPS)
// the results are bit-identical to it (tests/test_hip_parity.py::test_encoder_fragment_layout_*); the shipped order alternates the taps (see k_encf_gemm).
#include "rade_devutil.h"

#define EF_TILE RD_EF_TILE

// bias / scale / activation of 4 consecutive output channels of one row, and the store: float32 row-major, or the two binary16 planes at their fragment position
template <bool LIBM_TANH>
__device__ __forceinline__ void encf_emit(const rd_encf_args &a, int b, int t, int ch0, f32x4 v, f32x4 sc, f32x4 bs)
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float x = v[c] * sc[c] + bs[c];
        if (a.act == 1) x = clamp1(LIBM_TANH ? tanhf(x) : gate_tanh(x));
        v[c] = x;
    }
    if (a.yf) {
        const int col = a.ycol + ch0;                    // a multiple of 4; the lane's 4 channels are elements 4 (col / 4 % 2) .. + 3 of one 8-element k-half
        _Float16 *o = (_Float16 *)a.yf + ((size_t)b * a.NQ + 1 + (t >> 5)) * EF_TILE + (col >> 4) * 1024 + ((col >> 3) & 1) * 256 + (t & 31) * 8 + (col & 4);
        f16x4 hi, lo;
#pragma unroll
        for (int c = 0; c < 4; c++) { const float x = 256.0f * v[c]; const _Float16 h = (_Float16)x; hi[c] = h; lo[c] = (_Float16)(x - (float)h); }
        *(f16x4 *)o = hi; *(f16x4 *)(o + 512) = lo;
    } else {
        *(f32x4 *)(a.y + (size_t)b * a.y_sb + (size_t)t * a.y_st + ch0) = v;
    }
}

// One wavefront = RT time tiles of 32 rows of one stream x NT tiles of 32 output columns.  Operands in flight: XS k-blocks of activations and WS k-blocks of
// weights, XS a multiple of WS; a slot is refilled right behind the matrix instructions that read it (no register copies).  Shipped: XS = WS = 3 (124 registers).
// What bounds it, measured (profiles/r05_ab_enc_fragments.txt; tools/ubench/mfma_load_overlap.hip = this loop without the arithmetic around it): a k-block costs a
// wavefront 2 KB of activations + 3 KB of weights; with everything L2-resident the loop runs at the L2 -> L1 rate (32.7 TB/s over the chip: 0.36 us per k-block at two
// wavefronts per SIMD, matrix instructions hidden underneath), with the activations from the Infinity Cache / HBM at 6-7 TB/s of that stream (0.6 us).  So the launches
// read nothing twice from beyond L2 that need not be (TI and a.pair below: 0.526 -> 0.486 ms per pass), and what remains is the 5.5 GB a pass moves through the L1s
// (3.5 GB of it weights: a 32-row tile re-streams its layer's weights) + ~10 us per launch.  Measured and not kept: activations prefetched further ahead than weights
// (XS 4 / 6 / 8 over WS 2: 0.536 / 0.566 / 0.575 against 0.526 -- loads return in order, so a wait for a young weight fragment waits for every older activation load),
// 2 / 4 adjacent row tiles per workgroup (ENCF_WPB: L1 hits cost the same L1 cycles, -2 %), weights through LDS (tools/experiments/encf_gemm_lds_weights.inc: +1 % alone with
// the old tap order, -3 % in the pipeline: four-wavefront workgroups), a staggered start of the wavefronts, back-to-back against plane-by-plane matrix instructions (-0.2 %).
#ifndef ENCF_WPB
#define ENCF_WPB 1          /* wavefronts per workgroup = adjacent row tiles against the same column tiles (developer switch; nothing shared in the source) */
#endif
// TI: the two conv taps alternate per k-block (tap 0 of k-block j, tap 1 of k-block j, ...) instead of all of tap 0 and then all of tap 1: tap 1's fragments
// are tap 0's rows two (or one) further on -- the same cache lines, read again while they are still in L1 / L2 instead of 16..98 KB of streaming later (by then
// from the Infinity Cache or HBM again: the launch is bound by that stream).  The float32 sums are formed in another order: last-bit differences against the
// float32-row kernels, inside every parity bar; $RADE_ENCF_SEQ_TAPS keeps the sequential order (the bit-equality tests).
// a.pair: a product with two column groups (a GRU input projection, 192 columns) as ONE 1-D grid in which the two wavefronts of a row tile are blocks i and i + 8
// -- the same XCD (blocks are observed to go to XCD i % 8), dispatched back to back -- so the second one's activation reads hit that XCD's L2.
template <int NT, int RT, bool SINGLE, int XS, int WS, bool TI>
__global__ __launch_bounds__(64 * ENCF_WPB) void k_encf_gemm(rd_encf_args a)
{
    static_assert(XS % WS == 0, "activation slots are a multiple of the weight slots");
    const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
    const int tpq = (a.T + 31) >> 5;                     // time tiles with rows of this call
    const int tq = (tpq + RT - 1) / RT;
    int wid = blockIdx.x * ENCF_WPB + (threadIdx.x >> 6), cg = blockIdx.y;
    if (a.pair) { const int i = blockIdx.x; wid = ((i >> 4) << 3) | (i & 7); cg = (i >> 3) & 1; }
    if (wid >= a.B * tq) return;
    const int b = wid / tq, qt0 = (wid - b * tq) * RT;
    const int ntt = (a.N + 31) >> 5;
    const int nt0 = cg * NT;
    const _Float16 *sb = (const _Float16 *)a.xf + (size_t)b * a.NQ * EF_TILE;
    const _Float16 *p1[RT], *p0[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) {
        const int qt = min(qt0 + q, tpq - 1);
        p1[q] = sb + (size_t)(1 + qt) * EF_TILE + half * 256 + r * 8;
        const int rr = r - a.dil;                        // tap 0 = the row dil steps earlier: the same tile, or the last rows of the tile before (tile 0 = history)
        p0[q] = rr >= 0 ? sb + (size_t)(1 + qt) * EF_TILE + half * 256 + rr * 8 : sb + (size_t)qt * EF_TILE + half * 256 + (32 + rr) * 8;
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
    const _Float16 *wbase = (const _Float16 *)a.Wp16 + ((size_t)nt0 * planes * 64 + lane) * 8;
    const size_t wstep = (size_t)ntt * planes * 64 * 8;
    f16x8 xh[XS][RT], xl[XS][RT], wh[WS][NT], wl[WS][NT];
    auto fetch_x = [&](int st, int kb_) {
        const int kb = min(kb_, nkb - 1);
#pragma unroll
        for (int q = 0; q < RT; q++) {
            const _Float16 *p = TI ? ((kb & 1) ? p1[q] : p0[q]) + (size_t)(kb >> 1) * 1024 : (kb < nkb0 ? p0[q] + (size_t)kb * 1024 : p1[q] + (size_t)(kb - nkb0) * 1024);
            xh[st][q] = *(const f16x8 *)p; xl[st][q] = *(const f16x8 *)(p + 512);
        }
    };
    auto fetch_w = [&](int st, int kb_) {
        const int kq = min(kb_, nkb - 1);
        const int kb = TI ? ((kq & 1) ? nkb0 : 0) + (kq >> 1) : kq;       // the weight's K axis stays [tap 0 | tap 1]
#pragma unroll
        for (int i = 0; i < NT; i++) {
            wh[st][i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * planes * 64 * 8);
            if (!SINGLE) wl[st][i] = *(const f16x8 *)(wbase + kb * wstep + (size_t)i * 2 * 64 * 8 + 64 * 8);
        }
    };
    auto products = [&](int sx, int sw) {
        // plane by plane over the tiles: an accumulator's next instruction is NT x RT instructions away (back to back per accumulator measured the same: -0.2 %)
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sw][i], xl[sx][q], acc[q][i], 0, 0, 0);
        if (!SINGLE) {
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int i = 0; i < NT; i++) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[sw][i], xh[sx][q], acc[q][i], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int i = 0; i < NT; i++) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sw][i], xh[sx][q], acc[q][i], 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < XS; s++) fetch_x(s, s);
#pragma unroll
    for (int s = 0; s < WS; s++) fetch_w(s, s);
    int kb = 0;
#pragma unroll 1
    for (; kb + XS <= nkb; kb += XS) {
#pragma unroll
        for (int s = 0; s < XS; s++) {
            __builtin_amdgcn_sched_barrier(0);
            products(s, s % WS);
            __builtin_amdgcn_sched_barrier(0);
            fetch_x(s, kb + XS + s);
            fetch_w(s % WS, kb + WS + s);
        }
    }
    {   // the last nkb % XS k-blocks: their activations are in flight already, the weights keep cycling through their slots
        const int rem = nkb - kb;
#pragma unroll
        for (int s = 0; s < XS - 1; s++)
            if (s < rem) {
                __builtin_amdgcn_sched_barrier(0);
                products(s, s % WS);
                __builtin_amdgcn_sched_barrier(0);
                if (s + WS < XS - 1) fetch_w(s % WS, kb + WS + s);
            }
    }
    // epilogue: lane (row r, k-half) holds channels 32 i + 8 g + 4 half + c of its row in acc[.][i][4 g + c]
#pragma unroll
    for (int q = 0; q < RT; q++) {
        const int qt = qt0 + q, t = 32 * qt + r;
        if (qt >= tpq || t >= a.T) continue;
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int ch0 = 32 * (nt0 + i) + 8 * g + 4 * half;
                if (ch0 >= a.N) continue;
                const f32x4 bs = a.bias ? *(const f32x4 *)(a.bias + ch0) : (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
                f32x4 sc = { 0x1p-18f, 0x1p-18f, 0x1p-18f, 0x1p-18f };        // two planes of 2^10 w x rows of 2^8 x
                if (SINGLE) { sc = *(const f32x4 *)(a.Wscale + ch0); sc *= 0x1p-8f; }     // integers x column scale, rows carry 2^8
                const f32x4 v = { acc[q][i][4 * g], acc[q][i][4 * g + 1], acc[q][i][4 * g + 2], acc[q][i][4 * g + 3] };
                encf_emit<false>(a, b, t, ch0, v, sc, bs);
            }
    }
}

// conv_l and the product that consumes its output (the next GRU's input projection, or z_dense behind conv_5) in ONE launch: a workgroup = one 32-row time tile,
// wavefront w < 3 owns conv columns 32 w .. 32 w + 31 and column tile w of the second product, a fourth wavefront (NB = 3) the second product's tiles 3 .. 5
// (GRU input: 192 columns).  All wavefronts walk the tile's k-blocks together, so the tile is fetched from HBM once (the others hit in L1 / L2) instead of four
// times by three launches; the conv output goes to HBM for the later layers and, as operand fragments through 12 KB of LDS, into the second product's last 96
// k-columns.  The two conv taps are accumulated alternately (tap 0, tap 1 per k-block), not one after the other as the separate kernels do: the sums differ
// from theirs in the last bits (float32 addition order), inside every parity bar.
template <bool GS, int NB, int ST>
__global__ __launch_bounds__(NB ? 256 : 192) void k_encf_fused(rd_encf_fused_args a)
{
    __shared__ __attribute__((aligned(16))) _Float16 cx[6 * 1024];          // the tile's conv output: [k-block 6][plane][k-half][row][8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
    const int tpq = (a.T + 31) >> 5;
    const int b = blockIdx.x / tpq, qt = blockIdx.x - b * tpq;
    const int t = 32 * qt + r;
    _Float16 *sb = (_Float16 *)a.xf + (size_t)b * a.NQ * EF_TILE;
    const _Float16 *p1 = sb + (size_t)(1 + qt) * EF_TILE + half * 256 + r * 8;
    const int rr = r - a.dil;
    const _Float16 *p0 = rr >= 0 ? sb + (size_t)(1 + qt) * EF_TILE + half * 256 + rr * 8 : sb + (size_t)qt * EF_TILE + half * 256 + (32 + rr) * 8;
    const int nkb = a.cin >> 4;
    constexpr int gp = GS ? 1 : 2;
    const int gtt = (a.Ng + 31) >> 5;
    const size_t wcstep = (size_t)3 * 64 * 8, wgstep = (size_t)gtt * gp * 64 * 8;
    f32x16 accg[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) accg[i][j] = 0.0f;
    if (wave < 3) {
        f32x16 accc;
#pragma unroll
        for (int j = 0; j < 16; j++) accc[j] = 0.0f;
        const _Float16 *wc = (const _Float16 *)a.Wc + ((size_t)wave * 64 + lane) * 8;
        const _Float16 *wg = (const _Float16 *)a.Wg + ((size_t)wave * gp * 64 + lane) * 8;
        f16x8 x0h[ST], x0l[ST], x1h[ST], x1l[ST], c0[ST], c1[ST], gh[ST], gl[ST];
        auto fetch = [&](int st, int kb_) {
            const int kb = min(kb_, nkb - 1);
            x0h[st] = *(const f16x8 *)(p0 + (size_t)kb * 1024); x0l[st] = *(const f16x8 *)(p0 + (size_t)kb * 1024 + 512);
            x1h[st] = *(const f16x8 *)(p1 + (size_t)kb * 1024); x1l[st] = *(const f16x8 *)(p1 + (size_t)kb * 1024 + 512);
            c0[st] = *(const f16x8 *)(wc + kb * wcstep); c1[st] = *(const f16x8 *)(wc + (nkb + kb) * wcstep);
            gh[st] = *(const f16x8 *)(wg + kb * wgstep);
            if (!GS) gl[st] = *(const f16x8 *)(wg + kb * wgstep + 64 * 8);
        };
        auto products = [&](int st) {
            accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c0[st], x0l[st], accc, 0, 0, 0);
            accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c0[st], x0h[st], accc, 0, 0, 0);
            accg[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[st], x1l[st], accg[0], 0, 0, 0);
            accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1[st], x1l[st], accc, 0, 0, 0);
            if (!GS) accg[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[st], x1h[st], accg[0], 0, 0, 0);
            accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1[st], x1h[st], accc, 0, 0, 0);
            accg[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[st], x1h[st], accg[0], 0, 0, 0);
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
        {
            const int rem = nkb - kb;
#pragma unroll
            for (int s = 0; s < ST - 1; s++) if (s < rem) products(s);
        }
        // conv epilogue: bias, tanh, clamp; the two planes into LDS (for everybody's last k-blocks) and into the concat buffer
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int ch0 = 32 * wave + 8 * g + 4 * half;
            const f32x4 bs = *(const f32x4 *)(a.Wc_bias + ch0);
            f32x4 sc = *(const f32x4 *)(a.Wc_scale + ch0); sc *= 0x1p-8f;
            f16x4 hi, lo;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float y = clamp1(gate_tanh(accc[4 * g + c] * sc[c] + bs[c]));
                const float x = 256.0f * y; const _Float16 h = (_Float16)x; hi[c] = h; lo[c] = (_Float16)(x - (float)h);
            }
            const int fo = (2 * wave + (g >> 1)) * 1024 + (g & 1) * 256 + r * 8 + 4 * half;
            *(f16x4 *)(cx + fo) = hi; *(f16x4 *)(cx + fo + 512) = lo;
            if (t < a.T) {
                _Float16 *o = sb + (size_t)(1 + qt) * EF_TILE + (size_t)nkb * 1024 + fo;       // column cin + ch0: k-block cin / 16 + 2 wave + g / 2
                *(f16x4 *)o = hi; *(f16x4 *)(o + 512) = lo;
            }
        }
    } else if (NB) {
        const _Float16 *wg = (const _Float16 *)a.Wg + ((size_t)3 * gp * 64 + lane) * 8;
        f16x8 x1h[ST], x1l[ST], gh[ST][3], gl[ST][3];
        auto fetch = [&](int st, int kb_) {
            const int kb = min(kb_, nkb - 1);
            x1h[st] = *(const f16x8 *)(p1 + (size_t)kb * 1024); x1l[st] = *(const f16x8 *)(p1 + (size_t)kb * 1024 + 512);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                gh[st][i] = *(const f16x8 *)(wg + kb * wgstep + (size_t)i * gp * 64 * 8);
                if (!GS) gl[st][i] = *(const f16x8 *)(wg + kb * wgstep + (size_t)i * gp * 64 * 8 + 64 * 8);
            }
        };
        auto products = [&](int st) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[st][i], x1l[st], accg[i], 0, 0, 0);
                if (!GS) accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[st][i], x1h[st], accg[i], 0, 0, 0);
                accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[st][i], x1h[st], accg[i], 0, 0, 0);
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
        {
            const int rem = nkb - kb;
#pragma unroll
            for (int s = 0; s < ST - 1; s++) if (s < rem) products(s);
        }
    }
    __syncthreads();
    // the second product's last 96 k-columns = the conv output, from LDS
    const int nmine = wave < 3 ? 1 : NB, tile0 = wave < 3 ? wave : 3;
    const _Float16 *wgt = (const _Float16 *)a.Wg + (size_t)nkb * wgstep + ((size_t)tile0 * gp * 64 + lane) * 8;
#pragma unroll
    for (int k2 = 0; k2 < 6; k2++) {
        const f16x8 xh = *(const f16x8 *)(cx + k2 * 1024 + half * 256 + r * 8), xl = *(const f16x8 *)(cx + k2 * 1024 + 512 + half * 256 + r * 8);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i >= nmine) break;
            const f16x8 wh = *(const f16x8 *)(wgt + k2 * wgstep + (size_t)i * gp * 64 * 8);
            accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accg[i], 0, 0, 0);
            if (!GS) { const f16x8 wl = *(const f16x8 *)(wgt + k2 * wgstep + (size_t)i * gp * 64 * 8 + 64 * 8); accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accg[i], 0, 0, 0); }
            accg[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accg[i], 0, 0, 0);
        }
    }
    if (t >= a.T) return;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i >= nmine) break;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int ch0 = 32 * (tile0 + i) + 8 * g + 4 * half;
            if (ch0 >= a.Ng) continue;
            const f32x4 bs = a.Wg_bias ? *(const f32x4 *)(a.Wg_bias + ch0) : (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
            f32x4 sc = { 0x1p-18f, 0x1p-18f, 0x1p-18f, 0x1p-18f };
            if (GS) { sc = *(const f32x4 *)(a.Wg_scale + ch0); sc *= 0x1p-8f; }
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; c++) { float x = accg[i][4 * g + c] * sc[c] + bs[c]; if (a.g_act == 1) x = clamp1(gate_tanh(x)); v[c] = x; }
            *(f32x4 *)(a.y + (size_t)b * a.y_sb + (size_t)t * a.y_st + ch0) = v;
        }
    }
}

extern "C" int rd_launch_encf_fused(const rd_encf_fused_args *a, rd_stream_t s)
{
    if (a->B <= 0 || a->T <= 0) return 0;
    if ((a->cin & 31) || a->dil < 1 || a->dil > 2 || (a->Ng & 3) || !a->Wc_scale) return -1;
    const dim3 grid(a->B * ((a->T + 31) >> 5));
    hipStream_t st = (hipStream_t)s;
    static int deep = -1; if (deep < 0) deep = getenv("RADE_ENCF_FST") ? atoi(getenv("RADE_ENCF_FST")) : 3;       // developer switch
    if (a->Ng == 192 && a->Wg_scale) { if (deep == 2) hipLaunchKernelGGL((k_encf_fused<true, 3, 2>), grid, dim3(256), 0, st, *a); else hipLaunchKernelGGL((k_encf_fused<true, 3, 3>), grid, dim3(256), 0, st, *a); }
    else if (a->Ng <= 96 && !a->Wg_scale) { if (deep == 2) hipLaunchKernelGGL((k_encf_fused<false, 0, 2>), grid, dim3(192), 0, st, *a); else hipLaunchKernelGGL((k_encf_fused<false, 0, 3>), grid, dim3(192), 0, st, *a); }
    else return -1;
    return (int)hipGetLastError();
}

// dense_1 reads raw features -- the one encoder operand that is not tanh-bounded -- so it stays on v_mfma_f32_32x32x2_f32 (as k_gemm<2> has it), transposed like
// the kernel above; xin [B][T][Kin] float32 rows, 64 outputs into columns 0..63 of the fragment buffer
__global__ __launch_bounds__(64) void k_encf_dense1(rd_encf_args a)
{
    constexpr int NT = 2;
    const int lane = threadIdx.x, r = lane & 31, half = lane >> 5;
    const int tpq = (a.T + 31) >> 5;
    const int b = blockIdx.x / tpq, qt = blockIdx.x - b * tpq;
    const int t = 32 * qt + r;
    const float *p = a.xin + ((size_t)b * a.T + min(t, a.T - 1)) * a.Kin + 4 * half;
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    const float *wp = a.Wp + (size_t)lane * 4;
    const size_t wstep = (size_t)NT * 256;
    const int nkb = a.Kin >> 3;
    f32x4 av = *(const f32x4 *)p;
    f32x4 bv[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) bv[i] = *(const f32x4 *)(wp + i * 256);
    for (int kb = 0; kb < nkb; kb++) {
        f32x4 an = av; f32x4 bn[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) bn[i] = bv[i];
        if (kb + 1 < nkb) {
            an = *(const f32x4 *)(p + (kb + 1) * 8);
#pragma unroll
            for (int i = 0; i < NT; i++) bn[i] = *(const f32x4 *)(wp + wstep + i * 256);
        }
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < NT; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[i][s], av[s], acc[i], 0, 0, 0);
        av = an;
#pragma unroll
        for (int i = 0; i < NT; i++) bv[i] = bn[i];
        wp += wstep;
    }
    if (t >= a.T) return;
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int ch0 = 32 * i + 8 * g + 4 * half;
            const f32x4 bs = *(const f32x4 *)(a.bias + ch0);
            const f32x4 v = { acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3] };
            encf_emit<true>(a, b, t, ch0, v, (f32x4){ 1.0f, 1.0f, 1.0f, 1.0f }, bs);
        }
}

// The conv history between calls stays where the row-layout path keeps it (the two float32 history rows of enc_x: what k_batch_reset zeroes and what a call with
// few rows reads), so both paths can follow each other on one engine.  dir 0, before a pass: those rows -> rows 30, 31 of the history tile.  dir 1, after a pass
// over T steps: steps T - 2, T - 1 -> the history tile and the float32 rows (2^-8 (hi + lo): exact, and splitting it again gives hi, lo back).
__global__ __launch_bounds__(256) void k_encf_hist(unsigned short *xf_, int NQ, float *x32, long x32_sb, int T, int dir)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, tid = threadIdx.x;
    _Float16 *sb = (_Float16 *)xf_ + (size_t)b * NQ * EF_TILE;
    float *xr = x32 + (size_t)b * x32_sb;
    static_assert(RD_ENC_W % 16 == 0 && 2 * (RD_ENC_W / 8) <= 256, "two history rows in 8-column pieces, one piece per thread");
    constexpr int PW = RD_ENC_W / 8;                     // 108 pieces per row
    const int k = tid / PW, c8 = tid - k * PW;           // history row k = 0, 1 (step -2, -1), columns 8 c8 .. 8 c8 + 7
    const bool on = tid < 2 * PW;
    const int col = 8 * c8;
    const size_t fo = (size_t)(col >> 4) * 1024 + ((col >> 3) & 1) * 256;
    f16x8 hi = {}, lo = {};
    if (on) {
        if (dir == 0) {
            const f32x4 u0 = *(const f32x4 *)(xr + (size_t)k * RD_ENC_W + col), u1 = *(const f32x4 *)(xr + (size_t)k * RD_ENC_W + col + 4);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float x = 256.0f * (e < 4 ? u0[e & 3] : u1[e & 3]); const _Float16 h = (_Float16)x; hi[e] = h; lo[e] = (_Float16)(x - (float)h); }
        } else {
            const int ts = T - 2 + k;                     // source step; negative = still in the history tile (a call of one step)
            const _Float16 *s = ts >= 0 ? sb + (size_t)(1 + (ts >> 5)) * EF_TILE + fo + (ts & 31) * 8 : sb + fo + (32 + ts) * 8;
            hi = *(const f16x8 *)s; lo = *(const f16x8 *)(s + 512);
        }
    }
    __syncthreads();
    if (on) {
        _Float16 *d = sb + fo + (30 + k) * 8;
        *(f16x8 *)d = hi; *(f16x8 *)(d + 512) = lo;
        if (dir == 1) {
            f32x4 u0, u1;
#pragma unroll
            for (int e = 0; e < 8; e++) { const float x = ((float)hi[e] + (float)lo[e]) * 0x1p-8f; if (e < 4) u0[e & 3] = x; else u1[e & 3] = x; }
            *(f32x4 *)(xr + (size_t)k * RD_ENC_W + col) = u0; *(f32x4 *)(xr + (size_t)k * RD_ENC_W + col + 4) = u1;
        }
    }
}

extern "C" int rd_launch_encf_hist(unsigned short *xf, int NQ, float *x32, long x32_sb, int B, int T, int dir, rd_stream_t s)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_encf_hist, dim3(B), dim3(256), 0, (hipStream_t)s, xf, NQ, x32, x32_sb, T, dir);
    return (int)hipGetLastError();
}

extern "C" int rd_launch_encf_dense1(const rd_encf_args *a, rd_stream_t s)
{
    if (a->B <= 0 || a->T <= 0) return 0;
    if (a->N != 64 || (a->Kin & 7) || !a->yf) return -1;
    hipLaunchKernelGGL(k_encf_dense1, dim3(a->B * ((a->T + 31) >> 5)), dim3(64), 0, (hipStream_t)s, *a);
    return (int)hipGetLastError();
}

extern "C" int rd_launch_encf_gemm(const rd_encf_args *a, rd_stream_t s)
{
    if (a->B <= 0 || a->T <= 0) return 0;
    const int ntt = (a->N + 31) >> 5, tpq = (a->T + 31) >> 5;
    if ((a->K0 & 15) || (a->K1 & 15) || ntt % 3 || (a->N & 3) || a->dil < 0 || a->dil > 2) return -1;
    hipStream_t st = (hipStream_t)s;
    static int xs = -1;
    if (xs < 0) xs = getenv("RADE_ENCF_XS") ? atoi(getenv("RADE_ENCF_XS")) : 3;               // developer switch (A/B builds)
    const int seq = a->seq_taps, nopair = a->no_pair;                                          // per-engine switches (rade_engine.c: $RADE_ENCF_SEQ_TAPS, $RADE_ENCF_NO_PAIR)
    rd_encf_args b = *a;
    const int ntile = a->B * tpq;
    b.pair = (ntt == 6 && !nopair && ENCF_WPB == 1) ? 1 : 0;
    const dim3 g = b.pair ? dim3(((ntile + 7) / 8) * 16, 1) : dim3((ntile + ENCF_WPB - 1) / ENCF_WPB, ntt / 3), blk(64 * ENCF_WPB);
    const bool ti = a->K0 > 0 && a->K0 == a->K1 && !seq;
    static int rt2 = -1; if (rt2 < 0) rt2 = getenv("RADE_ENCF_RT2") ? 1 : 0;                  // developer switch: two row tiles per wavefront (half the weight bytes per product)
    if (a->Wscale && rt2 && ENCF_WPB == 1) {
        const int nt2 = a->B * ((tpq + 1) / 2);
        const dim3 g2 = b.pair ? dim3(((nt2 + 7) / 8) * 16, 1) : dim3(nt2, ntt / 3);
        if (ti) hipLaunchKernelGGL((k_encf_gemm<3, 2, true, 3, 3, true>), g2, blk, 0, st, b);
        else hipLaunchKernelGGL((k_encf_gemm<3, 2, true, 3, 3, false>), g2, blk, 0, st, b);
        return (int)hipGetLastError();
    }
    if (a->Wscale) {
        if (ti) hipLaunchKernelGGL((k_encf_gemm<3, 1, true, 3, 3, true>), g, blk, 0, st, b);
        else if (xs == 2) hipLaunchKernelGGL((k_encf_gemm<3, 1, true, 2, 2, false>), g, blk, 0, st, b);
        else if (xs == 4) hipLaunchKernelGGL((k_encf_gemm<3, 1, true, 4, 2, false>), g, blk, 0, st, b);
        else hipLaunchKernelGGL((k_encf_gemm<3, 1, true, 3, 3, false>), g, blk, 0, st, b);
    } else {
        hipLaunchKernelGGL((k_encf_gemm<3, 1, false, 2, 2, false>), g, blk, 0, st, b);
    }
    return (int)hipGetLastError();
}
