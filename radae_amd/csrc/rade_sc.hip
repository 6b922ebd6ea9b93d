// Batched single-carrier BPSK modem for BBFM symbols on gfx950.
//
// Reference behaviour: /root/reference/radae/dsp.py:532-562 (gen_rn_coeffs) and :579-860 (class single_carrier), driven by
// sc_tx.py:58-75 / sc_rx.py:83-112.  One workgroup per stream; the receiver runs every frame the available samples
// allow inside one launch (filter -> fine timing -> decimate -> phase tracker -> frame sync), all state in HBM between
// launches.  Arithmetic follows what NumPy 2 gives the reference: complex64 buffers, complex128 wherever the reference
// mixes in float64 scalars (filter dot products, timing, phase tracker).
//
//   k_sc_tx   grid (frames, streams) x 384: zero-stuffed symbols -> 24-tap root-Nyquist FIR -> LO
//   k_sc_rx   grid (streams) x 256: per frame  LO^-1 -> FIR -> envelope line at Rs -> linear-interpolated symbols ->
//             squared-symbol phase window (21) with pi-jump tracking -> 16-symbol sync-word correlation / error count
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rade_batch.h"

#define SC_NSYNC 16
#define SC_NFRAME 96
#define SC_NPAY 80
#define SC_NTAP 24
#define SC_M 4
#define SC_NPHASE 21
#define SC_SP 5                      /* nominal sample point, dsp.py:611 */
#define SC_NOUT ((SC_NFRAME + 2) * SC_M)
#define SC_NT 256
#define SC_PI 3.14159265358979323846

struct sc_stream {                   // receiver + transmitter state of one stream (dsp.py:605-627)
    float2 tx_mem[SC_NTAP], rx_mem[SC_NTAP];
    float2 rx_filt_out[SC_NOUT];
    float2 rx_symb_buf[2 * SC_NFRAME];
    double2 phase_mem[SC_NPHASE];
    double2 tx_lo, rx_lo;
    double phase_fine, phase_coarse, phase_ambiguity, g;
    int state, fs_s, bad_fs, nin;
    float max_cs_re, max_cs_im; float norm_rx_timing, pad;
};

struct rade_sc {
    int B, device;
    double omega;                    // 2 pi fcentre / Fs
    double rrc[SC_NTAP];
    double *d_rrc;
    sc_stream *st;
    rade_sc_status *d_status;
};

__constant__ float c_sync[SC_NSYNC] = { 1, 1, 1, 1, 1, -1, 1, 1, -1, -1, 1, 1, -1, -1, -1, -1 };   // dsp.py:592-593

// ---------------------------------------------------------------------------------------------------------------------
// transmitter (dsp.py:636-662).  Frame f of a call filters [tail of frame f-1's zero-stuffed input | its own]; the tail of
// the previous frame is a function of its last symbols, so frames are independent work items.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sc_symbol(const float *symbs, int m) { return m < SC_NSYNC ? c_sync[m] : symbs[m - SC_NSYNC]; }

__global__ __launch_bounds__(SC_NFRAME * SC_M) void k_sc_tx(sc_stream *st, const double *rrc, double omega, const float *symbs, int n_frames, float2 *out, long out_stride)
{
    __shared__ float2 fin[SC_NTAP + SC_NFRAME * SC_M + 1];
    const int f = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
    const float *sy = symbs + ((size_t)b * n_frames + f) * SC_NPAY;
    if (i < SC_NTAP) {
        float2 v = make_float2(0.0f, 0.0f);
        if (f == 0) v = st[b].tx_mem[i];
        else if ((i & 3) == 0) v = make_float2(SC_M * sc_symbol(sy - SC_NPAY, SC_NFRAME - SC_NTAP / SC_M + (i >> 2)), 0.0f);
        fin[i] = v;
    }
    fin[SC_NTAP + i] = (i & 3) == 0 ? make_float2(SC_M * sc_symbol(sy, i >> 2), 0.0f) : make_float2(0.0f, 0.0f);
    __syncthreads();
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int k = 0; k < SC_NTAP; k++) { const float2 v = fin[i + 1 + k]; ar += (double)v.x * rrc[k]; ai += (double)v.y * rrc[k]; }
    const float2 y = make_float2((float)ar, (float)ai);              // tx_filt_out is complex64
    const double2 lo0 = st[b].tx_lo;
    double sn, cs; sincos(omega * (double)(f * SC_NFRAME * SC_M + i), &sn, &cs);
    const double lr = lo0.x * cs - lo0.y * sn, li = lo0.x * sn + lo0.y * cs;
    out[(size_t)b * out_stride + (size_t)f * SC_NFRAME * SC_M + i] = make_float2((float)((double)y.x * lr - (double)y.y * li), (float)((double)y.x * li + (double)y.y * lr));
}

__global__ void k_sc_tx_finish(sc_stream *st, double omega, const float *symbs, int n_frames)
{   // filter memory = tail of the last frame's stuffed input; LO phasor advanced and renormalised (dsp.py:649, :652-657)
    const int b = blockIdx.x, i = threadIdx.x;
    const float *sy = symbs + ((size_t)b * n_frames + n_frames - 1) * SC_NPAY;
    if (i < SC_NTAP) st[b].tx_mem[i] = (i & 3) == 0 ? make_float2(SC_M * sc_symbol(sy, SC_NFRAME - SC_NTAP / SC_M + (i >> 2)), 0.0f) : make_float2(0.0f, 0.0f);
    if (i == 0) {
        const double2 lo0 = st[b].tx_lo;
        double sn, cs; sincos(omega * (double)n_frames * SC_NFRAME * SC_M, &sn, &cs);
        double lr = lo0.x * cs - lo0.y * sn, li = lo0.x * sn + lo0.y * cs;
        const double a = hypot(lr, li);
        st[b].tx_lo = make_double2(lr / a, li / a);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// receiver (dsp.py:747-860)
// ---------------------------------------------------------------------------------------------------------------------
struct ScShared {
    float2 fin[SC_NTAP + SC_NFRAME * SC_M + 2];
    float2 rf[SC_NOUT], rf2[SC_NOUT];
    double2 buf[SC_NPHASE + SC_NFRAME];          // phase window memory | this frame's symbols
    double fine[SC_NFRAME], phase[SC_NFRAME];
    float2 sb[2 * SC_NFRAME];
    double red[4][SC_NT / 64];
    float cabs[SC_NFRAME]; float2 cs[SC_NFRAME];
    unsigned long long bal[2][2]; float wbest[2]; int wbests[2];
    int nin, state, fs_s, bad_fs, go;
    double2 rx_lo; double phase_fine, phase_coarse, phase_amb, g; float2 max_cs; float norm;
};

__global__ __launch_bounds__(SC_NT) void k_sc_rx(sc_stream *stg, const double *rrc, double omega, const float2 *rx, long rx_stride, int n_avail, int max_frames,
                                                 float2 *payload, float *zhat, rade_sc_frame *frames, rade_sc_status *status)
{
    __shared__ ScShared sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sc_stream *S = &stg[b];
    for (int i = tid; i < SC_NOUT; i += SC_NT) sh.rf[i] = S->rx_filt_out[i];
    for (int i = tid; i < 2 * SC_NFRAME; i += SC_NT) sh.sb[i] = S->rx_symb_buf[i];
    if (tid < SC_NTAP) sh.fin[tid] = S->rx_mem[tid];
    if (tid < SC_NPHASE) sh.buf[tid] = S->phase_mem[tid];
    if (tid == 0) {
        sh.nin = S->nin; sh.state = S->state; sh.fs_s = S->fs_s; sh.bad_fs = S->bad_fs; sh.rx_lo = S->rx_lo; sh.phase_fine = S->phase_fine;
        sh.phase_coarse = S->phase_coarse; sh.phase_amb = S->phase_ambiguity; sh.g = S->g; sh.max_cs = make_float2(S->max_cs_re, S->max_cs_im); sh.norm = S->norm_rx_timing;
    }
    __syncthreads();
    const float2 *x = rx + (size_t)b * rx_stride;
    int pos = 0, nf = 0;
    while (true) {
        const int nin = sh.nin;
        if (pos + nin > n_avail || nf >= max_frames) break;
        // ---- LO down-mix (:753-758) into the filter input behind its 24-sample memory
        const double2 lo0 = sh.rx_lo;
        for (int i = tid; i < nin; i += SC_NT) {
            double sn, cs; sincos(-omega * (double)i, &sn, &cs);
            const double lr = lo0.x * cs - lo0.y * sn, li = lo0.x * sn + lo0.y * cs;
            const float2 v = x[pos + i];
            sh.fin[SC_NTAP + i] = make_float2((float)((double)v.x * lr - (double)v.y * li), (float)((double)v.x * li + (double)v.y * lr));
        }
        __syncthreads();
        // ---- root-Nyquist FIR (:761-766): keep the newest (98 M - nin) outputs, append nin new ones
        const int keep = SC_NOUT - nin;
        for (int i = tid; i < SC_NOUT; i += SC_NT) {
            float2 y;
            if (i < keep) y = sh.rf[i + nin];
            else {
                const int j = i - keep; double ar = 0.0, ai = 0.0;
#pragma unroll
                for (int k = 0; k < SC_NTAP; k++) { const float2 v = sh.fin[j + 1 + k]; ar += (double)v.x * rrc[k]; ai += (double)v.y * rrc[k]; }
                y = make_float2((float)ar, (float)ai);
            }
            sh.rf2[i] = y;
        }
        __syncthreads();
        for (int i = tid; i < SC_NOUT; i += SC_NT) sh.rf[i] = sh.rf2[i];
        float2 memv = make_float2(0.0f, 0.0f);
        if (tid < SC_NTAP) memv = sh.fin[nin + tid];
        __syncthreads();
        if (tid < SC_NTAP) sh.fin[tid] = memv;
        // ---- fine timing (:668-704): x = sum_n |rf[5+n]| e^{-j 2 pi n / 4}
        double xr = 0.0, xi = 0.0;
        for (int n = tid; n < SC_NOUT - SC_SP; n += SC_NT) {
            const float2 v = sh.rf[SC_SP + n];
            const double e = (double)hypotf(v.x, v.y);
            const int q = n & 3;
            if (q == 0) xr += e; else if (q == 1) xi -= e; else if (q == 2) xr -= e; else xi += e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { xr += __shfl_xor(xr, off); xi += __shfl_xor(xi, off); }
        if (lane == 0) { sh.red[0][wave] = xr; sh.red[1][wave] = xi; }
        __syncthreads();
        xr = 0.0; xi = 0.0;
        for (int w = 0; w < SC_NT / 64; w++) { xr += sh.red[0][w]; xi += sh.red[1][w]; }
        const double norm = atan2(xi, xr) / (2.0 * SC_PI);
        const double corr = -norm * SC_M;
        const int low = (int)floor(corr); const double fract = corr - (double)low;
        if (tid < SC_NFRAME) {                           // linear interpolation at the estimated instant, complex128
            const int s0 = SC_SP + low + SC_M * tid;
            const float2 a = sh.rf[s0], c = sh.rf[s0 + 1];
            sh.buf[SC_NPHASE + tid] = make_double2((double)a.x * (1.0 - fract) + (double)c.x * fract, (double)a.y * (1.0 - fract) + (double)c.y * fract);
        }
        __syncthreads();
        // ---- phase (:707-742): angle of the 21-symbol sum of squared symbols, halved; pi jumps tracked by one thread
        if (tid < SC_NFRAME) {
            double ar = 0.0, ai = 0.0;
            for (int k = 1; k <= SC_NPHASE; k++) { const double2 v = sh.buf[tid + k]; ar += v.x * v.x - v.y * v.y; ai += 2.0 * v.x * v.y; }
            sh.fine[tid] = atan2(ai, ar) / 2.0;
        }
        __syncthreads();
        // pi jumps (:722-726) as a prefix count over the frame: +1 where the fine estimate drops by more than 0.9 pi, -1 where it
        // rises by as much; ballots give every symbol the number of jumps up to and including itself
        {
            bool jp = false, jm = false;
            if (tid < SC_NFRAME) {
                const double d = sh.fine[tid] - (tid ? sh.fine[tid - 1] : sh.phase_fine);
                jp = d < -0.9 * SC_PI; jm = d > 0.9 * SC_PI;
            }
            const unsigned long long bp = __ballot(jp), bm = __ballot(jm);
            if (lane == 0 && wave < 2) { sh.bal[wave][0] = bp; sh.bal[wave][1] = bm; }
            __syncthreads();
            if (tid < SC_NFRAME) {
                const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
                int c = __popcll(sh.bal[wave][0] & le) - __popcll(sh.bal[wave][1] & le);
                if (wave == 1) c += __popcll(sh.bal[0][0]) - __popcll(sh.bal[0][1]);
                const double pc = sh.phase_coarse + (double)c * SC_PI;
                sh.phase[tid] = pc + sh.fine[tid];
                if (tid == SC_NFRAME - 1) { sh.red[2][0] = pc; sh.red[3][0] = sh.fine[tid]; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            sh.phase_coarse = sh.red[2][0]; sh.phase_fine = sh.red[3][0];
            int nn = SC_NFRAME * SC_M;                     // :697-702; M/4 = one sample
            if (norm < -0.35) nn += SC_M / 4;
            if (norm > 0.35) nn -= SC_M / 4;
            sh.nin = nn; sh.norm = (float)norm;
            double sn, cs; sincos(-omega * (double)nin, &sn, &cs);
            double lr = lo0.x * cs - lo0.y * sn, li = lo0.x * sn + lo0.y * cs; const double a = hypot(lr, li);
            sh.rx_lo = make_double2(lr / a, li / a);
        }
        __syncthreads();
        float2 oldsym = make_float2(0.0f, 0.0f), newsym = oldsym; double2 pm = make_double2(0.0, 0.0);
        if (tid < SC_NFRAME) {
            const double2 v = sh.buf[tid + SC_NPHASE / 2];
            double sn, cs; sincos(-sh.phase[tid], &sn, &cs);
            newsym = make_float2((float)(v.x * cs - v.y * sn), (float)(v.x * sn + v.y * cs));
            oldsym = sh.sb[SC_NFRAME + tid];
        }
        if (tid < SC_NPHASE) pm = sh.buf[SC_NFRAME + tid];
        __syncthreads();
        if (tid < SC_NFRAME) { sh.sb[tid] = oldsym; sh.sb[SC_NFRAME + tid] = newsym; }
        if (tid < SC_NPHASE) sh.buf[tid] = pm;
        __syncthreads();
        // ---- frame sync (:773-826)
        const int state0 = sh.state;
        if (state0 == 0) {
            if (tid < SC_NFRAME) {                          // normalised correlation with the sync word at symbol offset tid
                float nr = 0.0f, ni = 0.0f, en = 0.0f;
                for (int k = 0; k < SC_NSYNC; k++) { const float2 v = sh.sb[tid + k]; const float p = c_sync[k] * 0.25f; nr += v.x * p; ni -= v.y * p; en += v.x * v.x + v.y * v.y; }
                const float den = sqrtf(en) + 1e-12f;
                const float2 c = make_float2(nr / den, ni / den);
                sh.cs[tid] = c; sh.cabs[tid] = hypotf(c.x, c.y);
            }
            __syncthreads();
            {   // first maximum of |Cs| over the 96 offsets (strict > in ascending s, dsp.py:789): per-wave shuffles, two partials
                float v = tid < SC_NFRAME ? sh.cabs[tid] : -1.0f; int si = tid < SC_NFRAME ? tid : 1 << 20;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(v, off); const int oi = __shfl_xor(si, off);
                    if (ov > v || (ov == v && oi < si)) { v = ov; si = oi; }
                }
                if (lane == 0 && wave < 2) { sh.wbest[wave] = v; sh.wbests[wave] = si; }
            }
            __syncthreads();
            if (tid == 0) {
                float best = 0.0f; int ms = 0; float2 mc = make_float2(0.0f, 0.0f);
                for (int w = 0; w < 2; w++) if (sh.wbest[w] > best) { best = sh.wbest[w]; ms = sh.wbests[w]; }
                if (best > 0.0f) mc = sh.cs[ms];
                sh.max_cs = mc;
                if (best >= 0.5f) {
                    sh.state = 1; sh.fs_s = ms; sh.bad_fs = 0;
                    sh.phase_amb = mc.x < 0.0f ? SC_PI : 0.0;
                    float e = 0.0f; for (int k = 0; k < SC_NSYNC; k++) { const float2 v = sh.sb[ms + k]; e += v.x * v.x + v.y * v.y; }
                    sh.g = 1.0 / (sqrt((double)(e / SC_NSYNC)) + 1e-12);
                }
            }
        } else if (tid == 0) {
            const int fs = sh.fs_s; const double sgn = sh.phase_amb != 0.0 ? -1.0 : 1.0;
            int nerr = 0; float e = 0.0f;
            for (int k = 0; k < SC_NSYNC; k++) {
                const float2 v = sh.sb[fs + k];
                // exp(j pi) = -1 + 1.2e-16 j: the product's real part decides, its imaginary part only breaks exact ties
                const double pr = sgn * (double)v.x * c_sync[k], pi = sgn * (double)v.y * c_sync[k];
                if (pr < 0.0 || (pr == 0.0 && pi < 0.0)) nerr++;
                e += v.x * v.x + v.y * v.y;
            }
            sh.bad_fs = nerr > 2 ? sh.bad_fs + 1 : 0;
            if (sh.bad_fs >= 3) sh.state = 0;
            sh.g = 1.0 / (sqrt((double)(e / SC_NSYNC)) + 1e-12);
        }
        __syncthreads();
        // ---- outputs: payload symbols of the frame-sync position with the pi ambiguity resolved (:828-829); z_hat = g Re (sc_rx.py:99-101)
        {
            const int fs = sh.fs_s; const float sgn = sh.phase_amb != 0.0 ? -1.0f : 1.0f; const float g = (float)sh.g;
            const size_t o = ((size_t)b * max_frames + nf) * SC_NPAY;
            if (tid < SC_NPAY) {
                const float2 v = sh.sb[fs + SC_NSYNC + tid];
                if (payload) payload[o + tid] = make_float2(sgn * v.x, sgn * v.y);
                if (zhat) zhat[o + tid] = sh.state == 1 ? g * (sgn * v.x) : 0.0f;
            }
            if (tid == 0 && frames) {
                rade_sc_frame fr; memset(&fr, 0, sizeof fr);
                fr.state = sh.state; fr.nin = sh.nin; fr.fs_s = fs; fr.norm_rx_timing = sh.norm; fr.g = g; fr.max_cs_re = sh.max_cs.x; fr.max_cs_im = sh.max_cs.y;
                fr.phase_ambiguity = (float)sh.phase_amb;
                frames[(size_t)b * max_frames + nf] = fr;
            }
        }
        pos += nin; nf++;
        __syncthreads();
    }
    // ---- state back to HBM
    for (int i = tid; i < SC_NOUT; i += SC_NT) S->rx_filt_out[i] = sh.rf[i];
    for (int i = tid; i < 2 * SC_NFRAME; i += SC_NT) S->rx_symb_buf[i] = sh.sb[i];
    if (tid < SC_NTAP) S->rx_mem[tid] = sh.fin[tid];
    if (tid < SC_NPHASE) S->phase_mem[tid] = sh.buf[tid];
    if (tid == 0) {
        S->nin = sh.nin; S->state = sh.state; S->fs_s = sh.fs_s; S->bad_fs = sh.bad_fs; S->rx_lo = sh.rx_lo; S->phase_fine = sh.phase_fine; S->phase_coarse = sh.phase_coarse;
        S->phase_ambiguity = sh.phase_amb; S->g = sh.g; S->max_cs_re = sh.max_cs.x; S->max_cs_im = sh.max_cs.y; S->norm_rx_timing = sh.norm;
        rade_sc_status r; memset(&r, 0, sizeof r);
        r.n_frames = nf; r.consumed = pos; r.state = sh.state; r.nin = sh.nin; r.fs_s = sh.fs_s; r.g = (float)sh.g; r.max_cs_re = sh.max_cs.x; r.max_cs_im = sh.max_cs.y;
        r.norm_rx_timing = sh.norm; r.phase_ambiguity = (float)sh.phase_amb;
        status[b] = r;
    }
}

__global__ void k_sc_reset(sc_stream *st, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    sc_stream z; memset(&z, 0, sizeof z);
    z.tx_lo = make_double2(1.0, 0.0); z.rx_lo = make_double2(1.0, 0.0); z.g = 1.0; z.nin = SC_NFRAME * SC_M;
    st[b] = z;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static void sc_rrc(double alpha, double Rs, double Fs, double *h /* [24] */)
{   // gen_rn_coeffs (dsp.py:532-562): raised cosine -> 4096-point spectrum -> square root of the magnitude -> back
    const int Nfft = 4096, M = (int)(Fs / Rs), Nsym = SC_NTAP / SC_M;
    const double T = 1.0 / Fs, Ts = 1.0 / Rs, a = -Nsym * Ts / 2, bnd = Nsym * Ts / 2;
    int len = (int)ceil((bnd - a) / T);
    if (len > 64) len = 64;
    double g[64];
    for (int i = 0; i < len; i++) {
        const double n = a + i * T;
        const double sden = SC_PI * n / Ts, sinc = fabs(sden) < 1e-10 ? 1.0 : sin(SC_PI * n / Ts) / sden;
        const double cden = 1 - (2 * alpha * n / Ts) * (2 * alpha * n / Ts), cs = fabs(cden) < 1e-10 ? SC_PI / 4 : cos(alpha * SC_PI * n / Ts) / cden;
        g[i] = sinc * cs;
    }
    double *rr = (double *)malloc(sizeof(double) * 2 * Nfft);
    for (int k = 0; k < Nfft; k++) {
        double re = 0, im = 0;
        for (int i = 0; i < len; i++) { const double ph = -2 * SC_PI * (double)((long)k * i % Nfft) / Nfft; re += g[i] * cos(ph); im += g[i] * sin(ph); }
        re /= M; im /= M;
        if (hypot(re, im) < 0.02) { re *= 0.001; im *= 0.001; }
        const double mag = sqrt(hypot(re, im)), ang = atan2(im, re);
        rr[2 * k] = mag * cos(ang); rr[2 * k + 1] = mag * sin(ang);
    }
    for (int n = 0; n < SC_NTAP; n++) {
        double re = 0;
        for (int k = 0; k < Nfft; k++) { const double ph = 2 * SC_PI * (double)((long)k * n % Nfft) / Nfft; re += rr[2 * k] * cos(ph) - rr[2 * k + 1] * sin(ph); }
        h[n] = re / Nfft;
    }
    free(rr);
}

extern "C" {

rade_sc *rade_sc_open(int n_streams, double Rs, double Fs, double fcentreHz, double alpha, int device)
{
    if (n_streams <= 0 || Rs <= 0 || Fs != SC_M * Rs) { fprintf(stderr, "rade_sc_open: need n_streams > 0 and Fs = 4 Rs\n"); return NULL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) { fprintf(stderr, "rade_sc_open: no usable GPU (this library has no CPU path)\n"); return NULL; }
    if (hipSetDevice(device) != hipSuccess) return NULL;
    rade_sc *h = (rade_sc *)calloc(1, sizeof *h);
    h->B = n_streams; h->device = device; h->omega = 2 * SC_PI * fcentreHz / Fs;
    sc_rrc(alpha, Rs, Fs, h->rrc);
    if (hipMalloc((void **)&h->d_rrc, sizeof h->rrc) != hipSuccess || hipMalloc((void **)&h->st, sizeof(sc_stream) * n_streams) != hipSuccess ||
        hipMalloc((void **)&h->d_status, sizeof(rade_sc_status) * n_streams) != hipSuccess) { rade_sc_close(h); return NULL; }
    hipMemcpy(h->d_rrc, h->rrc, sizeof h->rrc, hipMemcpyHostToDevice);
    rade_sc_reset(h);
    return h;
}

void rade_sc_close(rade_sc *h)
{
    if (!h) return;
    if (h->d_rrc) hipFree(h->d_rrc);
    if (h->st) hipFree(h->st);
    if (h->d_status) hipFree(h->d_status);
    free(h);
}

void rade_sc_reset(rade_sc *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    hipLaunchKernelGGL(k_sc_reset, dim3((h->B + 63) / 64), dim3(64), 0, 0, h->st, h->B);
    hipDeviceSynchronize();
}

int rade_sc_n_streams(const rade_sc *h) { return h ? h->B : 0; }
int rade_sc_n_tx_out(const rade_sc *h) { (void)h; return SC_NFRAME * SC_M; }
int rade_sc_nin_max(const rade_sc *h) { (void)h; return SC_NFRAME * SC_M + SC_M / 4; }
int rade_sc_n_payload(const rade_sc *h) { (void)h; return SC_NPAY; }
void rade_sc_rrc(const rade_sc *h, double *taps_out) { if (h && taps_out) memcpy(taps_out, h->rrc, sizeof h->rrc); }

int rade_sc_tx(rade_sc *h, const float *symbs_dev, int n_frames, void *iq_out_dev, long iq_stride, void *stream)
{
    if (!h || !symbs_dev || !iq_out_dev || n_frames <= 0 || iq_stride < (long)n_frames * SC_NFRAME * SC_M) return -1;
    (void)hipSetDevice(h->device);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sc_tx, dim3(n_frames, h->B), dim3(SC_NFRAME * SC_M), 0, s, h->st, h->d_rrc, h->omega, symbs_dev, n_frames, (float2 *)iq_out_dev, iq_stride);
    hipLaunchKernelGGL(k_sc_tx_finish, dim3(h->B), dim3(64), 0, s, h->st, h->omega, symbs_dev, n_frames);
    return hipGetLastError() == hipSuccess ? n_frames * SC_NFRAME * SC_M : -1;
}

int rade_sc_rx(rade_sc *h, const void *rx_dev, long rx_stride, int n_avail, int max_frames, void *payload_out_dev, float *zhat_out_dev,
               rade_sc_frame *frames_out_dev, rade_sc_status *status_host, void *stream)
{
    if (!h || !rx_dev || n_avail < 0 || max_frames <= 0 || rx_stride < n_avail) return -1;
    (void)hipSetDevice(h->device);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sc_rx, dim3(h->B), dim3(SC_NT), 0, s, h->st, h->d_rrc, h->omega, (const float2 *)rx_dev, rx_stride, n_avail, max_frames,
                       (float2 *)payload_out_dev, zhat_out_dev, frames_out_dev, h->d_status);
    if (hipGetLastError() != hipSuccess) return -1;
    if (status_host) {
        if (hipMemcpyAsync(status_host, h->d_status, sizeof(rade_sc_status) * h->B, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
    }
    return 0;
}

}   // extern "C"
