/*
 * rade_engine.c -- batched RADE engine (plain C host code driving the HIP kernels through the
 * launch shims in rade_dev.h).  Implements include/rade_batch.h.
 *
 * Execution model: the encoder / decoder are evaluated layer by layer over
 * ALL streams and ALL time steps of a chunk: the feed-forward part of every layer is one skinny-N
 * f32-MFMA GEMM with M = B*T rows, and only the 64/96-wide GRU recurrences run as serial scans.
 * The receiver runs one workgroup per stream for a whole rade_batch_rx call; the sync state machine consumes the
 * decoder's aux bits (UW errors, radae_rxe.py:220-224, :306-312), so the decoder runs inside that workgroup
 * right before every unique-word decision (k_rx_sync2 -> rx2_decode_pending, rade_rx.hip).
 */
#define _GNU_SOURCE          /* sched_getaffinity / CPU_COUNT */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <time.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "rade_batch.h"
#include "rade_api.h"
#include "rade_host.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rade: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); goto fail; } } while (0)

/* ---- how rade_batch_rx waits for its stream (the one blocking call of the batched API) -----------------------------------------------------
 * hipStreamSynchronize spins by default: right for one engine per core, wrong when the host has fewer cores than engines in flight -- 8 GPUs x 3
 * batches in flight are 24 host threads, and a container's CPU quota may be 16 (seen on the 1-GPU lease) -- the spinning threads then take the
 * cores the other engines' launch paths need.  Policy: spin while the engines open in this process fit the CPUs the process may use, otherwise
 * sleep until the launch is done (sleep_until_event below: naps between hipEventQuery calls -- waiting on a hipEventBlockingSync event does not sleep on this runtime).
 * $RADE_SYNC=spin|block overrides. */
static int g_engines_open;                 /* engines alive in this process (atomic) */
double rade_host_cpu_quota(void)
{   /* CPUs this process may use: the smaller of its affinity mask and the cgroup v2 quota (cpu.max = "quota period" or "max period") */
    cpu_set_t set; double n = 1.0;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (double)CPU_COUNT(&set);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[64]; double per = 0.0;
        if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0.0) { const double c = atof(q) / per; if (c > 0.0 && c < n) n = c; }
        fclose(f);
    }
    return n;
}
/* 1 = sleep (sleep_until_event), 0 = spin: a pure function of the two counts (tests/test_host_cpu.py) */
int rade_sync_policy(int engines_open, double cpu_quota) { return (double)engines_open > cpu_quota; }
/* $RADE_SYNC_PEERS = processes that share this process's CPUs and hold as many engines each (one process per GPU under torchrun: the quota is the
 * container's, the engine count this process's -- bench.py sets it to LOCAL_WORLD_SIZE): the policy then compares engines x peers with the quota */
static int g_sync_mode = 2, g_sync_peers = 1; static double g_sync_quota = 1.0;    /* written once (pthread_once), read by every engine's host thread */
static pthread_once_t g_sync_once = PTHREAD_ONCE_INIT;
static void sync_policy_init(void)
{
    const char *e = getenv("RADE_SYNC"), *p = getenv("RADE_SYNC_PEERS");
    g_sync_quota = rade_host_cpu_quota(); g_sync_peers = p && atoi(p) > 1 ? atoi(p) : 1;
    g_sync_mode = e && !strcmp(e, "block") ? 1 : (e && !strcmp(e, "spin") ? 0 : 2);
}
static int sync_blocking_now(void)
{
    pthread_once(&g_sync_once, sync_policy_init);
    return g_sync_mode == 2 ? rade_sync_policy(__atomic_load_n(&g_engines_open, __ATOMIC_RELAXED) * g_sync_peers, g_sync_quota) : g_sync_mode;
}

typedef struct { float *wp, *bias; unsigned short *wp16, *wa16; float *wscale, *wscale16; int N, K; } dev_lin;   /* wscale16: column scales when wp16 is one plane of integers */

/* every public entry point runs on its engine's device, whatever device the calling thread had current (one host thread may
 * drive several engines, and an engine may be called from a thread other than the one that opened it) */
#define ON_DEV(h) do { if (h) (void)hipSetDevice((h)->device); } while (0)

#define RADE_PROF_MAXEV 256   /* launches recorded per profiled interval before the events are drained */
#define RADE_PROF_MAXIV 4096  /* launch intervals kept per profiling session (rade_batch_profile_intervals) */
struct rade_batch {
    int B, max_tx_mf, device, flags, trace_cap, Tcap;
    int R, dec_rows;                      /* do_radae_rx calls per stream per sync launch; 3R decoder slots */
    int unsync_off_after;                 /* int(disable_unsync * Fs / Nmf) or -1 */
    unsigned short *corr16, *corrq16, *corra16, *wfwd16, *bpf16; double *vm; int rx_lds, rx_census;   /* dynamic LDS of a receiver launch; phase mask of the -DRX2_CENSUS developer build */
    int feat_in, enc_kpad, bottleneck1;   /* 84 (model19: 4x21) or 80 (model05/bbfm: 4x20); tanh on z when bottleneck 1 */
    float *dec2_x, *dec2_gi, *dec2_hbuf, *dec2_h[5];   /* stand-alone decoder (rade_batch_decode) */
    rd_tables *d_tab;
    /* weights */
    dev_lin enc_dense1, enc_zdense, dec_dense1, dec_output, enc_gin[5], dec_gin[5], enc_conv[5], dec_conv[5], dec_glu[5];
    float *enc_whh[5], *enc_bhh[5], *dec_whh[5], *dec_bhh[5];
    unsigned short *dec_whq[5]; float *dec_whs[5];      /* decoder W_hh as matrix-core fragments (int8-exact) + row scales; NULL when the blob's recurrent weights are not int8 x scale */
    /* transmit side */
    float *enc_xin, *enc_x, *enc_gi, *enc_h[5], *enc_z, *eoo, *eoo_bits;
    unsigned short *enc_xf; int enc_nq, enc_unfused, enc_seq_taps, enc_no_pair;
    int enc_hist_frag;                   /* the history tile of enc_xf holds what enc_x's two float32 history rows hold (set by a fragment pass, cleared by a reset or a float32-row pass) */   /* the concat buffer as matrix-core operand fragments (rade_enc.hip: [B][enc_nq][RD_EF_TILE] binary16), engines with enough rows for the batched GEMMs only */
    /* optional Tx band-pass filter + clip (RADE_BATCH_TX_BPF; radae_txe.py:74-83): filter state per stream, its initial value, the modulator's raw output, block phases */
    int bypass_dec;                          /* RADE_BATCH_BYPASS_DEC */
    rd_bpf_state *tx_bpf, *tx_bpf_init; void *tx_raw; float *tx_chain; float *eoo_filt;   /* eoo_filt [B][Neoo] c64: the end-of-over frame as transmitted (filtered + clipped) for the channel's with_eoo */
    void *chan_scratch; void *chan_mp;        /* chan_mp [B][max_tx_mf * 960] c64: multipath output of the fused modulator (rade_batch_tx_channel), allocated on first use */
    /* receive side */
    rd_rx_stream *rx_st; rd_rx_round *rx_round;
    int *rx_avail, *rx_acc, *rx_progress, *rx_status;
    float *zrows, *dec_x, *dec_gi, *dec_hbuf, *dec_h[5], *feat84, *dtcache;
    void *rx_filt; float *bpf_chain; long filt_cap; int chain_stride;   /* band-pass pre-pass of an invocation: filtered samples [B][filt_cap] c64 and block phases [B][chain_stride] c64, grown on demand */
    rd_rx_trace *trace; float *trace_z;
    long long *wg_cycles;            /* [B] per-stream cycles of the last receiver launch */
    int *h_small;                    /* pinned host scratch */
    unsigned *lcg_seeds;             /* host copy for resets */
    unsigned *d_lcg_seeds;
    /* optional per-kernel-class timing with HIP events (bench.py roofline leg; never on in timed runs) */
    int prof_on, prof_cnt; hipEvent_t prof_ev[2 * RADE_PROF_MAXEV]; int prof_cls[RADE_PROF_MAXEV]; double prof_fl[RADE_PROF_MAXEV];
    double prof_ms[RADE_PROF_NCLASS], prof_flops[RADE_PROF_NCLASS]; long prof_n[RADE_PROF_NCLASS];
    /* optional: absolute start / end of every profiled launch relative to a caller-supplied event (launches of several engines on one time axis) */
    hipEvent_t prof_ref; int iv_n; int iv_cls[RADE_PROF_MAXIV]; float iv_t0[RADE_PROF_MAXIV], iv_t1[RADE_PROF_MAXIV];
    long rx_calls_search, rx_calls_sync;
    /* encoder in two time chunks on two HIP streams (encode_core): the side stream and the events that order the chunks */
    int enc_chunks; hipStream_t enc_side; hipEvent_t ev_fork, ev_join, ev_scan[5];
    hipEvent_t ev_block;             /* the event rade_batch_rx sleeps on (sleep_until_event) when the host has fewer CPUs than engines (sync_blocking_now) */
    long n_sync_block, n_sync_spin;  /* waits of either kind so far (rade_batch_sync_counts) */
    double wait_est_us;              /* how long the sleeping wait of rade_batch_rx lasted lately (running average): the next one sleeps through most of that before it polls */
};

static const int ENC_IN[5] = { 64, 224, 384, 544, 704 };    /* GRU input widths (radae_base.py:240-248) */
static const int ENC_DIL[5] = { 1, 2, 2, 2, 2 };
static const int DEC_IN[5] = { 96, 224, 352, 480, 608 };    /* radae_base.py:378-386 */

static void *dev_upload(const void *src, size_t bytes)
{
    void *d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    if (hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return NULL; }
    return d;
}
static void *dev_zeros(size_t bytes)
{
    void *d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    /* complete before returning: the memset runs on the null stream, which is not ordered against a caller's non-blocking stream (buffers
     * allocated on first use inside a stream-ordered call would otherwise be zeroed on top of what that call's kernels wrote) */
    if (hipMemset(d, 0, bytes) != hipSuccess || hipStreamSynchronize(NULL) != hipSuccess) { hipFree(d); return NULL; }
    return d;
}

/* pack W[N][K] (optionally padding K up to Kpad with zero columns) and upload */
static int upload_lin(dev_lin *d, const float *w, const float *b, const float *row_scale, int N, int K, int Kpad)
{
    float *wsrc = (float *)w, *tmp = NULL;
    if (Kpad != K) {
        tmp = calloc((size_t)N * Kpad, sizeof(float));
        for (int o = 0; o < N; o++) memcpy(tmp + (size_t)o * Kpad, w + (size_t)o * K, sizeof(float) * K);
        wsrc = tmp;
    }
    const long n = rd_packed_size(N, Kpad);
    float *packed = malloc(sizeof(float) * n);
    rd_pack_weights(wsrc, N, Kpad, packed);
    d->wp = dev_upload(packed, sizeof(float) * n);
    d->bias = b ? dev_upload(b, sizeof(float) * N) : NULL;
    d->N = N; d->K = Kpad; d->wp16 = NULL; d->wscale16 = NULL;
    if (Kpad % 16 == 0) {                  /* binary16 copy for the f16 matrix cores (k_gemm16): int8-exact layers as ONE plane of integers + column scales, others as two planes */
        const long n16 = rd_packed16_size(N, Kpad);
        unsigned short *p16 = malloc(sizeof(unsigned short) * n16);
        float *sc = calloc((size_t)((N + 31) / 32) * 32, sizeof(float));
        long nq = -1;
        if (p16 && sc && row_scale && !getenv("RADE_NO_INT8_EXACT")) nq = rd_pack_weights_q16(wsrc, row_scale, N, Kpad, p16, sc);
        if (nq > 0) {
            d->wp16 = dev_upload(p16, sizeof(unsigned short) * nq);
            d->wscale16 = dev_upload(sc, sizeof(float) * (size_t)((N + 31) / 32) * 32);
        } else {
            if (!p16 || rd_pack_weights_f16x2(wsrc, N, Kpad, p16) < 0) {
                fprintf(stderr, "rade: a weight exceeds the range of the split-binary16 operand planes (|w| < 63.9)\n");
                free(p16); free(sc); free(packed); free(tmp); return -1;
            }
            d->wp16 = dev_upload(p16, sizeof(unsigned short) * n16);
        }
        free(p16); free(sc);
    }
    d->wa16 = NULL; d->wscale = NULL;
    if (Kpad % 32 == 0) {                  /* A-operand layout of the in-kernel decoder's 16x16x32 products */
        const long na = rd_packed16a_size(N, Kpad);
        unsigned short *pa = malloc(sizeof(unsigned short) * na);
        float *sc = malloc(sizeof(float) * (size_t)((N + 15) / 16) * 16);
        long nq = -1;
        if (pa && sc && row_scale && !getenv("RADE_NO_INT8_EXACT")) nq = rd_pack_weights_q16_a16(wsrc, row_scale, N, Kpad, pa, sc);
        if (nq > 0) {                      /* int8 in the blob: the integers themselves in ONE binary16 plane (exact), scales apart: half the bytes to stream */
            d->wa16 = dev_upload(pa, sizeof(unsigned short) * nq);
            d->wscale = dev_upload(sc, sizeof(float) * (size_t)((N + 15) / 16) * 16);
        } else {
            if (!pa || rd_pack_weights_f16x2_a16(wsrc, N, Kpad, pa) < 0) { free(pa); free(sc); free(packed); free(tmp); return -1; }
            d->wa16 = dev_upload(pa, sizeof(unsigned short) * na);
        }
        free(pa); free(sc);
    }
    free(packed); free(tmp);
    return (d->wp && (!b || d->bias)) ? 0 : -1;
}

/* start-of-utterance state, ONE launch (k_batch_reset) for the directions asked for; the trace buffers and the Tx filter state only where they exist */
static void reset_on(rade_batch *h, int tx, int rx, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    rd_reset_args r;
    memset(&r, 0, sizeof r);
    r.B = h->B;
    if (rx) {
        r.st = h->rx_st; r.seeds = h->d_lcg_seeds; r.foff_err = (h->flags & RADE_FOFF_TEST) ? 10.0 : 0.0;   /* rade_api.c:263-264 */
        r.dec_h = h->dec_h[0];                             /* the five layers' states are one allocation */
        /* only a stream's history row has to start from zero: every other row of dec_x is written before it is read */
        r.dec_x = h->dec_x; r.dec_x_sb = (long)(1 + h->dec_rows) * RD_DEC_W;
        if (h->trace) { hipMemsetAsync(h->trace, 0, sizeof(rd_rx_trace) * (size_t)h->B * h->trace_cap, st); hipMemsetAsync(h->trace_z, 0, sizeof(float) * (size_t)h->B * h->trace_cap * RD_ZMF, st); }
    }
    if (tx) {
        h->enc_hist_frag = 0;
        if (h->tx_bpf) hipMemcpyAsync(h->tx_bpf, h->tx_bpf_init, sizeof(rd_bpf_state) * h->B, hipMemcpyDeviceToDevice, st);
        r.enc_h = h->enc_h[0];
        /* the two history rows of each stream (conv taps before the first frame); rows 2.. are written layer by layer before they are read */
        r.enc_x = h->enc_x; r.enc_x_sb = (long)(2 + h->Tcap) * RD_ENC_W;
    }
    rd_launch_reset(&r, st);
}

void rade_batch_rx_reset(rade_batch *h) { ON_DEV(h); reset_on(h, 0, 1, NULL); hipDeviceSynchronize(); }

/* stream-ordered reset of both directions (start of a new batch of utterances) */
void rade_batch_reset(rade_batch *h, void *stream) { ON_DEV(h); reset_on(h, 1, 1, stream); }

void rade_batch_rx_set_lcg(rade_batch *h, const unsigned *seeds_host)
{
    ON_DEV(h);
    for (int b = 0; b < h->B; b++) h->lcg_seeds[b] = seeds_host ? seeds_host[b] : 1u;
    hipMemcpy(h->d_lcg_seeds, h->lcg_seeds, sizeof(unsigned) * h->B, hipMemcpyHostToDevice);
    rade_batch_rx_reset(h);
}

void rade_batch_tx_reset(rade_batch *h) { ON_DEV(h); reset_on(h, 1, 0, NULL); hipDeviceSynchronize(); }

rade_batch *rade_batch_open_mem(const void *blob, size_t blob_len, const rade_batch_config *cfg)
{
    rade_batch *h = NULL;
    rd_model m; int have_model = 0;
    if (!cfg || cfg->n_streams <= 0 || cfg->max_tx_mf <= 0) { fprintf(stderr, "rade: bad batch config\n"); return NULL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "rade: no HIP device available -- this library has no CPU fallback\n");
        return NULL;
    }
    CHK(hipSetDevice(cfg->device));
    if (rd_model_parse(blob, blob_len, &m)) return NULL;
    have_model = 1;
    h = calloc(1, sizeof *h);
    if (!h) { rd_model_free(&m); return NULL; }
    __atomic_add_fetch(&g_engines_open, 1, __ATOMIC_RELAXED);
    h->B = cfg->n_streams; h->max_tx_mf = cfg->max_tx_mf; h->device = cfg->device; h->flags = cfg->flags;
    h->trace_cap = cfg->rx_trace_calls; h->Tcap = 3 * cfg->max_tx_mf;
    h->unsync_off_after = cfg->disable_unsync != 0.0f ? (int)((double)cfg->disable_unsync * 8000.0 / RD_NMF) : -1;    /* radae_rxe.py:279-280 */
    const size_t B = (size_t)h->B, T = (size_t)h->Tcap;
    /* do_radae_rx calls per stream and launch (the per-launch bookkeeping arrays hold RD_RX_ROUND_MAX); RADE_ROUND_CALLS
     * lowers it for tests: the call loop then takes several launches with identical results */
    h->R = getenv("RADE_ROUND_CALLS") ? atoi(getenv("RADE_ROUND_CALLS")) : RD_RX_ROUND_MAX;
    if (h->R < 1) h->R = 1;
    if (h->R > RD_RX_ROUND_MAX) h->R = RD_RX_ROUND_MAX;
    /* rows a stream may hold before its decoder stage runs: 8 frames wait for a unique-word check, up to 7 more can be
     * left over from before a loss of sync */
    h->dec_rows = getenv("RADE_DEC_ROWS") ? atoi(getenv("RADE_DEC_ROWS")) : 48;
    if (h->dec_rows < 3) h->dec_rows = 3;
    if (h->dec_rows > 63) h->dec_rows = 63;            /* the stage's LDS row flags hold DQ_PEND_MAX = 64 entries (rade_kernels.hip) */
    const size_t DR = (size_t)h->dec_rows;

    rd_tables *tab = malloc(sizeof *tab);
    rd_tables_fill(tab);
    h->d_tab = dev_upload(tab, sizeof *tab);
    {
        unsigned short *c16 = malloc(sizeof(unsigned short) * 5 * 10 * 2 * 64 * 8);
        if (c16) { rd_corr16_table_fill(tab, c16); h->corr16 = dev_upload(c16, sizeof(unsigned short) * 5 * 10 * 2 * 64 * 8); free(c16); }
    }
    {   /* the same correlator in two stages: 16 polynomial moments, then their expansion to the 40 frequencies (rade_host.c) */
        unsigned short *q16 = malloc(sizeof(unsigned short) * 2 * 10 * 2 * 64 * 8), *a16 = malloc(sizeof(unsigned short) * 5 * 2 * 64 * 8);
        if (q16 && a16) {
            rd_corrq16_table_fill(tab, q16); h->corrq16 = dev_upload(q16, sizeof(unsigned short) * 2 * 10 * 2 * 64 * 8);
            rd_corra16_table_fill(tab, a16); h->corra16 = dev_upload(a16, sizeof(unsigned short) * 5 * 2 * 64 * 8);
        }
        free(q16); free(a16);
    }
    {   /* the demodulator DFT matrix as matrix-core operands (k_rx_sync2) */
        unsigned short *w16 = malloc(sizeof(unsigned short) * 2 * 10 * 2 * 64 * 8);
        if (w16) { rd_wfwd16_table_fill(tab, w16); h->wfwd16 = dev_upload(w16, sizeof(unsigned short) * 2 * 10 * 2 * 64 * 8); free(w16); }
    }
    {   /* the band-pass taps as matrix-core operands (k_rx_bpf) */
        unsigned short *b16 = malloc(sizeof(unsigned short) * 4 * 2 * 64 * 8);
        if (b16) { rd_bpf16_table_fill(tab, b16); h->bpf16 = dev_upload(b16, sizeof(unsigned short) * 4 * 2 * 64 * 8); free(b16); }
    }
    free(tab);
    {   /* ((n - 79.5) / 80)^m, m = 0..7, by repeated multiplication (the order the kernels build their LDS copy in) */
        double vm[8][RD_M];
        for (int n = 0; n < RD_M; n++) { const double nu = ((double)n - 79.5) / 80.0; double v = 1.0; for (int m = 0; m < 8; m++) { vm[m][n] = v; v *= nu; } }
        h->vm = dev_upload(vm, sizeof vm);
    }
    /* the receiver kernel asks for more dynamic LDS than the 64 KB default: raised here, once per engine and before any launch (the
     * attribute belongs to the device's code object; setting it again from another engine's thread is harmless).  RADE_RX2_SOLO=1
     * (developer switch) asks for more than half a CU's LDS, i.e. one workgroup per CU. */
    h->rx_lds = rd_rx_sync_prepare(getenv("RADE_RX2_SOLO") != NULL);
    h->rx_census = getenv("RADE_RX2_CENSUS") ? atoi(getenv("RADE_RX2_CENSUS")) : 0;      /* only acts in -DRX2_CENSUS builds */
    if (!h->d_tab || !h->corr16 || !h->corrq16 || !h->corra16 || !h->vm || !h->wfwd16 || !h->bpf16 || h->rx_lds <= 0) goto fail;

    int err = 0;
    h->feat_in = m.enc_dense1.n_in; h->enc_kpad = (h->feat_in + 15) & ~15; h->bottleneck1 = (cfg->flags & RADE_BATCH_BOTTLENECK1) != 0;
    err |= upload_lin(&h->enc_dense1, m.enc_dense1.w, m.enc_dense1.b, NULL, 64, h->feat_in, h->enc_kpad);
    err |= upload_lin(&h->enc_zdense, m.enc_zdense.w, m.enc_zdense.b, NULL, 80, 864, 864);
    err |= upload_lin(&h->dec_dense1, m.dec_dense1.w, m.dec_dense1.b, NULL, 96, 80, 80);
    err |= upload_lin(&h->dec_output, m.dec_output.w, m.dec_output.b, NULL, h->feat_in, 736, 736);
    for (int l = 0; l < 5 && !err; l++) {
        err |= upload_lin(&h->enc_gin[l], m.enc_gru[l].w_ih, m.enc_gru[l].b_ih, m.enc_gru[l].s_ih, 192, ENC_IN[l], ENC_IN[l]);
        err |= upload_lin(&h->dec_gin[l], m.dec_gru[l].w_ih, m.dec_gru[l].b_ih, m.dec_gru[l].s_ih, 288, DEC_IN[l], DEC_IN[l]);
        err |= upload_lin(&h->enc_conv[l], m.enc_conv[l].w, m.enc_conv[l].b, m.enc_conv[l].row_scale, 96, m.enc_conv[l].n_in, m.enc_conv[l].n_in);
        err |= upload_lin(&h->dec_conv[l], m.dec_conv[l].w, m.dec_conv[l].b, m.dec_conv[l].row_scale, 32, m.dec_conv[l].n_in, m.dec_conv[l].n_in);
        err |= upload_lin(&h->dec_glu[l], m.dec_glu[l].w, NULL, m.dec_glu[l].row_scale, 96, 96, 96);
        h->enc_whh[l] = dev_upload(m.enc_gru[l].w_hh, sizeof(float) * 192 * 64);
        h->enc_bhh[l] = dev_upload(m.enc_gru[l].b_hh, sizeof(float) * 192);
        h->dec_whh[l] = dev_upload(m.dec_gru[l].w_hh, sizeof(float) * 288 * 96);
        h->dec_bhh[l] = dev_upload(m.dec_gru[l].b_hh, sizeof(float) * 288);
        if (m.dec_gru[l].s_hh && !getenv("RADE_NO_SCAN_MFMA")) {
            unsigned short *pa = malloc(sizeof(unsigned short) * rd_packed16a_size(288, 96)); float *scl = malloc(sizeof(float) * 288);
            const long nq = (pa && scl) ? rd_pack_weights_q16_a16(m.dec_gru[l].w_hh, m.dec_gru[l].s_hh, 288, 96, pa, scl) : -1;
            if (nq > 0) { h->dec_whq[l] = dev_upload(pa, sizeof(unsigned short) * nq); h->dec_whs[l] = dev_upload(scl, sizeof(float) * 288); }
            free(pa); free(scl);
        }
        if (!h->enc_whh[l] || !h->enc_bhh[l] || !h->dec_whh[l] || !h->dec_bhh[l]) err = -1;
    }
    if (err) { fprintf(stderr, "rade: weight upload failed\n"); goto fail; }

    h->enc_xin = dev_zeros(sizeof(float) * B * T * RD_ENC_IN);
    h->enc_x = dev_zeros(sizeof(float) * B * (2 + T) * RD_ENC_W);
    h->enc_gi = dev_zeros(sizeof(float) * B * T * 192);
    h->enc_nq = 1 + (h->Tcap + 31) / 32;
    /* conv_l and the product behind it as separate launches, unless $RADE_ENCF_FUSED (read per engine) asks for k_encf_fused: 7 launches instead of 12 and 6 % less GEMM time
     * alone, but 4-wavefront workgroups with 12 KB of LDS that find fewer places beside the receivers: -1.1 % +- 1.2 (124 registers) / -2.8 % +- 1.2 (152) frames/s in the
     * pipelined bench against the separate launches (profiles/r05_ab_enc_fragments.txt) */
    h->enc_unfused = getenv("RADE_ENCF_FUSED") == NULL;
    h->enc_seq_taps = getenv("RADE_ENCF_SEQ_TAPS") != NULL;    /* developer switches, read per engine (rade_enc.hip: k_encf_gemm) */
    h->enc_no_pair = getenv("RADE_ENCF_NO_PAIR") != NULL;
    if (B * T > 16384 && !getenv("RADE_ENC_ROWS")) {       /* $RADE_ENC_ROWS: the float32-row path (k_gemm16p) for every size: A/B and the equality test */
        h->enc_xf = dev_zeros(sizeof(unsigned short) * B * h->enc_nq * RD_EF_TILE);      /* (NULL = no memory for it: the float32-row kernels serve every call) */
        if (!h->enc_xf && !getenv("RADE_VERBOSE_0"))
            fprintf(stderr, "rade: no device memory for the encoder's fragment buffer (%.1f MB): the float32-row kernels serve every call (slower encoder GEMMs, same results to the last bits documented in rade_batch.h)\n",
                    1e-6 * sizeof(unsigned short) * (double)B * h->enc_nq * RD_EF_TILE);
    }
    h->enc_z = dev_zeros(sizeof(float) * B * T * RD_LATENT);
    h->eoo = dev_zeros(sizeof(float) * B * RD_NEOO * 2);
    h->eoo_bits = dev_zeros(sizeof(float) * B * RD_NEOOBITS);
    h->bypass_dec = (cfg->flags & RADE_BATCH_BYPASS_DEC) != 0;
    if (cfg->flags & RADE_BATCH_TX_BPF) {
        const long nraw = (long)(cfg->max_tx_mf > 2 ? cfg->max_tx_mf : 2) * RD_NMF;          /* (>= the 1152-sample end-of-over frame) */
        rd_bpf_state *init = calloc(B, sizeof *init);
        if (!init) goto fail;
        for (size_t b = 0; b < B; b++) { init[b].mem_len = 100; init[b].phase[0] = 1.0f; }    /* complex_bpf.__init__: dsp.py:54-60 */
        h->tx_bpf_init = dev_upload(init, sizeof(rd_bpf_state) * B); h->tx_bpf = dev_upload(init, sizeof(rd_bpf_state) * B);
        free(init);
        h->tx_raw = dev_zeros(sizeof(float) * 2 * B * nraw);
        h->tx_chain = dev_zeros(sizeof(float) * 2 * B * (cfg->max_tx_mf + 8));
        h->eoo_filt = dev_zeros(sizeof(float) * 2 * B * RD_NEOO);
        if (!h->tx_bpf_init || !h->tx_bpf || !h->tx_raw || !h->tx_chain || !h->eoo_filt) goto fail;
    }
    h->chan_scratch = dev_zeros(sizeof(double) * B * (1 + (cfg->max_tx_mf > 64 ? cfg->max_tx_mf : 64)) * 2);   /* rd_chan_args.scratch */
    h->enc_h[0] = dev_zeros(sizeof(float) * 5 * B * 64); h->dec_h[0] = dev_zeros(sizeof(float) * 5 * B * 96);
    if (!h->enc_h[0] || !h->dec_h[0]) goto fail;
    for (int l = 1; l < 5; l++) { h->enc_h[l] = h->enc_h[0] + (size_t)l * B * 64; h->dec_h[l] = h->dec_h[0] + (size_t)l * B * 96; }
    h->rx_st = dev_zeros(sizeof(rd_rx_stream) * B);
    h->rx_round = dev_zeros(sizeof(rd_rx_round) * B);
    h->rx_avail = dev_zeros(sizeof(int) * B); /* progress word, per-stream counters and status in ONE block, in the order of the host copy (rade_batch_rx: one transfer back per invocation) */
    h->rx_progress = dev_zeros(sizeof(int) * (8 + B * 8));
    if (h->rx_progress) { h->rx_acc = h->rx_progress + 8; h->rx_status = h->rx_progress + 8 + 4 * B; }
    h->wg_cycles = dev_zeros(sizeof(long long) * B);
    h->zrows = dev_zeros(sizeof(float) * B * DR * RD_LATENT);
    h->dec_x = dev_zeros(sizeof(float) * B * (1 + DR) * RD_DEC_W);
    h->dec_gi = dev_zeros(sizeof(float) * B * DR * 288);
    h->dec_hbuf = dev_zeros(sizeof(float) * B * DR * 96);
    h->feat84 = dev_zeros(sizeof(float) * B * DR * 84);
    h->dec2_x = dev_zeros(sizeof(float) * B * (1 + T) * RD_DEC_W);
    h->dec2_gi = dev_zeros(sizeof(float) * B * T * 288);
    h->dec2_hbuf = dev_zeros(sizeof(float) * B * T * 96);
    h->dec2_h[0] = dev_zeros(sizeof(float) * 5 * B * 96);
    if (!h->dec2_h[0]) goto fail;
    for (int l = 1; l < 5; l++) h->dec2_h[l] = h->dec2_h[0] + (size_t)l * B * 96;
    if (!h->dec2_x || !h->dec2_gi || !h->dec2_hbuf) goto fail;
    h->dtcache = dev_zeros(sizeof(float) * B * 2 * RD_NMF * RD_NFC);
    if (!h->enc_xin || !h->enc_x || !h->enc_gi || !h->enc_z || !h->eoo || !h->eoo_bits || !h->chan_scratch || !h->rx_st || !h->rx_round || !h->rx_avail ||
        !h->rx_acc || !h->rx_progress || !h->rx_status || !h->zrows || !h->dec_x || !h->dec_gi || !h->dec_hbuf || !h->feat84 || !h->dtcache) {
        fprintf(stderr, "rade: device allocation failed\n"); goto fail;
    }
    if (h->trace_cap > 0) {
        h->trace = dev_zeros(sizeof(rd_rx_trace) * B * h->trace_cap);
        h->trace_z = dev_zeros(sizeof(float) * B * h->trace_cap * RD_ZMF);
        if (!h->trace || !h->trace_z) goto fail;
    }
    CHK(hipHostMalloc((void **)&h->h_small, sizeof(int) * (8 + B * 8), 0));
    h->lcg_seeds = malloc(sizeof(unsigned) * B);
    for (size_t b = 0; b < B; b++) h->lcg_seeds[b] = 1u;
    h->d_lcg_seeds = dev_upload(h->lcg_seeds, sizeof(unsigned) * B);
    if (!h->d_lcg_seeds) goto fail;
    for (int i = 0; i < 2 * RADE_PROF_MAXEV; i++) CHK(hipEventCreate(&h->prof_ev[i]));
    CHK(hipEventCreateWithFlags(&h->ev_block, hipEventBlockingSync | hipEventDisableTiming));
    h->enc_chunks = getenv("RADE_ENC_CHUNKS") ? atoi(getenv("RADE_ENC_CHUNKS")) : 1;     /* measured (profiles/r03_tx_side_ab.json): 2 chunks gain 3 % with one batch in flight, lose 7 % with two (the default) */
    if (h->enc_chunks != 1) {
        h->enc_chunks = 2;
        CHK(hipStreamCreateWithFlags(&h->enc_side, hipStreamNonBlocking));
        CHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        for (int l = 0; l < 5; l++) CHK(hipEventCreateWithFlags(&h->ev_scan[l], hipEventDisableTiming));
    }
    rade_batch_rx_reset(h);
    if (rd_launch_eoo_build(h->d_tab, NULL, h->eoo, h->B, NULL)) goto fail;
    CHK(hipDeviceSynchronize());
    rd_model_free(&m);
    return h;
fail:
    if (have_model) rd_model_free(&m);
    if (h) rade_batch_close(h);
    return NULL;
}

rade_batch *rade_batch_open(const char *blob_path, const rade_batch_config *cfg)
{
    FILE *f = blob_path ? fopen(blob_path, "rb") : NULL;
    if (!f) { fprintf(stderr, "rade: cannot open weight blob %s\n", blob_path ? blob_path : "(null)"); return NULL; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *buf = malloc(n);
    if (fread(buf, 1, n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
    fclose(f);
    rade_batch *h = rade_batch_open_mem(buf, n, cfg);
    free(buf);
    return h;
}

static void free_lin(dev_lin *d) { if (d->wp) hipFree(d->wp); if (d->bias) hipFree(d->bias); if (d->wp16) hipFree(d->wp16); if (d->wa16) hipFree(d->wa16); if (d->wscale) hipFree(d->wscale); if (d->wscale16) hipFree(d->wscale16); }
void rade_batch_close(rade_batch *h)
{
    if (!h) return;
    ON_DEV(h);
    void *bufs[] = { h->d_tab, h->enc_xin, h->enc_x, h->enc_xf, h->enc_gi, h->enc_z, h->eoo, h->eoo_bits, h->chan_scratch, h->rx_st, h->rx_round, h->rx_avail,
                     h->rx_progress /* + rx_acc, rx_status */, h->wg_cycles, h->zrows, h->dec_x, h->dec_gi, h->dec_hbuf, h->feat84, h->trace, h->trace_z, h->d_lcg_seeds, h->dtcache, h->dec2_x, h->dec2_gi, h->dec2_hbuf, h->rx_filt, h->bpf_chain, h->bpf16, h->tx_bpf, h->tx_bpf_init, h->tx_raw, h->tx_chain, h->eoo_filt, h->corr16, h->corrq16, h->corra16, h->vm, h->chan_mp, h->wfwd16 };
    for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; i++) if (bufs[i]) hipFree(bufs[i]);
    free_lin(&h->enc_dense1); free_lin(&h->enc_zdense); free_lin(&h->dec_dense1); free_lin(&h->dec_output);
    for (int l = 0; l < 5; l++) {
        free_lin(&h->enc_gin[l]); free_lin(&h->dec_gin[l]); free_lin(&h->enc_conv[l]); free_lin(&h->dec_conv[l]); free_lin(&h->dec_glu[l]);
        void *p[] = { h->enc_whh[l], h->enc_bhh[l], h->dec_whh[l], h->dec_bhh[l], l ? NULL : h->enc_h[0], l ? NULL : h->dec_h[0], l ? NULL : h->dec2_h[0], h->dec_whq[l], h->dec_whs[l] };
        for (int i = 0; i < 9; i++) if (p[i]) hipFree(p[i]);
    }
    if (h->h_small) hipHostFree(h->h_small);
    if (h->ev_block) hipEventDestroy(h->ev_block);
    __atomic_sub_fetch(&g_engines_open, 1, __ATOMIC_RELAXED);
    if (h->enc_side) hipStreamDestroy(h->enc_side);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    for (int l = 0; l < 5; l++) if (h->ev_scan[l]) hipEventDestroy(h->ev_scan[l]);
    for (int i = 0; i < 2 * RADE_PROF_MAXEV; i++) if (h->prof_ev[i]) hipEventDestroy(h->prof_ev[i]);
    free(h->lcg_seeds);
    free(h);
}

int rade_batch_n_streams(const rade_batch *h) { return h->B; }
const rd_tables *rd_batch_tables(const rade_batch *h) { return h->d_tab; }
int rd_batch_has_tx_bpf(const rade_batch *h) { return h->tx_bpf != NULL; }

/* ---- per-kernel-class timing (HIP events on the launch stream) -------------------------------- */
/* Events are only recorded while the work is queued and read back afterwards (prof_drain): a profiled step runs back to
 * back like a timed one, without a host synchronisation (and the clock ramp-down that follows it) after every kernel. */
static void prof_drain(rade_batch *h)
{
    for (int i = 0; i < h->prof_cnt; i++) {
        float ms = 0;
        if (hipEventSynchronize(h->prof_ev[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]) == hipSuccess) {
            h->prof_ms[h->prof_cls[i]] += ms; h->prof_flops[h->prof_cls[i]] += h->prof_fl[i]; h->prof_n[h->prof_cls[i]]++;
            float t0 = 0;
            if (h->prof_ref && h->iv_n < RADE_PROF_MAXIV && hipEventElapsedTime(&t0, h->prof_ref, h->prof_ev[2 * i]) == hipSuccess) {
                h->iv_cls[h->iv_n] = h->prof_cls[i]; h->iv_t0[h->iv_n] = t0; h->iv_t1[h->iv_n] = t0 + ms; h->iv_n++;
            }
        }
    }
    h->prof_cnt = 0;
}
#define PROF_BEGIN(h, st) do { if ((h)->prof_on) { if ((h)->prof_cnt >= RADE_PROF_MAXEV) prof_drain(h); hipEventRecord((h)->prof_ev[2 * (h)->prof_cnt], (hipStream_t)(st)); } } while (0)
#define PROF_END(h, st, cls, fl) do { if ((h)->prof_on) { hipEventRecord((h)->prof_ev[2 * (h)->prof_cnt + 1], (hipStream_t)(st)); \
    (h)->prof_cls[(h)->prof_cnt] = (cls); (h)->prof_fl[(h)->prof_cnt] = (fl); (h)->prof_cnt++; } } while (0)

void rade_batch_profile(rade_batch *h, int enable)
{
    ON_DEV(h);
    if (!enable && h->prof_on) prof_drain(h);
    h->prof_on = enable;
    if (enable) { h->prof_cnt = 0; h->iv_n = 0; memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_flops, 0, sizeof h->prof_flops); memset(h->prof_n, 0, sizeof h->prof_n); }
}
/* launches of class `cls` since profiling was enabled as [start, end] in ms after `ref_event` (a hipEvent_t of the caller, recorded and
 * complete before the first launch): call rade_batch_profile_ref before enabling, read the intervals after disabling */
void rade_batch_profile_ref(rade_batch *h, void *ref_event) { h->prof_ref = (hipEvent_t)ref_event; }
int rade_batch_profile_intervals(rade_batch *h, int cls, float *t0_ms, float *t1_ms, int max)
{
    ON_DEV(h);
    prof_drain(h);
    int n = 0;
    for (int i = 0; i < h->iv_n && n < max; i++) if (h->iv_cls[i] == cls) { t0_ms[n] = h->iv_t0[i]; t1_ms[n] = h->iv_t1[i]; n++; }
    return n;
}
int rade_batch_profile_get(rade_batch *h, int cls, double *ms, double *work, long *launches)
{
    ON_DEV(h);
    if (cls < 0 || cls >= RADE_PROF_NCLASS) return -1;
    prof_drain(h);
    *ms = h->prof_ms[cls]; *work = h->prof_flops[cls]; *launches = h->prof_n[cls];
    return 0;
}

/* ---- one GEMM launch helper ------------------------------------------------------------------ */
static int gemm(rade_batch *hh, const dev_lin *w, const float *a1, long a1_sb, long a1_st, int K1, const float *a0, long a0_sb, long a0_st, int K0,
                const int *reset, const int *n_rows, float *y, long y_sb, long y_st, int B, int T, int act, void *stream)
{
    rd_gemm_args g;
    memset(&g, 0, sizeof g);
    g.a1 = a1; g.a1_sb = a1_sb; g.a1_st = a1_st; g.K1 = K1; g.a0 = a0; g.a0_sb = a0_sb; g.a0_st = a0_st; g.K0 = K0;
    g.reset = reset; g.reset_sb = hh->dec_rows; g.n_rows = n_rows; g.Wp = w->wp; g.Wp16 = w->wp16; g.Wscale = w->wp16 ? w->wscale16 : NULL; g.bias = w->bias; g.y = y; g.y_sb = y_sb; g.y_st = y_st; g.N = w->N; g.B = B; g.T = T; g.act = act;
    if (K0 + K1 != w->K) { fprintf(stderr, "rade: internal GEMM shape error (%d+%d != %d)\n", K0, K1, w->K); return -1; }
    PROF_BEGIN(hh, stream);
    const int rc = rd_launch_gemm(&g, stream);
    PROF_END(hh, stream, RADE_PROF_GEMM, 2.0 * (double)B * T * (K0 + K1) * w->N);
    return rc;
}

/* ---- CoreEncoderStatefull.forward over T steps for all streams (radae_base.py:260-286); xin = [B][T][enc_kpad] ----
 * Layer by layer over all B x T rows: the feed-forward pieces are GEMMs, the five W_hh recurrences serial scans (one workgroup of four
 * wavefronts per stream: latency-bound, most of the chip idle).  With enough rows the T steps are cut into two time chunks that run
 * the same layer sequence on two HIP streams, the second chunk one recurrence behind the first (it needs that layer's GRU state and
 * the conv history rows the first chunk leaves): chunk 0's GEMMs of layer l+1 then fill the chip while chunk 1's scan of layer l
 * waits on its serial chain, and vice versa.  No extra passes, same kernels, same arithmetic per row: results are bit-identical to
 * the one-chunk order ($RADE_ENC_CHUNKS=1). */
/* The same pass with the concat buffer kept as matrix-core operand fragments (rade_enc.hip): taken when the call has enough rows for the batched GEMMs
 * (the float32-row path below serves short calls with k_gemm_splitk); the conv history crosses calls in enc_x's two float32 rows either way. */
static int encf_gemm(rade_batch *h, const dev_lin *w, int K1, int K0, int dil, float *y, long y_sb, long y_st, int ycol, int T, int act, void *stream)
{
    rd_encf_args g;
    memset(&g, 0, sizeof g);
    g.xf = h->enc_xf; g.NQ = h->enc_nq; g.B = h->B; g.T = T; g.K0 = K0; g.K1 = K1; g.dil = dil;
    g.Wp16 = w->wp16; g.Wscale = w->wscale16; g.bias = w->bias; g.N = w->N; g.act = act;
    g.y = y; g.y_sb = y_sb; g.y_st = y_st; if (!y) { g.yf = h->enc_xf; g.ycol = ycol; }
    g.seq_taps = h->enc_seq_taps; g.no_pair = h->enc_no_pair;
    if (K0 + K1 != w->K || !w->wp16) { fprintf(stderr, "rade: internal GEMM shape error (%d+%d != %d)\n", K0, K1, w->K); return -1; }
    PROF_BEGIN(h, stream);
    const int rc = rd_launch_encf_gemm(&g, stream);
    PROF_END(h, stream, RADE_PROF_GEMM, 2.0 * (double)h->B * T * (K0 + K1) * w->N);
    return rc;
}
static int encode_core_frag(rade_batch *h, int T, float *z, void *stream)
{
    const int B = h->B;
    /* history rows: float32 -> planes only when the float32 rows are the newer ones (after a reset or a short call).  After a fragment pass the tile already holds the
     * planes themselves: rebuilding them from 2^-8 (hi + lo) would re-split a pair whose low plane is exactly half an ulp of the high one differently (round to even) --
     * the same sum, other partial products, last-bit differences against the float32-row kernels in about one stream of 400 per call */
    int e = h->enc_hist_frag ? 0 : rd_launch_encf_hist(h->enc_xf, h->enc_nq, h->enc_x, (long)(2 + h->Tcap) * RD_ENC_W, B, T, 0, stream);
    {
        rd_encf_args g;
        memset(&g, 0, sizeof g);
        g.xf = h->enc_xf; g.NQ = h->enc_nq; g.B = B; g.T = T; g.bias = h->enc_dense1.bias; g.N = 64; g.act = 1; g.yf = h->enc_xf; g.ycol = 0;
        g.xin = h->enc_xin; g.Kin = h->enc_kpad; g.Wp = h->enc_dense1.wp;
        PROF_BEGIN(h, stream); e |= rd_launch_encf_dense1(&g, stream); PROF_END(h, stream, RADE_PROF_GEMM, 2.0 * (double)B * T * h->enc_kpad * 64);
    }
    const int unfused = h->enc_unfused;
    e |= encf_gemm(h, &h->enc_gin[0], ENC_IN[0], 0, 0, h->enc_gi, (long)T * 192, 192, 0, T, 0, stream);
    for (int l = 0; l < 5 && !e; l++) {
        const int in = ENC_IN[l], cin = in + 64;
        rd_scan_args s = { h->enc_gi, (long)T * 192, 192, h->enc_whh[l], h->enc_bhh[l], h->enc_h[l], NULL, 0, 0, NULL, 0, NULL, B, T, 64, h->enc_xf, h->enc_nq, in };
        PROF_BEGIN(h, stream); e |= rd_launch_gru_scan(&s, stream); PROF_END(h, stream, RADE_PROF_SCAN, 2.0 * B * T * 192 * 64);
        const dev_lin *nx = l < 4 ? &h->enc_gin[l + 1] : &h->enc_zdense;          /* what reads the conv's output next */
        if (unfused) {
            e |= encf_gemm(h, &h->enc_conv[l], cin, cin, ENC_DIL[l], NULL, 0, 0, cin, T, 1, stream);
            if (l < 4) e |= encf_gemm(h, nx, ENC_IN[l + 1], 0, 0, h->enc_gi, (long)T * 192, 192, 0, T, 0, stream);
            else e |= encf_gemm(h, nx, 864, 0, 0, z, (long)T * RD_LATENT, RD_LATENT, 0, T, h->bottleneck1 ? 1 : 0, stream);
            continue;
        }
        rd_encf_fused_args f;
        memset(&f, 0, sizeof f);
        f.xf = h->enc_xf; f.NQ = h->enc_nq; f.B = B; f.T = T; f.cin = cin; f.dil = ENC_DIL[l];
        f.Wc = h->enc_conv[l].wp16; f.Wc_scale = h->enc_conv[l].wscale16; f.Wc_bias = h->enc_conv[l].bias;
        f.Wg = nx->wp16; f.Wg_scale = nx->wscale16; f.Wg_bias = nx->bias; f.Ng = nx->N;
        if (l < 4) { f.y = h->enc_gi; f.y_sb = (long)T * 192; f.y_st = 192; }
        else { f.y = z; f.y_sb = (long)T * RD_LATENT; f.y_st = RD_LATENT; f.g_act = h->bottleneck1 ? 1 : 0; }
        if (h->enc_conv[l].K != 2 * cin || nx->K != cin + 96 || !f.Wc || !f.Wg) { fprintf(stderr, "rade: internal GEMM shape error (fused layer %d)\n", l); return -1; }
        PROF_BEGIN(h, stream);
        e |= rd_launch_encf_fused(&f, stream);
        PROF_END(h, stream, RADE_PROF_GEMM, 2.0 * (double)B * T * ((double)2 * cin * 96 + (double)(cin + 96) * nx->N));
    }
    e |= rd_launch_encf_hist(h->enc_xf, h->enc_nq, h->enc_x, (long)(2 + h->Tcap) * RD_ENC_W, B, T, 1, stream);
    h->enc_hist_frag = !e;
    return e;
}

static int encode_core(rade_batch *h, int T, float *z, void *stream)
{
    const int B = h->B, W = RD_ENC_W;
    if (h->enc_xf && (long)B * T > 16384 && h->enc_chunks != 2) return encode_core_frag(h, T, z, stream);
    h->enc_hist_frag = 0;
    const long xsb = (long)(2 + h->Tcap) * W;
    float *x0 = h->enc_x + 2 * W;              /* time row 0 of each stream; rows -2,-1 hold the conv history */
    hipStream_t S[2] = { (hipStream_t)stream, h->enc_side };
    /* small jobs (single-stream ABI, tests with a handful of rows) stay on one stream: nothing to overlap, and rade_tx() captures its
     * launches into a hipGraph on the caller's stream */
    const int nC = (h->enc_chunks == 2 && (long)B * T > 16384 && T >= 16) ? 2 : 1;
    const int tsplit = nC == 2 ? ((T / 2 + 3) & ~3) : T;
    const int t0[2] = { 0, tsplit }, tn[2] = { tsplit, T - tsplit };
    int e = 0;
    if (nC == 2) { if (hipEventRecord(h->ev_fork, S[0]) != hipSuccess || hipStreamWaitEvent(S[1], h->ev_fork, 0) != hipSuccess) return -1; }
    /* dense1 reads raw features: the one encoder operand that is not tanh-bounded, so it stays on the f32 matrix cores
     * (the 2^8-scaled binary16 planes of the split-f16 kernels overflow beyond +-255.9) */
    dev_lin d1 = h->enc_dense1; d1.wp16 = NULL; d1.wscale16 = NULL;
    for (int c = 0; c < nC; c++)
        e |= gemm(h, &d1, h->enc_xin + (long)t0[c] * h->enc_kpad, (long)T * h->enc_kpad, h->enc_kpad, h->enc_kpad, NULL, 0, 0, 0, NULL, NULL, x0 + (long)t0[c] * W, xsb, W, B, tn[c], 1, S[c]);
    for (int l = 0; l < 5 && !e; l++) {
        const int in = ENC_IN[l], cin = in + 64;
        for (int c = 0; c < nC && !e; c++) {
            float *x = x0 + (long)t0[c] * W, *gi = h->enc_gi + (long)t0[c] * 192;
            e |= gemm(h, &h->enc_gin[l], x, xsb, W, in, NULL, 0, 0, 0, NULL, NULL, gi, (long)T * 192, 192, B, tn[c], 0, S[c]);
            if (c == 1 && hipStreamWaitEvent(S[1], h->ev_scan[l], 0) != hipSuccess) return -1;      /* GRU state + history rows of chunk 0 */
            rd_scan_args s = { gi, (long)T * 192, 192, h->enc_whh[l], h->enc_bhh[l], h->enc_h[l], x + in, xsb, W, NULL, 0, NULL, B, tn[c], 64 };
            PROF_BEGIN(h, S[c]); e |= rd_launch_gru_scan(&s, S[c]); PROF_END(h, S[c], RADE_PROF_SCAN, 2.0 * B * tn[c] * 192 * 64);
            if (c == 0 && nC == 2 && hipEventRecord(h->ev_scan[l], S[0]) != hipSuccess) return -1;
            e |= gemm(h, &h->enc_conv[l], x, xsb, W, cin, x - (long)ENC_DIL[l] * W, xsb, W, cin, NULL, NULL, x + cin, xsb, W, B, tn[c], 1, S[c]);
        }
    }
    /* bottleneck 1: z = tanh(z_dense) (radae_base.py:281-284); bottleneck 3: linear */
    for (int c = 0; c < nC; c++)
        e |= gemm(h, &h->enc_zdense, x0 + (long)t0[c] * W, xsb, W, 864, NULL, 0, 0, 0, NULL, NULL, z + (long)t0[c] * RD_LATENT, (long)T * RD_LATENT, RD_LATENT, B, tn[c], h->bottleneck1 ? 1 : 0, S[c]);
    if (nC == 2) { if (hipEventRecord(h->ev_join, S[1]) != hipSuccess || hipStreamWaitEvent(S[0], h->ev_join, 0) != hipSuccess) return -1; }
    e |= rd_launch_carry_rows(h->enc_x, B, h->Tcap, W, 2, T, NULL, stream);
    return e;
}

/* the optional Tx band-pass filter + magnitude clip (radae_txe.py:130-132, :141-143) over the n samples the modulator left in tx_raw: the receiver's
 * filtering pass with frames as blocks (len0 = the first frame: 960, or the 1152-sample end-of-over frame), every sample consumed */
static int tx_bpf_pass_adv(rade_batch *h, int n, int len0, void *out, long out_stride, void *stream, int advance)
{
    rd_bpf_args ba;
    memset(&ba, 0, sizeof ba);
    ba.state = h->tx_bpf; ba.state_stride = sizeof(rd_bpf_state); ba.len0_const = len0; ba.avail_const = n;
    ba.tab = h->d_tab; ba.bpf16 = h->bpf16; ba.x = h->tx_raw; ba.x_stride = (long)(h->max_tx_mf > 2 ? h->max_tx_mf : 2) * RD_NMF; ba.y = out; ba.y_stride = out_stride;
    ba.chain = h->tx_chain; ba.chain_stride = h->max_tx_mf + 8; ba.n_blocks = 1 + (n > len0 ? (n - len0 + RD_NMF - 1) / RD_NMF : 0); ba.B = h->B; ba.clip = 1; ba.advance = advance;
    PROF_BEGIN(h, stream);
    const int rc = rd_launch_bpf(&ba, stream);
    PROF_END(h, stream, RADE_PROF_BPF, 8.0 * 101.0 * (double)h->B * n);
    return rc;
}
static int tx_bpf_pass(rade_batch *h, int n, int len0, void *out, long out_stride, void *stream) { return tx_bpf_pass_adv(h, n, len0, out, out_stride, stream, 1); }

/* ---- transmit (radae_txe.py:108-135 for n_mf modem frames and B streams at once) -------------- */
int rade_batch_tx(rade_batch *h, const float *features_dev, int n_mf, void *iq_out_dev, long iq_stride, float *z_out_dev, void *stream)
{
    ON_DEV(h);
    if (!h || n_mf <= 0 || n_mf > h->max_tx_mf || h->feat_in != 84) return -1;
    const int B = h->B, T = 3 * n_mf;
    float *z = z_out_dev ? z_out_dev : h->enc_z;
    int e = 0;
    e |= rd_launch_enc_pack(features_dev, h->enc_xin, B, T, stream);
    e |= encode_core(h, T, z, stream);
    void *mod_out = h->tx_bpf ? h->tx_raw : iq_out_dev; const long mod_stride = h->tx_bpf ? (long)(h->max_tx_mf > 2 ? h->max_tx_mf : 2) * RD_NMF : iq_stride;
    PROF_BEGIN(h, stream); e |= rd_launch_ofdm_mod(h->d_tab, z, mod_out, mod_stride, B, n_mf, stream); PROF_END(h, stream, RADE_PROF_MOD, 8.0 * B * n_mf * 5 * 30 * 160);
    if (h->tx_bpf) e |= tx_bpf_pass(h, n_mf * RD_NMF, RD_NMF, iq_out_dev, iq_stride, stream);
    return e ? -1 : n_mf * RD_NMF;
}

/* radae_txe.py --bypass_enc (:124-126): latents in, the modulator (+ Tx band-pass filter) as rade_batch_tx runs it */
int rade_batch_tx_latents(rade_batch *h, const float *z_dev, int n_mf, void *iq_out_dev, long iq_stride, void *stream)
{
    ON_DEV(h);
    if (!h || !z_dev || !iq_out_dev || n_mf <= 0 || n_mf > h->max_tx_mf) return -1;
    const int B = h->B;
    int e = 0;
    void *mod_out = h->tx_bpf ? h->tx_raw : iq_out_dev; const long mod_stride = h->tx_bpf ? (long)(h->max_tx_mf > 2 ? h->max_tx_mf : 2) * RD_NMF : iq_stride;
    PROF_BEGIN(h, stream); e |= rd_launch_ofdm_mod(h->d_tab, z_dev, mod_out, mod_stride, B, n_mf, stream); PROF_END(h, stream, RADE_PROF_MOD, 8.0 * B * n_mf * 5 * 30 * 160);
    if (h->tx_bpf) e |= tx_bpf_pass(h, n_mf * RD_NMF, RD_NMF, iq_out_dev, iq_stride, stream);
    return e ? -1 : n_mf * RD_NMF;
}

/* ---- core encoder / decoder alone (the rade_core_encoder / rade_core_decoder level, src/rade_core.h:42-46) ---- */
int rade_batch_encode(rade_batch *h, const float *features_dev, int n_steps, float *z_out_dev, void *stream)
{
    ON_DEV(h);
    if (!h || n_steps <= 0 || n_steps > h->Tcap || !z_out_dev) return -1;
    int e = rd_launch_pad_rows(features_dev, h->enc_xin, (long)h->B * n_steps, h->feat_in, h->enc_kpad, stream);
    e |= encode_core(h, n_steps, z_out_dev, stream);
    return e ? -1 : n_steps;
}

int rade_batch_tx_set_eoo_bits(rade_batch *h, const float *bits_host)
{
    ON_DEV(h);
    if (bits_host && hipMemcpy(h->eoo_bits, bits_host, sizeof(float) * h->B * RD_NEOOBITS, hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (rd_launch_eoo_build(h->d_tab, bits_host ? h->eoo_bits : NULL, h->eoo, h->B, NULL)) return -1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

int rade_batch_tx_eoo(rade_batch *h, void *iq_out_dev, long iq_stride, void *stream)
{
    ON_DEV(h);
    if (!h->tx_bpf) return rd_launch_copy_eoo(h->eoo, iq_out_dev, iq_stride, h->B, stream) ? -1 : RD_NEOO;
    int e = rd_launch_copy_eoo(h->eoo, h->tx_raw, (long)(h->max_tx_mf > 2 ? h->max_tx_mf : 2) * RD_NMF, h->B, stream);
    e |= tx_bpf_pass(h, RD_NEOO, RD_NEOO, iq_out_dev, iq_stride, stream);
    return e ? -1 : RD_NEOO;
}

/* ---- channel ----------------------------------------------------------------------------------- */
float rade_sigma_from_EbNodB(float EbNodB)
{   /* radae.py:567-573 (bottleneck 3): sigma = sqrt(Fs/(EbNo*Rb)), Rb = latent_dim/Tz */
    const float EbNo = powf(10.0f, EbNodB / 10.0f), Rb = (float)(80.0 / (0.01 * 4));
    return powf(8000.0f / (EbNo * Rb), 0.5f);
}

int rade_batch_channel(rade_batch *h, const void *tx_dev, long tx_stride, void *rx_out_dev, long rx_stride, const rade_channel_params *p, void *stream)
{
    ON_DEV(h);
    rd_chan_args a;
    memset(&a, 0, sizeof a);
    a.tab = h->d_tab; a.tx = tx_dev; a.tx_stride = tx_stride; a.rx = rx_out_dev; a.rx_stride = rx_stride; a.G = p->G_dev; a.noise = p->noise_dev;
    a.eoo = h->eoo; a.scratch = h->chan_scratch; a.B = h->B; a.n_sig = p->n_sig; a.n_pre = p->n_pre; a.n_post = p->n_post; a.with_eoo = p->with_eoo;
    a.sigma = p->sigma; a.freq_offset = p->freq_offset; a.df_dt = p->df_dt; a.seed = p->seed;
    a.sine_amp = p->sine_amp; a.sine_freq = p->sine_freq; a.rx_gain = p->rx_gain != 0.0f ? p->rx_gain : 1.0f;
    if (h->tx_bpf && p->with_eoo) {        /* what radae_tx --txbpf transmits after the last frame: the end-of-over frame through the Tx band-pass filter and the clip,
                                            * the filter state carried on from the frames before it (radae_txe.py:138-144) */
        /* a pure channel call: the filter state is READ, not advanced (two passes over the same tx -- two SNR points, a two-pass bench -- see the same EOO);
         * tx_raw is the modulator's scratch and holds nothing a caller can see */
        if (rd_launch_copy_eoo(h->eoo, h->tx_raw, (long)(h->max_tx_mf > 2 ? h->max_tx_mf : 2) * RD_NMF, h->B, stream) ||
            tx_bpf_pass_adv(h, RD_NEOO, RD_NEOO, h->eoo_filt, RD_NEOO, stream, 0)) return -1;
        a.eoo = h->eoo_filt;
    }
    PROF_BEGIN(h, stream);
    if (rd_launch_channel(&a, stream)) return -1;
    PROF_END(h, stream, RADE_PROF_CHAN, 0.0);
    return p->n_pre + p->n_sig + (p->with_eoo ? RD_NEOO : 0) + p->n_post;
}

/* ---- transmit + channel in one pass (RADAE.forward, radae.py:529-589: latents -> OFDM -> multipath -> noise) -------------------
 * The modulator applies the two-path model while a frame's samples are in LDS and leaves the power sums, so tx is never re-read and
 * k_chan_power does not run; without G (AWGN only) it is the two calls back to back. */
int rade_batch_tx_channel(rade_batch *h, const float *features_dev, int n_mf, void *iq_out_dev, long iq_stride, void *rx_out_dev, long rx_stride,
                          const rade_channel_params *p, void *stream)
{
    ON_DEV(h);
    if (!h || !p || n_mf <= 0 || n_mf > h->max_tx_mf || h->feat_in != 84 || p->n_sig != n_mf * RD_NMF || !rx_out_dev) return -1;
    if (!p->G_dev || h->tx_bpf) {          /* (the Tx band-pass filter sits between the modulator and the channel: the two calls back to back) */
        if (!iq_out_dev) return -1;
        if (rade_batch_tx(h, features_dev, n_mf, iq_out_dev, iq_stride, NULL, stream) != p->n_sig) return -1;
        return rade_batch_channel(h, iq_out_dev, iq_stride, rx_out_dev, rx_stride, p, stream);
    }
    if (!h->chan_mp && !(h->chan_mp = dev_zeros(sizeof(float) * 2 * (size_t)h->B * h->max_tx_mf * RD_NMF))) return -1;
    const int B = h->B, T = 3 * n_mf;
    int e = rd_launch_enc_pack(features_dev, h->enc_xin, B, T, stream);
    e |= encode_core(h, T, h->enc_z, stream);
    PROF_BEGIN(h, stream);
    e |= rd_launch_ofdm_mod_mp(h->d_tab, h->enc_z, iq_out_dev, iq_stride, B, n_mf, p->G_dev, h->chan_mp, (double *)h->chan_scratch + 2 * (size_t)B, stream);
    PROF_END(h, stream, RADE_PROF_MOD, 8.0 * B * n_mf * 5 * 30 * 160);
    if (e) return -1;
    rd_chan_args a;
    memset(&a, 0, sizeof a);
    a.tab = h->d_tab; a.tx = NULL; a.tx_stride = 0; a.rx = rx_out_dev; a.rx_stride = rx_stride; a.G = p->G_dev; a.noise = p->noise_dev; a.mp = h->chan_mp;
    a.eoo = h->eoo; a.scratch = h->chan_scratch; a.B = B; a.n_sig = p->n_sig; a.n_pre = p->n_pre; a.n_post = p->n_post; a.with_eoo = p->with_eoo;
    a.sigma = p->sigma; a.freq_offset = p->freq_offset; a.df_dt = p->df_dt; a.seed = p->seed;
    a.sine_amp = p->sine_amp; a.sine_freq = p->sine_freq; a.rx_gain = p->rx_gain != 0.0f ? p->rx_gain : 1.0f;
    PROF_BEGIN(h, stream);
    if (rd_launch_channel(&a, stream)) return -1;
    PROF_END(h, stream, RADE_PROF_CHAN, 0.0);
    return p->n_pre + p->n_sig + (p->with_eoo ? RD_NEOO : 0) + p->n_post;
}

int rade_batch_multipath_gen(rade_batch *h, const float *fir_taps_host, int n_taps, int low_ratio, int n_out,
                             const void *noise_low_dev, unsigned long long seed, void *G_out_dev, void *stream)
{
    ON_DEV(h);
    if (!h || !fir_taps_host || n_taps <= 0 || n_taps > 1024 || !G_out_dev) return -1;
    float *taps = dev_upload(fir_taps_host, sizeof(float) * n_taps);
    if (!taps) return -1;
    void *ybuf = NULL;                                  /* more low-rate points than the kernel keeps in LDS (lmr60: Fs / 16): they live in HBM for the call */
    if (rd_multipath_gen_needs_scratch(low_ratio, n_out)) {
        const size_t n_low = (size_t)(n_out + low_ratio - 1) / low_ratio;
        if (hipMalloc(&ybuf, sizeof(double) * 2 * 2 * n_low * h->B) != hipSuccess) { hipFree(taps); return -1; }
    }
    const int rc = rd_launch_multipath_gen(taps, n_taps, low_ratio, n_out, noise_low_dev, seed, G_out_dev, ybuf, h->B, stream);
    hipStreamSynchronize((hipStream_t)stream);          /* the tap buffer is released right away */
    hipFree(taps);
    if (ybuf) hipFree(ybuf);
    return rc ? -1 : n_out;
}

/* multipath_samples.m:33-40: the rate-Rs channel matrix from rate-Fs Doppler samples */
int rade_batch_multipath_h(rade_batch *h, const void *G_dev, int n_g, int fs_over_rs, int n_sym, int Nc, float delay_s, float Rs, int want_complex, float *H_out_dev, void *stream)
{
    ON_DEV(h);
    if (!h || !G_dev || !H_out_dev) return -1;
    return rd_launch_multipath_h(G_dev, n_g, fs_over_rs, n_sym, Nc, delay_s * Rs, want_complex, H_out_dev, h->B, stream) ? -1 : n_sym;
}

/* ---- receive ----------------------------------------------------------------------------------- */
/* CoreDecoderStatefull.forward (radae_base.py:400-416) over T time slots of every stream.  x = [B][1+Tcap][736]
 * (slot 0 = conv history), rows beyond n_rows[b] are skipped, rst flags zero the state before a step. */
static int decoder_layers(rade_batch *h, const float *z, int T, int Tio, int Tcap, float *xbuf, float *gi, float *hbuf, float **hstate,
                          const int *nr, const int *rst, float *out, void *stream)
{
    const int B = h->B, W = RD_DEC_W;
    const long xsb = (long)(1 + Tcap) * W;
    float *x = xbuf + W;
    int e = 0;
    dev_lin d1 = h->dec_dense1; d1.wp16 = NULL; d1.wscale16 = NULL;       /* z_hat is unbounded: f32 matrix cores (see encode_core) */
    e |= gemm(h, &d1, z, (long)Tio * RD_LATENT, RD_LATENT, RD_LATENT, NULL, 0, 0, 0, NULL, nr, x, xsb, W, B, T, 1, stream);
    for (int l = 0; l < 5 && !e; l++) {
        const int in = DEC_IN[l];
        e |= gemm(h, &h->dec_gin[l], x, xsb, W, in, NULL, 0, 0, 0, NULL, nr, gi, (long)T * 288, 288, B, T, 0, stream);
        rd_scan_args s = { gi, (long)T * 288, 288, h->dec_whh[l], h->dec_bhh[l], hstate[l], hbuf, (long)T * 96, 96, rst, h->dec_rows, nr, B, T, 96 };
        PROF_BEGIN(h, stream); e |= rd_launch_gru_scan(&s, stream); PROF_END(h, stream, RADE_PROF_SCAN, 2.0 * B * T * 288 * 96);
        e |= gemm(h, &h->dec_glu[l], hbuf, (long)T * 96, 96, 96, NULL, 0, 0, 0, NULL, nr, x + in, xsb, W, B, T, 2, stream);
        const int cin = in + 96;
        e |= gemm(h, &h->dec_conv[l], x, xsb, W, cin, x - W, xsb, W, cin, rst, nr, x + cin, xsb, W, B, T, 1, stream);
    }
    e |= gemm(h, &h->dec_output, x, xsb, W, 736, NULL, 0, 0, 0, NULL, nr, out, (long)Tio * h->feat_in, h->feat_in, B, T, 0, stream);
    return e;
}

/* pointers of the receiver's in-kernel decoder stage (rx_decode_pending -> ds_layers) */
static void fill_dec_args(const rade_batch *h, rd_decs_args *d)
{
    memset(d, 0, sizeof *d);
    const long DR = h->dec_rows;
    d->z = h->zrows; d->z_sb = DR * RD_LATENT; d->x = h->dec_x + RD_DEC_W; d->x_sb = (1 + DR) * RD_DEC_W;
    d->gi = h->dec_gi; d->gi_sb = DR * 288; d->hbuf = h->dec_hbuf; d->hb_sb = DR * 96;
    d->out = h->feat84; d->out_sb = DR * h->feat_in; d->out_w = h->feat_in;
    d->B = h->B;
#define LIN(dst, src) do { (dst).wp = (src).wp; (dst).bias = (src).bias; (dst).wp16 = (src).wp16; (dst).wa16 = (src).wa16; (dst).wscale = (src).wscale; (dst).N = (src).N; (dst).K = (src).K; } while (0)
    LIN(d->dense1, h->dec_dense1); LIN(d->output, h->dec_output);
    for (int l = 0; l < 5; l++) { LIN(d->gin[l], h->dec_gin[l]); LIN(d->glu[l], h->dec_glu[l]); LIN(d->conv[l], h->dec_conv[l]); d->whh[l] = h->dec_whh[l]; d->bhh[l] = h->dec_bhh[l]; d->h[l] = h->dec_h[l];
                                  d->whq[l] = (h->dec_whq[l] && h->dec_whs[l]) ? h->dec_whq[l] : NULL; d->whs[l] = h->dec_whs[l]; }
#undef LIN
}

int rade_batch_decode(rade_batch *h, const float *z_dev, int n_steps, float *features_out_dev, int reset_state, void *stream)
{
    ON_DEV(h);
    if (!h || n_steps <= 0 || n_steps > h->Tcap || !features_out_dev) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (reset_state) {
        hipMemsetAsync(h->dec2_h[0], 0, sizeof(float) * 5 * h->B * 96, st);
        hipMemsetAsync(h->dec2_x, 0, sizeof(float) * (size_t)h->B * (1 + h->Tcap) * RD_DEC_W, st);
    }
    int e = decoder_layers(h, z_dev, n_steps, n_steps, h->Tcap, h->dec2_x, h->dec2_gi, h->dec2_hbuf, h->dec2_h, NULL, NULL, features_out_dev, stream);
    e |= rd_launch_carry_rows(h->dec2_x, h->B, h->Tcap, RD_DEC_W, 1, n_steps, NULL, stream);
    return e ? -1 : n_steps;
}

int rade_batch_channel_symbol(rade_batch *h, const float *z_dev, const float *H_dev, const float *noise_dev, float *z_hat_dev, int n_steps,
                              int mode, float p0, float p1, unsigned long long seed, void *stream)
{
    ON_DEV(h);
    if (!h || n_steps <= 0 || (mode != 0 && mode != 1)) return -1;
    return rd_launch_chan_symbol(z_dev, H_dev, noise_dev, z_hat_dev, (long)h->B * n_steps * RD_LATENT, mode, p0, p1, seed, stream) ? -1 : n_steps;
}

/* The wait of rade_batch_rx when the host is short of CPUs (sync_blocking_now): SLEEP until the receiver launch is done.  hipEventSynchronize on a
 * hipEventBlockingSync event does not do that on this runtime -- measured (tools/host_threads_cpu.py, round 5): a lane thread sitting in it burns its whole
 * wall time, 3.0 cores busy for three engines against 3.8 with hipStreamSynchronize -- so the thread sleeps itself: through three quarters of what the last
 * measured waits took (a receiver launch lasts milliseconds and as long as the one before it), then in short naps between hipEventQuery calls.  A nap costs a wake-up,
 * not a core; the launch's end is noticed at most one nap (plus the timer slack) late, which the other engines' batches in flight cover. */
static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec; }
static void nap_us(double us) { if (us > 0.0) { struct timespec t = { (time_t)(us / 1e6), (long)(1e3 * (us - 1e6 * (double)(time_t)(us / 1e6))) }; nanosleep(&t, NULL); } }
/* The estimate must not feed on its own nap: a wait that ends INSIDE the long nap says only "the launch took at most the nap", so the estimate is halved (not
 * averaged with a dt that is mostly the nap -- that form decayed 6 % per call and an outlier cost seconds of oversleeping); only a wait whose first query after the nap
 * still found the launch running measures the launch, and only those are averaged in, an outlier counted as at most 8x the present estimate.  The long nap is capped
 * (a receiver launch lasts milliseconds), and the engine's first wait -- lazy code-object load included -- does not seed the estimate. */
#define RADE_NAP_CAP_US 20000.0
static int sleep_until_event(rade_batch *h, hipEvent_t ev)
{
    const double t0 = now_us();
    int overslept = 0;
    hipError_t e = hipEventQuery(ev);
    if (e == hipErrorNotReady && h->wait_est_us > 150.0) {
        const double nap = 0.75 * h->wait_est_us;
        nap_us(nap < RADE_NAP_CAP_US ? nap : RADE_NAP_CAP_US); e = hipEventQuery(ev); overslept = e == hipSuccess;
    }
    while (e == hipErrorNotReady) { nap_us(40.0); e = hipEventQuery(ev); }
    if (e != hipSuccess) { fprintf(stderr, "rade: %s while waiting for the receiver launch\n", hipGetErrorString(e)); return -1; }
    double dt = now_us() - t0;
    if (overslept) h->wait_est_us *= 0.5;
    else if (h->n_sync_block > 0) {
        if (h->wait_est_us > 0.0 && dt > 8.0 * h->wait_est_us) dt = 8.0 * h->wait_est_us;
        h->wait_est_us = h->wait_est_us > 0.0 ? 0.75 * h->wait_est_us + 0.25 * dt : dt;
    }
    return 0;
}
/* test hook (tests/test_host_cpu.py): the estimate's update rule driven with synthetic waits -- `ready_after_us` is when the launch really ends; returns the
 * time this wait would have kept the thread beyond that (the oversleep) and updates *est like sleep_until_event does */
double rade_wait_model_step(double *est, double ready_after_us, int first)
{
    double t = 0.0; int overslept = 0;
    if (ready_after_us > 0.0 && *est > 150.0) { const double nap = 0.75 * *est; t = nap < RADE_NAP_CAP_US ? nap : RADE_NAP_CAP_US; overslept = t >= ready_after_us; }
    while (t < ready_after_us) t += 40.0;
    double dt = t;
    if (overslept) *est *= 0.5;
    else if (!first) { if (*est > 0.0 && dt > 8.0 * *est) dt = 8.0 * *est; *est = *est > 0.0 ? 0.75 * *est + 0.25 * dt : dt; }
    return t - ready_after_us;
}

int rade_batch_rx(rade_batch *h, const void *rx_dev, long rx_stride, const int *n_avail_host, int max_calls,
                  float *features_out_dev, long feat_stride, float *eoo_out_dev, rade_rx_status *status_host, void *stream)
{
    ON_DEV(h);
    const int row_floats = h && h->bypass_dec ? RD_ZMF : RD_FEAT_MF;       /* RADE_BATCH_BYPASS_DEC: 240 latents per valid modem frame instead of 432 feature floats */
    if (!h || !n_avail_host || max_calls <= 0 || h->feat_in != 84 || !features_out_dev || feat_stride < row_floats) return -1;
    const int B = h->B;
    hipStream_t st = (hipStream_t)stream;
    int *hs = h->h_small;
    /* complex_bpf.bpf for every sample of this invocation, ahead of the receiver launches (rade_rx.hip: k_rx_bpf).  The buffers follow the
     * largest invocation seen (a few times the input: 8 bytes per sample and stream); growing them waits for the device. */
    int max_avail = 0;
    for (int b = 0; b < B; b++) if (n_avail_host[b] > max_avail) max_avail = n_avail_host[b];
    if (max_avail > 0 && !rx_dev) return -1;
    const int n_blocks = max_avail > 0 ? 2 + (max_avail - 1) / 800 : 0;          /* blocks of >= 800 samples (a first block after a slip), the tail included */
    if (max_avail > h->filt_cap || n_blocks + 3 > h->chain_stride) {
        CHK(hipDeviceSynchronize());
        if (h->rx_filt) hipFree(h->rx_filt);
        if (h->bpf_chain) hipFree(h->bpf_chain);
        h->filt_cap = ((long)max_avail + 1023) & ~1023L; h->chain_stride = (int)(h->filt_cap / 800) + 8;
        /* (no memset: a hipMemset on the null stream is not ordered against the caller's non-blocking stream and could land on top of the
         * pre-pass's results; every entry that is read is written by the pre-pass first) */
        h->rx_filt = NULL; h->bpf_chain = NULL;
        if (hipMalloc(&h->rx_filt, sizeof(float) * 2 * (size_t)B * h->filt_cap) != hipSuccess) h->rx_filt = NULL;
        if (hipMalloc((void **)&h->bpf_chain, sizeof(float) * 2 * (size_t)B * h->chain_stride) != hipSuccess) h->bpf_chain = NULL;
        if (!h->rx_filt || !h->bpf_chain) { h->filt_cap = 0; h->chain_stride = 0; fprintf(stderr, "rade: device allocation failed (receiver pre-pass buffers)\n"); return -1; }
    }
    CHK(hipMemcpyAsync(h->rx_avail, n_avail_host, sizeof(int) * B, hipMemcpyHostToDevice, st));
    PROF_BEGIN(h, st);
    {
        rd_bpf_args ba;
        memset(&ba, 0, sizeof ba);
        ba.state = &h->rx_st->bpf; ba.state_stride = sizeof(rd_rx_stream); ba.len0 = &h->rx_st->nin; ba.len0_stride = sizeof(rd_rx_stream); ba.avail = h->rx_avail;
        ba.tab = h->d_tab; ba.bpf16 = h->bpf16; ba.x = rx_dev; ba.x_stride = rx_stride; ba.y = h->rx_filt; ba.y_stride = h->filt_cap;
        ba.chain = h->bpf_chain; ba.chain_stride = h->chain_stride; ba.n_blocks = n_blocks; ba.B = B;
        ba.zero_acc = h->rx_acc; ba.zero_progress = h->rx_progress;        /* the invocation's counters are cleared by the pre-pass's first kernel */
        if (rd_launch_bpf(&ba, st)) goto fail;
    }
    PROF_END(h, st, RADE_PROF_BPF, 8.0 * 101.0 * (double)B * max_avail);
    int counters_clear = n_blocks > 0;
    if (!counters_clear) CHK(hipMemsetAsync(h->rx_acc, 0, sizeof(int) * B * 4, st));
    rd_sync_args sa;
    memset(&sa, 0, sizeof sa);
    sa.tab = h->d_tab; sa.st = h->rx_st; sa.round = h->rx_round; sa.rx = rx_dev; sa.rx_stride = rx_stride; sa.rxf = h->rx_filt; sa.rxf_stride = h->filt_cap; sa.bpf16 = h->bpf16; sa.bpf_chain = h->bpf_chain; sa.chain_stride = h->chain_stride; sa.avail = h->rx_avail; sa.acc = h->rx_acc;
    sa.max_calls = max_calls; sa.round_calls = h->R; sa.dec_rows = h->dec_rows; sa.unsync_off_after = h->unsync_off_after;
    sa.corr16 = h->corr16; sa.corrq16 = h->corrq16; sa.corra16 = h->corra16; sa.zrows = h->zrows; sa.status = h->rx_status; sa.eoo_out = eoo_out_dev; sa.dtcache = h->dtcache;
    sa.trace = h->trace; sa.trace_z = h->trace_z; sa.trace_cap = h->trace_cap; sa.progress = h->rx_progress; sa.wg_cycles = h->wg_cycles; sa.B = B; sa.vm = h->vm; sa.wfwd16 = h->wfwd16; sa.variant = h->rx_census << 8; sa.lds_bytes = h->rx_lds;
    fill_dec_args(h, &sa.dec); sa.features_out = features_out_dev; sa.feat_stride = feat_stride;
    sa.bypass_dec = h->bypass_dec;
    sa.feat_cap = (int)(feat_stride / row_floats);      /* the kernel never writes past the caller's rows: a stream pauses once its buffer is full (status.consumed tells how far it got) */
    /* one launch normally takes every stream through all of its samples (calls, decoder, output); the loop only
     * continues when a stream ran into the per-launch call limit */
    for (;;) {
        if (!counters_clear) CHK(hipMemsetAsync(h->rx_progress, 0, sizeof(int) * 4, st));
        counters_clear = 0;
        PROF_BEGIN(h, st);
        if (rd_launch_rx_sync(&sa, st)) goto fail;
        PROF_END(h, st, RADE_PROF_SYNC, 0.0);
        /* progress word and (normally final) per-stream results come back in one transfer (one device block in the host copy's order) */
        CHK(hipMemcpyAsync(hs, h->rx_progress, sizeof(int) * (status_host ? 8 + 8 * (size_t)B : 4), hipMemcpyDeviceToHost, st));
        if (sync_blocking_now()) { CHK(hipEventRecord(h->ev_block, st)); if (sleep_until_event(h, h->ev_block)) goto fail; h->n_sync_block++; }
        else { CHK(hipStreamSynchronize(st)); h->n_sync_spin++; }
        if (hs[0] == 0 || hs[1] == 0) break;    /* nothing done, or no stream stopped at the per-launch limit */
    }
    if (status_host) {
        const int *acc = hs + 8, *sts = hs + 8 + 4 * B;
        for (int b = 0; b < B; b++) {
            rade_rx_status *s = status_host + b;
            s->consumed = acc[4 * b]; s->n_calls = acc[4 * b + 1]; s->n_valid = acc[4 * b + 2]; s->has_eoo = acc[4 * b + 3] > 0;
            s->nin = sts[4 * b]; s->sync = sts[4 * b + 1]; s->snr_dB = sts[4 * b + 2]; s->state = sts[4 * b + 3];
        }
    }
    return 0;
fail:
    return -1;
}

/* how many of this engine's rade_batch_rx waits slept / spun (measurement aid) */
void rade_batch_sync_counts(const rade_batch *h, long *blocking, long *spinning) { if (blocking) *blocking = h->n_sync_block; if (spinning) *spinning = h->n_sync_spin; }

/* shader-clock cycles every stream's workgroup spent in the most recent receiver launch (measurement aid: the launch lasts as
 * long as its slowest stream) */
int rade_batch_rx_stream_cycles(rade_batch *h, long long *out_host)
{
    if (!h || !out_host || !h->wg_cycles) return -1;
    ON_DEV(h);
    return hipMemcpy(out_host, h->wg_cycles, sizeof(long long) * h->B, hipMemcpyDeviceToHost) == hipSuccess ? h->B : -1;
}

/* test / measurement aid: the band-pass filtered samples (complex_bpf.bpf, dsp.py:63-102) the most recent rade_batch_rx invocation's receiver read
 * for stream b, first n of them -> out_host [n] complex64; returns the count copied */
int rade_batch_rx_filtered(rade_batch *h, int b, void *out_host, int n)
{
    if (!h || b < 0 || b >= h->B || !out_host || !h->rx_filt) return -1;
    ON_DEV(h);
    if (n > h->filt_cap) n = (int)h->filt_cap;
    if (n <= 0) return 0;
    return hipMemcpy(out_host, (const char *)h->rx_filt + sizeof(float) * 2 * (size_t)b * h->filt_cap, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? n : -1;
}

int rade_batch_rx_get_trace(rade_batch *h, int b, rade_rx_trace *out, float *z_hat_out, int max_calls)
{
    ON_DEV(h);
    if (!h->trace || b < 0 || b >= h->B) return -1;
    rd_rx_stream *tmp = malloc(sizeof *tmp);
    hipMemcpy(tmp, h->rx_st + b, sizeof *tmp, hipMemcpyDeviceToHost);
    int n = tmp->mf - 1;
    free(tmp);
    if (n > h->trace_cap) n = h->trace_cap;
    if (n > max_calls) n = max_calls;
    if (n <= 0) return 0;
    if (out) hipMemcpy(out, h->trace + (size_t)b * h->trace_cap, sizeof(rd_rx_trace) * n, hipMemcpyDeviceToHost);
    if (z_hat_out) hipMemcpy(z_hat_out, h->trace_z + (size_t)b * h->trace_cap * RD_ZMF, sizeof(float) * n * RD_ZMF, hipMemcpyDeviceToHost);
    return n;
}
