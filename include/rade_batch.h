/*
 * rade_batch.h -- batched (many independent streams) form of the RADE hot path, C ABI.
 *
 * Additive extension of rade_api.h: the single-stream API cannot express "256 utterances on one
 * MI355X", which is the workload BASELINE.json names.  Every call processes the same step for all
 * B streams; streams never exchange data, so sharding across GPUs is one
 * engine per device with a disjoint set of streams.
 *
 * All *_dev pointers are device (HBM) addresses; `stream` is a hipStream_t passed as void*
 * (NULL = the legacy default stream).  No torch / C++ types cross this boundary.
 *
 * Reference behaviour implemented per entry point:
 *   rade_batch_tx          radae_txe.py:108-135 (do_radae_tx) + rade_api.c:403-445, n_mf modem frames at once,
 *                          encoder state carried across calls (radae_base.py:97-129)
 *   rade_batch_tx_eoo      radae_txe.py:138-144, radae.py:208-219, :441-455
 *   rade_batch_channel     radae.py:529-589 (rate-Fs multipath, offsets, AWGN; power normalised per
 *                          stream = the reference's batch-1 behaviour) + inference.py:263-284 (EOO / noise framing)
 *   rade_batch_rx          radae_rxe.py:171-330 (do_radae_rx) looped like radae_rxe.py:349-356 /
 *                          src/radae_rx.c:42-53, incl. rade_api.c:480-513 decoder + UW accounting
 */
#ifndef RADE_BATCH_H
#define RADE_BATCH_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rade_batch rade_batch;
#define RADE_BATCH_BOTTLENECK1 0x100
/* Accepted and ignored since round 4: the receiver kernel (k_rx_sync2: 256 threads and at most 80 KB of LDS per stream, so that the
 * workgroups of two launches share a CU) is the only one; rounds 2-3 selected it with this flag beside a one-stream-per-CU kernel. */
#define RADE_BATCH_RX_TWO_PER_CU 0x200

/* Tx band-pass filter + magnitude clip on every transmitted frame and the end-of-over frame: radae_tx(..., txbpf_en=True) / `radae_tx.py --txbpf`
 * (radae_txe.py:74-83, :130-132, :141-143; ctest radae_tx_basic).  Off by default, as in the reference. */
#define RADE_BATCH_TX_BPF 0x400

/* Receiver without the core decoder: `radae_rxe.py --bypass_dec` (radae_rxe.py:67, :113-123, :300-302, :315), the mode the reference's rade_api.c drives when an
 * external C decoder follows (rade_api.c:491-513).  Every valid modem frame appends its 240 equalised latents (3 x 80, what receiver_one returned) to
 * features_out instead of 432 feature floats -- rows are 240 floats, feat_stride / 240 is the capacity -- and, as in the reference, the aux-bit (UW) errors are
 * never summed in this mode (sum_uw_errors sits behind the decoder), so a UW failure cannot end sync; everything else of do_radae_rx is unchanged. */
#define RADE_BATCH_BYPASS_DEC 0x800

typedef struct {
    int n_streams;        /* B */
    int max_tx_mf;        /* largest n_mf a single rade_batch_tx call may carry */
    int device;           /* HIP device ordinal */
    int flags;            /* RADE_FOFF_TEST from rade_api.h is honoured; RADE_BATCH_BOTTLENECK1: z = tanh(.) (model05, bbfm); RADE_BATCH_TX_BPF; RADE_BATCH_BYPASS_DEC */
    int rx_trace_calls;   /* >0: keep a per-call trace of this many do_radae_rx calls per stream (tests) */
    float disable_unsync; /* test mode of radae_rxe.py --disable_unsync (:277-281, :337): after this many seconds in sync the receiver no longer
                           * drops back to search (pilot loss, end-of-over, UW failure); 0 = normal operation */
} rade_batch_config;

/* blob = DNNw weight file (weights/model19_check3.bin).  Returns NULL on failure (message on stderr). */
rade_batch *rade_batch_open(const char *blob_path, const rade_batch_config *cfg);
rade_batch *rade_batch_open_mem(const void *blob, size_t blob_len, const rade_batch_config *cfg);
void rade_batch_close(rade_batch *h);
int rade_batch_n_streams(const rade_batch *h);

/* Arithmetic of the encoder depends on the SIZE of the call, in the last bits only: calls with more than 16384 rows (B x 3 n_mf) run the batched kernels on operand
 * fragments (rade_enc.hip), whose two conv taps alternate per k-block -- another summation order of the same float32 partial products than the float32-row kernels that
 * serve smaller calls ($RADE_ENCF_SEQ_TAPS restores the sequential order, $RADE_ENC_ROWS the row kernels for every size).  Both are inside every parity bar (latents
 * < 2e-5 of full scale against the float32 oracle, tests/test_hip_parity.py), but the same stream encoded in a batch of 32 and in a batch of 256 is not bit-identical.
 * The batched path also keeps one bit of HOST state per engine (whether the history tile of its fragment buffer is current): rade_batch_tx / rade_batch_encode are
 * stream-ordered but not capturable into a hipGraph that is replayed across resets or mixed with short calls (rade_tx() of rade_api.h, which IS captured, always takes
 * the row path). */
/* ---- transmit ------------------------------------------------------------------------------
 * features_dev : [B][n_mf*12][36] float32 (first 20 of each 36 used; aux symbol -1 added inside)
 * iq_out_dev   : stream b written at iq_out_dev + b*iq_stride (units: complex samples), n_mf*960 samples
 * z_out_dev    : optional [B][n_mf*3][80] latents (NULL to skip)
 * returns n_mf*960 or <0 on error */
int rade_batch_tx(rade_batch *h, const float *features_dev, int n_mf, void *iq_out_dev, long iq_stride,
                  float *z_out_dev, void *stream);
/* `radae_txe.py --bypass_enc` (radae_txe.py:124-126; rade_api.c with RADE_USE_C_ENCODER): latents from an external core encoder, z_dev [B][3 n_mf][80] float32,
 * straight into the OFDM modulator (and the Tx band-pass filter when the engine has it).  The engine's own encoder state is not touched.  Returns n_mf * 960. */
int rade_batch_tx_latents(rade_batch *h, const float *z_dev, int n_mf, void *iq_out_dev, long iq_stride, void *stream);
/* bits_host: [B][180] +-1 floats, or NULL to restore the default (all-zero data symbols) EOO frame */
int rade_batch_tx_set_eoo_bits(rade_batch *h, const float *bits_host);
/* writes the 1152-sample end-of-over frame of every stream; returns 1152 */
int rade_batch_tx_eoo(rade_batch *h, void *iq_out_dev, long iq_stride, void *stream);
void rade_batch_tx_reset(rade_batch *h);

/* ---- core encoder / decoder alone ---------------------------------------------------------------
 * The rade_core_encoder / rade_core_decoder level (src/rade_core.h:42-46, test_rade_enc.c / test_rade_dec.c),
 * and what the non-OFDM configurations (inference.py rate-Rs, bbfm.py) run.  The feature width comes from
 * the blob: 4 x 21 (model19: caller supplies the aux symbol) or 4 x 20 (model05, bbfm).
 * features_dev: [B][n_steps][4*feat_dim] -> z_out_dev [B][n_steps][80]; state carried across calls
 * (rade_batch_tx_reset clears it).  rade_batch_decode: z [B][n_steps][80] -> features [B][n_steps][4*feat_dim]. */
int rade_batch_encode(rade_batch *h, const float *features_dev, int n_steps, float *z_out_dev, void *stream);
int rade_batch_decode(rade_batch *h, const float *z_dev, int n_steps, float *features_out_dev, int reset_state, void *stream);
/* symbol-domain channels: mode 0 = rate-Rs AWGN/multipath magnitudes (radae.py:604-634; H_dev [B][n_steps*40] per
 * QPSK symbol or NULL, p0 = sigma); mode 1 = BBFM FM-demodulator SNR model (bbfm.py:157-197; H_dev [B][n_steps*80]
 * or NULL, p0 = CNRdB, p1 = Gfm dB).  noise_dev: [B][n_steps*80] float32 (already scaled per component) or NULL -> Philox(seed) */
int rade_batch_channel_symbol(rade_batch *h, const float *z_dev, const float *H_dev, const float *noise_dev, float *z_hat_dev, int n_steps,
                              int mode, float p0, float p1, unsigned long long seed, void *stream);

/* ---- channel simulator ---------------------------------------------------------------------- */
typedef struct {
    int n_sig;            /* signal samples per stream (multiple of 960) */
    int n_pre, n_post;    /* noise-only samples before / after (inference.py --prepend_noise/--append_noise) */
    int with_eoo;         /* append the stream's EOO frame, phase-continued (inference.py --end_of_over).  On an engine with RADE_BATCH_TX_BPF the appended frame is the EOO
                           * through the Tx band-pass filter + clip, continuing from the filter state the transmitted frames left (radae_txe.py:138-144); the channel call
                           * only READS that state (calling it twice gives the same samples).  Do not call rade_batch_tx_eoo first: that advances the state past the EOO. */
    float sigma;          /* AWGN std-dev, see rade_sigma_from_EbNodB */
    float freq_offset;    /* Hz */
    float df_dt;          /* Hz/s */
    const void *G_dev;    /* [B][n_sig][2] complex64 Doppler samples (G1,G2) or NULL = (1,0) */
    const void *noise_dev;/* [B][n_total] complex64, unit variance, or NULL: generate (Philox) from seed; seed 0 = no noise.
                           * The generated sequence is a property of the BUILD, not of the seed alone: Philox4x32-10 keyed by (seed, stream) with one counter per
                           * PAIR of samples (words 0-1 -> sample 2p, 2-3 -> sample 2p + 1; round 3: one counter per sample), Box-Muller on the hardware
                           * log2 / sqrt / sin / cos units (round 4).  Seeded statistics are reproducible within a build; anything that must be comparable
                           * across builds (parity tests) passes noise_dev. */
    unsigned long long seed;
    float sine_amp, sine_freq;   /* complex tone added over the whole output, inference.py:285-288 (--sine_amp/--sine_freq); 0 = none */
    float rx_gain;               /* final scale, inference.py:289 (--rx_gain); 0 is taken as 1 */
} rade_channel_params;
/* n_total = n_pre + n_sig + (with_eoo ? 1152 : 0) + n_post samples written per stream; returns n_total */
int rade_batch_channel(rade_batch *h, const void *tx_dev, long tx_stride, void *rx_out_dev, long rx_stride,
                       const rade_channel_params *p, void *stream);

/* transmit and channel in one pass (what RADAE.forward does, radae.py:529-589): features [B][12 n_mf][36] -> received samples, with
 * p->n_sig == 960 n_mf.  With p->G_dev the modulator applies the two-path model itself and leaves the power sums (no second pass over
 * tx and G); iq_out_dev (the clean transmit samples) is then optional.  Returns n_total like rade_batch_channel, or <0. */
int rade_batch_tx_channel(rade_batch *h, const float *features_dev, int n_mf, void *iq_out_dev, long iq_stride, void *rx_out_dev, long rx_stride,
                          const rade_channel_params *p, void *stream);

/* ---- Watterson / Doppler-spread sample generator on the device (doppler_spread.m:7-50, multipath_samples.m:10-31):
 * per stream two independent paths G1, G2 = complex Gaussian noise at the low rate Fs/low_ratio through the
 * n_taps Gaussian-PSD FIR (taps designed by the caller, e.g. radae_amd/channel_tools.py), linearly interpolated to
 * Fs, scaled by hf_gain = 1/sqrt(var G1 + var G2).
 * noise_low_dev : optional [B][2][n_low + n_taps] complex64 unit-variance-per-component input noise with
 *                 n_low = max(ceil(n_out / low_ratio), 2); NULL: Philox from seed
 * G_out_dev     : [B][n_out][2] complex64, the layout rade_channel_params.G_dev takes.  Returns n_out or <0. */
int rade_batch_multipath_gen(rade_batch *h, const float *fir_taps_host, int n_taps, int low_ratio, int n_out,
                             const void *noise_low_dev, unsigned long long seed, void *G_out_dev, void *stream);
/* The rate-Rs channel matrix multipath_samples.m:33-40 derives from the same Doppler samples (what `multipath_samples("lmr60", 8000, 2000, 1, ...)` writes for the
 * BBFM model, BBFM.md:37, and the `h_*.f32` files of the rate-Rs RADE model): H[b][t][c] = G1[t M] + G2[t M] exp(-j 2 pi c delay Rs), M = Fs / Rs = fs_over_rs,
 * as magnitudes (float32 [B][n_sym][Nc], the default file form) or complex (want_complex: [B][n_sym][Nc][2]).  G_dev [B][n_g][2] complex64 as
 * rade_batch_multipath_gen leaves it (hf_gain included), n_g > (n_sym - 1) M.  Returns n_sym or <0. */
int rade_batch_multipath_h(rade_batch *h, const void *G_dev, int n_g, int fs_over_rs, int n_sym, int Nc, float delay_s, float Rs, int want_complex,
                           float *H_out_dev, void *stream);
float rade_sigma_from_EbNodB(float EbNodB);

/* ---- receive --------------------------------------------------------------------------------
 * rx_dev + b*rx_stride points at the first sample stream b has NOT yet consumed; n_avail_host[b]
 * samples are readable there.  Each stream consumes whole do_radae_rx calls (rade_nin() samples
 * each) while enough samples remain and fewer than max_calls calls were made in this invocation.
 * features_out_dev: stream b at + b*feat_stride floats; each valid modem frame appends 432 floats.  feat_stride / 432 is the
 *   stream's capacity in frames: a stream that has filled it makes no further call in this invocation (status consumed < available),
 *   so nothing is ever written past a stream's rows; the caller continues from rx_dev + consumed with fresh rows.
 * eoo_out_dev: [B][180] soft bits of the most recent end-of-over frame (NULL to skip).
 * status_host: [B] records filled on return (the call synchronises `stream`). */
typedef struct {
    int consumed;         /* samples consumed by this invocation */
    int n_calls;          /* do_radae_rx calls made */
    int n_valid;          /* calls that produced features (432 floats each) */
    int has_eoo;          /* an end-of-over frame was decoded */
    int nin;              /* samples the next call needs (rade_nin) */
    int sync;             /* rade_sync */
    int snr_dB;           /* rade_snrdB_3k_est */
    int state;            /* 0 search 1 candidate 2 sync */
} rade_rx_status;
int rade_batch_rx(rade_batch *h, const void *rx_dev, long rx_stride, const int *n_avail_host, int max_calls,
                  float *features_out_dev, long feat_stride, float *eoo_out_dev, rade_rx_status *status_host,
                  void *stream);
void rade_batch_rx_reset(rade_batch *h);
/* stream-ordered reset of encoder and receiver state of every stream (a new batch of utterances starts) */
void rade_batch_reset(rade_batch *h, void *stream);
/* seed of the documented LCG that picks the 48 rows check_pilots refreshes (dsp.py:291-295 uses an
 * unseeded np.random.randint); seeds_host[B] or NULL for all-ones */
void rade_batch_rx_set_lcg(rade_batch *h, const unsigned *seeds_host);

/* per-call trace record (tests): mirrors what radae_rxe.py prints per frame at -v 2 */
typedef struct {
    int state_before, state_after, nin_before, nin_after, ret, tmax, f_ind_max, valid_count;
    int uw_errors, synced_count, snr_int, pad;
    double fmax, Dthresh, Dtmax12, Dtmax12_eoo;
    float snrdB_3k_est; float pad2;
} rade_rx_trace;
/* copies up to max_calls records + 240-float z_hat rows per call of stream b to host; returns #calls traced */
int rade_batch_rx_get_trace(rade_batch *h, int b, rade_rx_trace *out, float *z_hat_out, int max_calls);

/* Host-side wait policy of rade_batch_rx (the one call that waits for its stream): hipStreamSynchronize spins, which is right while every engine's
 * host thread has a core; when more engines are open in the process than the process has CPUs (affinity mask and cgroup quota: 8 GPUs x 3 batches
 * in flight = 24 threads under a 16-core quota) the wait SLEEPS instead (naps between hipEventQuery calls: rade_engine.c sleep_until_event; hipEventSynchronize on a hipEventBlockingSync event was measured to burn a core per waiting thread on this runtime).  $RADE_SYNC=spin|block overrides; $RADE_SYNC_PEERS=n: n processes with as many engines each share these CPUs (one process per GPU: the count is engines x n).
 * rade_sync_policy is the rule itself (1 = block); rade_host_cpu_quota what it is fed with; rade_batch_sync_counts what an engine did so far. */
double rade_host_cpu_quota(void);
int rade_sync_policy(int engines_open, double cpu_quota);
void rade_batch_sync_counts(const rade_batch *h, long *blocking, long *spinning);
/* test aid: the band-pass filtered samples (complex_bpf.bpf, dsp.py:63-102) the receiver of the most recent rade_batch_rx invocation read for
 * stream b -> out_host [n] complex64 (host memory); returns the number copied, < 0 on error */
int rade_batch_rx_filtered(rade_batch *h, int b, void *out_host, int n);
/* measurement aid: shader-clock cycles each stream's workgroup spent in the most recent receiver launch -> out_host[B]; returns B */
int rade_batch_rx_stream_cycles(rade_batch *h, long long *out_host);

/* ---- measurement hook (bench.py): per-kernel-class HIP-event timing, off by default ---------- */
enum { RADE_PROF_GEMM = 0, RADE_PROF_SCAN, RADE_PROF_MOD, RADE_PROF_CHAN, RADE_PROF_SYNC, RADE_PROF_BPF, RADE_PROF_NCLASS };
void rade_batch_profile(rade_batch *h, int enable);
/* accumulated since enable: device milliseconds, algorithmic FLOPs, launches */
int rade_batch_profile_get(rade_batch *h, int cls, double *ms, double *work, long *launches);
/* every profiled launch of a class as [start, end] in ms after a caller-supplied hipEvent_t (so that launches of several engines, which
 * may overlap on the device, can be put on one time axis): set the reference before enabling, read after disabling; returns the count */
void rade_batch_profile_ref(rade_batch *h, void *ref_event);
int rade_batch_profile_intervals(rade_batch *h, int cls, float *t0_ms, float *t1_ms, int max);

/* ---- several GPUs from one host process (BASELINE.json configs[3]: 2048 utterances over 8 MI355X) -------------------
 * The n_streams_total independent utterances are sharded contiguously over the devices of device_mask (bit g = HIP device g),
 * ceil(total / n_dev) per device; there is no data-path collective.  rade_multi_open reads the DNNw blob once, moves it to the
 * other devices with ONE ncclBroadcast (RCCL over xGMI; librccl.so is bound at run time and only required for n_dev > 1) and opens
 * one engine per device (one host thread each).  Per-device work is driven through the ordinary rade_batch_* calls on
 * rade_multi_engine(m, i); rade_multi_foreach runs a callback on one host thread per device; rade_multi_allreduce_sum adds job
 * statistics (frames, loss sums, bit errors) with ONE ncclAllReduce.  NULL / <0 on failure, message on stderr. */
typedef struct rade_multi rade_multi;
typedef int (*rade_multi_fn)(int i, rade_batch *engine, int first_stream, int n_streams, void *arg);
rade_multi *rade_multi_open(const char *blob_path, int n_streams_total, int max_tx_mf, unsigned long long device_mask, int flags);
void rade_multi_close(rade_multi *m);
int rade_multi_n_devices(const rade_multi *m);
const char *rade_multi_transport(const rade_multi *m);            /* "rccl" or "none (single device)" */
rade_batch *rade_multi_engine(rade_multi *m, int i, int *device, int *first_stream, int *n_streams);
void rade_multi_shard(int n_total, int n_dev, int i, int *first, int *count);      /* the sharding rule, also usable on its own */
int rade_multi_foreach(rade_multi *m, rade_multi_fn fn, void *arg);
int rade_multi_allreduce_sum(rade_multi *m, const double *per_device /* [n_dev][n] */, int n, double *out /* [n] */);

/* ---- single-carrier modem for BBFM symbols ------------------------------------------------
 * Batched form of the reference's `single_carrier` class (radae/dsp.py:579-860; drivers sc_tx.py:58-75,
 * sc_rx.py:83-112): BPSK at Rs symbols/s, Fs = 4 Rs, 16-symbol frame-sync word + 80 payload symbols per frame,
 * 24-tap root-Nyquist filters (gen_rn_coeffs, dsp.py:532-562), fine timing from the symbol-rate line of the
 * envelope, squared-symbol phase tracker, frame-sync state machine.  One independent modem per stream; all state
 * lives on the device between calls. */
typedef struct rade_sc rade_sc;
typedef struct {          /* end-of-call state of one stream */
    int n_frames;         /* frames demodulated by this call */
    int consumed;         /* samples consumed */
    int state;            /* 0 search, 1 sync */
    int nin;              /* samples the next frame needs: 96 M - 1, 96 M or 96 M + 1 (dsp.py:697-702) */
    int fs_s;             /* frame-sync position inside the two-frame symbol buffer */
    float g;              /* amplitude normalisation from the sync word (dsp.py:805-806) */
    float max_cs_re, max_cs_im, norm_rx_timing, phase_ambiguity;
} rade_sc_status;
typedef struct {          /* per-frame record, what sc_rx.py prints at -v 2 */
    int state, nin, fs_s, pad;
    float norm_rx_timing, g, max_cs_re, max_cs_im, phase_ambiguity, pad2[3];
} rade_sc_frame;
/* Rs, Fs, fcentreHz, alpha as single_carrier.__init__ (dsp.py:581); Fs must equal 4 Rs.  NULL without a GPU. */
rade_sc *rade_sc_open(int n_streams, double Rs, double Fs, double fcentreHz, double alpha, int device);
void rade_sc_close(rade_sc *h);
void rade_sc_reset(rade_sc *h);
int rade_sc_n_streams(const rade_sc *h);
int rade_sc_n_tx_out(const rade_sc *h);       /* 384 samples per frame */
int rade_sc_nin_max(const rade_sc *h);        /* 385 */
int rade_sc_n_payload(const rade_sc *h);      /* 80 */
void rade_sc_rrc(const rade_sc *h, double *taps_out /* [24] */);
/* single_carrier.tx (dsp.py:636-662) for n_frames frames per stream: symbs_dev [B][n_frames][80] float ->
 * iq_out_dev [B][iq_stride] complex64, n_frames * 384 samples each.  Returns samples per stream or -1. */
int rade_sc_tx(rade_sc *h, const float *symbs_dev, int n_frames, void *iq_out_dev, long iq_stride, void *stream);
/* single_carrier.rx (dsp.py:773-829) over every whole frame in the first n_avail samples of each stream (at most
 * max_frames): payload_out_dev [B][max_frames][80] complex64 = the returned symbols, zhat_out_dev [B][max_frames][80]
 * = g * Re(payload) where the modem is in sync after the frame, else 0 (sc_rx.py:99-101); frames_out_dev
 * [B][max_frames]; any of the three may be NULL.  status_host[B] is filled after a stream synchronisation. */
int rade_sc_rx(rade_sc *h, const void *rx_dev, long rx_stride, int n_avail, int max_frames, void *payload_out_dev, float *zhat_out_dev,
               rade_sc_frame *frames_out_dev, rade_sc_status *status_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif
