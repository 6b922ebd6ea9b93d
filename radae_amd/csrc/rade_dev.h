/*
 * rade_dev.h -- data layouts shared by the C host (rade_engine.c / rade_tables.c) and the HIP
 * kernels (rade_kernels.hip), and the thin launch shims the host calls ("FFI" between plain C and
 * device code).  Everything here is POD; float2-like pairs are spelled as float[2] so that plain C
 * can fill them.
 */
#ifndef RADE_DEV_H
#define RADE_DEV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numerology of RADE V1 / model19_check3 (radae/radae.py:128-234) */
#define RD_M        160   /* samples per OFDM symbol body */
#define RD_NCP      32    /* cyclic prefix */
#define RD_SYM      192
#define RD_NC       30    /* carriers */
#define RD_NS       4     /* data symbols per modem frame */
#define RD_NMF      960   /* samples per modem frame */
#define RD_NEOO     1152
#define RD_RXBUF    2112  /* 2*Nmf + M + Ncp */
#define RD_NFC      40    /* coarse frequency bins */
#define RD_NQ       16    /* polynomial moments of the two-stage pilot correlator (rade_host.c: rd_corrq16_table_fill) */
#define RD_NTAP     101
#define RD_NINMAX   1120
#define RD_LATENT   80
#define RD_ZMF      240   /* latent floats per modem frame */
#define RD_FEAT_MF  432   /* 12 frames x 36 */
#define RD_NEOOBITS 180
#define RD_ENC_W    864   /* encoder concat width */
#define RD_DEC_W    736
#define RD_ENC_IN   96    /* 84 padded to a multiple of 16 (k-block of the f16 matrix instructions) */
#define RD_RX_ROUND_MAX 128   /* capacity: do_radae_rx calls per stream per sync-kernel launch (engine picks R <= this) */
#define RD_DEC_ROWS_MAX (3 * RD_RX_ROUND_MAX)

/* constant tables, one copy in HBM (filled by rade_tables.c) */
typedef struct {
    float Winv[RD_NC][RD_M][2];        /* radae.py:175-178 */
    float Wfwd[RD_M][RD_NC][2];        /* :179 */
    float P[RD_NC], Pend[RD_NC];       /* real parts (imag = 0), :182-185 */
    float p[RD_M][2], pend[RD_M][2];   /* :183,:186 */
    float eoo[RD_NEOO][2];             /* default EOO frame :208-219 */
    float Pmat[RD_NC][2][3][2];        /* dsp.py:400-412 */
    float eq_rot[RD_NC][2];            /* exp(-1j*w[c]*20), dsp.py:433 */
    float bpf_h[RD_NTAP + 3];          /* dsp.py:46-49 */
    float bpf_E[RD_NEOO][2];           /* phase_vec_exp, dsp.py:61: exp(-j alpha (i + 1)); 1152 entries (the transmit filter also takes the end-of-over frame) */
    float p_w[RD_M][RD_NFC][2];        /* acquisition.p_w, dsp.py:166-173 */
    double fcoarse[RD_NFC];            /* dsp.py:163 */
    float pilot_gain;                  /* radae.py:196-199 */
    float snr_c1, snr_c2;              /* 10log10(Rs*Nc/3000), 10log10((M+Ncp)/M)  dsp.py:453 */
    float pad;
} rd_tables;

/* what complex_bpf carries from call to call (dsp.py:54-61, :96-99): one record per stream and filter (receiver input filter; transmit filter) */
typedef struct {
    int mem_len;                        /* samples of memory: Ntap - 1 = 100 before the first call, Ntap + 1 = 102 afterwards (dsp.py:55 against :96) */
    int grid_off;                       /* receiver: the stream left the block grid of the invocation's pre-pass (a timing slip inside the invocation) and filters its own samples until the invocation ends */
    float phase[2];                     /* the complex64 mixer phase after the last sample filtered */
    float mem[102][2];                  /* the last mem_len baseband samples, oldest first */
} rd_bpf_state;

/* per-stream receiver state (HBM, one record per stream, touched by exactly one workgroup) */
typedef struct {
    int state, nin, tmax, tmax_candidate, valid_count, uw_errors, synced_count, mf;
    int f_ind_max, dec_reset_pending, n_out_frames, pad0;
    uint32_t lcg; int has_eoo, dt_valid;
    int pad1;
    double fmax, foff_err, rx_phase[2];
    double rx_theta;                 /* k_rx_sync2 keeps rx_phase as its angle (rx_phase = e^{j rx_theta}); carried exactly, so results do not depend on how calls are cut into launches */
    double Dthresh, Dtmax12, Dtmax12_eoo;
    float snr_est, pad2[3];
    unsigned rxmax[4];                  /* float bits: largest |re|,|im| of the filtered samples of the last three calls (check_pilots operand scale) */
    long long consumed;                 /* samples consumed since reset */
    rd_bpf_state bpf;                   /* input band-pass filter (radae_rxe.py:104-109, :194-195) */
    float rx_buf[RD_RXBUF][2];
    float rowsum1[RD_NMF], rowsum2[RD_NMF];   /* sum_f |Dt1[t,f]|, |Dt2[t,f]|: all check_pilots ever reads back */
} rd_rx_stream;

/* per-stream, per-launch bookkeeping of the receiver kernel (row flags for its decoder stage, per-call trace indices) */
typedef struct {
    int n_calls;                        /* calls made this round */
    int n_rows;                         /* decoder steps emitted (3 per valid call) */
    int uw_from_row;                    /* aux-bit errors count from this row on (sync entry resets) */
    int consumed;                       /* samples consumed this round */
    int row_reset[RD_DEC_ROWS_MAX];     /* 1: decoder state is reset before this step */
    int call_ret[RD_RX_ROUND_MAX];      /* bit0 valid bit1 eoo */
    int call_row_lo[RD_RX_ROUND_MAX], call_row_hi[RD_RX_ROUND_MAX]; /* trace patching of uw_errors */
    int call_trace_idx[RD_RX_ROUND_MAX];
    int out_base;                       /* valid frames of this invocation written to features_out so far */
    int pad[3];
} rd_rx_round;

/* same layout as rade_rx_trace in include/rade_batch.h */
typedef struct {
    int state_before, state_after, nin_before, nin_after, ret, tmax, f_ind_max, valid_count;
    int uw_errors, synced_count, snr_int, pad;
    double fmax, Dthresh, Dtmax12, Dtmax12_eoo;
    float snrdB_3k_est; float pad2;
} rd_rx_trace;

/* ---- launch shims (defined in rade_kernels.hip / rade_rx.hip) -------------------------------------------- */
typedef void *rd_stream_t;

/* Y[r, n] = act(sum_k A[r,k] W[n,k] + bias[n]) on f32 MFMA; rows r = b*T + t.
 * A row (b,t) = [tap0 | tap1]: tap1 at a1 + b*a1_sb + t*a1_st (K1 floats), tap0 (K0 floats, may be 0)
 * at a0 + b*a0_sb + t*a0_st, or the zero row when reset[b*reset_sb+t] != 0.  Wp = packed weights
 * (rd_pack_weights).  act: 0 none, 1 tanh+clamp, 2 GLU (y = a1[r][n] * sigmoid(acc), clamp). */
typedef struct {
    const float *a1; long a1_sb, a1_st; int K1;
    const float *a0; long a0_sb, a0_st; int K0;
    const int *reset; int reset_sb;    /* optional [B][reset_sb] (reset_sb >= T) */
    const int *n_rows;                 /* optional [B]: rows t >= n_rows[b] are skipped */
    const float *Wp; const float *bias;
    const unsigned short *Wp16;        /* optional rd_pack_weights_f16x2 copy: used for large row counts when K0, K1 are multiples of 16 */
    const float *Wscale;               /* non-NULL: Wp16 is ONE plane of integers (rd_pack_weights_q16, int8-exact layer) and Wscale[n] the column's scale */
    float *y; long y_sb, y_st; int N;  /* N valid outputs (<= 32*NT) */
    int B, T, act;
} rd_gemm_args;
int rd_launch_gemm(const rd_gemm_args *a, rd_stream_t s);
/* packs W[N][K] (row-major) into the fragment order the GEMM kernel streams; returns #floats (K,N padded) */
long rd_pack_weights(const float *W, int N, int K, float *out);
long rd_packed_size(int N, int K);

/* GRU recurrence over T steps, one workgroup per stream.  gi[b][t][3H] = W_ih x + b_ih (from the GEMM).
 * h state [B][H] in/out.  Writes clamp(h_t) to out + b*out_sb + t*out_st. */
typedef struct {
    const float *gi; long gi_sb, gi_st;
    const float *Whh; const float *bhh;   /* [3H][H], [3H], torch gate order r,z,n */
    float *h;                             /* [B][H] */
    float *out; long out_sb, out_st;
    const int *reset; int reset_sb;       /* optional [B][reset_sb]: zero h before step */
    const int *n_rows;                    /* optional [B] */
    int B, T, H;
    unsigned short *outf; int outf_NQ, outf_col;   /* optional (rade_enc.hip): clamp(h_t) as two binary16 planes into columns outf_col.. of the encoder's fragment buffer instead of out */
} rd_scan_args;
int rd_launch_gru_scan(const rd_scan_args *a, rd_stream_t s);

/* ---- the batched encoder on activations stored as matrix-core operand fragments (rade_enc.hip) ----
 * xf [B][NQ][RD_EF_TILE] binary16: per stream a history tile (rows 30, 31 = steps -2, -1) and one tile per 32 steps, a tile = [col / 16][plane hi / lo][(col % 16) / 8][t % 32][col % 8]
 * of 2^8 x = hi + lo.  Y^T = W X^T: K0 columns of the rows dil steps earlier (0: none), then K1 columns of the rows themselves; outputs as float32 rows (y) or into columns ycol.. of
 * the same kind of buffer (yf). */
#define RD_EF_KB   (RD_ENC_W / 16)
#define RD_EF_TILE (RD_EF_KB * 1024)
typedef struct {
    const unsigned short *xf; int NQ;
    int B, T, K0, K1, dil;
    const unsigned short *Wp16; const float *Wscale; const float *bias; int N, act;   /* as rd_gemm_args; act 0 none, 1 tanh + clamp */
    float *y; long y_sb, y_st;
    unsigned short *yf; int ycol;
    int seq_taps, no_pair;                          /* developer switches: conv taps one after the other (the float32-row kernels' summation order); no XCD pairing of column groups */
    int pair;                                       /* set by rd_launch_encf_gemm: the two column groups of a 192-column product as blocks i, i + 8 of a 1-D grid (same XCD) */
    const float *xin; int Kin; const float *Wp;     /* rd_launch_encf_dense1 only: float32 input rows [B][T][Kin], rd_pack_weights copy of dense_1 */
} rd_encf_args;
int rd_launch_encf_gemm(const rd_encf_args *a, rd_stream_t s);
/* conv_l (2 x cin -> 96 columns at column cin of xf, taps dil steps apart, tanh) and the product over the cin + 96 columns that then exist (the next GRU's input
 * projection: Ng = 192, one-plane weights; z_dense: Ng = 80, two planes) in one launch; y float32 rows */
typedef struct {
    unsigned short *xf; int NQ; int B, T, cin, dil;
    const unsigned short *Wc; const float *Wc_scale, *Wc_bias;
    const unsigned short *Wg; const float *Wg_scale, *Wg_bias; int Ng, g_act;
    float *y; long y_sb, y_st;
} rd_encf_fused_args;
int rd_launch_encf_fused(const rd_encf_fused_args *a, rd_stream_t s);
int rd_launch_encf_dense1(const rd_encf_args *a, rd_stream_t s);
/* conv history between calls: dir 0 = the float32 history rows (x32 + b * x32_sb, 2 x RD_ENC_W) -> the history tile; dir 1 = steps T - 2, T - 1 -> history tile and float32 rows */
int rd_launch_encf_hist(unsigned short *xf, int NQ, float *x32, long x32_sb, int B, int T, int dir, rd_stream_t s);

/* encoder input packing: features [B][T*4][36] -> [B][T][88] = 4 x (20 feats, aux -1), zero pad */
int rd_launch_enc_pack(const float *features, float *xin, int B, int T, rd_stream_t s);
int rd_launch_pad_rows(const float *src, float *dst, long R, int K, int Kpad, rd_stream_t s);
int rd_launch_chan_symbol(const float *z, const float *H, const float *noise, float *out, long n_real, int mode, float p0, float p1, unsigned long long seed, rd_stream_t s);
/* x is [B][nhist+Tcap][W]: copy time rows [T, T+nhist) (or [n_rows[b], ..)) of each stream to rows [0, nhist) */
int rd_launch_carry_rows(float *x, int B, int Tcap, int W, int nhist, int T, const int *n_rows, rd_stream_t s);
/* z [B][n_mf*3][80] -> tx [b*stride + mf*960 ...] (transmitter_one, dsp.py:340-378) */
int rd_launch_ofdm_mod(const rd_tables *tab, const float *z, void *tx, long tx_stride, int B, int n_mf, rd_stream_t s);
/* the same with the two-path multipath model applied on the way out: mp [B][n_mf*960] c64, part [B][n_mf][2] sums of |tx|^2, |mp|^2; tx optional */
int rd_launch_ofdm_mod_mp(const rd_tables *tab, const float *z, void *tx, long tx_stride, int B, int n_mf, const void *G, void *mp, double *part, rd_stream_t s);
/* EOO data symbols: bits [B][180] -> eoo frames [B][1152] (radae.py:441-455); bits NULL = defaults */
int rd_launch_eoo_build(const rd_tables *tab, const float *bits, float *eoo, int B, rd_stream_t s);
int rd_launch_copy_eoo(const float *eoo, void *out, long stride, int B, rd_stream_t s);

typedef struct {
    const rd_tables *tab; const void *tx; long tx_stride; void *rx; long rx_stride;
    const void *G; const void *noise; const float *eoo; void *scratch; /* >= B * (1 + max(64, n_sig / 960)) * 2 doubles: [B][4] floats (gain, final phase), then the partial power sums */
    const void *mp;                    /* optional [B][n_sig] c64: the multipath output k_ofdm_mod_mp left (scratch then holds its n_sig / 960 per-frame power sums per stream) */
    int B, n_sig, n_pre, n_post, with_eoo; float sigma, freq_offset, df_dt; unsigned long long seed;
    float sine_amp, sine_freq, rx_gain;
} rd_chan_args;
int rd_launch_channel(const rd_chan_args *a, rd_stream_t s);
/* Doppler-spread generator: taps_dev [n_taps] f32, noise optional, G [B][n_out][2] c64 */
int rd_launch_multipath_gen(const float *taps_dev, int n_taps, int low_ratio, int n_out, const void *noise_low, unsigned long long seed, void *G, void *ybuf, int B, rd_stream_t s);
int rd_multipath_gen_needs_scratch(int low_ratio, int n_out);
int rd_launch_multipath_h(const void *G, int n_g, int M, int n_sym, int Nc, float dRs, int want_complex, float *H, int B, rd_stream_t s);

/* CoreDecoderStatefull.forward (radae_base.py:388-430) for the pending rows of one stream, run inside the receiver
 * kernel by the stream's own workgroup (rx_decode_pending -> ds_layers): buffers and weights of that stage. */
typedef struct { const float *wp, *bias; const unsigned short *wp16, *wa16; const float *wscale; int N, K; } rd_lin;
/* wp16: rd_pack_weights_f16x2; wa16 (in-kernel decoder): rd_pack_weights_q16_a16 when wscale != NULL (int8-exact layer: one plane of
 * integers + per-column scales), else rd_pack_weights_f16x2_a16 (two planes) */
typedef struct {
    const float *z; long z_sb;                 /* [B][.][80] latent rows */
    float *x; long x_sb;                       /* [B][1 + Tcap][736]; x points at row 0 of stream 0, row -1 = conv history (the in-kernel stage keeps every other row in LDS) */
    float *gi; long gi_sb;                     /* [B][.][288] */
    float *hbuf; long hb_sb;                   /* [B][.][96] */
    float *h[5];                               /* GRU states [B][96] */
    float *out; long out_sb; int out_w;        /* [B][.][84] */
    rd_lin dense1, gin[5], glu[5], conv[5], output;
    const float *whh[5], *bhh[5];
    const unsigned short *whq[5]; const float *whs[5];   /* W_hh as int8-exact A-operand fragments (rd_pack_weights_q16_a16) + row scales, or NULL: k_rx_sync2 runs the recurrence on the matrix cores (dq2_scan_mfma) */
    int B;
} rd_decs_args;

typedef struct {
    const rd_tables *tab; rd_rx_stream *st; rd_rx_round *round;
    const void *rx; long rx_stride; const int *avail;   /* [B] samples readable at rx + b*stride (raw input: the receiver only reads it to rebuild the filter memory) */
    void *rxf; long rxf_stride;                          /* band-pass filtered samples of the invocation (k_rx_bpf), same indexing as rx */
    const unsigned short *bpf16;                         /* rd_bpf16_table_fill(): [4][2][64][8] binary16, the band-pass taps as matrix-core A operands */
    const void *bpf_chain; int chain_stride;             /* [B][chain_stride] float2: (nin0, mem_len0) + the block phases of the pre-pass */
    int *acc;                                            /* [B][4] this invocation: consumed, calls, valid, eoo */
    int max_calls;                                       /* call budget per stream per invocation */
    int unsync_off_after;                                /* radae_rxe.py:277-281: synced_count beyond which the unsync paths are disabled; < 0 = never */
    int round_calls, dec_rows;                           /* R calls per stream per launch (<= RD_RX_ROUND_MAX), 3R decoder slots */
    rd_decs_args dec;                                    /* the decoder runs inside the stream's workgroup (rx_decode_pending) */
    float *features_out; long feat_stride;               /* [B][cap][432] */
    int feat_cap;                                        /* valid modem frames features_out holds per stream: a stream stops making calls once it has produced that many */
    int bypass_dec;                                      /* radae_rxe.py --bypass_dec (:300-302, :315): rows of features_out are the 240 latents of a valid modem frame, no decoder, no UW accounting */
    const unsigned short *corr16;                        /* rd_corr16_table_fill(): [5][10][2][64][8] binary16 (the one-stage correlator: -DRX2_ONE_STAGE builds) */
    const unsigned short *corrq16, *corra16;             /* rd_corrq16_table_fill() [2][10][2][64][8] / rd_corra16_table_fill() [5][2][64][8]: the two-stage pilot correlator */
    float *zrows;                                        /* [B][dec_rows][80] */
    float *dtcache;                                      /* [B][960][40] |Dt2| surface of the previous detect_pilots call */
    int *status;                                         /* [B][4]: nin, sync, snr_int, state */
    float *eoo_out;                                      /* [B][180] or NULL */
    rd_rx_trace *trace; float *trace_z; int trace_cap;   /* optional */
    int *progress;                                       /* [4]: calls made by this launch, unused x3 */
    long long *wg_cycles;                                /* [B] shader-clock cycles each stream's workgroup spent in the launch (or NULL) */
    const unsigned short *wfwd16;                        /* [2][10][2][64][8] binary16: the forward DFT matrix (wr, -wi rows) as matrix-core A operands (rd_wfwd16_table_fill; k_rx_sync2's demodulator) */
    const double *vm;                                    /* [8][160] ((n - 79.5) / 80)^m: refine()'s moment powers (k_rx_sync2 reads them from L2) */
    int variant;                                         /* bits 8..: phase mask of the -DRX2_CENSUS developer build (tools/rx2_census.sh); 0 otherwise */
    int lds_bytes;                                       /* dynamic LDS of the launch: what rd_rx_sync_prepare() returned */
    int B;
} rd_sync_args;
int rd_launch_rx_sync(const rd_sync_args *a, rd_stream_t s);
/* once per device before the first launch (rade_batch_open): raises the kernel's dynamic-LDS limit; returns the bytes to launch with, < 0 on error */
int rd_rx_sync_prepare(int solo);

/* complex_bpf.bpf (dsp.py:63-102) over n samples of every stream in one pass (rade_rx.hip: k_bpf_chain + k_bpf_fir [+ k_bpf_advance]): the receiver's
 * input filter ahead of the receiver launches of an invocation, and the transmitter's optional output filter (radae_txe.py:74-83).  Streams are cut into
 * blocks as the reference cuts them into calls: the first block len0 samples, every later one 960. */
typedef struct {
    rd_bpf_state *state; long state_stride;      /* record of stream b at (char *)state + b * state_stride */
    const int *len0; long len0_stride;           /* first block length of stream b at (char *)len0 + b * len0_stride (the receiver: its nin), or NULL: len0_const */
    int len0_const;
    const int *avail; int avail_const;           /* samples of stream b: device array, or NULL: avail_const for every stream */
    const rd_tables *tab; const unsigned short *bpf16;   /* rd_bpf16_table_fill(): the taps as matrix-core operands */
    const void *x; long x_stride; void *y; long y_stride;   /* complex64 in / out, stream b at + b * stride (samples); must not overlap */
    void *chain; int chain_stride;               /* [B][chain_stride] float2 scratch, chain_stride >= n_blocks + 3: (len0, mem_len) + the block phases */
    int n_blocks;                                /* blocks of the longest stream */
    int B;
    int clip;                                    /* 1: magnitude limited to 1 after the filter (radae_txe.py:132) */
    int advance;                                 /* 1: every sample is consumed -- leave the state complex_bpf would hold (the receiver kernel does that itself, by what it consumed) */
    int *zero_acc, *zero_progress;               /* optional: [B][4] / [4] ints that k_bpf_chain clears on its way (the receiver's per-invocation counters: two memsets less in the stream) */
} rd_bpf_args;
int rd_launch_bpf(const rd_bpf_args *a, rd_stream_t s);
/* start-of-utterance state of every stream in ONE launch (one workgroup per stream): the receiver record (radae_rxe.py:128-142) when st != NULL, and any of
 * the GRU states [5][B][H] / conv history rows that are given (NULL = leave alone) */
typedef struct {
    rd_rx_stream *st; const unsigned *seeds; double foff_err;
    float *dec_h; float *dec_x; long dec_x_sb;      /* [5][B][96]; history row of stream b at dec_x + b * dec_x_sb, RD_DEC_W floats */
    float *enc_h; float *enc_x; long enc_x_sb;      /* [5][B][64]; the two history rows of stream b at enc_x + b * enc_x_sb, 2 * RD_ENC_W floats */
    int B;
} rd_reset_args;
int rd_launch_reset(const rd_reset_args *a, rd_stream_t s);

/* ---- one core encoder / decoder step of ONE stream as one launch (rade_core_step.hip; include/rade_core.h) ----
 * A layer = row-major W[N][K] (K a multiple of 8, zero padded): wq != NULL: ONE binary16 plane of the integers q of an int8 layer
 * (w = q * scale[n], exact); else float32 wf (scale NULL). */
typedef struct { const unsigned short *wq; const float *wf; const float *scale; const float *bias; int N, K; } rd_mv;
typedef struct {
    rd_mv dense1, gin[5], ghh[5], glu[5], conv[5], out;
    int is_enc, n_in, n_out;           /* floats in / out per step (80 or 84 features, 80 latents) */
    int W, H, in0, conv_out;           /* concat width 864 / 736, GRU width 64 / 96, dense1 outputs 64 / 96, conv outputs 96 / 32 */
    int dil[5];                        /* conv dilation (1 or 2): tap 0 reads row t - dil */
    float *hist;                       /* [2][W]: concat rows t-1, t-2 of the previous steps */
    float *h;                          /* [5][H] GRU states */
    const float *in; float *out_vec;   /* device-visible buffers (pinned host memory in rade_core.c) */
    unsigned *done; unsigned seq;      /* optional completion word (device-visible host memory): set to seq after out_vec is written */
    const rd_tables *tab; float *iq_out;   /* rd_launch_tx_frame only: the constant tables and the 960 complex output samples (device-visible) */
} rd_core_args;
int rd_launch_core_step(const rd_core_args *a, rd_stream_t s);
/* a whole modem frame of the transmitter in one launch: in = 3 x n_in packed feature rows -> three encoder steps -> OFDM modulator -> iq_out[960] */
int rd_launch_tx_frame(const rd_core_args *a_dev /* the record in DEVICE memory */, unsigned seq, rd_stream_t s);


#ifdef __cplusplus
}
#endif
#endif
