/*
 * rade_oracle.h -- CPU restatement of the RADAE streaming hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C, single-stream, deterministic restatement of the reference's
 * algorithm.  It is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg and by
 * nothing else; the product (radae_amd/csrc) never links it.  Pinned against golden vectors captured
 * from the imported reference (tests/golden/, generator oracle/gen_golden.py) -- see
 * tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 */
#ifndef RADE_ORACLE_H
#define RADE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_c32;

/* numerology of model19_check3 (radae/radae.py:128-234) */
enum {
    ORC_M = 160, ORC_NCP = 32, ORC_NC = 30, ORC_NS = 4, ORC_NZMF = 3, ORC_LATENT = 80,
    ORC_SYM = 192, ORC_NMF = 960, ORC_NEOO = 1152, ORC_RXBUF = 2112, ORC_NFCOARSE = 40,
    ORC_NTAP = 101, ORC_NIN_MAX = 1120, ORC_NFEAT = 21, ORC_NFEAT_TOTAL = 36, ORC_NEOO_BITS = 180
};

/* ---- model (DNNw blob) ------------------------------------------------------------------- */
typedef struct orc_model orc_model;
orc_model *orc_model_load(const char *blob_path);
void orc_model_free(orc_model *m);
/* named fp32 tensor access in the layouts documented in rade_oracle.c (for the blob-reader test) */
int orc_model_tensor(const orc_model *m, const char *name, const float **data, int *n);

/* ---- constants --------------------------------------------------------------------------- */
/* copies a named constant table into out (floats; complex tables interleaved re,im); returns the
 * number of floats written or -1.  names: w Winv Wfwd P Pend p pend p_cp pend_cp eoo Pmat bpf_h
 * bpf_phase_vec_exp acq_p_w acq_fcoarse pilot_gain bpf_alpha */
int orc_get_const(const char *name, float *out, int max_floats);

/* ---- core encoder / decoder (radae_base.py:223-286, 358-430; src/rade_enc.c, rade_dec.c) --- */
typedef struct orc_enc_state orc_enc_state;
typedef struct orc_dec_state orc_dec_state;
orc_enc_state *orc_enc_new(void);
orc_dec_state *orc_dec_new(void);
void orc_enc_reset(orc_enc_state *s);
void orc_dec_reset(orc_dec_state *s);
void orc_enc_free(orc_enc_state *s);
void orc_dec_free(orc_dec_state *s);
void orc_core_encoder(const orc_model *m, orc_enc_state *s, float z[80], const float features[84]);
void orc_core_decoder(const orc_model *m, orc_dec_state *s, float features[84], const float z_hat[80]);
void orc_core_encoder_b1(const orc_model *m, orc_enc_state *s, float z[80], const float *features);   /* bottleneck 1 */
int orc_model_feat_width(const orc_model *m);                                                          /* 84 or 80 */
void orc_channel_rs(float *z_hat, const float *z, const float *H, const float *noise, int n, float sigma);
void orc_channel_bbfm(float *z_hat, const float *z, const float *H, const float *noise, int n, float CNRdB, float Gfm);
/* state peek for tests: layer 1..5 */
const float *orc_enc_gru_state(const orc_enc_state *s, int layer);
const float *orc_dec_gru_state(const orc_dec_state *s, int layer);

/* ---- transmitter (radae_txe.py:47-144, dsp.py:323-378, radae.py:208-219,441-455) ---------- */
typedef struct orc_tx orc_tx;
orc_tx *orc_tx_new(const orc_model *m);
void orc_tx_free(orc_tx *t);
/* 12 feature frames x 36 floats -> 960 IQ samples; z_out (240 floats) optional */
void orc_tx_frame(orc_tx *t, orc_c32 tx_out[960], const float features_in[432], float *z_out);
void orc_ofdm_mod(orc_c32 tx_out[960], const float z[240]);
void orc_tx_set_eoo_bits(orc_tx *t, const float bits[180]);
void orc_tx_set_txbpf(orc_tx *t, int enable);      /* radae_tx(..., txbpf_en=True): Tx band-pass filter + magnitude clip on every frame and the EOO frame */
void orc_tx_eoo(orc_tx *t, orc_c32 out[1152]);

/* ---- channel simulator (radae.py:529-589, inference.py:155-171,263-284), batch of one ------ */
/* G: n x 2 complex (G1,G2) or NULL for the identity channel; noise: n complex unit-variance or NULL */
void orc_channel(orc_c32 *rx, const orc_c32 *tx, int n, const orc_c32 *G, const orc_c32 *noise,
                 float sigma, float freq_offset, float df_dt, orc_c32 *final_phase);
float orc_sigma_from_EbNodB(float EbNodB);
/* EOO frame through the same offsets: eoo*lin_phase*final_phase + sigma*noise (inference.py:263-275) */
void orc_channel_eoo(orc_c32 *rx, const orc_c32 *eoo, int n, const orc_c32 *noise, float sigma,
                     float freq_offset, float df_dt, orc_c32 final_phase);

/* ---- receiver (radae_rxe.py:56-330, dsp.py:39-102,152-320,383-526, rade_api.c:463-539) ----- */
typedef struct orc_rx orc_rx;
orc_rx *orc_rx_new(const orc_model *m);
void orc_rx_free(orc_rx *r);
void orc_rx_set_lcg(orc_rx *r, unsigned seed);      /* row-refresh generator, see gen_golden.py */
void orc_rx_set_foff_err(orc_rx *r, double hz);     /* RADE_FOFF_TEST (rade_api.c:263-264) */
void orc_rx_set_disable_unsync(orc_rx *r, double seconds);   /* radae_rxe.py --disable_unsync (:277-281, :337) */
int orc_rx_nin(const orc_rx *r);
int orc_rx_sync(const orc_rx *r);
int orc_rx_snr(const orc_rx *r);
/* one do_radae_rx call: consumes orc_rx_nin() samples; returns bit0 valid, bit1 end-of-over.
 * features_out[432] filled when valid; eoo_out[180] when end-of-over; z_hat_out[240] optional. */
int orc_rx_frame(orc_rx *r, float features_out[432], float eoo_out[180], const orc_c32 *rx_in, float *z_hat_out);

typedef struct {
    int state, nin, tmax, f_ind_max, valid_count, uw_errors, synced_count, mf;
    double fmax, Dthresh, Dtmax12, Dtmax12_eoo;
    float snrdB_3k_est;
} orc_rx_trace;
void orc_rx_get_trace(const orc_rx *r, orc_rx_trace *t);

/* stand-alone pieces for stage-wise tests */
typedef struct orc_bpf orc_bpf;
orc_bpf *orc_bpf_new(void);
void orc_bpf_free(orc_bpf *b);
void orc_bpf_run(orc_bpf *b, orc_c32 *out, const orc_c32 *in, int n);

/* ---- loss (radae_base.py:50-68, loss.py:64-91) -------------------------------------------- */
double orc_distortion_loss(const float *y_true, const float *y_pred, int nframes, int dim, int stride);
/* loss.py time alignment: slide features_hat over features, return min loss and its start */
double orc_find_loss(const float *features, int n, const float *features_hat, int n_hat, int stride, int *start);

#ifdef __cplusplus
}
#endif
#endif
