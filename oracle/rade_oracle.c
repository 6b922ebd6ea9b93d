/*
 * rade_oracle.c -- CPU restatement of the RADAE streaming hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99 + libm, single stream, deterministic.  Compile with -ffp-contract=off so that float
 * expressions round the way NumPy / PyTorch CPU kernels do (no FMA contraction).
 * The arithmetic type of every step follows the reference (float32 / complex64, with the few
 * complex128 temporaries NumPy creates); see the per-function citations.
 *
 * Pinned by tests/test_oracle_golden.py against tests/golden/*.npz (captured from the imported
 * reference by oracle/gen_golden.py).
 */
#include "rade_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI_D 3.14159265358979323846

/* ------------------------------------------------------------------------------------------ */
/* small complex helpers (explicit float arithmetic, no C99 _Complex so rounding is visible)    */
typedef struct { double re, im; } c64d;
static inline orc_c32 c32(float re, float im) { orc_c32 r = { re, im }; return r; }
static inline orc_c32 cmulf(orc_c32 a, orc_c32 b) { return c32(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline orc_c32 caddf(orc_c32 a, orc_c32 b) { return c32(a.re + b.re, a.im + b.im); }
static inline orc_c32 csubf(orc_c32 a, orc_c32 b) { return c32(a.re - b.re, a.im - b.im); }
static inline orc_c32 cscalef(orc_c32 a, float s) { return c32(a.re * s, a.im * s); }
static inline orc_c32 cconjf_(orc_c32 a) { return c32(a.re, -a.im); }
static inline float cabsf_(orc_c32 a) { return hypotf(a.re, a.im); }
static inline orc_c32 cexpjf(float ang) { return c32(cosf(ang), sinf(ang)); } /* exp(1j*ang) in f32 */
static inline c64d cmuld(c64d a, c64d b) { c64d r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float clamp1(float x) { return x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x); } /* radae_base.n() without the noise */

/* ============================================================================================
 * 1. DNNw blob reader -> fp32 tensors
 *    format: src/write_rade_weights.c:51-74; layouts: weight-exchange/wexchange/c_export/common.py
 *    :59-69 (8x4 blocks) :140-176 (sparse idx) :263-271 (scale/127) :290-293 (dense W.T)
 *    :307-311 (conv taps) :360-368 (gate swap rzn->zrn)
 * ==========================================================================================*/
typedef struct { char name[48]; int type, size; const unsigned char *data; } rec_t;

typedef struct { int n_in, n_out; float *w, *b; } lin_t;              /* w[out][in] */
typedef struct { int n_in, hid; float *w_ih, *w_hh, *b_ih, *b_hh; } gru_t; /* torch gate order r,z,n */

struct orc_model {
    unsigned char *blob; long blob_len;
    rec_t *recs; int nrec;
    lin_t enc_dense1, enc_zdense, dec_dense1, dec_output;
    gru_t enc_gru[5], dec_gru[5];
    lin_t enc_conv[5], dec_conv[5];   /* w[out][2*in], column = tap*in + i, tap 0 = older */
    lin_t dec_glu[5];
};

static const rec_t *find_rec(const orc_model *m, const char *a, const char *b)
{
    char nm[96];
    snprintf(nm, sizeof nm, "%s%s", a, b);
    for (int i = 0; i < m->nrec; i++) if (!strcmp(m->recs[i].name, nm)) return &m->recs[i];
    fprintf(stderr, "oracle: record %s missing\n", nm);
    abort();
}

static void load_dense_float(const orc_model *m, const char *name, lin_t *l)
{
    const rec_t *rb = find_rec(m, name, "_bias"), *rw = find_rec(m, name, "_weights_float");
    l->n_out = rb->size / 4; l->n_in = rw->size / 4 / l->n_out;
    l->b = malloc(sizeof(float) * l->n_out); memcpy(l->b, rb->data, sizeof(float) * l->n_out);
    l->w = malloc(sizeof(float) * l->n_in * l->n_out);
    const float *src = (const float *)rw->data;          /* (n_in, n_out) */
    for (int o = 0; o < l->n_out; o++) for (int i = 0; i < l->n_in; i++) l->w[o * l->n_in + i] = src[i * l->n_out + o];
}

static void load_dense_int8(const orc_model *m, const char *name, lin_t *l)
{
    const rec_t *rb = find_rec(m, name, "_bias"), *rs = find_rec(m, name, "_scale"), *rq = find_rec(m, name, "_weights_int8");
    l->n_out = rb->size / 4; l->n_in = rq->size / l->n_out;
    l->b = malloc(sizeof(float) * l->n_out); memcpy(l->b, rb->data, sizeof(float) * l->n_out);
    l->w = malloc(sizeof(float) * l->n_in * l->n_out);
    const float *sc = (const float *)rs->data; const signed char *q = (const signed char *)rq->data;
    int nib = l->n_in / 4;
    for (int og = 0; og < l->n_out / 8; og++) for (int ib = 0; ib < nib; ib++) for (int o8 = 0; o8 < 8; o8++) for (int i4 = 0; i4 < 4; i4++) {
        int o = og * 8 + o8, i = ib * 4 + i4;
        float s = sc[o] * 127.0f;
        l->w[o * l->n_in + i] = (float)q[((og * nib + ib) * 8 + o8) * 4 + i4] * s;
    }
}

/* n_in from the architecture: it is not in the blob (common.py:274 compiles it in) and cannot be inferred when no group keeps the last input block */
static void load_sparse_int8(const orc_model *m, const char *name, lin_t *l, int n_in)
{
    const rec_t *rb = find_rec(m, name, "_bias"), *rs = find_rec(m, name, "_scale"), *rq = find_rec(m, name, "_weights_int8"),
                *ri = find_rec(m, name, "_weights_idx");
    l->n_out = rb->size / 4;
    const int *idx = (const int *)ri->data; const float *sc = (const float *)rs->data; const signed char *q = (const signed char *)rq->data;
    int p = 0;
    l->n_in = n_in;
    l->b = malloc(sizeof(float) * l->n_out); memcpy(l->b, rb->data, sizeof(float) * l->n_out);
    l->w = calloc((size_t)l->n_in * l->n_out, sizeof(float));
    p = 0; int blk = 0;
    for (int g = 0; g < l->n_out / 8; g++) {
        int cnt = idx[p++];
        for (int k = 0; k < cnt; k++, blk++) {
            int j = idx[p++];
            for (int o8 = 0; o8 < 8; o8++) for (int i4 = 0; i4 < 4; i4++) {
                int o = g * 8 + o8;
                l->w[o * l->n_in + j + i4] = (float)q[blk * 32 + o8 * 4 + i4] * (sc[o] * 127.0f);
            }
        }
    }
}

static void unswap_rows(float *a, int hid, int cols)
{   /* exporter z,r,n -> torch r,z,n */
    float *tmp = malloc(sizeof(float) * hid * cols);
    memcpy(tmp, a, sizeof(float) * hid * cols);
    memcpy(a, a + hid * cols, sizeof(float) * hid * cols);
    memcpy(a + hid * cols, tmp, sizeof(float) * hid * cols);
    free(tmp);
}

static void load_gru(const orc_model *m, const char *name, gru_t *g, int n_in)
{
    char nm[64]; lin_t a, b;
    snprintf(nm, sizeof nm, "%s_input", name); load_sparse_int8(m, nm, &a, n_in);
    snprintf(nm, sizeof nm, "%s_recurrent", name); load_dense_int8(m, nm, &b);
    g->n_in = a.n_in; g->hid = b.n_in;
    g->w_ih = a.w; g->b_ih = a.b; g->w_hh = b.w; g->b_hh = b.b;
    unswap_rows(g->w_ih, g->hid, g->n_in); unswap_rows(g->w_hh, g->hid, g->hid);
    unswap_rows(g->b_ih, g->hid, 1); unswap_rows(g->b_hh, g->hid, 1);
}

orc_model *orc_model_load(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    orc_model *m = calloc(1, sizeof *m);
    fseek(f, 0, SEEK_END); m->blob_len = ftell(f); fseek(f, 0, SEEK_SET);
    m->blob = malloc(m->blob_len);
    if (fread(m->blob, 1, m->blob_len, f) != (size_t)m->blob_len) { fclose(f); free(m->blob); free(m); return NULL; }
    fclose(f);
    m->recs = malloc(sizeof(rec_t) * 512);
    long off = 0;
    while (off + 64 <= m->blob_len) {
        const unsigned char *h = m->blob + off;
        int32_t ver, type, size, block;
        if (memcmp(h, "DNNw", 4)) { orc_model_free(m); return NULL; }
        memcpy(&ver, h + 4, 4); memcpy(&type, h + 8, 4); memcpy(&size, h + 12, 4); memcpy(&block, h + 16, 4);
        rec_t *r = &m->recs[m->nrec++];
        memset(r->name, 0, sizeof r->name); memcpy(r->name, h + 20, 44);
        r->type = type; r->size = size; r->data = h + 64;
        off += 64 + block;
    }
    load_dense_float(m, "enc_dense1", &m->enc_dense1); load_dense_float(m, "enc_zdense", &m->enc_zdense);
    load_dense_float(m, "dec_dense1", &m->dec_dense1); load_dense_float(m, "dec_output", &m->dec_output);
    for (int i = 0; i < 5; i++) {
        char nm[32];
        snprintf(nm, sizeof nm, "enc_gru%d", i + 1); load_gru(m, nm, &m->enc_gru[i], 64 + 160 * i);      /* radae_base.py:239-249 */
        snprintf(nm, sizeof nm, "dec_gru%d", i + 1); load_gru(m, nm, &m->dec_gru[i], 96 + 128 * i);      /* radae_base.py:377-391 */
        snprintf(nm, sizeof nm, "enc_conv%d", i + 1); load_dense_int8(m, nm, &m->enc_conv[i]);
        snprintf(nm, sizeof nm, "dec_conv%d", i + 1); load_dense_int8(m, nm, &m->dec_conv[i]);
        snprintf(nm, sizeof nm, "dec_glu%d", i + 1); load_dense_int8(m, nm, &m->dec_glu[i]);
    }
    return m;
}

static void free_lin(lin_t *l) { free(l->w); free(l->b); }
void orc_model_free(orc_model *m)
{
    if (!m) return;
    free_lin(&m->enc_dense1); free_lin(&m->enc_zdense); free_lin(&m->dec_dense1); free_lin(&m->dec_output);
    for (int i = 0; i < 5; i++) {
        free(m->enc_gru[i].w_ih); free(m->enc_gru[i].w_hh); free(m->enc_gru[i].b_ih); free(m->enc_gru[i].b_hh);
        free(m->dec_gru[i].w_ih); free(m->dec_gru[i].w_hh); free(m->dec_gru[i].b_ih); free(m->dec_gru[i].b_hh);
        free_lin(&m->enc_conv[i]); free_lin(&m->dec_conv[i]); free_lin(&m->dec_glu[i]);
    }
    free(m->recs); free(m->blob); free(m);
}

int orc_model_tensor(const orc_model *m, const char *name, const float **data, int *n)
{
    char side[4]; int idx; char rest[32];
#define RET(p, cnt) do { *data = (p); *n = (cnt); return 0; } while (0)
    if (!strcmp(name, "enc_dense1_w")) RET(m->enc_dense1.w, m->enc_dense1.n_in * m->enc_dense1.n_out);
    if (!strcmp(name, "enc_dense1_b")) RET(m->enc_dense1.b, m->enc_dense1.n_out);
    if (!strcmp(name, "enc_zdense_w")) RET(m->enc_zdense.w, m->enc_zdense.n_in * m->enc_zdense.n_out);
    if (!strcmp(name, "enc_zdense_b")) RET(m->enc_zdense.b, m->enc_zdense.n_out);
    if (!strcmp(name, "dec_dense1_w")) RET(m->dec_dense1.w, m->dec_dense1.n_in * m->dec_dense1.n_out);
    if (!strcmp(name, "dec_dense1_b")) RET(m->dec_dense1.b, m->dec_dense1.n_out);
    if (!strcmp(name, "dec_output_w")) RET(m->dec_output.w, m->dec_output.n_in * m->dec_output.n_out);
    if (!strcmp(name, "dec_output_b")) RET(m->dec_output.b, m->dec_output.n_out);
    if (sscanf(name, "%3[a-z]_gru%d_%31s", side, &idx, rest) == 3 && idx >= 1 && idx <= 5) {
        const gru_t *g = !strcmp(side, "enc") ? &m->enc_gru[idx - 1] : &m->dec_gru[idx - 1];
        if (!strcmp(rest, "w_ih")) RET(g->w_ih, 3 * g->hid * g->n_in);
        if (!strcmp(rest, "w_hh")) RET(g->w_hh, 3 * g->hid * g->hid);
        if (!strcmp(rest, "b_ih")) RET(g->b_ih, 3 * g->hid);
        if (!strcmp(rest, "b_hh")) RET(g->b_hh, 3 * g->hid);
    }
    if (sscanf(name, "%3[a-z]_conv%d_%31s", side, &idx, rest) == 3 && idx >= 1 && idx <= 5) {
        const lin_t *c = !strcmp(side, "enc") ? &m->enc_conv[idx - 1] : &m->dec_conv[idx - 1];
        if (!strcmp(rest, "w")) RET(c->w, c->n_in * c->n_out);
        if (!strcmp(rest, "b")) RET(c->b, c->n_out);
    }
    if (sscanf(name, "dec_glu%d_%31s", &idx, rest) == 2 && idx >= 1 && idx <= 5 && !strcmp(rest, "w"))
        RET(m->dec_glu[idx - 1].w, 96 * 96);
#undef RET
    return -1;
}

/* ============================================================================================
 * 2. constants -- radae/radae.py:128-234 (numerology, DFT matrices, pilots, EOO),
 *    radae/dsp.py:40-61 (BPF), :153-176 (acquisition p_w), :400-416 (Pmat)
 * ==========================================================================================*/
static struct {
    int ready;
    float w[ORC_NC];
    orc_c32 Winv[ORC_NC][ORC_M], Wfwd[ORC_M][ORC_NC];
    orc_c32 P[ORC_NC], Pend[ORC_NC], p[ORC_M], pend[ORC_M], p_cp[ORC_SYM], pend_cp[ORC_SYM];
    double pilot_gain; float pilot_gain_f;
    orc_c32 eoo[ORC_NEOO];
    orc_c32 Pmat[ORC_NC][2][3];
    orc_c32 eq_rot[ORC_NC];                 /* exp(-1j*w[c]*a), a = 20 samples (dsp.py:433) */
    float bpf_h[ORC_NTAP]; float bpf_alpha; float bpf_B; orc_c32 bpf_pv[ORC_NEOO];   /* 1152 entries: the transmit filter also takes the end-of-over frame */
    orc_c32 p_w[ORC_M][ORC_NFCOARSE]; double fcoarse[ORC_NFCOARSE];
} K;

static orc_c32 pa_limit(orc_c32 x)
{   /* tanh(|x|)*exp(1j*angle(x)) -- radae.py:218, dsp.py:377 */
    float mag = cabsf_(x), ang = atan2f(x.im, x.re);
    float t = tanhf(mag);
    return c32(t * cosf(ang), t * sinf(ang));
}

static void consts_init(void)
{
    if (K.ready) return;
    static const float barker13[13] = { 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1 };
    const float two_pi_f = (float)(2.0 * PI_D);
    for (int c = 0; c < ORC_NC; c++) K.w[c] = (two_pi_f * (float)(15 + c)) / 160.0f;   /* radae.py:172-174 */
    for (int c = 0; c < ORC_NC; c++) for (int n = 0; n < ORC_M; n++) {
        float arg = (float)n * K.w[c];
        float cs = (float)cos((double)arg), sn = (float)sin((double)arg);
        K.Winv[c][n] = c32(cs / 160.0f, sn / 160.0f);                                 /* :178 */
        K.Wfwd[n][c] = c32(cs, -sn);                                                   /* :179 */
    }
    const float sqrt2f = (float)pow(2.0, 0.5);
    for (int c = 0; c < ORC_NC; c++) {                                                 /* :48-56, :182-185 */
        K.P[c] = c32(sqrt2f * barker13[c % 13], 0.0f);
        K.Pend[c] = (c & 1) ? c32(-K.P[c].re, 0.0f) : K.P[c];
    }
    for (int n = 0; n < ORC_M; n++) {                                                  /* :183, :186 */
        orc_c32 a = c32(0, 0), b = c32(0, 0);
        for (int c = 0; c < ORC_NC; c++) { a = caddf(a, cmulf(K.P[c], K.Winv[c][n])); b = caddf(b, cmulf(K.Pend[c], K.Winv[c][n])); }
        K.p[n] = a; K.pend[n] = b;
    }
    for (int n = 0; n < ORC_M; n++) { K.p_cp[ORC_NCP + n] = K.p[n]; K.pend_cp[ORC_NCP + n] = K.pend[n]; }
    for (int n = 0; n < ORC_NCP; n++) { K.p_cp[n] = K.p[ORC_M - ORC_NCP + n]; K.pend_cp[n] = K.pend[ORC_M - ORC_NCP + n]; }
    K.pilot_gain = pow(10.0, -2.0 / 20.0) * 160.0 / pow(30.0, 0.5);                   /* :196-199 */
    K.pilot_gain_f = (float)K.pilot_gain;
    /* EOO frame [P][Pend][0 0 0][Pend] * pilot_gain, PA limited -- :208-219 */
    memset(K.eoo, 0, sizeof K.eoo);
    for (int n = 0; n < ORC_SYM; n++) {
        K.eoo[n] = K.p_cp[n]; K.eoo[ORC_SYM + n] = K.pend_cp[n]; K.eoo[ORC_NMF + n] = K.pend_cp[n];
    }
    for (int n = 0; n < ORC_NEOO; n++) K.eoo[n] = pa_limit(cscalef(K.eoo[n], K.pilot_gain_f));
    /* LS pilot estimator matrices -- dsp.py:400-412 (plain transpose, not Hermitian) */
    for (int c = 0; c < ORC_NC; c++) {
        int cm = c == 0 ? 1 : (c == ORC_NC - 1 ? ORC_NC - 2 : c);
        c64d e[3];
        for (int k = 0; k < 3; k++) { float ang = -(K.w[cm - 1 + k] * 20.0f); e[k].re = (float)cos((double)ang); e[k].im = (float)sin((double)ang); }
        /* A = [[1,e0],[1,e1],[1,e2]];  AtA = [[3, s1],[s1, s2]] */
        c64d s1 = { e[0].re + e[1].re + e[2].re, e[0].im + e[1].im + e[2].im }, s2 = { 0, 0 };
        for (int k = 0; k < 3; k++) { c64d q = cmuld(e[k], e[k]); s2.re += q.re; s2.im += q.im; }
        c64d three = { 3, 0 }, det = cmuld(three, s2), s1s1 = cmuld(s1, s1);
        det.re -= s1s1.re; det.im -= s1s1.im;
        double dn = det.re * det.re + det.im * det.im;
        c64d idet = { det.re / dn, -det.im / dn };
        /* inv = idet * [[s2, -s1],[-s1, 3]];  Pmat = inv * At, At = [[1,1,1],[e0,e1,e2]] */
        for (int k = 0; k < 3; k++) {
            c64d r0 = { s2.re - (s1.re * e[k].re - s1.im * e[k].im), s2.im - (s1.re * e[k].im + s1.im * e[k].re) };
            c64d r1 = { -s1.re + 3 * e[k].re, -s1.im + 3 * e[k].im };
            r0 = cmuld(idet, r0); r1 = cmuld(idet, r1);
            K.Pmat[c][0][k] = c32((float)r0.re, (float)r0.im);
            K.Pmat[c][1][k] = c32((float)r1.re, (float)r1.im);
        }
        float ang = -(K.w[c] * 20.0f);
        K.eq_rot[c] = c32((float)cos((double)ang), (float)sin((double)ang));
    }
    /* BPF -- radae_rxe.py:104-109, dsp.py:40-61 (all float32 as NumPy evaluates it) */
    float bandwidth = (((1.2f * (K.w[ORC_NC - 1] - K.w[0])) * 8000.0f) / (float)(2.0 * PI_D));
    float centre = (((K.w[ORC_NC - 1] + K.w[0]) * 8000.0f) / (float)(2.0 * PI_D)) / 2.0f;
    K.bpf_B = bandwidth / 8000.0f;
    K.bpf_alpha = ((float)(2.0 * PI_D) * centre) / 8000.0f;
    for (int i = 0; i < ORC_NTAP; i++) {
        float x = (float)(i - 50) * K.bpf_B;
        float y = (float)PI_D * (x == 0.0f ? 1.0e-20f : x);
        float s = (float)sin((double)y) / y;
        K.bpf_h[i] = K.bpf_B * s;
    }
    for (int k = 0; k < ORC_NEOO; k++) {
        float arg = (float)((double)K.bpf_alpha * (double)(k + 1));
        K.bpf_pv[k] = c32((float)cos((double)arg), (float)-sin((double)arg));
    }
    /* acquisition -- dsp.py:163-173 (complex128 product stored as complex64) */
    for (int fi = 0; fi < ORC_NFCOARSE; fi++) {
        K.fcoarse[fi] = -50.0 + 2.5 * fi;
        double w = 2.0 * PI_D * K.fcoarse[fi] / 8000.0;
        for (int n = 0; n < ORC_M; n++) {
            c64d wv = { cos(w * n), sin(w * n) }, pp = { K.p[n].re, K.p[n].im }, r = cmuld(wv, pp);
            K.p_w[n][fi] = c32((float)r.re, (float)r.im);
        }
    }
    K.ready = 1;
}

int orc_get_const(const char *name, float *out, int max_floats)
{
    consts_init();
    const void *src = NULL; int n = 0;
#define C(nm, ptr, cnt) if (!strcmp(name, nm)) { src = (ptr); n = (cnt); }
    C("w", K.w, ORC_NC) C("Winv", K.Winv, 2 * ORC_NC * ORC_M) C("Wfwd", K.Wfwd, 2 * ORC_M * ORC_NC)
    C("P", K.P, 2 * ORC_NC) C("Pend", K.Pend, 2 * ORC_NC) C("p", K.p, 2 * ORC_M) C("pend", K.pend, 2 * ORC_M)
    C("p_cp", K.p_cp, 2 * ORC_SYM) C("pend_cp", K.pend_cp, 2 * ORC_SYM) C("eoo", K.eoo, 2 * ORC_NEOO)
    C("Pmat", K.Pmat, 2 * ORC_NC * 6) C("bpf_h", K.bpf_h, ORC_NTAP) C("bpf_phase_vec_exp", K.bpf_pv, 2 * ORC_NIN_MAX)
    C("acq_p_w", K.p_w, 2 * ORC_M * ORC_NFCOARSE) C("pilot_gain", &K.pilot_gain_f, 1) C("bpf_alpha", &K.bpf_alpha, 1)
#undef C
    if (!strcmp(name, "acq_fcoarse")) { if (max_floats < ORC_NFCOARSE) return -1; for (int i = 0; i < ORC_NFCOARSE; i++) out[i] = (float)K.fcoarse[i]; return ORC_NFCOARSE; }
    if (!src || n > max_floats) return -1;
    memcpy(out, src, sizeof(float) * n);
    return n;
}

/* ============================================================================================
 * 3. core encoder / decoder -- radae_base.py:223-286, :358-430 (one 40 ms step per call, as the
 *    C twins src/rade_enc.c:55-114 and src/rade_dec.c:50-102 do)
 * ==========================================================================================*/
static void dense(const lin_t *l, float *out, const float *in)
{
    for (int o = 0; o < l->n_out; o++) {
        const float *w = l->w + (size_t)o * l->n_in; float acc = 0.0f;
        for (int i = 0; i < l->n_in; i++) acc += w[i] * in[i];
        out[o] = acc + l->b[o];
    }
}

static void gru_step(const gru_t *g, float *h, const float *x)
{   /* torch.nn.GRU cell, gate order r,z,n; h' = (h - n)*z + n */
    int H = g->hid; float gi[3 * 96], gh[3 * 96];
    for (int o = 0; o < 3 * H; o++) {
        const float *w = g->w_ih + (size_t)o * g->n_in; float a = 0.0f;
        for (int i = 0; i < g->n_in; i++) a += w[i] * x[i];
        gi[o] = a + g->b_ih[o];
        const float *u = g->w_hh + (size_t)o * H; float b = 0.0f;
        for (int i = 0; i < H; i++) b += u[i] * h[i];
        gh[o] = b + g->b_hh[o];
    }
    for (int j = 0; j < H; j++) {
        float r = sigmoidf_(gh[j] + gi[j]);
        float z = sigmoidf_(gh[H + j] + gi[H + j]);
        float n = tanhf(gi[2 * H + j] + gh[2 * H + j] * r);
        h[j] = (h[j] - n) * z + n;
    }
}

/* Conv1DStatefull, kernel 2: out = tanh(W0*hist[oldest] + W1*x + b); hist keeps `dil` frames */
static void conv_step(const lin_t *c, int dil, float *hist, float *out, const float *x)
{
    int in = c->n_in / 2;
    for (int o = 0; o < c->n_out; o++) {
        const float *w0 = c->w + (size_t)o * c->n_in, *w1 = w0 + in; float a = 0.0f;
        for (int i = 0; i < in; i++) a += w0[i] * hist[i];
        for (int i = 0; i < in; i++) a += w1[i] * x[i];
        out[o] = tanhf(a + c->b[o]);
    }
    if (dil > 1) memmove(hist, hist + in, sizeof(float) * in * (dil - 1));
    memcpy(hist + (size_t)in * (dil - 1), x, sizeof(float) * in);
}

static const int ENC_DIL[5] = { 1, 2, 2, 2, 2 };
struct orc_enc_state { float gru[5][64]; float conv[5][2 * 768]; };
struct orc_dec_state { float gru[5][96]; float conv[5][704]; };
orc_enc_state *orc_enc_new(void) { return calloc(1, sizeof(orc_enc_state)); }
orc_dec_state *orc_dec_new(void) { return calloc(1, sizeof(orc_dec_state)); }
void orc_enc_reset(orc_enc_state *s) { memset(s, 0, sizeof *s); }
void orc_dec_reset(orc_dec_state *s) { memset(s, 0, sizeof *s); }
void orc_enc_free(orc_enc_state *s) { free(s); }
void orc_dec_free(orc_dec_state *s) { free(s); }
const float *orc_enc_gru_state(const orc_enc_state *s, int layer) { return s->gru[layer - 1]; }
const float *orc_dec_gru_state(const orc_dec_state *s, int layer) { return s->gru[layer - 1]; }

void orc_core_encoder(const orc_model *m, orc_enc_state *s, float z[80], const float features[84])
{
    float buf[864]; int n = 0;
    dense(&m->enc_dense1, buf, features);
    for (int i = 0; i < 64; i++) buf[i] = clamp1(tanhf(buf[i]));
    n = 64;
    for (int l = 0; l < 5; l++) {
        gru_step(&m->enc_gru[l], s->gru[l], buf);
        for (int i = 0; i < 64; i++) buf[n + i] = clamp1(s->gru[l][i]);
        n += 64;
        conv_step(&m->enc_conv[l], ENC_DIL[l], s->conv[l], buf + n, buf);
        for (int i = 0; i < 96; i++) buf[n + i] = clamp1(buf[n + i]);
        n += 96;
    }
    dense(&m->enc_zdense, z, buf);      /* bottleneck 3: linear (radae_base.py:281-284) */
}

/* bottleneck 1 (model05, bbfm): z = tanh(z_dense(x)) -- radae_base.py:281-282.  features has 4*feature_dim floats
 * (80 for the 20-feature models), taken from the blob's enc_dense1 width. */
void orc_core_encoder_b1(const orc_model *m, orc_enc_state *s, float z[80], const float *features)
{
    orc_core_encoder(m, s, z, features);
    for (int i = 0; i < 80; i++) z[i] = tanhf(z[i]);
}
int orc_model_feat_width(const orc_model *m) { return m->enc_dense1.n_in; }

/* symbol-domain channels (radae.py:604-634 rate-Rs with bottleneck 1; bbfm.py:157-197) on n real symbols */
void orc_channel_rs(float *z_hat, const float *z, const float *H /* per QPSK symbol or NULL */, const float *noise, int n, float sigma)
{
    for (int i = 0; i < n; i++) z_hat[i] = z[i] * (H ? H[i >> 1] : 1.0f) + sigma * (noise ? noise[i] : 0.0f);
}
void orc_channel_bbfm(float *z_hat, const float *z, const float *H, const float *noise, int n, float CNRdB, float Gfm)
{
    for (int i = 0; i < n; i++) {
        float cnr = 20.0f * log10f(H ? H[i] : 1.0f) + CNRdB;
        float snr = fmaxf(cnr - 12.0f, 0.0f) + 12.0f + Gfm;
        snr += -fmaxf(-(cnr - 12.0f), 0.0f) * (1.0f + Gfm / 3.0f);
        float sigma = 1.0f / powf(powf(10.0f, snr / 10.0f), 0.5f);
        float v = z[i] + sigma * (noise ? noise[i] : 0.0f);
        z_hat[i] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
    }
}

void orc_core_decoder(const orc_model *m, orc_dec_state *s, float features[84], const float z_hat[80])
{
    float buf[736]; int n = 0; float gate[96];
    dense(&m->dec_dense1, buf, z_hat);
    for (int i = 0; i < 96; i++) buf[i] = clamp1(tanhf(buf[i]));
    n = 96;
    for (int l = 0; l < 5; l++) {
        gru_step(&m->dec_gru[l], s->gru[l], buf);
        float hc[96];
        for (int i = 0; i < 96; i++) hc[i] = clamp1(s->gru[l][i]);
        const lin_t *g = &m->dec_glu[l];                 /* GLU: x*sigmoid(Wx), no bias (radae_base.py:149-153) */
        for (int o = 0; o < 96; o++) { float a = 0.0f; for (int i = 0; i < 96; i++) a += g->w[o * 96 + i] * hc[i]; gate[o] = a; }
        for (int i = 0; i < 96; i++) buf[n + i] = clamp1(hc[i] * sigmoidf_(gate[i]));
        n += 96;
        conv_step(&m->dec_conv[l], 1, s->conv[l], buf + n, buf);
        for (int i = 0; i < 32; i++) buf[n + i] = clamp1(buf[n + i]);
        n += 32;
    }
    dense(&m->dec_output, features, buf);
}

/* ============================================================================================
 * 4. transmitter -- radae_txe.py:108-144, dsp.py:340-378, radae.py:441-455
 * ==========================================================================================*/
struct orc_bpf { orc_c32 mem[102]; int mem_len; orc_c32 phase; };     /* complex_bpf's state (section 6) */
struct orc_tx { const orc_model *m; orc_enc_state enc; orc_c32 eoo[ORC_NEOO]; int txbpf_en; struct orc_bpf txbpf; };

orc_tx *orc_tx_new(const orc_model *m)
{
    consts_init();
    orc_tx *t = calloc(1, sizeof *t);
    t->m = m; memcpy(t->eoo, K.eoo, sizeof K.eoo);
    t->txbpf.mem_len = 100; t->txbpf.phase = c32(1, 0);
    return t;
}
void orc_tx_free(orc_tx *t) { free(t); }

static void ofdm_symbols_to_time(orc_c32 *out, const orc_c32 sym[][ORC_NC], int nsym)
{   /* (nsym,30) x Winv (30,160) -> +CP -> flatten; dsp.py:362-372 */
    for (int s = 0; s < nsym; s++) {
        orc_c32 *o = out + s * ORC_SYM;
        for (int n = 0; n < ORC_M; n++) {
            orc_c32 a = c32(0, 0);
            for (int c = 0; c < ORC_NC; c++) a = caddf(a, cmulf(sym[s][c], K.Winv[c][n]));
            o[ORC_NCP + n] = a;
        }
        for (int n = 0; n < ORC_NCP; n++) o[n] = o[ORC_M + n];
    }
}

void orc_ofdm_mod(orc_c32 tx_out[960], const float z[240])
{
    consts_init();
    orc_c32 sym[ORC_NS + 1][ORC_NC];
    for (int c = 0; c < ORC_NC; c++) sym[0][c] = cscalef(K.P[c], K.pilot_gain_f);      /* dsp.py:355 */
    for (int k = 0; k < 120; k++) sym[1 + k / ORC_NC][k % ORC_NC] = c32(z[2 * k], z[2 * k + 1]); /* :341,:348-354 */
    ofdm_symbols_to_time(tx_out, (const orc_c32(*)[ORC_NC])sym, ORC_NS + 1);
    for (int n = 0; n < ORC_NMF; n++) tx_out[n] = pa_limit(tx_out[n]);                 /* :376-377 */
}

/* radae_txe.py:74-83, :130-132, :141-143 (--txbpf): the transmit samples through the same complex_bpf the receiver uses on its input, then
 * np.clip(abs(tx), 0, 1) * np.exp(1j * np.angle(tx)) */
void orc_tx_set_txbpf(orc_tx *t, int enable) { t->txbpf_en = enable; }
static void tx_bpf_clip(orc_tx *t, orc_c32 *tx, int n)
{
    orc_c32 y[ORC_NEOO];
    orc_bpf_run(&t->txbpf, y, tx, n);
    for (int i = 0; i < n; i++) {
        const float mag = (float)hypot((double)y[i].re, (double)y[i].im), ang = atan2f(y[i].im, y[i].re);
        const float m = mag > 1.0f ? 1.0f : mag;
        tx[i] = c32(m * cosf(ang), m * sinf(ang));
    }
}

void orc_tx_frame(orc_tx *t, orc_c32 tx_out[960], const float features_in[432], float *z_out)
{
    float z[240], feat[84];
    for (int c = 0; c < ORC_NZMF; c++) {                                               /* rade_api.c:426-434 == radae_txe.py:114-123 */
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 20; j++) feat[i * 21 + j] = features_in[(c * 4 + i) * 36 + j];
            feat[i * 21 + 20] = -1.0f;
        }
        orc_core_encoder(t->m, &t->enc, z + 80 * c, feat);
    }
    if (z_out) memcpy(z_out, z, sizeof z);
    orc_ofdm_mod(tx_out, z);
    if (t->txbpf_en) tx_bpf_clip(t, tx_out, ORC_NMF);                                 /* radae_txe.py:130-132 */
}

void orc_tx_set_eoo_bits(orc_tx *t, const float bits[180])
{   /* radae.py:441-455: 90 QPSK symbols -> 3 OFDM symbols placed after [P][Pend] */
    orc_c32 sym[ORC_NS - 1][ORC_NC]; orc_c32 td[(ORC_NS - 1) * ORC_SYM];
    for (int k = 0; k < 90; k++) sym[k / ORC_NC][k % ORC_NC] = c32(bits[2 * k], bits[2 * k + 1]);
    ofdm_symbols_to_time(td, (const orc_c32(*)[ORC_NC])sym, ORC_NS - 1);
    for (int n = 0; n < (ORC_NS - 1) * ORC_SYM; n++) t->eoo[2 * ORC_SYM + n] = pa_limit(cscalef(td[n], K.pilot_gain_f));
}
void orc_tx_eoo(orc_tx *t, orc_c32 out[1152]) { memcpy(out, t->eoo, sizeof t->eoo); if (t->txbpf_en) tx_bpf_clip(t, out, ORC_NEOO); }   /* radae_txe.py:138-144 */

/* ============================================================================================
 * 5. channel -- radae.py:529-589 with batch 1 (global mean == per utterance), inference.py:263-275
 * ==========================================================================================*/
float orc_sigma_from_EbNodB(float EbNodB)
{   /* radae.py:567-573, bottleneck 3: sigma = sqrt(Fs/(EbNo*Rb)), Rb = 80/0.04 */
    float EbNo = powf(10.0f, EbNodB / 10.0f);
    float Rb = (float)(80.0 / (0.01 * 4));
    return powf(8000.0f / (EbNo * Rb), 0.5f);
}

static float omega_at(int i, float freq_offset, float df_dt)
{   /* freq = f0*1 + df_dt*arange/Fs ; omega = freq*2*pi/Fs, float32 (radae.py:547-549) */
    float freq = freq_offset + (df_dt * (float)i) / 8000.0f;
    return ((freq * 2.0f) * (float)PI_D) / 8000.0f;
}

void orc_channel(orc_c32 *rx, const orc_c32 *tx, int n, const orc_c32 *G, const orc_c32 *noise,
                 float sigma, float freq_offset, float df_dt, orc_c32 *final_phase)
{
    const int d = 16;
    orc_c32 *mp = malloc(sizeof(orc_c32) * n);
    for (int i = 0; i < n; i++) {
        orc_c32 g0 = G ? G[2 * i] : c32(1, 0);
        mp[i] = cmulf(tx[i], g0);
    }
    if (G) for (int i = d; i < n; i++) mp[i] = caddf(mp[i], cmulf(tx[i - d], G[2 * (i - d) + 1]));
    double p_tx = 0, p_mp = 0;
    for (int i = 0; i < n; i++) { float a = cabsf_(tx[i]), b = cabsf_(mp[i]); p_tx += (double)(a * a); p_mp += (double)(b * b); }
    float tx_power = (float)(p_tx / n), mp_power = (float)(p_mp / n);
    float mp_gain = powf(tx_power / mp_power, 0.5f);
    double acc = 0.0;                        /* torch.cumsum(float32) accumulates in double on CPU */
    orc_c32 lp = c32(1, 0);
    for (int i = 0; i < n; i++) {
        orc_c32 v = cscalef(mp[i], mp_gain);
        if (freq_offset != 0.0f) {           /* radae.py:546 `if self.freq_offset:` */
            acc += (double)omega_at(i, freq_offset, df_dt);
            lp = cexpjf((float)acc);
            v = cmulf(v, lp);
        }
        if (noise) v = caddf(v, cscalef(noise[i], sigma));
        rx[i] = v;
    }
    if (final_phase) *final_phase = lp;
    free(mp);
}

void orc_channel_eoo(orc_c32 *rx, const orc_c32 *eoo, int n, const orc_c32 *noise, float sigma,
                     float freq_offset, float df_dt, orc_c32 final_phase)
{
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        acc += (double)omega_at(i, freq_offset, df_dt);
        orc_c32 v = cmulf(cmulf(eoo[i], cexpjf((float)acc)), final_phase);
        if (noise) v = caddf(v, cscalef(noise[i], sigma));
        rx[i] = v;
    }
}

/* ============================================================================================
 * 6. receiver DSP
 * ==========================================================================================*/
/* ---- complex_bpf.bpf, dsp.py:63-102.  Quirk kept: memory is Ntap-1 = 100 samples on the first
 *      call and Ntap+1 = 102 afterwards (:55 vs :96) while the window always starts at index 0. */
orc_bpf *orc_bpf_new(void) { consts_init(); orc_bpf *b = calloc(1, sizeof *b); b->mem_len = 100; b->phase = c32(1, 0); return b; }
void orc_bpf_free(orc_bpf *b) { free(b); }

void orc_bpf_run(orc_bpf *b, orc_c32 *out, const orc_c32 *in, int n)
{
    orc_c32 xm[102 + ORC_NEOO], pv[ORC_NEOO];
    int ml = b->mem_len;
    memcpy(xm, b->mem, sizeof(orc_c32) * ml);
    for (int i = 0; i < n; i++) { pv[i] = cmulf(b->phase, K.bpf_pv[i]); xm[ml + i] = cmulf(in[i], pv[i]); }
    for (int i = 0; i < n; i++) {
        float ar = 0.0f, ai = 0.0f;
        for (int k = 0; k < ORC_NTAP; k++) { ar += xm[i + k].re * K.bpf_h[k]; ai += xm[i + k].im * K.bpf_h[k]; }
        out[i] = cmulf(c32(ar, ai), cconjf_(pv[i]));
    }
    int tot = ml + n;
    memmove(b->mem, xm + tot - 102, sizeof(orc_c32) * 102);   /* [-Ntap-1:] */
    b->mem_len = 102;
    b->phase = pv[n - 1];
}

/* ---- acquisition, dsp.py:152-320 --------------------------------------------------------- */
typedef struct {
    float absDt1[ORC_NMF][ORC_NFCOARSE], absDt2[ORC_NMF][ORC_NFCOARSE]; /* only |Dt| is ever read back */
    double Dthresh, Dtmax12, Dtmax12_eoo; int f_ind_max;
    uint32_t lcg;
} acq_t;

static void corr_row(const orc_c32 *rxc /* conj(rx)+t */, float *absrow)
{   /* np.matmul(conj(rx)[t:t+M], p_w) then abs -- dsp.py:207-209 */
    orc_c32 acc[ORC_NFCOARSE];
    for (int f = 0; f < ORC_NFCOARSE; f++) acc[f] = c32(0, 0);
    for (int m = 0; m < ORC_M; m++) {
        orc_c32 x = rxc[m];
        for (int f = 0; f < ORC_NFCOARSE; f++) { orc_c32 w = K.p_w[m][f]; acc[f].re += x.re * w.re - x.im * w.im; acc[f].im += x.re * w.im + x.im * w.re; }
    }
    for (int f = 0; f < ORC_NFCOARSE; f++) absrow[f] = cabsf_(acc[f]);
}

static float mean_abs(const float a[ORC_NMF][ORC_NFCOARSE])
{
    double s = 0; for (int t = 0; t < ORC_NMF; t++) for (int f = 0; f < ORC_NFCOARSE; f++) s += a[t][f];
    return (float)(s / (ORC_NMF * ORC_NFCOARSE));
}

static float sigma_r_of(const acq_t *a)
{   /* dsp.py:218-220, float32 scalars */
    const float k = (float)pow(PI_D / 2.0, 0.5);
    float s1 = mean_abs(a->absDt1) / k, s2 = mean_abs(a->absDt2) / k;
    return (s1 + s2) / 2.0f;
}

static int detect_pilots(acq_t *a, const orc_c32 *rx, int *tmax_out, double *fmax_out)
{
    orc_c32 rxc[ORC_RXBUF];
    for (int i = 0; i < ORC_RXBUF; i++) rxc[i] = cconjf_(rx[i]);
    float Dtmax12 = 0.0f; int f_ind_max = 0, tmax = 0; double fmax = 0;
    for (int t = 0; t < ORC_NMF; t++) {
        corr_row(rxc + t, a->absDt1[t]); corr_row(rxc + t + ORC_NMF, a->absDt2[t]);
        float lmax = -1.0f; int larg = 0;
        for (int f = 0; f < ORC_NFCOARSE; f++) { float v = a->absDt1[t][f] + a->absDt2[t][f]; if (v > lmax) { lmax = v; larg = f; } }
        if (lmax > Dtmax12) { Dtmax12 = lmax; f_ind_max = larg; fmax = K.fcoarse[larg]; tmax = t; }
    }
    float sr = sigma_r_of(a);
    a->Dthresh = (double)(2.0f * sr) * sqrt(-log(0.00001 / 5.0));                     /* :221 */
    a->Dtmax12 = Dtmax12; a->f_ind_max = f_ind_max;
    *tmax_out = tmax; *fmax_out = fmax;
    return (double)Dtmax12 > a->Dthresh;
}

/* np.arange(start, stop, step) for doubles: len = ceil((stop-start)/step), v[i] = start + i*((start+step)-start) */
static int arange_d(double start, double stop, double step, double *v, int maxn)
{
    int len = (int)ceil((stop - start) / step);
    if (len > maxn) len = maxn;
    double delta = (start + step) - start;
    for (int i = 0; i < len; i++) v[i] = start + i * delta;
    return len;
}

/* diagnostic (tools/parity_sweep.py): how far the runner-up cell of the most recent refine() arg-max was below the winner, relative to the winner, and where it
 * was -- the reference takes the FIRST maximum of float32 magnitudes of complex128 sums, so two cells within float32 rounding of each other are a tie that a
 * different (equally valid) summation order resolves the other way */
double orc_debug_refine_margin = 1.0; int orc_debug_refine_second_t = 0; double orc_debug_refine_second_f = 0.0;

static void refine(const orc_c32 *rx, int *tmax, double *fmax, int t0, int t1, const double *fr, int nf)
{   /* dsp.py:233-270; complex128 dot products rounded to complex64, |Dt1+Dt2| in float32 */
    float Dtmax = 0.0f; int tbest = *tmax; double fbest = *fmax;
    float second = 0.0f; int tsec = *tmax; double fsec = *fmax;
    for (int fi = 0; fi < nf; fi++) {
        double w = 2.0 * PI_D * fr[fi] / 8000.0;
        c64d wp1[ORC_M], wp2[ORC_M];
        c64d rot = { cos(-w * ORC_NMF), sin(-w * ORC_NMF) };
        for (int n = 0; n < ORC_M; n++) {
            c64d wv = { cos(-w * n), sin(-w * n) }, pc = { K.p[n].re, -(double)K.p[n].im };
            wp1[n] = cmuld(wv, pc); wp2[n] = cmuld(cmuld(wv, rot), pc);
        }
        for (int t = t0; t < t1; t++) {
            c64d a = { 0, 0 }, b = { 0, 0 };
            for (int n = 0; n < ORC_M; n++) {
                c64d x = { rx[t + n].re, rx[t + n].im }, y = { rx[t + ORC_NMF + n].re, rx[t + ORC_NMF + n].im };
                c64d q = cmuld(x, wp1[n]), r = cmuld(y, wp2[n]);
                a.re += q.re; a.im += q.im; b.re += r.re; b.im += r.im;
            }
            orc_c32 s = caddf(c32((float)a.re, (float)a.im), c32((float)b.re, (float)b.im));
            float v = cabsf_(s);
            if (v > Dtmax) { second = Dtmax; tsec = tbest; fsec = fbest; Dtmax = v; tbest = t; fbest = fr[fi]; }
            else if (v > second) { second = v; tsec = t; fsec = fr[fi]; }
        }
    }
    orc_debug_refine_margin = Dtmax > 0.0f ? ((double)Dtmax - (double)second) / (double)Dtmax : 1.0; orc_debug_refine_second_t = tsec; orc_debug_refine_second_f = fsec;
    *tmax = tbest; *fmax = fbest;
}

static double corr_abs_d(const orc_c32 *rx, double w, const orc_c32 *ref)
{   /* |dot(conj(w_vec*rx), ref)| in complex128, dsp.py:307 */
    c64d acc = { 0, 0 };
    for (int n = 0; n < ORC_M; n++) {
        c64d wv = { cos(-w * n), sin(-w * n) }, x = { rx[n].re, rx[n].im }, q = cmuld(wv, x);
        q.im = -q.im;
        c64d r = { ref[n].re, ref[n].im }, pr = cmuld(q, r);
        acc.re += pr.re; acc.im += pr.im;
    }
    return hypot(acc.re, acc.im);
}

static void check_pilots(acq_t *a, const orc_c32 *rx, int tmax, double fmax, int *valid, int *endofover)
{   /* dsp.py:273-320 */
    orc_c32 rxc[ORC_RXBUF];
    for (int i = 0; i < ORC_RXBUF; i++) rxc[i] = cconjf_(rx[i]);
    for (int i = 0; i < 48; i++) {                       /* int(0.05*960) random row refreshes, LCG instead of np.random */
        a->lcg = a->lcg * 1664525u + 1013904223u;
        int t = (int)((a->lcg >> 8) % ORC_NMF);
        corr_row(rxc + t, a->absDt1[t]); corr_row(rxc + t + ORC_NMF, a->absDt2[t]);
    }
    float sr = sigma_r_of(a);
    double Dthresh = (double)(2.0f * sr) * sqrt(-log(0.0001 / 5.0));
    double Dthresh_eoo = (double)(2.0f * sr) * sqrt(-log(0.00001 / 5.0));
    double w = 2.0 * PI_D * fmax / 8000.0;
    double D = corr_abs_d(rx + tmax, w, K.p) + corr_abs_d(rx + tmax + ORC_NMF, w, K.p);
    double De = corr_abs_d(rx + tmax + ORC_M + ORC_NCP, w, K.pend) + corr_abs_d(rx + tmax + ORC_NMF, w, K.pend);
    *valid = D > Dthresh; *endofover = De > Dthresh_eoo;
    a->Dthresh = Dthresh; a->Dtmax12 = D; a->Dtmax12_eoo = De;
}

/* ---- receiver_one, dsp.py:418-526 --------------------------------------------------------- */
static void est_pilot_row(orc_c32 out[ORC_NC], const orc_c32 row[ORC_NC])
{   /* dsp.py:418-435 */
    for (int c = 0; c < ORC_NC; c++) {
        int cm = c == 0 ? 1 : (c == ORC_NC - 1 ? ORC_NC - 2 : c);
        orc_c32 h[3], g0 = c32(0, 0), g1 = c32(0, 0);
        for (int k = 0; k < 3; k++) { float pr = K.P[cm - 1 + k].re; h[k] = c32(row[cm - 1 + k].re / pr, row[cm - 1 + k].im / pr); }
        for (int k = 0; k < 3; k++) { g0 = caddf(g0, cmulf(K.Pmat[c][0][k], h[k])); g1 = caddf(g1, cmulf(K.Pmat[c][1][k], h[k])); }
        out[c] = caddf(g0, cmulf(g1, K.eq_rot[c]));
    }
}

static void receiver_one(float *snr_state, float *z_hat, const orc_c32 rx[1152], int endofover)
{
    orc_c32 sym[6][ORC_NC];
    for (int s = 0; s < 6; s++) for (int c = 0; c < ORC_NC; c++) {                     /* :497-501 window [16:176] */
        orc_c32 a = c32(0, 0); const orc_c32 *x = rx + s * ORC_SYM + ORC_NCP - 16;
        for (int n = 0; n < ORC_M; n++) a = caddf(a, cmulf(x[n], K.Wfwd[n][c]));
        sym[s][c] = a;
    }
    if (!endofover) {
        orc_c32 rp[2][ORC_NC];
        est_pilot_row(rp[0], sym[0]); est_pilot_row(rp[1], sym[5]);
        /* update_snr_est, :438-456 */
        float S1 = 0.0f, S2 = 0.0f;
        for (int c = 0; c < ORC_NC; c++) {
            float ph = atan2f(rp[0][c].im, rp[0][c].re);
            orc_c32 r = cmulf(sym[0][c], cexpjf(-ph));
            float a = cabsf_(sym[0][c]); S1 += a * a; float b = fabsf(r.im); S2 += b * b;
        }
        S2 += 1e-12f;
        float snr_est = S1 / (2.0f * S2) - 1.0f;
        if (snr_est <= 0.0f) snr_est = 0.1f;
        float snrdB = 10.0f * log10f(snr_est);
        snrdB = (snrdB - 2.513f) / 0.8070f;
        float snr3k = snrdB + (float)(10.0 * log10(50.0 * 30 / 3000.0)) + (float)(10.0 * log10(192.0 / 160.0));
        *snr_state = 0.9f * *snr_state + 0.1f * snr3k;
        /* linear interpolation phase EQ, :468-474 */
        for (int c = 0; c < ORC_NC; c++) {
            orc_c32 d = csubf(rp[1][c], rp[0][c]); orc_c32 slope = c32(d.re / 5.0f, d.im / 5.0f);
            for (int k = 1; k <= ORC_NS; k++) {
                orc_c32 ch = caddf(cscalef(slope, (float)k), rp[0][c]);
                float ang = atan2f(ch.im, ch.re);
                sym[k][c] = cmulf(sym[k][c], cexpjf(-ang));
            }
        }
        /* coarse magnitude, :477-482 */
        float acc = 0.0f;
        for (int i = 0; i < 2; i++) for (int c = 0; c < ORC_NC; c++) { float a = cabsf_(rp[i][c]); acc += a * a; }
        float mag = powf(acc / 60.0f, 0.5f) + 1e-6f;
        mag = (mag * fabsf(K.P[0].re)) / K.pilot_gain_f;
        for (int k = 1; k <= ORC_NS; k++) for (int c = 0; c < ORC_NC; c++) {
            int idx = (k - 1) * ORC_NC + c;
            z_hat[2 * idx] = sym[k][c].re / mag; z_hat[2 * idx + 1] = sym[k][c].im / mag;
        }
    } else {
        /* EOO branch :513-524 -- mean of three pilots per carrier, symbols 2..4 carry 180 soft bits */
        for (int c = 0; c < ORC_NC; c++) {
            float pp = K.P[c].re, pe = K.Pend[c].re;
            orc_c32 s = c32(sym[0][c].re / pp + sym[1][c].re / pe + sym[5][c].re / pe, sym[0][c].im / pp + sym[1][c].im / pe + sym[5][c].im / pe);
            float ang = atan2f(s.im, s.re); orc_c32 rot = cexpjf(-ang);
            for (int k = 2; k <= 4; k++) { orc_c32 v = cmulf(sym[k][c], rot); int idx = (k - 2) * ORC_NC + c; z_hat[2 * idx] = v.re; z_hat[2 * idx + 1] = v.im; }
        }
    }
}

/* ---- radae_rx.do_radae_rx state machine, radae_rxe.py:171-330 ------------------------------ */
enum { ST_SEARCH = 0, ST_CANDIDATE = 1, ST_SYNC = 2 };
struct orc_rx {
    const orc_model *m; orc_dec_state dec; orc_bpf bpf; acq_t acq;
    orc_c32 rx_buf[ORC_RXBUF];
    int state, nin, tmax, tmax_candidate, valid_count, uw_errors, synced_count, mf;
    double fmax, foff_err; c64d rx_phase; float snrdB_3k_est;
    double disable_unsync;                  /* radae_rxe.py:277-281: seconds of sync after which unsync paths are switched off (0 = normal) */
};

orc_rx *orc_rx_new(const orc_model *m)
{
    consts_init();
    orc_rx *r = calloc(1, sizeof *r);
    r->m = m; r->bpf.mem_len = 100; r->bpf.phase = c32(1, 0);
    r->nin = ORC_NMF; r->state = ST_SEARCH; r->mf = 1; r->rx_phase.re = 1.0; r->acq.lcg = 1;
    return r;
}
void orc_rx_free(orc_rx *r) { free(r); }
void orc_rx_set_lcg(orc_rx *r, unsigned seed) { r->acq.lcg = seed; }
void orc_rx_set_foff_err(orc_rx *r, double hz) { r->foff_err = hz; }
void orc_rx_set_disable_unsync(orc_rx *r, double seconds) { r->disable_unsync = seconds; }
int orc_rx_nin(const orc_rx *r) { return r->nin; }
int orc_rx_sync(const orc_rx *r) { return r->state == ST_SYNC; }
int orc_rx_snr(const orc_rx *r) { return (int)r->snrdB_3k_est; }
void orc_rx_get_trace(const orc_rx *r, orc_rx_trace *t)
{
    t->state = r->state; t->nin = r->nin; t->tmax = r->tmax; t->f_ind_max = r->acq.f_ind_max; t->valid_count = r->valid_count;
    t->uw_errors = r->uw_errors; t->synced_count = r->synced_count; t->mf = r->mf; t->fmax = r->fmax;
    t->Dthresh = r->acq.Dthresh; t->Dtmax12 = r->acq.Dtmax12; t->Dtmax12_eoo = r->acq.Dtmax12_eoo; t->snrdB_3k_est = r->snrdB_3k_est;
}

int orc_rx_frame(orc_rx *r, float features_out[432], float eoo_out[180], const orc_c32 *rx_in, float *z_hat_out)
{
    const int Nmf = ORC_NMF, M = ORC_M, Ncp = ORC_NCP, Nmf_unsync = 25;
    int valid_output = 0, endofover = 0, uw_fail = 0, candidate = 0;
    float z_hat[240];
    orc_c32 filt[ORC_NIN_MAX];
    int nin = r->nin;
    orc_bpf_run(&r->bpf, filt, rx_in, nin);                                            /* :193-195 */
    memmove(r->rx_buf, r->rx_buf + nin, sizeof(orc_c32) * (ORC_RXBUF - nin));          /* :196-197 */
    memcpy(r->rx_buf + ORC_RXBUF - nin, filt, sizeof(orc_c32) * nin);

    if (r->state == ST_SEARCH || r->state == ST_CANDIDATE) {
        candidate = detect_pilots(&r->acq, r->rx_buf, &r->tmax, &r->fmax);             /* :198-199 */
    } else {
        double fr[32]; int nf = arange_d(r->fmax - 1, r->fmax + 1, 0.1, fr, 32);      /* :202-206 */
        int t0 = r->tmax - 8 > 0 ? r->tmax - 8 : 0, t1 = r->tmax + 8;
        double fhat = r->fmax;
        refine(r->rx_buf, &r->tmax, &fhat, t0, t1, fr, nf);
        r->fmax = 0.9 * r->fmax + 0.1 * fhat;
        check_pilots(&r->acq, r->rx_buf, r->tmax, r->fmax, &candidate, &endofover);
        r->nin = Nmf;                                                                  /* :209-218 */
        if (r->tmax >= Nmf - M) { r->nin = Nmf + M; r->tmax -= M; }
        if (r->tmax < M) { r->nin = Nmf - M; r->tmax += M; }
        r->synced_count++;                                                             /* :220-224 */
        if (r->synced_count % (8000 / Nmf) == 0) { if (r->uw_errors > 7) uw_fail = 1; r->uw_errors = 0; }
        double w = 2.0 * PI_D * r->fmax / 8000.0;                                      /* :227-233 */
        c64d step = { cos(-w), sin(-w) };
        orc_c32 rx1[1152];
        const orc_c32 *src = r->rx_buf + r->tmax - Ncp;
        for (int n = 0; n < Nmf + M + Ncp; n++) {
            r->rx_phase = cmuld(r->rx_phase, step);
            rx1[n] = cmulf(src[n], c32((float)r->rx_phase.re, (float)r->rx_phase.im));
        }
        receiver_one(&r->snrdB_3k_est, z_hat, rx1, endofover);                         /* :236 */
        valid_output = !endofover;
    }

    int next_state = r->state;                                                         /* :248-297 */
    if (r->state == ST_SEARCH) {
        if (candidate) { next_state = ST_CANDIDATE; r->tmax_candidate = r->tmax; r->valid_count = 1; }
    } else if (r->state == ST_CANDIDATE) {
        if (candidate && abs(r->tmax - r->tmax_candidate) < Ncp) {
            r->valid_count++;
            if (r->valid_count > 3) {
                next_state = ST_SYNC;
                orc_dec_reset(&r->dec);
                r->synced_count = 0; uw_fail = 0; r->uw_errors = 0; r->valid_count = Nmf_unsync;
                double fr[96]; int nf = arange_d(r->fmax - 10, r->fmax + 10, 0.25, fr, 96);
                int t0 = r->tmax - 1 > 0 ? r->tmax - 1 : 0, t1 = r->tmax + 2;
                refine(r->rx_buf, &r->tmax, &r->fmax, t0, t1, fr, nf);
                r->fmax += r->foff_err; r->foff_err = 0;
            }
        } else next_state = ST_SEARCH;
    } else {
        int unsync_enable = 1;                                                         /* :277-281 (test mode --disable_unsync) */
        if (r->disable_unsync != 0.0 && r->synced_count > (int)(r->disable_unsync * 8000.0 / ORC_NMF)) unsync_enable = 0;
        if (candidate) r->valid_count = Nmf_unsync;
        else { r->valid_count--; if (unsync_enable && r->valid_count == 0) next_state = ST_SEARCH; }
        if (unsync_enable && (endofover || uw_fail)) next_state = ST_SEARCH;
    }
    r->state = next_state;
    if (r->state == ST_SEARCH) r->nin = Nmf;
    r->mf++;

    if (valid_output) {                                                                /* :300-319 == rade_api.c:480-513 */
        float feat[84]; int uw = 0;
        memset(features_out, 0, sizeof(float) * 432);
        for (int c = 0; c < ORC_NZMF; c++) {
            orc_core_decoder(r->m, &r->dec, feat, z_hat + 80 * c);
            for (int i = 0; i < 4; i++) for (int j = 0; j < 20; j++) features_out[(c * 4 + i) * 36 + j] = feat[i * 21 + j];
            if (feat[20] > 0) uw++;
        }
        r->uw_errors += uw;
        if (z_hat_out) memcpy(z_hat_out, z_hat, sizeof z_hat);
    }
    if (endofover) {                                                                   /* :321-323 */
        memcpy(eoo_out, z_hat, sizeof(float) * 180);
        if (z_hat_out) { memset(z_hat_out, 0, sizeof(float) * 240); memcpy(z_hat_out, z_hat, sizeof(float) * 180); }
    }
    return valid_output | (endofover << 1);
}

/* ============================================================================================
 * 7. loss -- radae_base.py:50-68, loss.py:64-91
 * ==========================================================================================*/
double orc_distortion_loss(const float *yt, const float *yp, int nframes, int dim, int stride)
{
    double tot = 0;
    for (int t = 0; t < nframes; t++) {
        const float *a = yt + (size_t)t * stride, *b = yp + (size_t)t * stride;
        float acc = 0.0f;
        float pitch_error = 2.0f * (b[18] - a[18]), corr_error = b[19] - a[19];
        float pw = a[19] + 0.5f; pw = pw > 0 ? pw * pw : 0.0f;
        float data_error = dim == 21 ? b[20] - a[20] : 0.0f;
        float extra = (float)(3.0 * (10.0 / 18.0)) * fabsf(pitch_error) * pw + (float)(1.0 / 18.0) * corr_error * corr_error + (float)(0.5 / 18.0) * data_error * data_error;
        for (int i = 0; i < 18; i++) { float e = b[i] - a[i]; acc += e * e + extra; }
        tot += (double)(acc / 18.0f);
    }
    return tot / nframes;
}

double orc_find_loss(const float *features, int n, const float *features_hat, int n_hat, int stride, int *start)
{
    double best = orc_distortion_loss(features, features_hat, n_hat, 20, stride); int bs = 0;
    for (int s = 0; s < n - n_hat; s++) {
        double l = orc_distortion_loss(features + (size_t)s * stride, features_hat, n_hat, 20, stride);
        if (l < best) { best = l; bs = s; }
    }
    if (start) *start = bs;
    return best;
}
