/*
 * rade_host.c -- host-side (plain C) model preparation for the HIP engine:
 *   1. constant tables of the OFDM modem (rd_tables), evaluated with the same float32/float64 steps
 *      NumPy/PyTorch take in the reference so the device kernels see bit-comparable constants;
 *   2. DNNw weight-blob reader -> fp32 matrices (the blob is int8 + per-row scales for most layers);
 *   3. weight packing into the MFMA fragment order k_gemm streams.
 *
 * Reference: radae/radae.py:128-234 (numerology/DFT/pilots/EOO), radae/dsp.py:40-61 (BPF),
 * :153-176 (p_w), :400-416 (Pmat); blob format src/write_rade_weights.c:51-74 and
 * weight-exchange/wexchange/c_export/common.py:59-69,140-176,263-271,290-293,307-311,360-368.
 */
#include "rade_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI_D 3.14159265358979323846

/* ---------------------------------------------------------------------------------------------
 * 1. tables
 * -------------------------------------------------------------------------------------------*/
static void pa_limit_host(float *re, float *im)
{   /* tanh(|x|)*exp(1j*angle(x)) in float32 (radae.py:218) */
    float mag = hypotf(*re, *im), ang = atan2f(*im, *re), t = tanhf(mag);
    *re = t * cosf(ang); *im = t * sinf(ang);
}

void rd_tables_fill(rd_tables *T)
{
    static const float barker13[13] = { 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1 };
    float w[RD_NC];
    memset(T, 0, sizeof *T);
    /* w = 2*pi*(15+arange(30))/160 as float32 tensor arithmetic (radae.py:172-174) */
    const float two_pi = (float)(2.0 * PI_D);
    for (int c = 0; c < RD_NC; c++) w[c] = (two_pi * (float)(15 + c)) / 160.0f;
    for (int c = 0; c < RD_NC; c++)
        for (int n = 0; n < RD_M; n++) {
            const float arg = (float)n * w[c];          /* float32 product, then exp(1j*arg) in float32 */
            const float cs = (float)cos((double)arg), sn = (float)sin((double)arg);
            T->Winv[c][n][0] = cs / 160.0f; T->Winv[c][n][1] = sn / 160.0f;
            T->Wfwd[n][c][0] = cs; T->Wfwd[n][c][1] = -sn;
        }
    const float r2 = (float)sqrt(2.0);
    for (int c = 0; c < RD_NC; c++) { T->P[c] = r2 * barker13[c % 13]; T->Pend[c] = (c & 1) ? -T->P[c] : T->P[c]; }
    for (int n = 0; n < RD_M; n++) {                    /* p = P @ Winv, pend = Pend @ Winv (complex64) */
        float ar = 0, ai = 0, br = 0, bi = 0;
        for (int c = 0; c < RD_NC; c++) {
            const float wr = T->Winv[c][n][0], wi = T->Winv[c][n][1];
            ar += T->P[c] * wr - 0.0f * wi; ai += T->P[c] * wi + 0.0f * wr;
            br += T->Pend[c] * wr - 0.0f * wi; bi += T->Pend[c] * wi + 0.0f * wr;
        }
        T->p[n][0] = ar; T->p[n][1] = ai; T->pend[n][0] = br; T->pend[n][1] = bi;
    }
    T->pilot_gain = (float)(pow(10.0, -2.0 / 20.0) * 160.0 / sqrt(30.0));
    /* default EOO frame: [p_cp][pend_cp][0][0][0][pend_cp] * pilot_gain, PA-limited */
    for (int n = 0; n < RD_SYM; n++) {
        const int src = n < RD_NCP ? RD_M - RD_NCP + n : n - RD_NCP;
        T->eoo[n][0] = T->p[src][0]; T->eoo[n][1] = T->p[src][1];
        T->eoo[RD_SYM + n][0] = T->pend[src][0]; T->eoo[RD_SYM + n][1] = T->pend[src][1];
        T->eoo[RD_NMF + n][0] = T->pend[src][0]; T->eoo[RD_NMF + n][1] = T->pend[src][1];
    }
    for (int n = 0; n < RD_NEOO; n++) { T->eoo[n][0] *= T->pilot_gain; T->eoo[n][1] *= T->pilot_gain; pa_limit_host(&T->eoo[n][0], &T->eoo[n][1]); }
    /* 3-pilot least-squares matrices Pmat[c] = inv(A^T A) A^T with A = [[1, e^{-j w a}]] (plain transpose) */
    for (int c = 0; c < RD_NC; c++) {
        const int cm = c == 0 ? 1 : (c == RD_NC - 1 ? RD_NC - 2 : c);
        double er[3], ei[3];
        for (int k = 0; k < 3; k++) { const float ang = -(w[cm - 1 + k] * 20.0f); er[k] = (float)cos((double)ang); ei[k] = (float)sin((double)ang); }
        double s1r = er[0] + er[1] + er[2], s1i = ei[0] + ei[1] + ei[2], s2r = 0, s2i = 0;
        for (int k = 0; k < 3; k++) { s2r += er[k] * er[k] - ei[k] * ei[k]; s2i += 2 * er[k] * ei[k]; }
        double dr = 3 * s2r - (s1r * s1r - s1i * s1i), di = 3 * s2i - 2 * s1r * s1i;
        double dn = dr * dr + di * di, ir = dr / dn, ii = -di / dn;
        for (int k = 0; k < 3; k++) {
            double a0r = s2r - (s1r * er[k] - s1i * ei[k]), a0i = s2i - (s1r * ei[k] + s1i * er[k]);
            double a1r = 3 * er[k] - s1r, a1i = 3 * ei[k] - s1i;
            T->Pmat[c][0][k][0] = (float)(ir * a0r - ii * a0i); T->Pmat[c][0][k][1] = (float)(ir * a0i + ii * a0r);
            T->Pmat[c][1][k][0] = (float)(ir * a1r - ii * a1i); T->Pmat[c][1][k][1] = (float)(ir * a1i + ii * a1r);
        }
        const float ang = -(w[c] * 20.0f);
        T->eq_rot[c][0] = (float)cos((double)ang); T->eq_rot[c][1] = (float)sin((double)ang);
    }
    /* input band-pass filter: 101-tap sinc low-pass at baseband, float32 like NumPy builds it */
    const float bandwidth = ((1.2f * (w[RD_NC - 1] - w[0])) * 8000.0f) / two_pi;
    const float centre = (((w[RD_NC - 1] + w[0]) * 8000.0f) / two_pi) / 2.0f;
    const float Bn = bandwidth / 8000.0f, alpha = (two_pi * centre) / 8000.0f;
    for (int i = 0; i < RD_NTAP; i++) {
        const float x = (float)(i - 50) * Bn;
        const float y = (float)PI_D * (x == 0.0f ? 1.0e-20f : x);
        T->bpf_h[i] = Bn * ((float)sin((double)y) / y);
    }
    for (int k = 0; k < RD_NEOO; k++) {
        const float arg = (float)((double)alpha * (double)(k + 1));
        T->bpf_E[k][0] = (float)cos((double)arg); T->bpf_E[k][1] = (float)-sin((double)arg);
    }
    /* coarse acquisition grid and frequency-shifted pilot replicas */
    for (int f = 0; f < RD_NFC; f++) {
        T->fcoarse[f] = -50.0 + 2.5 * f;
        const double wf = 2.0 * PI_D * T->fcoarse[f] / 8000.0;
        for (int n = 0; n < RD_M; n++) {
            const double cr = cos(wf * n), ci = sin(wf * n), pr = T->p[n][0], pi = T->p[n][1];
            T->p_w[n][f][0] = (float)(cr * pr - ci * pi); T->p_w[n][f][1] = (float)(cr * pi + ci * pr);
        }
    }
    T->snr_c1 = (float)(10.0 * log10(50.0 * 30 / 3000.0));
    T->snr_c2 = (float)(10.0 * log10(192.0 / 160.0));
}

/* ---------------------------------------------------------------------------------------------
 * 2. DNNw blob
 * -------------------------------------------------------------------------------------------*/
typedef struct { const char *name; int type, size; const unsigned char *data; } dnnw_rec;

static const dnnw_rec *rec_find(const dnnw_rec *r, int n, const char *base, const char *suffix)
{
    char nm[96];
    snprintf(nm, sizeof nm, "%s%s", base, suffix);
    for (int i = 0; i < n; i++) if (!strcmp(r[i].name, nm)) return &r[i];
    return NULL;
}

/* Every record is validated before anything is written: sizes against the shapes they imply, index walks against their
 * array, allocations against NULL.  The blob path comes from rade_open()'s argument or $RADE_MODEL_FILE, i.e. it is
 * untrusted input. */
static int lin_alloc(rd_linear *l)
{
    if (l->n_in <= 0 || l->n_out <= 0 || l->n_in > 65536 || l->n_out > 65536) return -1;
    l->b = malloc(sizeof(float) * (size_t)l->n_out); l->w = calloc((size_t)l->n_in * l->n_out, sizeof(float)); l->row_scale = NULL;
    if (!l->b || !l->w) { free(l->b); free(l->w); l->b = l->w = NULL; return -1; }
    return 0;
}

static int dequant_dense_float(const dnnw_rec *R, int n, const char *name, rd_linear *l)
{
    const dnnw_rec *b = rec_find(R, n, name, "_bias"), *w = rec_find(R, n, name, "_weights_float");
    if (!b || !w || b->size < 4 || b->size % 4 || w->size < 4 || w->size % 4) return -1;
    l->n_out = b->size / 4;
    if ((w->size / 4) % l->n_out) return -1;
    l->n_in = w->size / 4 / l->n_out;
    if (lin_alloc(l)) return -1;
    memcpy(l->b, b->data, sizeof(float) * l->n_out);
    const float *src = (const float *)w->data;                     /* stored as W^T: (n_in, n_out) */
    for (int i = 0; i < l->n_in; i++) for (int o = 0; o < l->n_out; o++) l->w[(size_t)o * l->n_in + i] = src[(size_t)i * l->n_out + o];
    return 0;
}

static int dequant_dense_int8(const dnnw_rec *R, int n, const char *name, rd_linear *l)
{
    const dnnw_rec *b = rec_find(R, n, name, "_bias"), *s = rec_find(R, n, name, "_scale"), *q = rec_find(R, n, name, "_weights_int8");
    if (!b || !s || !q || b->size < 32 || b->size % 32 || s->size != b->size || q->size <= 0) return -1;   /* n_out a multiple of 8 */
    l->n_out = b->size / 4;
    if (q->size % l->n_out) return -1;
    l->n_in = q->size / l->n_out;
    if (l->n_in % 4) return -1;
    if (lin_alloc(l)) return -1;
    memcpy(l->b, b->data, sizeof(float) * l->n_out);
    const float *sc = (const float *)s->data; const signed char *qq = (const signed char *)q->data;
    if ((l->row_scale = malloc(sizeof(float) * (size_t)l->n_out))) for (int o = 0; o < l->n_out; o++) l->row_scale[o] = sc[o] * 127.0f;
    const int blocks_in = l->n_in / 4;
    size_t pos = 0;                                                  /* walk the [n_out/8][n_in/4][8][4] tiles in storage order */
    for (int og = 0; og < l->n_out / 8; og++)
        for (int ib = 0; ib < blocks_in; ib++)
            for (int o8 = 0; o8 < 8; o8++)
                for (int i4 = 0; i4 < 4; i4++, pos++) {
                    const int o = og * 8 + o8;
                    l->w[(size_t)o * l->n_in + ib * 4 + i4] = (float)qq[pos] * (sc[o] * 127.0f);
                }
    return 0;
}

/* n_in comes from the architecture: the blob does not hold it (the reference compiles nb_inputs into linear_init(), wexchange/c_export/common.py:274), and
 * a trailing 4-input block that no output group keeps leaves no trace in the index list (tests/golden/dnnw_export_A.bin has such layers) */
static int dequant_blocksparse_int8(const dnnw_rec *R, int n, const char *name, rd_linear *l, int n_in)
{
    const dnnw_rec *b = rec_find(R, n, name, "_bias"), *s = rec_find(R, n, name, "_scale"), *q = rec_find(R, n, name, "_weights_int8"),
                   *ix = rec_find(R, n, name, "_weights_idx");
    if (!b || !s || !q || !ix || b->size < 32 || b->size % 32 || s->size != b->size || ix->size < 4 || ix->size % 4 || q->size < 0) return -1;
    l->n_out = b->size / 4;
    const int *idx = (const int *)ix->data; const int nidx = ix->size / 4;
    const float *sc = (const float *)s->data; const signed char *qq = (const signed char *)q->data;
    /* first walk: validate every count and column index against n_in, count the 8x4 blocks */
    int p = 0; long nblk = 0;
    if (n_in <= 0 || n_in % 4) return -1;
    for (int g = 0; g < l->n_out / 8; g++) {
        if (p >= nidx) return -1;
        const int cnt = idx[p++];
        if (cnt < 0 || cnt > nidx - p) return -1;
        for (int k = 0; k < cnt; k++, p++) {
            const int j = idx[p];
            if (j < 0 || j % 4 || j + 4 > n_in) return -1;
        }
        nblk += cnt;
    }
    if (nblk * 32 != (long)q->size || p != nidx) return -1;
    l->n_in = n_in;
    if (lin_alloc(l)) return -1;
    memcpy(l->b, b->data, sizeof(float) * l->n_out);
    if ((l->row_scale = malloc(sizeof(float) * (size_t)l->n_out))) for (int o = 0; o < l->n_out; o++) l->row_scale[o] = sc[o] * 127.0f;
    size_t pos = 0;
    p = 0;
    for (int g = 0; g < l->n_out / 8; g++) {
        const int cnt = idx[p++];
        for (int k = 0; k < cnt; k++) {
            const int j = idx[p++];
            for (int o8 = 0; o8 < 8; o8++) for (int i4 = 0; i4 < 4; i4++, pos++) {
                const int o = g * 8 + o8;
                l->w[(size_t)o * l->n_in + j + i4] = (float)qq[pos] * (sc[o] * 127.0f);
            }
        }
    }
    return 0;
}

static void swap_first_two_thirds(float *a, int third)
{   /* exporter wrote gates as z,r,n; torch (and our kernels) use r,z,n */
    for (int i = 0; i < third; i++) { const float t = a[i]; a[i] = a[third + i]; a[third + i] = t; }
}

static int load_gru(const dnnw_rec *R, int n, const char *name, rd_gru *g, int n_in)
{
    char nm[64]; rd_linear in, rec;
    snprintf(nm, sizeof nm, "%s_input", name); if (dequant_blocksparse_int8(R, n, nm, &in, n_in)) return -1;
    snprintf(nm, sizeof nm, "%s_recurrent", name); if (dequant_dense_int8(R, n, nm, &rec)) { free(in.w); free(in.b); free(in.row_scale); return -1; }
    if (rec.n_out != 3 * rec.n_in || in.n_out != rec.n_out) { free(in.w); free(in.b); free(in.row_scale); free(rec.w); free(rec.b); free(rec.row_scale); return -1; }
    g->hid = rec.n_in; g->n_in = in.n_in; g->w_ih = in.w; g->b_ih = in.b; g->w_hh = rec.w; g->b_hh = rec.b; g->s_ih = in.row_scale; g->s_hh = rec.row_scale;
    swap_first_two_thirds(g->w_ih, g->hid * g->n_in); swap_first_two_thirds(g->w_hh, g->hid * g->hid);
    swap_first_two_thirds(g->b_ih, g->hid); swap_first_two_thirds(g->b_hh, g->hid);
    if (g->s_ih) swap_first_two_thirds(g->s_ih, g->hid);
    if (g->s_hh) swap_first_two_thirds(g->s_hh, g->hid);
    return 0;
}

int rd_model_parse(const void *blob, size_t len, rd_model *m)
{
    const unsigned char *p = blob;
    dnnw_rec recs[256]; char names[256][48]; int n = 0;
    size_t off = 0;
    memset(m, 0, sizeof *m);
    while (off + 64 <= len && n < 256) {
        int ver, type, size, block;
        if (memcmp(p + off, "DNNw", 4)) { fprintf(stderr, "rade: bad weight blob (offset %zu)\n", off); return -1; }
        memcpy(&ver, p + off + 4, 4); memcpy(&type, p + off + 8, 4); memcpy(&size, p + off + 12, 4); memcpy(&block, p + off + 16, 4);
        if (ver != 0 || size < 0 || block < size || (size_t)block > len - off - 64) { fprintf(stderr, "rade: corrupt or truncated weight blob (record %d)\n", n); return -1; }
        memset(names[n], 0, 48); memcpy(names[n], p + off + 20, 44);
        recs[n].name = names[n]; recs[n].type = type; recs[n].size = size; recs[n].data = p + off + 64;
        n++; off += 64 + block;
    }
    /* the architecture this engine is built for (radae_base.py:239-251, :377-393): GRU input widths = the running concat */
    static const int enc_in[5] = { 64, 224, 384, 544, 704 }, dec_in[5] = { 96, 224, 352, 480, 608 };
    int err = 0;
    err |= dequant_dense_float(recs, n, "enc_dense1", &m->enc_dense1);
    err |= dequant_dense_float(recs, n, "enc_zdense", &m->enc_zdense);
    err |= dequant_dense_float(recs, n, "dec_dense1", &m->dec_dense1);
    err |= dequant_dense_float(recs, n, "dec_output", &m->dec_output);
    for (int i = 0; i < 5 && !err; i++) {
        char nm[32];
        snprintf(nm, sizeof nm, "enc_gru%d", i + 1); err |= load_gru(recs, n, nm, &m->enc_gru[i], enc_in[i]);
        snprintf(nm, sizeof nm, "dec_gru%d", i + 1); err |= load_gru(recs, n, nm, &m->dec_gru[i], dec_in[i]);
        snprintf(nm, sizeof nm, "enc_conv%d", i + 1); err |= dequant_dense_int8(recs, n, nm, &m->enc_conv[i]);
        snprintf(nm, sizeof nm, "dec_conv%d", i + 1); err |= dequant_dense_int8(recs, n, nm, &m->dec_conv[i]);
        snprintf(nm, sizeof nm, "dec_glu%d", i + 1); err |= dequant_dense_int8(recs, n, nm, &m->dec_glu[i]);
    }
    if (err) { fprintf(stderr, "rade: weight blob is missing layers\n"); rd_model_free(m); return -1; }
    /* sanity: every other layer shape of that architecture */
    for (int i = 0; i < 5; i++)
        if (m->enc_gru[i].n_in != enc_in[i] || m->enc_gru[i].hid != 64 || m->dec_gru[i].n_in != dec_in[i] || m->dec_gru[i].hid != 96 ||
            m->enc_conv[i].n_in != 2 * (enc_in[i] + 64) || m->enc_conv[i].n_out != 96 || m->dec_conv[i].n_in != 2 * (dec_in[i] + 96) || m->dec_conv[i].n_out != 32) {
            fprintf(stderr, "rade: unexpected layer shape in weight blob (layer %d)\n", i + 1); rd_model_free(m); return -1;
        }
    for (int i = 0; i < 5; i++)
        if (m->dec_glu[i].n_in != 96 || m->dec_glu[i].n_out != 96) { fprintf(stderr, "rade: unexpected GLU shape in weight blob (layer %d)\n", i + 1); rd_model_free(m); return -1; }
    /* every dimension the engine's upload (rade_engine.c:upload_lin) hard-codes is checked here: a foreign or crafted blob with smaller
     * layers must not get as far as the packers, which read N x K floats */
    if ((m->enc_dense1.n_in != 84 && m->enc_dense1.n_in != 80) || m->enc_dense1.n_out != 64 || m->enc_zdense.n_in != 864 || m->enc_zdense.n_out != 80 ||
        m->dec_dense1.n_in != 80 || m->dec_dense1.n_out != 96 || m->dec_output.n_in != 736 || m->dec_output.n_out != m->enc_dense1.n_in) {   /* 4 x 21 features (model19, aux symbol) or 4 x 20 (model05, bbfm) */
        fprintf(stderr, "rade: unexpected dense layer shape in weight blob\n"); rd_model_free(m); return -1;
    }
    return 0;
}

static void lin_free(rd_linear *l) { free(l->w); free(l->b); free(l->row_scale); l->w = l->b = l->row_scale = NULL; }
void rd_model_free(rd_model *m)
{
    lin_free(&m->enc_dense1); lin_free(&m->enc_zdense); lin_free(&m->dec_dense1); lin_free(&m->dec_output);
    for (int i = 0; i < 5; i++) {
        free(m->enc_gru[i].w_ih); free(m->enc_gru[i].w_hh); free(m->enc_gru[i].b_ih); free(m->enc_gru[i].b_hh); free(m->enc_gru[i].s_ih); free(m->enc_gru[i].s_hh);
        free(m->dec_gru[i].w_ih); free(m->dec_gru[i].w_hh); free(m->dec_gru[i].b_ih); free(m->dec_gru[i].b_hh); free(m->dec_gru[i].s_ih); free(m->dec_gru[i].s_hh);
        lin_free(&m->enc_conv[i]); lin_free(&m->dec_conv[i]); lin_free(&m->dec_glu[i]);
    }
    memset(m, 0, sizeof *m);
}

/* ---------------------------------------------------------------------------------------------
 * 3. packing for k_gemm: out[((kb*ntt + nt)*64 + lane)*4 + s] = W[nt*32 + (lane&31)][8kb + 4(lane>>5) + s]
 * -------------------------------------------------------------------------------------------*/
long rd_packed_size(int N, int K) { return (long)((K + 7) / 8) * ((N + 31) / 32) * 256; }

long rd_pack_weights(const float *W, int N, int K, float *out)
{
    const int nkb = (K + 7) / 8, ntt = (N + 31) / 32;
    for (int kb = 0; kb < nkb; kb++)
        for (int nt = 0; nt < ntt; nt++)
            for (int lane = 0; lane < 64; lane++)
                for (int s = 0; s < 4; s++) {
                    const int nn = nt * 32 + (lane & 31), k = kb * 8 + 4 * (lane >> 5) + s;
                    out[(((size_t)kb * ntt + nt) * 64 + lane) * 4 + s] = (nn < N && k < K) ? W[(size_t)nn * K + k] : 0.0f;
                }
    return rd_packed_size(N, K);
}

/* float -> IEEE binary16 (round to nearest even, subnormals kept) and back: the two-plane split of the decoder weights */
static unsigned short f32_to_f16(float f)
{
    unsigned int x; memcpy(&x, &f, 4);
    const unsigned int sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);                 /* rounds to >= 65520: infinity */
    if (x < 0x38800000u) {                                                         /* below 2^-14: subnormal half */
        if (x < 0x33000000u) return (unsigned short)sign;                          /* < 2^-25 */
        const int e = (int)(x >> 23);                                              /* biased float exponent, 102..112 */
        unsigned int m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                                                 /* 14..24 */
        const unsigned int halfm = m >> shift, rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        unsigned int r = halfm + ((rem > mid || (rem == mid && (halfm & 1u))) ? 1u : 0u);
        return (unsigned short)(sign | r);
    }
    unsigned int r = ((x - 0x38000000u) >> 13);
    const unsigned int rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (unsigned short)(sign | r);
}
static float f16_to_f32(unsigned short h)
{
    const unsigned int sign = (unsigned int)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    unsigned int x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { float f = (float)m * 5.9604644775390625e-08f; memcpy(&x, &f, 4); x |= sign; }      /* m * 2^-24 */
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

/* Returns the packed size, or -1 when a weight does not fit the planes (|w| 2^10 must stay below binary16's 65504).
 * W[N][K] (K a multiple of 16) -> B operands of v_mfma_f32_32x32x16_f16 in two planes, 2^10 w = hi + lo to 22 bits
 * (the power-of-two scale keeps the low plane out of binary16's subnormal range; the GEMM epilogue undoes it):
 * out[kb][nt][plane][lane][j] = plane(1024 W[32 nt + lane%32][16 kb + 8 (lane/32) + j]) */
long rd_packed16_size(int N, int K) { return (long)(K / 16) * ((N + 31) / 32) * 2 * 64 * 8; }
long rd_pack_weights_f16x2(const float *W, int N, int K, unsigned short *out)
{
    const int nkb = K / 16, ntt = (N + 31) / 32;
    for (int kb = 0; kb < nkb; kb++)
        for (int nt = 0; nt < ntt; nt++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int nn = nt * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
                    const float w = nn < N ? 1024.0f * W[(size_t)nn * K + k] : 0.0f;
                    if (!(fabsf(w) < 65504.0f)) return -1;                         /* the high plane would be inf (or the weight is NaN) */
                    const unsigned short hi = f32_to_f16(w), lo = f32_to_f16(w - f16_to_f32(hi));
                    unsigned short *o = out + ((((size_t)kb * ntt + nt) * 2) * 64 + lane) * 8 + j;
                    o[0] = hi; o[64 * 8] = lo;
                }
    return rd_packed16_size(N, K);
}

/* The receiver's in-kernel decoder (dq_gemm_tile) uses v_mfma_f32_16x16x32_f16 with the weights as the A operand: 16 output
 * columns x 32 k per instruction, the same two planes of 2^10 w.  K must be a multiple of 32; N is padded to 16 with zeros.
 * out[ks][ct][plane][lane][j] = plane(1024 W[16 ct + lane%16][32 ks + 8 (lane/16) + j]).  Returns the size or -1 on overflow. */
long rd_packed16a_size(int N, int K) { return (long)(K / 32) * ((N + 15) / 16) * 2 * 64 * 8; }
long rd_pack_weights_f16x2_a16(const float *W, int N, int K, unsigned short *out)
{
    const int nks = K / 32, nct = (N + 15) / 16;
    for (int ks = 0; ks < nks; ks++)
        for (int ct = 0; ct < nct; ct++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int nn = ct * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
                    const float w = nn < N ? 1024.0f * W[(size_t)nn * K + k] : 0.0f;
                    if (!(fabsf(w) < 65504.0f)) return -1;
                    const unsigned short hi = f32_to_f16(w), lo = f32_to_f16(w - f16_to_f32(hi));
                    unsigned short *o = out + ((((size_t)ks * nct + ct) * 2) * 64 + lane) * 8 + j;
                    o[0] = hi; o[64 * 8] = lo;
                }
    return rd_packed16a_size(N, K);
}

static int q_of(float w, float sc, float *q)
{   /* w = q * sc with integer |q| <= 127, exactly? */
    *q = 0.0f;
    if (w == 0.0f) return 0;
    if (sc == 0.0f) return -1;
    *q = rintf(w / sc);
    return (fabsf(*q) <= 127.0f && *q * sc == w) ? 0 : -1;
}
/* chunk-major binary16 plane of the integers q of an int8 layer (w = q * row_scale[n] exactly) for the single-stream step kernel
 * (rade_core_step.hip): out[(c * Npad + n) * 8 + j] = q(W[n][8 c + j]), K zero-padded to Kpad (a multiple of 8), rows to Npad (a
 * multiple of 64).  Returns 0, or -1 when the layer is not int8-exact. */
int rd_chunkmajor_q16(const float *W, const float *row_scale, int N, int K, int Kpad, int Npad, unsigned short *out)
{
    if (!row_scale) return -1;
    for (int c = 0; c < Kpad / 8; c++)
        for (int n = 0; n < Npad; n++)
            for (int j = 0; j < 8; j++) {
                const int k = 8 * c + j;
                float q = 0.0f;
                if (n < N && k < K && q_of(W[(size_t)n * K + k], row_scale[n], &q)) return -1;
                out[((size_t)c * Npad + n) * 8 + j] = f32_to_f16(q);
            }
    return 0;
}
void rd_chunkmajor_f32(const float *W, int N, int K, int Kpad, int Npad, float *out)
{
    for (int c = 0; c < Kpad / 8; c++)
        for (int n = 0; n < Npad; n++)
            for (int j = 0; j < 8; j++) { const int k = 8 * c + j; out[((size_t)c * Npad + n) * 8 + j] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.0f; }
}

long rd_pack_weights_q16(const float *W, const float *row_scale, int N, int K, unsigned short *out, float *scale_out)
{
    const int nkb = K / 16, ntt = (N + 31) / 32;
    if (!row_scale) return -1;
    for (int n = 0; n < ntt * 32; n++) scale_out[n] = n < N ? row_scale[n] : 0.0f;
    for (int kb = 0; kb < nkb; kb++)
        for (int nt = 0; nt < ntt; nt++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int nn = nt * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
                    float q = 0.0f;
                    if (nn < N && q_of(W[(size_t)nn * K + k], row_scale[nn], &q)) return -1;
                    out[(((size_t)kb * ntt + nt) * 64 + lane) * 8 + j] = f32_to_f16(q);
                }
    return (long)nkb * ntt * 64 * 8;
}
long rd_pack_weights_q16_a16(const float *W, const float *row_scale, int N, int K, unsigned short *out, float *scale_out)
{
    const int nks = K / 32, nct = (N + 15) / 16;
    if (!row_scale) return -1;
    for (int n = 0; n < nct * 16; n++) scale_out[n] = n < N ? row_scale[n] : 0.0f;
    for (int ks = 0; ks < nks; ks++)
        for (int ct = 0; ct < nct; ct++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int nn = ct * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
                    float q = 0.0f;
                    if (nn < N) {
                        const float w = W[(size_t)nn * K + k];
                        if (w != 0.0f) {
                            if (row_scale[nn] == 0.0f) return -1;
                            q = rintf(w / row_scale[nn]);
                            if (!(fabsf(q) <= 127.0f) || q * row_scale[nn] != w) return -1;      /* not an int8-exact layer after all */
                        }
                    }
                    out[(((size_t)ks * nct + ct) * 64 + lane) * 8 + j] = f32_to_f16(q);
                }
    return (long)nks * nct * 64 * 8;
}

/* check_pilots rows on the f16 matrix cores: the realified pilot table Pm[n = 2f + c', k = 2m + c] = {pr, pi; pi, -pr}[c'][c]
 * of p_w[m][f] (dsp.py:207-208 as a real GEMM), split in two binary16 planes and laid out as the A operands of
 * v_mfma_f32_16x16x32_f16: out[nt][s][plane][lane][j] = plane(2^12 Pm[16 nt + lane%16][32 s + 8 (lane/16) + j]) */
void rd_corr16_table_fill(const rd_tables *T, unsigned short *out /* [5][10][2][64][8] */)
{
    for (int nt = 0; nt < 5; nt++)
        for (int s = 0; s < 10; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int n = 16 * nt + (lane & 15), f = n >> 1, cp = n & 1;
                    const int k = 32 * s + 8 * (lane >> 4) + j, m = k >> 1, c = k & 1;
                    const float pr = T->p_w[m][f][0], pi = T->p_w[m][f][1];
                    const float v = 4096.0f * (cp == 0 ? (c == 0 ? pr : pi) : (c == 0 ? pi : -pr));     /* 2^12: low plane stays normal */
                    const unsigned short hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                    unsigned short *o = out + ((((size_t)nt * 10 + s) * 2) * 64 + lane) * 8 + j;
                    o[0] = hi; o[64 * 8] = lo;
                }
}

/* ---- the pilot correlator in two stages (round 5) -------------------------------------------------------------------------------------------
 * p_w[m][f] = p[m] e^{j w_f m} for the 40 coarse frequencies |f| <= 50 Hz spans a window of 160 samples: the phase e^{j w_f (m - 79.5)} moves by at most
 * +-pi across it, i.e. as a function of x = (m - 79.5) / 80 it is band-limited to a time-bandwidth product of 2 and is reproduced to 1.7e-10 by its
 * projection on the first RD_NQ = 16 polynomials q_r orthonormal on the 160 grid points (14: 1.8e-8, 12: 1.5e-6):
 *     p_w[m][f] = sum_r alpha[r][f] s_r[m],   s_r[m] = p[m] q_r(x_m),   alpha[r][f] = e^{j w_f 79.5} sum_m q_r(x_m) e^{j w_f (m - 79.5)}
 * so Dt[t][f] = sum_m conj(rx[t + m]) p_w[m][f] = sum_r alpha[r][f] Mom_r[t] with the 16 "moments" Mom_r[t] = sum_m conj(rx[t + m]) s_r[m]: the big
 * product (K = 320 real) has 32 rows (two 16-row tiles) instead of 80 (five), and a second product with K = 32 expands the moments to the 40 frequencies.
 * On v_mfma_f32_16x16x32_f16: 2 x 10 x 3 + 5 x 5 = 85 matrix instructions per tile of 16 timings instead of 5 x 10 x 3 = 150.
 * rd_corrq16_table_fill: stage 1, the realified s_r (rows n' = 2 r + c', k = 2 m + c as in rd_corr16_table_fill), 2^12-scaled, two binary16 planes,
 *                        A-operand order: out[tile][s][plane][lane][j]
 * rd_corra16_table_fill: stage 2, A2[n = 2 f + c''][n'] = {ar, -ai; ai, ar} of alpha[r][f], 2^10-scaled, two planes.  Its K axis is ordered the way
 *                        stage 1's accumulators lie in a lane (C layout of two 16-row tiles: lane group g holds rows 4 g .. 4 g + 3 of either tile), so that
 *                        the moments go from accumulator registers to B-operand registers without leaving the lane:
 *                        k-slot (g, j): n' = 4 g + j (j < 4), 16 + 4 g + (j - 4) (j >= 4).   out[tile][plane][lane][j] */
static void corr_basis(const rd_tables *T, double q[RD_NQ][RD_M], double alr[RD_NQ][RD_NFC], double ali[RD_NQ][RD_NFC])
{
    /* orthonormal polynomials on x_m = (m - 79.5) / 80: Legendre recurrence, then two passes of modified Gram-Schmidt on the grid */
    double x[RD_M];
    for (int m = 0; m < RD_M; m++) x[m] = ((double)m - 79.5) / 80.0;
    for (int m = 0; m < RD_M; m++) { q[0][m] = 1.0; q[1][m] = x[m]; }
    for (int r = 1; r + 1 < RD_NQ; r++)
        for (int m = 0; m < RD_M; m++) q[r + 1][m] = ((2.0 * r + 1.0) * x[m] * q[r][m] - (double)r * q[r - 1][m]) / (r + 1.0);
    for (int r = 0; r < RD_NQ; r++) {
        for (int pass = 0; pass < 2; pass++)
            for (int u = 0; u < r; u++) {
                double d = 0.0;
                for (int m = 0; m < RD_M; m++) d += q[u][m] * q[r][m];
                for (int m = 0; m < RD_M; m++) q[r][m] -= d * q[u][m];
            }
        double n2 = 0.0;
        for (int m = 0; m < RD_M; m++) n2 += q[r][m] * q[r][m];
        const double inv = 1.0 / sqrt(n2);
        for (int m = 0; m < RD_M; m++) q[r][m] *= inv;
    }
    for (int f = 0; f < RD_NFC; f++) {
        const double wf = 2.0 * PI_D * T->fcoarse[f] / 8000.0, cr = cos(wf * 79.5), ci = sin(wf * 79.5);
        for (int r = 0; r < RD_NQ; r++) {
            double ar = 0.0, ai = 0.0;
            for (int m = 0; m < RD_M; m++) { ar += q[r][m] * cos(wf * ((double)m - 79.5)); ai += q[r][m] * sin(wf * ((double)m - 79.5)); }
            alr[r][f] = ar * cr - ai * ci; ali[r][f] = ar * ci + ai * cr;
        }
    }
}
void rd_corrq16_table_fill(const rd_tables *T, unsigned short *out /* [2][10][2][64][8] */)
{
    static double q[RD_NQ][RD_M], alr[RD_NQ][RD_NFC], ali[RD_NQ][RD_NFC];
    corr_basis(T, q, alr, ali);
    for (int nt = 0; nt < 2; nt++)
        for (int s = 0; s < 10; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int n = 16 * nt + (lane & 15), r = n >> 1, cp = n & 1;
                    const int k = 32 * s + 8 * (lane >> 4) + j, m = k >> 1, c = k & 1;
                    const double sr = (double)T->p[m][0] * q[r][m], si = (double)T->p[m][1] * q[r][m];
                    const float v = (float)(4096.0 * (cp == 0 ? (c == 0 ? sr : si) : (c == 0 ? si : -sr)));
                    const unsigned short hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                    unsigned short *o = out + ((((size_t)nt * 10 + s) * 2) * 64 + lane) * 8 + j;
                    o[0] = hi; o[64 * 8] = lo;
                }
}
void rd_corra16_table_fill(const rd_tables *T, unsigned short *out /* [5][2][64][8] */)
{
    static double q[RD_NQ][RD_M], alr[RD_NQ][RD_NFC], ali[RD_NQ][RD_NFC];
    corr_basis(T, q, alr, ali);
    for (int nt = 0; nt < 5; nt++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++) {
                const int n = 16 * nt + (lane & 15), f = n >> 1, cpp = n & 1;
                const int g = lane >> 4, np_ = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4), r = np_ >> 1, cp = np_ & 1;
                const double ar = alr[r][f], ai = ali[r][f];
                const float v = (float)(1024.0 * (cpp == 0 ? (cp == 0 ? ar : -ai) : (cp == 0 ? ai : ar)));
                const unsigned short hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                unsigned short *o = out + (((size_t)nt * 2) * 64 + lane) * 8 + j;
                o[0] = hi; o[64 * 8] = lo;
            }
}
/* test aid (tests/test_host_cpu.py): the two tables multiplied back together on the host, out[m][f] (re, im) ~ p_w[m][f]: max |difference| is returned */
double rd_corr_tables_check(const rd_tables *T)
{
    static double q[RD_NQ][RD_M], alr[RD_NQ][RD_NFC], ali[RD_NQ][RD_NFC];
    corr_basis(T, q, alr, ali);
    double worst = 0.0;
    for (int m = 0; m < RD_M; m++)
        for (int f = 0; f < RD_NFC; f++) {
            double re = 0.0, im = 0.0;
            for (int r = 0; r < RD_NQ; r++) {
                const double sr = (double)T->p[m][0] * q[r][m], si = (double)T->p[m][1] * q[r][m];
                re += alr[r][f] * sr - ali[r][f] * si; im += alr[r][f] * si + ali[r][f] * sr;
            }
            const double d = hypot(re - (double)T->p_w[m][f][0], im - (double)T->p_w[m][f][1]);
            if (d > worst) worst = d;
        }
    return worst;
}

/* The demodulator DFT of k_rx_sync2 on the f16 matrix cores (receiver_one, dsp.py:487-526: sym[s][c] = sum_n x_s[n] Wfwd[n][c]).  With the row
 * R[c][2n + comp] = (wr, -wi)[comp] the real part is R . (xr, xi) and the imaginary part R . (xi, -xr): ONE 32-row operand (30 carriers) serves both, the two
 * variants of the window are two groups of B columns.  Two binary16 planes in the A-operand order of v_mfma_f32_16x16x32_f16 like the pilot table above:
 * out[tile][s][plane][lane][j] = plane(2^12 R[16 tile + lane%16][32 s + 8 (lane/16) + j]); rows 30, 31 are zero. */
void rd_wfwd16_table_fill(const rd_tables *T, unsigned short *out /* [2][10][2][64][8] */)
{
    for (int tile = 0; tile < 2; tile++)
        for (int s = 0; s < 10; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int c = 16 * tile + (lane & 15);
                    const int k = 32 * s + 8 * (lane >> 4) + j, n = k >> 1, comp = k & 1;
                    float v = 0.0f;
                    if (c < RD_NC) v = 4096.0f * (comp == 0 ? T->Wfwd[n][c][0] : -T->Wfwd[n][c][1]);
                    const unsigned short hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                    unsigned short *o = out + ((((size_t)tile * 10 + s) * 2) * 64 + lane) * 8 + j;
                    o[0] = hi; o[64 * 8] = lo;
                }
}

/* complex_bpf's 101 real taps (dsp.py:46-49) as the A operands of the matrix-core FIR (rade_rx.hip: bpf_fir_tile): the Toeplitz rows
 * T[r][m] = 2^10 h[m - r] (0 outside 0 <= m - r <= 100), r = 0..15 outputs of a group, m = 0..127 window positions, in two binary16 planes
 * and the A-operand order of v_mfma_f32_16x16x32_f16: out[ks][plane][lane][j] = plane(T[lane % 16][32 ks + 8 (lane / 16) + j]) */
void rd_bpf16_table_fill(const rd_tables *T, unsigned short *out /* [4][2][64][8] */)
{
    for (int ks = 0; ks < 4; ks++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++) {
                const int r = lane & 15, m = 32 * ks + 8 * (lane >> 4) + j, t = m - r;
                const float v = (t >= 0 && t < RD_NTAP) ? 1024.0f * T->bpf_h[t] : 0.0f;
                const unsigned short hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                unsigned short *o = out + (((size_t)ks * 2) * 64 + lane) * 8 + j;
                o[0] = hi; o[64 * 8] = lo;
            }
}
