/* rade_host.h -- host-side model preparation (see rade_host.c) */
#ifndef RADE_HOST_H
#define RADE_HOST_H

#include <stddef.h>

#include "rade_dev.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int n_in, n_out; float *w, *b; float *row_scale; } rd_linear;   /* w[out][in] row-major; row_scale[out] != NULL: the layer is int8 in the blob, w[o][i] = q * row_scale[o] with integer |q| <= 127 */
typedef struct { int n_in, hid; float *w_ih, *w_hh, *b_ih, *b_hh; float *s_ih, *s_hh; } rd_gru; /* torch gate order r,z,n; s_*: row scales of the int8 weights */

typedef struct {
    rd_linear enc_dense1, enc_zdense, dec_dense1, dec_output;
    rd_gru enc_gru[5], dec_gru[5];
    rd_linear enc_conv[5], dec_conv[5];   /* w[out][2*in]: columns [0,in) = older tap, [in,2in) = current frame */
    rd_linear dec_glu[5];
} rd_model;

/* weight blob search order of rade_open(): hint, $RADE_MODEL_FILE, <library dir>/../weights/model19_check3.bin, cwd candidates (rade_api.c) */
const char *rd_find_default_model(const char *hint, char *buf, size_t n);

void rd_tables_fill(rd_tables *T);
int rd_model_parse(const void *blob, size_t len, rd_model *m);
void rd_model_free(rd_model *m);

long rd_packed16_size(int N, int K);
void rd_corr16_table_fill(const rd_tables *T, unsigned short *out);
void rd_corrq16_table_fill(const rd_tables *T, unsigned short *out);     /* [2][10][2][64][8]: stage 1 of the two-stage pilot correlator (rade_host.c) */
void rd_corra16_table_fill(const rd_tables *T, unsigned short *out);     /* [5][2][64][8]: stage 2 */
double rd_corr_tables_check(const rd_tables *T);
void rd_wfwd16_table_fill(const rd_tables *T, unsigned short *out);
void rd_bpf16_table_fill(const rd_tables *T, unsigned short *out);   /* [4][2][64][8] */
long rd_pack_weights_f16x2(const float *W, int N, int K, unsigned short *out);
/* int8-exact layers for k_gemm16: ONE plane of integers in the B-operand order of v_mfma_f32_32x32x16_f16, out[K/16][ceil(N/32)][64][8];
 * scale_out[32 ceil(N/32)].  Returns the size in halfs or -1 when W is not q * row_scale with integer |q| <= 127. */
long rd_pack_weights_q16(const float *W, const float *row_scale, int N, int K, unsigned short *out, float *scale_out);
int rd_chunkmajor_q16(const float *W, const float *row_scale, int N, int K, int Kpad, int Npad, unsigned short *out);
void rd_chunkmajor_f32(const float *W, int N, int K, int Kpad, int Npad, float *out);
long rd_packed16a_size(int N, int K);
/* int8-exact layers: ONE binary16 plane holding the integers q (exact), out[K/32][ceil(N/16)][64][8]; scale_out[16 ceil(N/16)] = row scale (0 past N).
 * Returns the plane's size in halfs, or -1 when W is not q * row_scale with integer |q| <= 127. */
long rd_pack_weights_q16_a16(const float *W, const float *row_scale, int N, int K, unsigned short *out, float *scale_out);
long rd_pack_weights_f16x2_a16(const float *W, int N, int K, unsigned short *out);
#ifdef __cplusplus
}
#endif
/* rade_core.c, for rade_api.c: rade_tx() as one launch per modem frame (NULL from open: fall back to the batched engine) */
void *rd_core_tx_open(const void *blob, int len, const rd_tables *d_tab);
void rd_core_tx_close(void *t);
void rd_core_tx_reset(void *t);
int rd_core_tx_frame(void *t, const float *features_in, float *tx_out);
/* rade_engine.c: the engine's constant tables on its device */
struct rade_batch;
const rd_tables *rd_batch_tables(const struct rade_batch *h);
int rd_batch_has_tx_bpf(const struct rade_batch *h);

#endif
