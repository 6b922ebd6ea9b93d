/*
 * rade_core.c -- include/rade_core.h on top of the batched HIP engine: one engine with a single stream per encoder /
 * decoder state, one 40 ms step per call (the call granularity of /root/reference/src/rade_enc.c:55-114 and
 * rade_dec.c:50-102), host buffers in and out.  The step itself is rade_batch_encode / rade_batch_decode with B = 1.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rade_core.h"
#include "rade_batch.h"
#include "rade_host.h"

/* hidden record BEHIND the NULL terminator of every list rade_parse_weights() returns: where the blob lives.  *list is the
 * malloc'ed pointer itself, so the reference harnesses' free(list) (test_rade_enc.c:115, test_rade_dec.c:115) is valid. */
#define RD_LIST_MAGIC "__rade_blob__"
/* the reference links compiled-in weights under these names; here they stand for "the default blob" */
const WeightArray radeenc_arrays[1] = { { NULL, -1, 0, NULL } };
const WeightArray radedec_arrays[1] = { { NULL, -1, 0, NULL } };

int rade_parse_weights(WeightArray **list, const void *data, int len)
{
    if (!list || !data || len < 64) return -1;
    const unsigned char *p = data;
    int n = 0;
    for (size_t off = 0; off + 64 <= (size_t)len; n++) {       /* first pass: count and validate the record headers */
        int size, block;
        if (memcmp(p + off, "DNNw", 4)) return -1;
        memcpy(&size, p + off + 12, 4); memcpy(&block, p + off + 16, 4);
        if (size < 0 || block < size || (size_t)block > (size_t)len - off - 64 || p[off + 63] != 0) return -1;
        off += 64 + (size_t)block;
    }
    WeightArray *l = calloc((size_t)n + 2, sizeof *l);
    if (!l) return -1;
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        int type, size, block;
        memcpy(&type, p + off + 8, 4); memcpy(&size, p + off + 12, 4); memcpy(&block, p + off + 16, 4);
        l[i].name = (const char *)(p + off + 20); l[i].type = type; l[i].size = size; l[i].data = p + off + 64;
        off += 64 + (size_t)block;
    }
    /* l[n] is the NULL terminator (calloc); the blob record sits behind it, out of reach of a walk over the list */
    l[n + 1].name = RD_LIST_MAGIC; l[n + 1].size = len; l[n + 1].data = data;
    *list = l;
    return n;
}

static int find_array(const WeightArray *a, const char *name)
{
    for (int i = 0; a[i].name; i++) if (!strcmp(a[i].name, name)) return a[i].size;
    return -1;
}

/* resolves a list to (blob, len); the default lists map the default blob file into memory once */
static int list_blob(const WeightArray *arrays, const void **blob, int *len)
{
    static void *def_blob; static int def_len;
    if (arrays && arrays != radeenc_arrays && arrays != radedec_arrays) {
        const WeightArray *hdr = arrays;
        while (hdr->name) hdr++;                               /* the terminator; the blob record follows it */
        hdr++;
        if (!hdr->name || strcmp(hdr->name, RD_LIST_MAGIC)) { fprintf(stderr, "rade_core: weight list was not made by rade_parse_weights()\n"); return -1; }
        *blob = hdr->data; *len = hdr->size;
        return 0;
    }
    if (!def_blob) {
        char buf[4096];
        const char *path = rd_find_default_model(NULL, buf, sizeof buf);
        FILE *f = path ? fopen(path, "rb") : NULL;
        if (!f) { fprintf(stderr, "rade_core: no default weight blob ($RADE_MODEL_FILE, weights/model19_check3.bin)\n"); return -1; }
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        void *b = n > 0 ? malloc((size_t)n) : NULL;
        if (!b || fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return -1; }
        fclose(f);
        def_blob = b; def_len = (int)n;
    }
    *blob = def_blob; *len = def_len;
    return 0;
}

static int init_model(const WeightArray *arrays, int dim, const void **blob, int *len)
{
    if (list_blob(arrays, blob, len)) return 1;
    WeightArray *l = NULL;
    if (rade_parse_weights(&l, *blob, *len) < 0) return 1;
    /* the dimension the blob was exported with: enc_dense1 is (input_dim x 64) floats, dec_output has output_dim biases */
    const int w = find_array(l, "enc_dense1_weights_float"), b = find_array(l, "dec_output_bias");
    free(l);
    if (w != dim * 64 * 4 || b != dim * 4) return 1;
    return 0;
}

int init_radeenc(RADEEnc *model, const WeightArray *arrays, int input_dim)
{
    if (!model || (input_dim != 80 && input_dim != 84)) return 1;
    memset(model, 0, sizeof *model);
    if (init_model(arrays, input_dim, &model->blob, &model->blob_len)) return 1;
    model->input_dim = input_dim; model->nb_z = RADE_LATENT_DIM;
    return 0;
}

int init_radedec(RADEDec *model, const WeightArray *arrays, int output_dim)
{
    if (!model || (output_dim != 80 && output_dim != 84)) return 1;
    memset(model, 0, sizeof *model);
    if (init_model(arrays, output_dim, &model->blob, &model->blob_len)) return 1;
    model->output_dim = output_dim; model->nb_z = RADE_LATENT_DIM;
    return 0;
}

/* device side of one state: a single-stream engine and its staging buffers */
/* One step is ~20 short dependent launches for a single stream (launch-bound), so after the first call the sequence
 * [input H2D, encoder / decoder kernels, output D2H] is captured once and replayed as a hipGraph, as rade_tx() does. */
typedef struct { rade_batch *eng; float *d_in, *d_out; int dim; hipStream_t gs; hipGraphExec_t graph; int calls, graph_off; float *h_in, *h_out; } core_dev;

static core_dev *dev_open(const void *blob, int len, int dim)
{
    core_dev *d = calloc(1, sizeof *d);
    if (!d) return NULL;
    rade_batch_config cfg = { 1, 1, 0, 0, 0, 0.0f };
    const char *dv = getenv("RADE_DEVICE");
    if (dv) cfg.device = atoi(dv);
    d->eng = rade_batch_open_mem(blob, (size_t)len, &cfg);
    d->dim = dim;
    if (!d->eng || hipMalloc((void **)&d->d_in, sizeof(float) * 96) != hipSuccess || hipMalloc((void **)&d->d_out, sizeof(float) * 96) != hipSuccess) {
        if (d->eng) rade_batch_close(d->eng);
        if (d->d_in) hipFree(d->d_in);
        free(d);
        return NULL;
    }
    if (getenv("RADE_NO_GRAPH") || hipStreamCreate(&d->gs) != hipSuccess || hipHostMalloc((void **)&d->h_in, sizeof(float) * 96, 0) != hipSuccess ||
        hipHostMalloc((void **)&d->h_out, sizeof(float) * 96, 0) != hipSuccess) { d->graph_off = 1; (void)hipGetLastError(); }
    return d;
}
static void dev_close(core_dev *d)
{
    if (!d) return;
    if (d->graph) hipGraphExecDestroy(d->graph);
    if (d->h_in) hipHostFree(d->h_in);
    if (d->h_out) hipHostFree(d->h_out);
    if (d->gs) hipStreamDestroy(d->gs);
    rade_batch_close(d->eng); hipFree(d->d_in); hipFree(d->d_out); free(d);
}

void rade_init_encoder(RADEEncState *s) { memset(s, 0, sizeof *s); }     /* rade_enc.c:39-43: the caller's memory may be uninitialised */
void rade_init_decoder(RADEDecState *s) { memset(s, 0, sizeof *s); }
void rade_free_encoder(RADEEncState *s) { if (s && s->initialized) dev_close(s->dev); if (s) memset(s, 0, sizeof *s); }
void rade_free_decoder(RADEDecState *s) { if (s && s->initialized) dev_close(s->dev); if (s) memset(s, 0, sizeof *s); }

/* one step through the engine: in[n_in] (host) -> out[n_out] (host); enc selects rade_batch_encode / rade_batch_decode */
static int core_step(core_dev *d, int enc, const float *in, int n_in, float *out, int n_out)
{
    int ok = 0;
    if (d->calls > 0 && !d->graph_off) {
        if (!d->graph) {                                       /* second call: record the sequence (nothing runs during capture) */
            hipGraph_t g = NULL;
            int c = hipStreamBeginCapture(d->gs, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (c) {
                c = hipMemcpyAsync(d->d_in, d->h_in, sizeof(float) * n_in, hipMemcpyHostToDevice, d->gs) == hipSuccess &&
                    (enc ? rade_batch_encode(d->eng, d->d_in, 1, d->d_out, d->gs) : rade_batch_decode(d->eng, d->d_in, 1, d->d_out, 0, d->gs)) == 1 &&
                    hipMemcpyAsync(d->h_out, d->d_out, sizeof(float) * n_out, hipMemcpyDeviceToHost, d->gs) == hipSuccess;
                if (hipStreamEndCapture(d->gs, &g) != hipSuccess) c = 0;
            }
            if (c && hipGraphInstantiate(&d->graph, g, NULL, NULL, 0) != hipSuccess) { c = 0; d->graph = NULL; }
            if (g) hipGraphDestroy(g);
            if (!c) { d->graph_off = 1; (void)hipGetLastError(); }
        }
        if (d->graph) {
            memcpy(d->h_in, in, sizeof(float) * n_in);
            if (hipGraphLaunch(d->graph, d->gs) == hipSuccess && hipStreamSynchronize(d->gs) == hipSuccess) { memcpy(out, d->h_out, sizeof(float) * n_out); ok = 1; }
        }
    }
    if (!ok) ok = hipMemcpy(d->d_in, in, sizeof(float) * n_in, hipMemcpyHostToDevice) == hipSuccess &&
                  (enc ? rade_batch_encode(d->eng, d->d_in, 1, d->d_out, NULL) : rade_batch_decode(d->eng, d->d_in, 1, d->d_out, 0, NULL)) == 1 &&
                  hipMemcpy(out, d->d_out, sizeof(float) * n_out, hipMemcpyDeviceToHost) == hipSuccess;
    d->calls++;
    return ok ? 0 : -1;
}

static void die(const char *what) { fprintf(stderr, "%s: device error (this library has no CPU fallback)\n", what); exit(1); }

void rade_core_encoder(RADEEncState *s, const RADEEnc *model, float *z, const float *features, int arch, int bottleneck)
{
    (void)arch;
    if (!s->initialized) {                        /* first step after rade_init_encoder(): zero state = a fresh engine */
        s->dev = dev_open(model->blob, model->blob_len, model->input_dim);
        if (!s->dev) die("rade_core_encoder");
        s->initialized = 1;
    }
    core_dev *d = s->dev;
    if (core_step(d, 1, features, d->dim, z, RADE_LATENT_DIM)) die("rade_core_encoder");
    if (bottleneck == 1) for (int i = 0; i < RADE_LATENT_DIM; i++) z[i] = tanhf(z[i]);      /* rade_enc.c:113 / radae_base.py:281-284 */
}

void rade_core_decoder(RADEDecState *s, const RADEDec *model, float *features, const float *z_hat, int arch)
{
    (void)arch;
    if (!s->initialized) {
        s->dev = dev_open(model->blob, model->blob_len, model->output_dim);
        if (!s->dev) die("rade_core_decoder");
        s->initialized = 1;
    }
    core_dev *d = s->dev;
    if (core_step(d, 0, z_hat, RADE_LATENT_DIM, features, d->dim)) die("rade_core_decoder");
}
