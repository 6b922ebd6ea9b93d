/*
 * rade_api.c -- the single-stream RADE C ABI (include/rade_api.h) on top of the batched HIP engine.
 *
 * One `struct rade` = one engine with B = 1 plus small device staging buffers.  Same call sequence,
 * element counts and return values as the reference implementation (/root/reference/src/rade_api.c),
 * but with no embedded interpreter: rade_tx()/rade_rx() launch HIP kernels.  A mutex per handle
 * replaces the reference's GIL serialisation (rade_api.c:409,470).
 */
#define __HIP_PLATFORM_AMD__ 1
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>

#include <assert.h>
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "rade_api.h"
#include "rade_batch.h"
#include "rade_dev.h"
#include "rade_host.h"

#define RADE_API_VERSION 1      /* rade_api.c:37 */

struct rade {
    rade_batch *eng;
    int flags, nin, sync, snr, device;
    float *d_feat_in, *d_feat_out, *d_eoo; void *d_iq, *d_rx;
    pthread_mutex_t lock;
    /* rade_tx is ~20 short launches for one stream (launch-bound): after the first call the sequence
     * [features H2D, encoder + modulator kernels, samples D2H] is captured once and replayed as a hipGraph */
    hipStream_t gs; hipGraphExec_t tx_graph; int tx_calls, tx_graph_off;
    float *h_feat; RADE_COMP *h_iq;          /* pinned staging buffers the graph copies from / to */
    void *txc;                               /* rade_tx as ONE launch per frame (rade_core.c: rd_core_tx_*); NULL: the batched engine's launch sequence above */
};

void rade_initialize(void) { /* reference: Py_InitializeEx (rade_api.c:329-332); HIP initialises lazily */ }
void rade_finalize(void) { }

static int is_blob(const char *path)
{
    char magic[4]; FILE *f = path ? fopen(path, "rb") : NULL;
    if (!f) return 0;
    int ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "DNNw", 4);
    fclose(f);
    return ok;
}

const char *rd_find_default_model(const char *hint, char *buf, size_t n)
{
    if (is_blob(hint)) return hint;
    const char *env = getenv("RADE_MODEL_FILE");
    if (is_blob(env)) return env;
    Dl_info info;
    if (dladdr((void *)rade_version, &info) && info.dli_fname) {     /* <repo>/radae_amd/libradehip.so -> <repo>/weights */
        snprintf(buf, n, "%s", info.dli_fname);
        char *slash = strrchr(buf, '/');
        if (slash) { snprintf(slash, n - (slash - buf), "/../weights/model19_check3.bin"); if (is_blob(buf)) return buf; }
    }
    static const char *cands[] = { "weights/model19_check3.bin", "model19_check3.bin", "../weights/model19_check3.bin" };
    for (size_t i = 0; i < sizeof cands / sizeof cands[0]; i++) if (is_blob(cands[i])) return cands[i];
    return NULL;
}

struct rade *rade_open(char model_file[], int flags)
{
    char buf[4096];
    const char *path = rd_find_default_model(model_file, buf, sizeof buf);
    if (!path) { fprintf(stderr, "rade_open: no DNNw weight blob found (tried \"%s\", $RADE_MODEL_FILE, weights/model19_check3.bin)\n", model_file ? model_file : ""); return NULL; }
    struct rade *r = calloc(1, sizeof *r);
    rade_batch_config cfg = { 1, 1, 0, flags, 0, 0.0f };
    const char *dev = getenv("RADE_DEVICE");
    if (dev) cfg.device = atoi(dev);
    r->eng = rade_batch_open(path, &cfg);
    if (!r->eng) { free(r); return NULL; }
    r->flags = flags; r->nin = RD_NMF; r->device = cfg.device;
    if (!rd_batch_has_tx_bpf(r->eng)) {      /* (the Tx band-pass option filters between modulator and output: that path stays on the engine) */
        FILE *f = fopen(path, "rb");
        if (f) {
            fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
            void *blob = n > 0 ? malloc((size_t)n) : NULL;
            if (blob && fread(blob, 1, (size_t)n, f) == (size_t)n) r->txc = rd_core_tx_open(blob, (int)n, rd_batch_tables(r->eng));
            free(blob); fclose(f);
        }
    }
    if (hipMalloc((void **)&r->d_feat_in, sizeof(float) * RD_FEAT_MF) || hipMalloc((void **)&r->d_feat_out, sizeof(float) * RD_FEAT_MF) ||
        hipMalloc((void **)&r->d_eoo, sizeof(float) * RD_NEOOBITS) || hipMalloc(&r->d_iq, sizeof(RADE_COMP) * RD_NEOO) ||
        hipMalloc(&r->d_rx, sizeof(RADE_COMP) * RD_NINMAX)) { rade_close(r); return NULL; }
    pthread_mutex_init(&r->lock, NULL);
    /* optional fast path of rade_tx (hipGraph replay): any failure here or later simply leaves the plain path in use */
    if (getenv("RADE_NO_GRAPH") || hipStreamCreate(&r->gs) != hipSuccess || hipHostMalloc((void **)&r->h_feat, sizeof(float) * RD_FEAT_MF, 0) != hipSuccess ||
        hipHostMalloc((void **)&r->h_iq, sizeof(RADE_COMP) * RD_NMF, 0) != hipSuccess) { r->tx_graph_off = 1; (void)hipGetLastError(); }
    if (!(flags & RADE_VERBOSE_0)) fprintf(stderr, "rade_open: model %s, HIP back end\n", path);
    return r;
}

void rade_close(struct rade *r)
{
    if (!r) return;
    if (r->txc) rd_core_tx_close(r->txc);
    if (r->tx_graph) hipGraphExecDestroy(r->tx_graph);
    if (r->h_feat) hipHostFree(r->h_feat);
    if (r->h_iq) hipHostFree(r->h_iq);
    if (r->gs) hipStreamDestroy(r->gs);
    if (r->eng) rade_batch_close(r->eng);
    void *p[] = { r->d_feat_in, r->d_feat_out, r->d_eoo, r->d_iq, r->d_rx };
    for (int i = 0; i < 5; i++) if (p[i]) hipFree(p[i]);
    free(r);
}

int rade_version(void) { return RADE_API_VERSION; }
int rade_n_tx_out(struct rade *r) { assert(r != NULL); return RD_NMF; }
int rade_n_tx_eoo_out(struct rade *r) { assert(r != NULL); return RD_NEOO; }
int rade_nin_max(struct rade *r) { assert(r != NULL); return RD_NINMAX; }
int rade_nin(struct rade *r) { assert(r != NULL); return r->nin; }
int rade_n_features_in_out(struct rade *r) { assert(r != NULL); return RD_FEAT_MF; }
int rade_n_eoo_bits(struct rade *r) { assert(r != NULL); return RD_NEOOBITS; }

void rade_tx_set_eoo_bits(struct rade *r, float eoo_bits[])
{
    assert(r != NULL); assert(eoo_bits != NULL);
    pthread_mutex_lock(&r->lock);
    (void)hipSetDevice(r->device);          /* any thread may call; the staging copies below must target the engine's device */
    rade_batch_tx_set_eoo_bits(r->eng, eoo_bits);
    pthread_mutex_unlock(&r->lock);
}

int rade_tx(struct rade *r, RADE_COMP tx_out[], float features_in[])
{
    assert(r != NULL); assert(features_in != NULL); assert(tx_out != NULL);
    int ret = 0;
    pthread_mutex_lock(&r->lock);
    (void)hipSetDevice(r->device);          /* any thread may call; the staging copies below must target the engine's device */
    if (r->txc) { if (rd_core_tx_frame(r->txc, features_in, (float *)tx_out) == 0) ret = RD_NMF; }
    else if (r->tx_calls > 0 && !r->tx_graph_off && r->gs && r->h_feat && r->h_iq) {
        if (!r->tx_graph) {                                   /* second call: record the sequence (nothing runs during capture) */
            hipGraph_t g = NULL;
            int ok = hipStreamBeginCapture(r->gs, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                ok = hipMemcpyAsync(r->d_feat_in, r->h_feat, sizeof(float) * RD_FEAT_MF, hipMemcpyHostToDevice, r->gs) == hipSuccess &&
                     rade_batch_tx(r->eng, r->d_feat_in, 1, r->d_iq, RD_NMF, NULL, r->gs) == RD_NMF &&
                     hipMemcpyAsync(r->h_iq, r->d_iq, sizeof(RADE_COMP) * RD_NMF, hipMemcpyDeviceToHost, r->gs) == hipSuccess;
                if (hipStreamEndCapture(r->gs, &g) != hipSuccess) ok = 0;
            }
            if (ok && hipGraphInstantiate(&r->tx_graph, g, NULL, NULL, 0) != hipSuccess) { ok = 0; r->tx_graph = NULL; }
            if (g) hipGraphDestroy(g);
            if (!ok) { r->tx_graph_off = 1; (void)hipGetLastError(); }
        }
        if (r->tx_graph) {
            memcpy(r->h_feat, features_in, sizeof(float) * RD_FEAT_MF);
            if (hipGraphLaunch(r->tx_graph, r->gs) == hipSuccess && hipStreamSynchronize(r->gs) == hipSuccess) {
                memcpy(tx_out, r->h_iq, sizeof(RADE_COMP) * RD_NMF); ret = RD_NMF;
            }
        }
    }
    if (!ret && !r->txc && hipMemcpy(r->d_feat_in, features_in, sizeof(float) * RD_FEAT_MF, hipMemcpyHostToDevice) == hipSuccess &&
        rade_batch_tx(r->eng, r->d_feat_in, 1, r->d_iq, RD_NMF, NULL, NULL) == RD_NMF &&
        hipMemcpy(tx_out, r->d_iq, sizeof(RADE_COMP) * RD_NMF, hipMemcpyDeviceToHost) == hipSuccess) ret = RD_NMF;
    r->tx_calls++;
    pthread_mutex_unlock(&r->lock);
    if (!ret) { fprintf(stderr, "rade_tx: device error\n"); exit(1); }       /* reference: check_error() exits (rade_api.c:93-102) */
    return ret;
}

int rade_tx_eoo(struct rade *r, RADE_COMP tx_eoo_out[])
{
    assert(r != NULL); assert(tx_eoo_out != NULL);
    int ret = 0;
    pthread_mutex_lock(&r->lock);
    (void)hipSetDevice(r->device);          /* any thread may call; the staging copies below must target the engine's device */
    if (rade_batch_tx_eoo(r->eng, r->d_iq, RD_NEOO, NULL) == RD_NEOO &&
        hipMemcpy(tx_eoo_out, r->d_iq, sizeof(RADE_COMP) * RD_NEOO, hipMemcpyDeviceToHost) == hipSuccess) ret = RD_NEOO;
    pthread_mutex_unlock(&r->lock);
    if (!ret) { fprintf(stderr, "rade_tx_eoo: device error\n"); exit(1); }
    return ret;
}

int rade_rx(struct rade *r, float features_out[], int *has_eoo_out, float eoo_out[], RADE_COMP rx_in[])
{
    assert(r != NULL); assert(features_out != NULL); assert(rx_in != NULL);
    rade_rx_status st; int ok = 0;
    pthread_mutex_lock(&r->lock);
    (void)hipSetDevice(r->device);          /* any thread may call; the staging copies below must target the engine's device */
    const int nin = r->nin;
    if (hipMemcpy(r->d_rx, rx_in, sizeof(RADE_COMP) * nin, hipMemcpyHostToDevice) == hipSuccess &&
        rade_batch_rx(r->eng, r->d_rx, RD_NINMAX, &nin, 1, r->d_feat_out, RD_FEAT_MF, r->d_eoo, &st, NULL) == 0) {
        ok = 1;
        if (st.n_valid) ok = hipMemcpy(features_out, r->d_feat_out, sizeof(float) * RD_FEAT_MF, hipMemcpyDeviceToHost) == hipSuccess;
        if (has_eoo_out) *has_eoo_out = 0;
        if (ok && st.has_eoo) {
            if (eoo_out) ok = hipMemcpy(eoo_out, r->d_eoo, sizeof(float) * RD_NEOOBITS, hipMemcpyDeviceToHost) == hipSuccess;
            if (has_eoo_out) *has_eoo_out = 1;
        }
        r->nin = st.nin; r->sync = st.sync; r->snr = st.snr_dB;             /* refreshed like rade_api.c:528-530 */
    }
    pthread_mutex_unlock(&r->lock);
    if (!ok) { fprintf(stderr, "rade_rx: device error\n"); exit(1); }
    return st.n_valid ? RD_FEAT_MF : 0;
}

int rade_sync(struct rade *r) { assert(r != NULL); return r->sync; }
float rade_freq_offset(struct rade *r) { assert(r != NULL); return 0; }     /* stub in the reference too (rade_api.c:547-550) */
int rade_snrdB_3k_est(struct rade *r) { assert(r != NULL); return r->snr; }
