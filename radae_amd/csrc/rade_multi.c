/*
 * rade_multi.c -- one host process driving several MI355X: utterances are sharded contiguously over the devices of a mask
 * (streams are independent, no data-path collective), the DNNw weight blob is read once and broadcast from the
 * first device to the others with ONE ncclBroadcast (RCCL, over xGMI), and job statistics meet in one ncclAllReduce.
 * RCCL is bound at run time (dlopen of librccl.so) so that single-GPU users of libradehip.so need not have it installed;
 * with more than one device its absence is an error, not a fallback.
 *
 * Reference counterpart: none -- the reference is single-stream CPU code; this is the multi-GPU form BASELINE.json's
 * north_star asks for ("shard independent utterances across the 8 GPUs of one node with RCCL broadcast of model19 weights").
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rade_batch.h"

#define RM_MAXDEV 64

struct rade_multi {
    int n_dev, n_total, per;
    int dev[RM_MAXDEV], first[RM_MAXDEV], count[RM_MAXDEV];
    rade_batch *eng[RM_MAXDEV];
    void *rccl;                      /* dlopen handle, NULL when a single device needs no collective */
    ncclComm_t comm[RM_MAXDEV];
    hipStream_t cs[RM_MAXDEV];       /* one collective stream per device */
    double *d_red[RM_MAXDEV];        /* per-device all-reduce buffer */
    int red_cap;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};

void rade_multi_shard(int n_total, int n_dev, int i, int *first, int *count)
{   /* contiguous balanced shards: n_total / n_dev each, the first n_total % n_dev devices one more (no device is left empty when
     * n_total >= n_dev: 10 streams on 8 devices = 2,2,1,1,1,1,1,1).  Config 4 = 2048 utterances: device g owns [256 g, 256 g + 256) */
    const int base = n_total / n_dev, extra = n_total % n_dev;
    const int lo = i * base + (i < extra ? i : extra);
    if (first) *first = lo;
    if (count) *count = base + (i < extra ? 1 : 0);
}

static int rccl_bind(rade_multi *m)
{
    const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
    for (size_t i = 0; i < sizeof names / sizeof names[0] && !m->rccl; i++) m->rccl = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!m->rccl) { fprintf(stderr, "rade_multi: librccl.so not found (%s) -- more than one device needs RCCL\n", dlerror()); return -1; }
#define BIND(field, sym) do { *(void **)&m->field = dlsym(m->rccl, sym); if (!m->field) { fprintf(stderr, "rade_multi: %s missing in librccl.so\n", sym); return -1; } } while (0)
    BIND(CommInitAll, "ncclCommInitAll"); BIND(CommDestroy, "ncclCommDestroy"); BIND(Broadcast, "ncclBroadcast"); BIND(AllReduce, "ncclAllReduce");
    BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    return 0;
}
#define NCHK(m, x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "rade_multi: RCCL error %s at %s:%d\n", (m)->GetErrorString(r_), __FILE__, __LINE__); goto fail; } } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rade_multi: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); goto fail; } } while (0)

typedef struct { rade_multi *m; int i; const void *blob; size_t len; rade_batch_config cfg; } open_job;
static void *open_thread(void *p)
{   /* engines open concurrently: each one parses the (already broadcast) bytes and uploads to its own device */
    open_job *j = p;
    j->m->eng[j->i] = rade_batch_open_mem(j->blob, j->len, &j->cfg);
    return NULL;
}

rade_multi *rade_multi_open(const char *blob_path, int n_streams_total, int max_tx_mf, unsigned long long device_mask, int flags)
{
    rade_multi *m = calloc(1, sizeof *m);
    unsigned char *blob = NULL, **shard_blob = NULL; void **d_blob = NULL;
    if (!m) return NULL;
    int ndev_hw = 0;
    if (hipGetDeviceCount(&ndev_hw) != hipSuccess || ndev_hw <= 0) { fprintf(stderr, "rade_multi: no HIP device available -- this library has no CPU fallback\n"); free(m); return NULL; }
    if (ndev_hw < 64 && (device_mask >> ndev_hw)) {          /* a mask bit for a device that does not exist would silently change the sharding */
        fprintf(stderr, "rade_multi: device mask %#llx selects device(s) beyond the %d present\n", device_mask, ndev_hw); free(m); return NULL;
    }
    for (int g = 0; g < RM_MAXDEV && g < ndev_hw; g++) if (device_mask & (1ull << g)) m->dev[m->n_dev++] = g;
    if (m->n_dev == 0 || n_streams_total < m->n_dev || max_tx_mf <= 0) { fprintf(stderr, "rade_multi: bad arguments (mask %#llx selects %d of %d devices, %d streams)\n", device_mask, m->n_dev, ndev_hw, n_streams_total); free(m); return NULL; }
    m->n_total = n_streams_total;
    for (int i = 0; i < m->n_dev; i++) rade_multi_shard(n_streams_total, m->n_dev, i, &m->first[i], &m->count[i]);

    FILE *f = blob_path ? fopen(blob_path, "rb") : NULL;
    if (!f) { fprintf(stderr, "rade_multi: cannot open weight blob %s\n", blob_path ? blob_path : "(null)"); free(m); return NULL; }
    fseek(f, 0, SEEK_END); const long len = ftell(f); fseek(f, 0, SEEK_SET);
    blob = len > 0 ? malloc((size_t)len) : NULL;
    if (!blob || fread(blob, 1, (size_t)len, f) != (size_t)len) { fclose(f); free(blob); free(m); return NULL; }
    fclose(f);

    shard_blob = calloc(m->n_dev, sizeof *shard_blob); d_blob = calloc(m->n_dev, sizeof *d_blob);
    if (!shard_blob || !d_blob) goto fail;
    const int use_rccl = m->n_dev > 1 || getenv("RADE_MULTI_FORCE_RCCL");
    if (use_rccl) {
        if (rccl_bind(m)) goto fail;
        NCHK(m, m->CommInitAll(m->comm, m->n_dev, m->dev));
        for (int i = 0; i < m->n_dev; i++) {
            HCHK(hipSetDevice(m->dev[i]));
            HCHK(hipStreamCreate(&m->cs[i]));
            HCHK(hipMalloc(&d_blob[i], (size_t)len));
        }
        /* only the root holds the file's bytes; everybody else receives them over the device fabric */
        HCHK(hipSetDevice(m->dev[0]));
        HCHK(hipMemcpy(d_blob[0], blob, (size_t)len, hipMemcpyHostToDevice));
        NCHK(m, m->GroupStart());
        for (int i = 0; i < m->n_dev; i++) NCHK(m, m->Broadcast(d_blob[i], d_blob[i], (size_t)len, ncclUint8, 0, m->comm[i], m->cs[i]));
        NCHK(m, m->GroupEnd());
        for (int i = 0; i < m->n_dev; i++) {
            HCHK(hipSetDevice(m->dev[i]));
            HCHK(hipStreamSynchronize(m->cs[i]));
            shard_blob[i] = malloc((size_t)len);
            if (!shard_blob[i]) goto fail;
            HCHK(hipMemcpy(shard_blob[i], d_blob[i], (size_t)len, hipMemcpyDeviceToHost));     /* each engine parses what ITS device received */
            HCHK(hipFree(d_blob[i])); d_blob[i] = NULL;
        }
    } else shard_blob[0] = blob;

    {
        pthread_t th[RM_MAXDEV]; open_job jobs[RM_MAXDEV];
        for (int i = 0; i < m->n_dev; i++) {
            rade_batch_config cfg = { m->count[i], max_tx_mf, m->dev[i], flags, 0, 0.0f };
            jobs[i] = (open_job){ m, i, shard_blob[i], (size_t)len, cfg };
            if (pthread_create(&th[i], NULL, open_thread, &jobs[i])) { th[i] = 0; open_thread(&jobs[i]); }
        }
        for (int i = 0; i < m->n_dev; i++) if (th[i]) pthread_join(th[i], NULL);
        for (int i = 0; i < m->n_dev; i++) if (!m->eng[i]) { fprintf(stderr, "rade_multi: engine on device %d failed to open\n", m->dev[i]); goto fail; }
    }
    for (int i = 0; i < m->n_dev; i++) if (shard_blob[i] != blob) free(shard_blob[i]);
    free(shard_blob); free(d_blob); free(blob);
    return m;
fail:
    if (shard_blob) for (int i = 0; i < m->n_dev; i++) if (shard_blob[i] != blob) free(shard_blob[i]);
    if (d_blob) for (int i = 0; i < m->n_dev; i++) if (d_blob[i]) { (void)hipSetDevice(m->dev[i]); (void)hipFree(d_blob[i]); }
    free(shard_blob); free(d_blob); free(blob);
    rade_multi_close(m);
    return NULL;
}

void rade_multi_close(rade_multi *m)
{
    if (!m) return;
    for (int i = 0; i < m->n_dev; i++) {
        if (m->eng[i]) rade_batch_close(m->eng[i]);
        (void)hipSetDevice(m->dev[i]);
        if (m->d_red[i]) (void)hipFree(m->d_red[i]);
        if (m->cs[i]) (void)hipStreamDestroy(m->cs[i]);
        if (m->comm[i] && m->CommDestroy) m->CommDestroy(m->comm[i]);
    }
    if (m->rccl) dlclose(m->rccl);
    free(m);
}

int rade_multi_n_devices(const rade_multi *m) { return m ? m->n_dev : 0; }
const char *rade_multi_transport(const rade_multi *m) { return m && m->rccl ? "rccl" : "none (single device)"; }

rade_batch *rade_multi_engine(rade_multi *m, int i, int *device, int *first_stream, int *n_streams)
{
    if (!m || i < 0 || i >= m->n_dev) return NULL;
    if (device) *device = m->dev[i];
    if (first_stream) *first_stream = m->first[i];
    if (n_streams) *n_streams = m->count[i];
    return m->eng[i];
}

/* sum over the devices of n doubles each (per_device[i*n + k]); out[n].  RCCL all-reduce on the device group. */
int rade_multi_allreduce_sum(rade_multi *m, const double *per_device, int n, double *out)
{
    if (!m || !per_device || !out || n <= 0) return -1;
    if (!m->rccl) {                       /* one device, no communicator: the sum of one term */
        memcpy(out, per_device, sizeof(double) * (size_t)n);
        return 0;
    }
    if (n > m->red_cap) {
        for (int i = 0; i < m->n_dev; i++) {
            HCHK(hipSetDevice(m->dev[i]));
            if (m->d_red[i]) HCHK(hipFree(m->d_red[i]));
            m->d_red[i] = NULL;
            HCHK(hipMalloc((void **)&m->d_red[i], sizeof(double) * (size_t)n));
        }
        m->red_cap = n;
    }
    for (int i = 0; i < m->n_dev; i++) {
        HCHK(hipSetDevice(m->dev[i]));
        HCHK(hipMemcpyAsync(m->d_red[i], per_device + (size_t)i * n, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, m->cs[i]));
    }
    NCHK(m, m->GroupStart());
    for (int i = 0; i < m->n_dev; i++) NCHK(m, m->AllReduce(m->d_red[i], m->d_red[i], (size_t)n, ncclFloat64, ncclSum, m->comm[i], m->cs[i]));
    NCHK(m, m->GroupEnd());
    HCHK(hipSetDevice(m->dev[0]));
    HCHK(hipMemcpyAsync(out, m->d_red[0], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, m->cs[0]));
    for (int i = 0; i < m->n_dev; i++) { HCHK(hipSetDevice(m->dev[i])); HCHK(hipStreamSynchronize(m->cs[i])); }
    return 0;
fail:
    return -1;
}

/* fn(i, engine, first_stream, n_streams, arg) on one host thread per device, all at once; returns the first negative return value, else the largest */
typedef struct { rade_multi *m; int i; rade_multi_fn fn; void *arg; int rc; } run_job;
static void *run_thread(void *p)
{
    run_job *j = p;
    (void)hipSetDevice(j->m->dev[j->i]);
    j->rc = j->fn(j->i, j->m->eng[j->i], j->m->first[j->i], j->m->count[j->i], j->arg);
    return NULL;
}
int rade_multi_foreach(rade_multi *m, rade_multi_fn fn, void *arg)
{
    if (!m || !fn) return -1;
    pthread_t th[RM_MAXDEV]; run_job jobs[RM_MAXDEV];
    for (int i = 0; i < m->n_dev; i++) {
        jobs[i] = (run_job){ m, i, fn, arg, 0 };
        if (pthread_create(&th[i], NULL, run_thread, &jobs[i])) { th[i] = 0; run_thread(&jobs[i]); }
    }
    int rc = 0;
    for (int i = 0; i < m->n_dev; i++) if (th[i]) pthread_join(th[i], NULL);
    for (int i = 0; i < m->n_dev; i++) { if (jobs[i].rc < 0) return jobs[i].rc; if (jobs[i].rc > rc) rc = jobs[i].rc; }
    return rc;
}
