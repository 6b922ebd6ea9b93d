/*
 * rade_multi_bench -- BASELINE.json configs[3] from plain C: B utterances per GPU (default 256, i.e. 2048 on 8 GPUs) x 1008 feature
 * frames through encode -> OFDM mod -> MPP Doppler-spread channel + AWGN 3 dB / -11 Hz -> sync / demod / EQ / decode, one host
 * process, one host thread per device, utterances sharded contiguously (utterance u: feature seed 1000 + u), the weight blob
 * broadcast over RCCL, statistics summed with one RCCL all-reduce (include/rade_batch.h: rade_multi_*).  Prints one JSON line.
 *
 * With --pipeline P (default 3, what bench.py times) every device keeps P batches in flight: P engines with their own state, each on its own
 * HIP stream and host thread, the steps dealt to them in turn (the receiver kernel takes half a CU per stream, so the workgroups of two
 * batches share a CU: DESIGN.md 3.5).  --pipeline 1 is the plain one-engine-per-device loop.
 *
 * usage: rade_multi_bench [--gpus N | --mask HEX] [--streams-per-gpu B] [--frames T] [--steps K] [--warmup W] [--pipeline P] [weights.bin]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <pthread.h>
#include <sys/resource.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rade_batch.h"

#define PI_D 3.14159265358979323846

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* xorshift64* + Box-Muller: the features only have to be "speech-like" AR(1) noise, not a particular sequence */
static uint64_t rng_next(uint64_t *s) { *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27; return *s * 2685821657736338717ull; }
static double rng_gauss(uint64_t *s)
{
    const double u1 = ((rng_next(s) >> 11) + 0.5) / 9007199254740992.0, u2 = ((rng_next(s) >> 11) + 0.5) / 9007199254740992.0;
    return sqrt(-2.0 * log(u1)) * cos(2.0 * PI_D * u2);
}
static void synth_features(uint64_t seed, int T, float *out /* [T][36] */)
{   /* x[t] = 0.9 x[t-1] + 0.436 N(0,1); f0 = 4 x0, f18 = 0.5 x18, f19 = clip(0.3 x19, +-0.5) */
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1; double x[20] = { 0 };
    memset(out, 0, sizeof(float) * (size_t)T * 36);
    for (int t = 0; t < T; t++)
        for (int j = 0; j < 20; j++) {
            x[j] = 0.9 * x[j] + 0.436 * rng_gauss(&s);
            double f = x[j];
            if (j == 0) f *= 4.0; else if (j == 18) f *= 0.5; else if (j == 19) { f *= 0.3; if (f > 0.5) f = 0.5; if (f < -0.5) f = -0.5; }
            out[(size_t)t * 36 + j] = (float)f;
        }
}
/* Gaussian-PSD Doppler filter by frequency sampling (doppler_spread.m:20-32; radae_amd/channel_tools.py): 100 taps at lowFs */
static void doppler_taps(double spread_hz, double low_fs, float *h /* [100] */)
{
    const int ntaps = 100, npt = 512, N = 2 * npt; const double sigma = spread_hz / 2.0;
    for (int n = 0; n < ntaps; n++) {
        double acc = 0.0;
        for (int k = 0; k <= npt; k++) {
            const double f = low_fs / 2.0 * k / npt, mag = 1.0 / (sigma * sqrt(2 * PI_D)) * exp(-f * f / (2 * sigma * sigma));
            const double ph = -PI_D * k * (ntaps - 1) / (2.0 * npt) + 2.0 * PI_D * k * n / N;
            acc += (k == 0 || k == npt ? 1.0 : 2.0) * mag * cos(ph);
        }
        h[n] = (float)(acc / N * (0.54 - 0.46 * cos(2.0 * PI_D * n / (ntaps - 1))));
    }
}

typedef struct {
    int T, n_mf, n_sig, n_pre, n_post, n_total, steps, warmup;
    float taps[100]; int low_ratio;
    pthread_barrier_t bar;
    volatile int failed;             /* set by any device thread whose step failed: the others skip their work but still reach both barriers */
    /* per device */
    float *feat[64], *fout[64]; void *G[64], *rx[64]; int *avail[64]; rade_rx_status *st[64];
    double t_step[64]; double stats[64][6];
    int pipeline, flags, device[64]; const char *blob;
} job;

static int setup(int i, rade_batch *e, int first, int n, void *arg)
{
    job *j = arg;
    float *hf = malloc(sizeof(float) * (size_t)n * j->T * 36);
    if (!hf) return -1;
    for (int b = 0; b < n; b++) synth_features(1000 + first + b, j->T, hf + (size_t)b * j->T * 36);
    int bad = hipMalloc((void **)&j->feat[i], sizeof(float) * (size_t)n * j->T * 36) || hipMalloc(&j->G[i], 16ull * n * j->n_sig) ||
              hipMalloc(&j->rx[i], 8ull * n * j->n_total) ||
              hipMalloc((void **)&j->fout[i], sizeof(float) * (size_t)n * (j->n_mf + 8) * 432);
    if (!bad) bad = hipMemcpy(j->feat[i], hf, sizeof(float) * (size_t)n * j->T * 36, hipMemcpyHostToDevice) != hipSuccess;
    free(hf);
    if (bad) {                       /* release what this device got (hipFree(NULL) is a no-op) */
        hipFree(j->feat[i]); hipFree(j->G[i]); hipFree(j->rx[i]); hipFree(j->fout[i]);
        j->feat[i] = j->fout[i] = NULL; j->G[i] = j->rx[i] = NULL;
        return -1;
    }
    if (rade_batch_multipath_gen(e, j->taps, 100, j->low_ratio, j->n_sig, NULL, 5000 + first, j->G[i], NULL) != j->n_sig) return -1;
    j->avail[i] = malloc(sizeof(int) * n); j->st[i] = malloc(sizeof(rade_rx_status) * n);
    for (int b = 0; b < n; b++) j->avail[i][b] = j->n_total;
    return 0;
}

static int one_step(job *j, int i, rade_batch *e, void *rx, float *fout, rade_rx_status *st, unsigned long long seed, void *stream)
{
    rade_channel_params p; memset(&p, 0, sizeof p);
    p.n_sig = j->n_sig; p.n_pre = j->n_pre; p.n_post = j->n_post; p.with_eoo = 1; p.sigma = rade_sigma_from_EbNodB(3.0f); p.freq_offset = -11.0f;
    p.G_dev = j->G[i]; p.seed = seed;
    rade_batch_reset(e, stream);
    if (rade_batch_tx_channel(e, j->feat[i], j->n_mf, NULL, 0, rx, j->n_total, &p, stream) != j->n_total) return -1;      /* transmit + channel in one pass */
    return rade_batch_rx(e, rx, j->n_total, j->avail[i], 1 << 20, fout, (long)(j->n_mf + 8) * 432, NULL, st, stream);
}

/* one of the P batches in flight on device i: its own engine (worker 0: the rade_multi one), stream, received-sample / feature / status buffers */
typedef struct { job *j; int i, w, n; rade_batch *e; int own_engine; hipStream_t s; void *rx; float *fout; rade_rx_status *st; int rc; double t_end; pthread_t th; } worker;

static void *worker_main(void *arg)
{
    worker *k = arg; job *j = k->j; const int P = j->pipeline;
    if (hipSetDevice(j->device[k->i]) != hipSuccess) { k->rc = -1; j->failed = 1; }
    /* a failed step must not leave the other threads waiting at a barrier for ever: the error is recorded, the work skipped, and every
     * thread still reaches both barriers */
    for (int w = k->w; w < j->warmup && !j->failed; w += P) if (one_step(j, k->i, k->e, k->rx, k->fout, k->st, 100 + w, k->s)) { j->failed = 1; k->rc = -1; }
    hipStreamSynchronize(k->s);
    pthread_barrier_wait(&j->bar);                               /* all batches of all devices start the timed region together */
    for (int s = k->w; s < j->steps && !j->failed; s += P) if (one_step(j, k->i, k->e, k->rx, k->fout, k->st, 1 + s, k->s)) { j->failed = 1; k->rc = -1; }
    hipStreamSynchronize(k->s);
    k->t_end = now_s();
    pthread_barrier_wait(&j->bar);
    return NULL;
}

static int run(int i, rade_batch *e, int first, int n, void *arg)
{
    job *j = arg; const int P = j->pipeline;
    worker wk[16]; memset(wk, 0, sizeof wk);
    int bad = 0;
    for (int w = 0; w < P; w++) {
        worker *k = &wk[w];
        k->j = j; k->i = i; k->w = w; k->n = n; k->e = e; k->rx = j->rx[i]; k->fout = j->fout[i]; k->st = j->st[i];
        if (w > 0) {                                             /* further batches in flight: an engine, buffers and a stream of their own */
            rade_batch_config c; memset(&c, 0, sizeof c);
            c.n_streams = n; c.max_tx_mf = j->n_mf; c.device = j->device[i]; c.flags = j->flags;
            k->e = rade_batch_open(j->blob, &c); k->own_engine = 1;
            k->st = malloc(sizeof(rade_rx_status) * n);
            bad |= !k->e || !k->st || hipMalloc(&k->rx, 8ull * n * j->n_total) != hipSuccess || hipMalloc((void **)&k->fout, sizeof(float) * (size_t)n * (j->n_mf + 8) * 432) != hipSuccess;
        }
        bad |= hipStreamCreate(&k->s) != hipSuccess;
    }
    if (bad) j->failed = 1;                                       /* the workers still run: they skip the work and meet the others at the barriers */
    double t0 = 0.0;
    for (int w = 1; w < P; w++) pthread_create(&wk[w].th, NULL, worker_main, &wk[w]);
    {   /* this thread is worker 0; the start time is taken when it leaves the first barrier */
        worker *k = &wk[0];
        for (int w = 0; w < j->warmup && !j->failed; w += P) if (one_step(j, i, k->e, k->rx, k->fout, k->st, 100 + w, k->s)) { j->failed = 1; k->rc = -1; }
        hipStreamSynchronize(k->s);
        pthread_barrier_wait(&j->bar);
        t0 = now_s();
        for (int s = 0; s < j->steps && !j->failed; s += P) if (one_step(j, i, k->e, k->rx, k->fout, k->st, 1 + s, k->s)) { j->failed = 1; k->rc = -1; }
        hipStreamSynchronize(k->s);
        k->t_end = now_s();
        pthread_barrier_wait(&j->bar);
    }
    for (int w = 1; w < P; w++) pthread_join(wk[w].th, NULL);
    double t1 = t0; int rc = 0;
    for (int w = 0; w < P; w++) { if (wk[w].t_end > t1) t1 = wk[w].t_end; rc |= wk[w].rc; }
    j->t_step[i] = (t1 - t0) / j->steps;
    if (!(rc || j->failed)) {
        const worker *last = &wk[(j->steps - 1) % P];             /* the batch that ran the last timed step */
        double *s = j->stats[i]; memset(s, 0, sizeof(double) * 6);
        for (int b = 0; b < n; b++) {
            const rade_rx_status *r = &last->st[b];
            s[0] += j->T; s[1] += 12.0 * r->n_valid; s[2] += r->n_calls; s[3] += r->n_valid + r->has_eoo; s[4] += r->has_eoo; s[5] += r->consumed;
        }
    }
    for (int w = 0; w < P; w++) {
        worker *k = &wk[w];
        if (k->s) hipStreamDestroy(k->s);
        if (k->own_engine) { if (k->e) rade_batch_close(k->e); hipFree(k->rx); hipFree(k->fout); free(k->st); }
    }
    (void)first;
    return (rc || j->failed) ? -1 : 0;
}

int main(int argc, char **argv)
{
    int gpus = 1, B = 256; unsigned long long mask = 0; const char *blob = NULL;
    job j; memset(&j, 0, sizeof j);
    j.T = 1008; j.steps = 21; j.warmup = 3; j.pipeline = 3;
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--mask") && a + 1 < argc) mask = strtoull(argv[++a], NULL, 16);
        else if (!strcmp(argv[a], "--streams-per-gpu") && a + 1 < argc) B = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--frames") && a + 1 < argc) j.T = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--steps") && a + 1 < argc) j.steps = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--warmup") && a + 1 < argc) j.warmup = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--pipeline") && a + 1 < argc) j.pipeline = atoi(argv[++a]);
        else blob = argv[a];
    }
    if (!mask) mask = gpus >= 64 ? ~0ull : (1ull << gpus) - 1;
    if (!blob) blob = getenv("RADE_MODEL_FILE") ? getenv("RADE_MODEL_FILE") : "weights/model19_check3.bin";
    int n_dev = 0; for (int g = 0; g < 64; g++) n_dev += (mask >> g) & 1;
    j.n_mf = j.T / 12; j.n_sig = j.n_mf * 960; j.n_pre = 8000; j.n_post = 1152; j.n_total = j.n_pre + j.n_sig + 1152 + j.n_post;
    j.low_ratio = 800; doppler_taps(1.0, 10.0, j.taps);         /* MPP: 1 Hz Doppler spread, lowFs = 10 Hz (multipath_samples.m:13) */
    if (j.pipeline < 1) j.pipeline = 1;
    if (j.pipeline > 16) j.pipeline = 16;
    if (j.pipeline > 3) setenv("GPU_MAX_HW_QUEUES", "8", 0);     /* HIP's default of 4 hardware queues per device makes more streams than that share a queue and serialise */
    j.flags = 0; j.blob = blob;
    rade_multi *m = rade_multi_open(blob, B * n_dev, j.n_mf, mask, j.flags);
    if (!m) return 1;
    n_dev = rade_multi_n_devices(m);
    for (int i = 0; i < n_dev; i++) rade_multi_engine(m, i, &j.device[i], NULL, NULL);
    pthread_barrier_init(&j.bar, NULL, n_dev * j.pipeline);
    if (rade_multi_foreach(m, setup, &j) < 0) { fprintf(stderr, "rade_multi_bench: setup failed\n"); return 1; }
    struct rusage ru0, ru1; getrusage(RUSAGE_SELF, &ru0);      /* (spans warm-up and timed steps of every device thread: CPU seconds per step below is an average over both) */
    if (rade_multi_foreach(m, run, &j) < 0) { fprintf(stderr, "rade_multi_bench: run failed\n"); return 1; }
    getrusage(RUSAGE_SELF, &ru1);
    const double cpu_s = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + 1e-6 * (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec)
                       + (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec);
    double flat[64 * 6], tot[6], tmax = 0.0;
    for (int i = 0; i < n_dev; i++) { memcpy(flat + 6 * i, j.stats[i], sizeof(double) * 6); if (j.t_step[i] > tmax) tmax = j.t_step[i]; }
    if (rade_multi_allreduce_sum(m, flat, 6, tot)) return 1;
    printf("{\"metric\": \"vocoder-feature frames/sec (enc+chan+dec), model19\", \"value\": %.1f, \"unit\": \"frames/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
           "\"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", \"dtype\": \"f32\", \"data\": \"synthetic\", "
           "\"config\": {\"workload\": \"model19_check3, %d utterances x %d frames sharded over %d GPU(s) from one C host process (configs[3] recipe)\", \"batches_in_flight_per_gpu\": %d, \"collectives\": \"%s: 1 broadcast (blob), 1 all-reduce (statistics)\"}, "
           "\"per_device_ms_per_step\": [", tot[0] / tmax, n_dev, j.steps, j.warmup, 1e3 * tmax, B * n_dev, j.T, n_dev, j.pipeline, rade_multi_transport(m));
    for (int i = 0; i < n_dev; i++) printf("%s%.4f", i ? ", " : "", 1e3 * j.t_step[i]);
    printf("], \"host\": {\"cpu_s_per_step\": %.6f, \"cpu_quota\": %.2f, \"engines_in_process\": %d, \"rx_wait\": \"%s\"}",
           cpu_s / (j.steps + j.warmup), rade_host_cpu_quota(), n_dev * j.pipeline, rade_sync_policy(n_dev * j.pipeline, rade_host_cpu_quota()) ? "sleep" : "spin");
    printf(", \"job_last_step\": {\"offered_frames\": %.0f, \"decoded_frames\": %.0f, \"rx_calls\": %.0f, \"sync_calls\": %.0f, \"eoo_detected_streams\": %.0f, \"samples_consumed\": %.0f}}\n",
           tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]);
    rade_multi_close(m);
    return 0;
}
