/*
 * rade_enc_filter -- core encoder filter: features.f32 (stride 36) on stdin -> z.f32 (80 floats per 40 ms step) on stdout,
 * over include/rade_core.h.
 *
 * Own implementation of the role /root/reference/src/test_rade_enc.c:20-110 plays (same command line
 * `bottleneck[1-3] auxdata[0-1] [weights_blob.bin]`, same wire formats: 4 x 36 floats in per step, the first 20 of each
 * frame used, aux symbol -1 appended when auxdata = 1; 80 floats out).
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rade_core.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s bottleneck[1-3] auxdata[0-1] [weights_blob.bin]\n", argv[0]); return 1; }
    const int bottleneck = atoi(argv[1]), auxdata = atoi(argv[2]);
    const int nb_total = 36, used = 20, nf = used + (auxdata ? 1 : 0), input_dim = nf * RADE_FRAMES_PER_STEP;
    RADEEnc model; RADEEncState st;
    WeightArray *list = NULL; void *data = NULL;
    if (argc > 3) {
        FILE *f = fopen(argv[3], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[3]); return 1; }
        fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        if (len <= 0 || len > INT_MAX) { fprintf(stderr, "bad weight blob %s (size %ld)\n", argv[3], len); return 1; }
        data = malloc((size_t)len);
        if (!data || fread(data, 1, (size_t)len, f) != (size_t)len || parse_weights(&list, data, (int)len) < 0) { fprintf(stderr, "bad weight blob %s\n", argv[3]); return 1; }
        fclose(f);
    }
    if (init_radeenc(&model, list ? list : radeenc_arrays, input_dim) != 0) { fprintf(stderr, "Error initialising encoder model (input_dim %d)\n", input_dim); return 1; }
    rade_init_encoder(&st);
    float in[RADE_FRAMES_PER_STEP * 36], feat[84], z[RADE_LATENT_DIM];
    long n = 0;
    while (fread(in, sizeof(float), RADE_FRAMES_PER_STEP * nb_total, stdin) == (size_t)(RADE_FRAMES_PER_STEP * nb_total)) {
        for (int i = 0; i < RADE_FRAMES_PER_STEP; i++) {
            for (int j = 0; j < used; j++) feat[i * nf + j] = in[i * nb_total + j];
            if (auxdata) feat[i * nf + used] = -1.0f;
        }
        rade_core_encoder(&st, &model, z, feat, 0, bottleneck);
        fwrite(z, sizeof(float), RADE_LATENT_DIM, stdout);
        fflush(stdout);                 /* per step, as test_rade_enc.c does: a live pipe (enc | tx) must not stall on the stdio buffer */
        n++;
    }
    fflush(stdout);
    fprintf(stderr, "%ld feature vectors processed\n", n);
    rade_free_encoder(&st);
    return 0;
}
