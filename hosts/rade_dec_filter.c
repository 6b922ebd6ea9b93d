/*
 * rade_dec_filter -- core decoder filter: z_hat.f32 (80 floats per step) on stdin -> features.f32 (stride 36) on stdout,
 * over include/rade_core.h.
 *
 * Own implementation of the role /root/reference/src/test_rade_dec.c:20-110 plays (same command line
 * `auxdata[0-1] [weights_blob.bin]`; per step 4 x 36 floats out: 20 features, the aux symbol in column 20 when auxdata = 1,
 * zeros elsewhere).
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rade_core.h"

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s auxdata[0-1] [weights_blob.bin]\n", argv[0]); return 1; }
    const int auxdata = atoi(argv[1]);
    const int nb_total = 36, used = 20, nf = used + (auxdata ? 1 : 0), output_dim = nf * RADE_FRAMES_PER_STEP;
    RADEDec model; RADEDecState st;
    WeightArray *list = NULL; void *data = NULL;
    if (argc > 2) {
        FILE *f = fopen(argv[2], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        if (len <= 0 || len > INT_MAX) { fprintf(stderr, "bad weight blob %s (size %ld)\n", argv[2], len); return 1; }
        data = malloc((size_t)len);
        if (!data || fread(data, 1, (size_t)len, f) != (size_t)len || parse_weights(&list, data, (int)len) < 0) { fprintf(stderr, "bad weight blob %s\n", argv[2]); return 1; }
        fclose(f);
    }
    if (init_radedec(&model, list ? list : radedec_arrays, output_dim) != 0) { fprintf(stderr, "Error initialising decoder model (output_dim %d)\n", output_dim); return 1; }
    rade_init_decoder(&st);
    float z[RADE_LATENT_DIM], feat[84], out[RADE_FRAMES_PER_STEP * 36];
    memset(out, 0, sizeof out);
    long n = 0;
    while (fread(z, sizeof(float), RADE_LATENT_DIM, stdin) == RADE_LATENT_DIM) {
        rade_core_decoder(&st, &model, feat, z, 0);
        for (int i = 0; i < RADE_FRAMES_PER_STEP; i++) {
            for (int j = 0; j < used; j++) out[i * nb_total + j] = feat[i * nf + j];
            if (auxdata) out[i * nb_total + used] = feat[i * nf + used];
        }
        fwrite(out, sizeof(float), RADE_FRAMES_PER_STEP * nb_total, stdout);
        fflush(stdout);                 /* per step, as test_rade_dec.c does: a live pipe (rx | dec | vocoder) must not stall on the stdio buffer */
        n++;
    }
    fflush(stdout);
    fprintf(stderr, "%ld latent vectors processed\n", n);
    rade_free_decoder(&st);
    return 0;
}
