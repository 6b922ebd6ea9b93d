/*
 * rade_rx_filter -- stdin IQ.f32 -> stdout features.f32 (stride 36), over rade_api.h.
 *
 * Own implementation of the role /root/reference/src/radae_rx.c:12-59 plays: reads rade_nin() samples
 * per call, writes 432 floats whenever a frame is decoded, appends end-of-over soft bits to eoo_rx.f32.
 * `rade_rx_filter <blob> 1` sets RADE_FOFF_TEST like `radae_rx 1` does.
 */
#include <stdio.h>
#include <stdlib.h>

#include "rade_api.h"

int main(int argc, char **argv)
{
    int flags = RADE_USE_C_DECODER | RADE_VERBOSE_0;
    if (argc > 2 && atoi(argv[2]) == 1) flags |= RADE_FOFF_TEST;
    rade_initialize();
    struct rade *r = rade_open(argc > 1 ? argv[1] : "", flags);
    if (!r) { fprintf(stderr, "rade_rx_filter: rade_open failed\n"); return 1; }
    const int nf = rade_n_features_in_out(r), nmax = rade_nin_max(r), nbits = rade_n_eoo_bits(r);
    float *feat = malloc(sizeof(float) * nf), *eoo = malloc(sizeof(float) * nbits);
    RADE_COMP *rx = malloc(sizeof(RADE_COMP) * nmax);
    FILE *fe = fopen("eoo_rx.f32", "wb");
    int nin = rade_nin(r), has_eoo = 0;
    while (fread(rx, sizeof(RADE_COMP), nin, stdin) == (size_t)nin) {
        if (rade_rx(r, feat, &has_eoo, eoo, rx)) { fwrite(feat, sizeof(float), nf, stdout); fflush(stdout); }
        if (has_eoo && fe) fwrite(eoo, sizeof(float), nbits, fe);
        nin = rade_nin(r);
    }
    fprintf(stderr, "rade_rx_filter: sync %d SNR3k %d dB\n", rade_sync(r), rade_snrdB_3k_est(r));
    if (fe) fclose(fe);
    free(feat); free(eoo); free(rx);
    rade_close(r);
    rade_finalize();
    return 0;
}
