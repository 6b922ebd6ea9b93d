/*
 * rade_tx_filter -- stdin features.f32 (stride 36) -> stdout IQ.f32 (..IQIQ.., float32), over rade_api.h.
 *
 * Own implementation of the role /root/reference/src/radae_tx.c:12-58 plays (same wire formats:
 * 432 floats in per modem frame, 960 complex samples out, EOO frame + one frame of silence at EOF,
 * optional eoo_tx.f32 side file with 180 +-1 floats).  Used by the pipe test; the reference's own
 * host compiles against include/rade_api.h unchanged.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rade_api.h"

int main(int argc, char **argv)
{
    rade_initialize();
    struct rade *r = rade_open(argc > 1 ? argv[1] : "", RADE_USE_C_ENCODER | RADE_VERBOSE_0);
    if (!r) { fprintf(stderr, "rade_tx_filter: rade_open failed\n"); return 1; }
    const int nf = rade_n_features_in_out(r), ntx = rade_n_tx_out(r), neoo = rade_n_tx_eoo_out(r), nbits = rade_n_eoo_bits(r);
    float *feat = malloc(sizeof(float) * nf), *bits = malloc(sizeof(float) * nbits);
    RADE_COMP *tx = malloc(sizeof(RADE_COMP) * (ntx > neoo ? ntx : neoo));
    FILE *fb = fopen("eoo_tx.f32", "rb");
    if (fb) { if (fread(bits, sizeof(float), nbits, fb) == (size_t)nbits) rade_tx_set_eoo_bits(r, bits); fclose(fb); }
    while (fread(feat, sizeof(float), nf, stdin) == (size_t)nf) {
        const int n = rade_tx(r, tx, feat);
        fwrite(tx, sizeof(RADE_COMP), n, stdout);
    }
    int n = rade_tx_eoo(r, tx);
    fwrite(tx, sizeof(RADE_COMP), n, stdout);
    memset(tx, 0, sizeof(RADE_COMP) * n);               /* silence so the receiver can finish the EOO frame */
    fwrite(tx, sizeof(RADE_COMP), n, stdout);
    fflush(stdout);
    free(feat); free(bits); free(tx);
    rade_close(r);
    rade_finalize();
    return 0;
}
