/*
 * rade_api.h -- single-stream RADE C ABI, implemented by libradehip.so (HIP / gfx950 back end).
 *
 * Drop-in boundary.  The entry points, argument orders, element counts and return conventions
 * below are the ones freedv-gui and the stock stdin/stdout hosts bind to in the reference
 * (/root/reference/src/rade_api.h:82-129; behaviour from /root/reference/src/rade_api.c:329-555).
 * The stock hosts src/radae_tx.c:12-58 and src/radae_rx.c:12-59 compile against this header and
 * run unmodified.  Nothing behind this header embeds Python: the modem DSP, the sync state machine
 * and the encoder/decoder all run as HIP kernels (see include/rade_batch.h for the batched form).
 *
 * Per-symbol map to the reference:
 *   rade_initialize/rade_finalize  rade_api.h:82,85   (reference starts/stops CPython; here HIP device init / no-op)
 *   rade_open / rade_close         :88-89             one handle = one Tx + one Rx; model_file is
 *                                                     ignored by the reference (rade_api.c:351); here a
 *                                                     readable DNNw blob path is honoured, anything else
 *                                                     falls back to $RADE_MODEL_FILE or the built-in path
 *   rade_version                   :92                returns 1 (rade_api.c:37)
 *   rade_n_tx_out ... n_eoo_bits   :95-99             960, 1152, 1120, 432, 180 (elements)
 *   rade_tx                        :103               432 floats (12 x 36, first 20 used) -> 960 IQ
 *   rade_tx_set_eoo_bits           :107               180 floats +-1
 *   rade_tx_eoo                    :111               1152 IQ
 *   rade_nin                       :114               samples the next rade_rx consumes (800/960/1120)
 *   rade_rx                        :120               returns 432 when features_out is valid else 0
 *   rade_sync, rade_freq_offset, rade_snrdB_3k_est :123-129  (freq_offset is a stub returning 0 in the
 *                                                     reference, rade_api.c:547-550; kept)
 */
#ifndef RADE_API_H_MI355X
#define RADE_API_H_MI355X

#include <sys/types.h>

#if defined(IS_BUILDING_RADE_API) && !defined(_WIN32)
#define RADE_EXPORT __attribute__((visibility("default")))
#else
#define RADE_EXPORT
#endif

#ifndef __RADE_COMP__
#define __RADE_COMP__
typedef struct { float real; float imag; } RADE_COMP;   /* interleaved I/Q, same memory as complex64 */
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define RADE_MODEM_SAMPLE_RATE  8000
#define RADE_SPEECH_SAMPLE_RATE 16000

/* rade_open() flag bits.  The two USE_C_* bits select the native core in the reference; here the
 * core is always native (HIP), so they are accepted and ignored. */
#define RADE_USE_C_ENCODER 0x1
#define RADE_USE_C_DECODER 0x2
#define RADE_FOFF_TEST     0x4   /* inject a 10 Hz error on first sync (UW false-sync test) */
#define RADE_VERBOSE_0     0x8   /* quiet */
/* Extension (not in the reference's header): rade_open() also honours RADE_BATCH_TX_BPF (0x400, include/rade_batch.h) -- the reference's
 * radae_tx(..., txbpf_en=True) / `radae_tx.py --txbpf` (radae_txe.py:74-83), which the reference's C API has no switch for: rade_tx() and
 * rade_tx_eoo() then return band-pass filtered and magnitude-clipped samples (the filter state carried from call to call). */

struct rade;

RADE_EXPORT void rade_initialize(void);
RADE_EXPORT void rade_finalize(void);

RADE_EXPORT struct rade *rade_open(char model_file[], int flags);
RADE_EXPORT void rade_close(struct rade *r);

RADE_EXPORT int rade_version(void);

RADE_EXPORT int rade_n_tx_out(struct rade *r);
RADE_EXPORT int rade_n_tx_eoo_out(struct rade *r);
RADE_EXPORT int rade_nin_max(struct rade *r);
RADE_EXPORT int rade_n_features_in_out(struct rade *r);
RADE_EXPORT int rade_n_eoo_bits(struct rade *r);

RADE_EXPORT int rade_tx(struct rade *r, RADE_COMP tx_out[], float features_in[]);
RADE_EXPORT void rade_tx_set_eoo_bits(struct rade *r, float eoo_bits[]);
RADE_EXPORT int rade_tx_eoo(struct rade *r, RADE_COMP tx_eoo_out[]);

RADE_EXPORT int rade_nin(struct rade *r);
RADE_EXPORT int rade_rx(struct rade *r, float features_out[], int *has_eoo_out, float eoo_out[], RADE_COMP rx_in[]);

RADE_EXPORT int rade_sync(struct rade *r);
RADE_EXPORT float rade_freq_offset(struct rade *r);
RADE_EXPORT int rade_snrdB_3k_est(struct rade *r);

#ifdef __cplusplus
}
#endif
#endif
