/*
 * rade_core.h -- the core encoder / decoder level of the RADE C path, MI355X back end.
 *
 * Replaces the interface of /root/reference/src/rade_core.h:42-46 (rade_init_encoder / rade_core_encoder /
 * rade_init_decoder / rade_core_decoder), the model initialisers rade_enc_data.h:101 / rade_dec_data.h:116
 * (init_radeenc / init_radedec with the run-time input / output dimension 80 <-> 84, README.md:582-588) and the weight-blob
 * walker the reference's harnesses use (Opus dnn/nnet.h parse_weights(): test_rade_enc.c:60-66, test_rade_dec.c:59-65).
 * Same call sequence, argument meaning and element counts; the arithmetic is the fp32 model on the GPU
 * (radae_amd/csrc: rade_batch_encode / rade_batch_decode with one stream), not the int8-activation CPU path.
 *
 * Differences a caller can see, all additive:
 *   - state structs hold a handle to device-resident GRU / conv state instead of the arrays themselves; they are still
 *     caller-allocated (stack or heap) and still start life with rade_init_encoder() / rade_init_decoder();
 *     rade_free_encoder() / rade_free_decoder() release the device side (the reference has nothing to release);
 *   - `arch` is ignored (the reference hard-wires arch = 0: rade_api.c:421, test_rade_enc.c:88);
 *   - there are no compiled-in weights: radeenc_arrays / radedec_arrays select the default blob
 *     ($RADE_MODEL_FILE or <library dir>/../weights/model19_check3.bin).
 * Errors: init_radeenc / init_radedec return non-zero like the reference's; rade_core_encoder / rade_core_decoder print and
 * exit(1) on a device error (no CPU fallback).
 */
#ifndef RADE_CORE_H
#define RADE_CORE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RADE_FRAMES_PER_STEP 4      /* rade_constants.h */
#define RADE_LATENT_DIM 80

/* layout of Opus' WeightArray (dnn/nnet.h): one record of a DNNw blob */
typedef struct { const char *name; int type; int size; const void *data; } WeightArray;
/* walks a DNNw blob (src/write_rade_weights.c:51-74); *list is the malloc'ed pointer itself (free(list) is valid, also while models initialised from it are in use), terminated by a NULL name; returns the number of
 * arrays or -1.  `data` must stay mapped while models initialised from the list are in use (as in the reference). */
int rade_parse_weights(WeightArray **list, const void *data, int len);
#ifndef RADE_CORE_KEEP_OPUS_NAMES_FREE
#define parse_weights rade_parse_weights
#endif

typedef struct RADEEnc { const void *blob; int blob_len; int input_dim; int nb_z; } RADEEnc;
typedef struct RADEDec { const void *blob; int blob_len; int output_dim; int nb_z; } RADEDec;
typedef struct RADEEncStruct { int initialized; void *dev; } RADEEncState;
typedef struct RADEDecStruct { int initialized; void *dev; } RADEDecState;

extern const WeightArray radeenc_arrays[];
extern const WeightArray radedec_arrays[];

/* input_dim / output_dim = 4 x 21 = 84 (model19_check3: auxdata) or 4 x 20 = 80; must match the blob */
int init_radeenc(RADEEnc *model, const WeightArray *arrays, int input_dim);
int init_radedec(RADEDec *model, const WeightArray *arrays, int output_dim);

void rade_init_encoder(RADEEncState *enc_state);
/* one 40 ms step: features[input_dim] -> z[80]; bottleneck 1: z = tanh(.), otherwise linear (rade_enc.c:55-114) */
void rade_core_encoder(RADEEncState *enc_state, const RADEEnc *model, float *z, const float *features, int arch, int bottleneck);

void rade_init_decoder(RADEDecState *dec_state);
/* one step: z_hat[80] -> features[output_dim] (rade_dec.c:50-102) */
void rade_core_decoder(RADEDecState *dec_state, const RADEDec *model, float *features, const float *z_hat, int arch);

/* additive: release the device-side state of a stream */
void rade_free_encoder(RADEEncState *enc_state);
void rade_free_decoder(RADEDecState *dec_state);
/* additive: zero GRU / conv state of an initialised stream, keeping its uploaded weights (rade_init_* on a state in use would leak them) */
void rade_reset_encoder(RADEEncState *enc_state);
void rade_reset_decoder(RADEDecState *dec_state);

#ifdef __cplusplus
}
#endif
#endif
