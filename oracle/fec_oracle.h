/*
 * oracle/fec_oracle.h -- TEST INFRASTRUCTURE ONLY: CPU restatement of the GSM L1 FEC soft decode
 * (Viterbi R=1/2 K=5, Fire/CRC parity, XCCH and RACH flows).  See fec_oracle.c for the references.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 */
#ifndef FEC_ORACLE_H
#define FEC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void fo_encode(const uint8_t *bits, int n, uint8_t *out);
void fo_viterbi_decode(const float *soft, int n, uint8_t *out, int nout);
uint64_t fo_parity(uint64_t coeff, unsigned psize, const uint8_t *bits, int n);
uint64_t fo_syndrome(uint64_t coeff, unsigned psize, const uint8_t *bits, int n);
void fo_lsb8msb(uint8_t *bits, int n);
float fo_wire(float v);
int fo_xcch_decode(const float *i4x114, uint8_t *u228, uint8_t *d184, uint64_t *syn);
void fo_xcch_encode(const uint8_t *frame23, const uint8_t *tsc26, uint8_t *bursts4x148);
int fo_rach_decode(const float *e36, uint8_t *u18, unsigned *bsic, unsigned *ra);
int fo_tch_decode(const float *c456, uint8_t *u189, uint8_t *d260);
void fo_tch_decode_batch(const float *soft, int stride, int nbursts, int wire, uint8_t *tch, uint8_t *tch_good,
                         uint8_t *facch, uint8_t *facch_ok, uint8_t *stolen, int nthreads);
void fo_xcch_decode_batch(const float *soft, int stride, int nblk, int wire, uint8_t *frames, uint8_t *ok, int nthreads);
void fo_rach_decode_batch(const float *soft, int stride, int n, int wire, uint8_t *out3, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
