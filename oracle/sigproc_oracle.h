/*
 * oracle/sigproc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the OpenBTS sigProcLib burst-processing path.
 * It is the parity checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (libtrxsig.so) never links or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the real reference compiled in place from /root/reference
 * (oracle/_ref, tests/test_oracle_vs_ref.py, build container only) and against
 * the golden vectors captured from that reference (tests/golden/, everywhere).
 *
 * All line references are to /root/reference/Transceiver/sigProcLib.cpp unless
 * another file is named.
 */
#ifndef SIGPROC_ORACLE_H
#define SIGPROC_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float r, i; } so_c32;

#define SO_MAXSPS 8
#define SO_TABLESIZE 1024

/* ConvType, sigProcLib.h:41-48 */
enum { SO_FULL_SPAN = 0, SO_OVERLAP_ONLY = 1, SO_START_ONLY = 2, SO_WITH_TAIL = 3,
       SO_NO_DELAY = 4, SO_CUSTOM = 5 /* Transceiver52M/sigProcLib.h:47 */ };

/* Library state: the reference keeps these in process globals
   (sigProcLib.cpp:39-59); here they live in a context so the oracle is
   re-entrant. */
typedef struct {
  int sps;
  int variant52m;                       /* 0 = Transceiver/, 1 = Transceiver52M/ */
  float cosT[SO_TABLESIZE + 2];         /* +1 as the reference, +1 guard for arg==1.0 (reads [1025]*0) */
  float sinT[SO_TABLESIZE + 2];
  so_c32 rot[157 * SO_MAXSPS];          /* GMSKRotation */
  so_c32 rev[157 * SO_MAXSPS];          /* GMSKReverseRotation */
  int pulse_len;
  float pulse[2 * SO_MAXSPS + 1];       /* generateGSMPulse(2,sps), real only */
  so_c32 mid[8][16 * SO_MAXSPS];        /* gMidambles[t]->sequence */
  float mid_toa[8];
  so_c32 mid_gain[8];
  so_c32 rach[41 * SO_MAXSPS];          /* gRACHSequence->sequence */
  float rach_toa;
  so_c32 rach_gain;
} so_ctx;

/* GSM constants restated from GSM/GSMCommon.cpp:44-57 (bit values only) */
extern const char so_training_sequence[8][27];
extern const char so_dummy_burst[149];
extern const char so_rach_synch[42];

size_t so_ctx_size(void);
/* sigProcLibSetup + generateGSMPulse(2,sps) + generateRACHSequence + generateMidamble(0..7) */
int so_setup(so_ctx *c, int sps, int variant52m);

float so_sinLookup(const so_ctx *c, float x);
float so_cosLookup(const so_ctx *c, float x);
so_c32 so_expjLookup(const so_ctx *c, float x);
float so_sinc(const so_ctx *c, float x);
/* the rest of sigProcLib.h's surface: dB / dBinv (ref:88-144), vectorNorm2 / vectorPower (:146-160), frequencyShift
   (:432-471; returns the final phase), addVector (:746-758), offsetVector (:760-777), resampleVector (:1213-1243, as it
   behaves: see the .c), gaussianNoise (:618-637, rand()) */
float so_dB(float x);
float so_dBinv(float x);
float so_vector_norm2(const so_c32 *x, int n);
float so_vector_power(const so_c32 *x, int n);
float so_frequency_shift(const so_ctx *c, const so_c32 *x, int n, float freq, float startPhase, int real_only, so_c32 *y);
void so_add_vector(so_c32 *x, int nx, const so_c32 *y, int ny);
/* the channeliser's mixer convention (see the .c): frequencyShift sample by sample with directly formed phases */
void so_mix_down(const so_ctx *c, const so_c32 *x, int n, long long n0, float freq, so_c32 *y);
void so_offset_vector(so_c32 *x, int n, so_c32 offset, int real_only);
int so_resample_vector(const so_c32 *x, int n, float expFactor, so_c32 endPoint, so_c32 *out);
void so_gaussian_noise(int length, float variance, so_c32 mean, so_c32 *out);

/* flags: bit0 = a realOnly, bit1 = b realOnly.  Returns output length or -1.
   start/len only used for SO_CUSTOM. */
int so_convolve(const so_c32 *a, int La, const so_c32 *b, int Lb, so_c32 *out,
                int span, int flags, unsigned startIx, unsigned len);
int so_correlate(const so_c32 *a, int La, const so_c32 *b, int Lb, so_c32 *out,
                 int span, int flags);
void so_scale_vector(so_c32 *x, int n, so_c32 s, int real_only);
void so_gmsk_rotate(const so_ctx *c, so_c32 *x, int n, int reverse, int real_only);
void so_delay_vector(const so_ctx *c, so_c32 *x, int n, float delay);
so_c32 so_interpolate_point(const so_ctx *c, const so_c32 *x, int n, float ix, int real_only);
so_c32 so_peak_detect(const so_ctx *c, const so_c32 *x, int n, float *peakIndex, float *avgPwr);

int so_modulate(const so_ctx *c, const char *bits, int nbits, const so_c32 *pulse,
                int pulse_len, int pulse_real, int guard, so_c32 *out);
int so_modulate_gsm(const so_ctx *c, const char *bits, int nbits, int guard, so_c32 *out);
int so_energy_detect(const so_c32 *x, int n, unsigned win, float thresh, float *avgPwr, int variant52m);
/* chan: 6*sps entries; *chan_len = 0 unless the reference would have allocated it */
int so_analyze_traffic(const so_ctx *c, const so_c32 *x, int n, unsigned tsc, float thresh,
                       unsigned maxTOA, so_c32 *amp, float *toa, int reqChan,
                       so_c32 *chan, int *chan_len, float *chan_off, float *peak_to_mean);
int so_detect_rach(const so_ctx *c, const so_c32 *x, int n, float thresh, so_c32 *amp,
                   float *toa, float *peak_to_mean);
int so_demodulate(const so_ctx *c, const so_c32 *x, int n, so_c32 amp, float toa, float *soft);

/* createLPF with the raw coefficient table passed in (the tables themselves are
   reference data: tests/golden/lpf_tables.npz) */
void so_create_lpf(const float *raw, int len, float gainDC, float *out);
int so_polyphase_resample(const so_c32 *x, int n, int P, int Q, const float *lpf, int L, so_c32 *out);

int so_design_dfe(const so_ctx *c, const so_c32 *chan, int nchan, float snr, int Nf, so_c32 *w, so_c32 *b);
int so_equalize(const so_ctx *c, const so_c32 *x, int n, float toa, const so_c32 *w, int nw,
                const so_c32 *b, int nb, float *soft);

/* Batched loops over packed bursts (off/len in samples), nthreads OpenMP threads.
   Used by the parity tests and as bench.py's cpu_baseline ("port"). */
int so_normal_batch(const so_ctx *c, const so_c32 *x, const int *off, const int *len, int B,
                    unsigned tsc, float thresh, unsigned char *ok, so_c32 *amp, float *toa,
                    float *soft, int nsoft, int nthreads);
int so_rach_batch(const so_ctx *c, const so_c32 *x, const int *off, const int *len, int B,
                  float thresh, unsigned char *ok, so_c32 *amp, float *toa,
                  float *soft, int nsoft, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
