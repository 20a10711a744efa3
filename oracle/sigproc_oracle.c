/*
 * oracle/sigproc_oracle.c -- TEST INFRASTRUCTURE ONLY (see sigproc_oracle.h).
 *
 * Plain-C restatement of the reference OpenBTS sigProcLib, written from the
 * behaviour of /root/reference/Transceiver/sigProcLib.cpp (cited per function
 * as "ref:<line>"; "ref52:<line>" = Transceiver52M/sigProcLib.cpp; Complex.h =
 * Transceiver/Complex.h).  The arithmetic is float32 with every product and sum
 * separately rounded, in the reference's evaluation order; build with
 * -ffp-contract=off (oracle/Makefile) so the compiler never fuses a
 * multiply-add -- the x86-64 reference build has no FMA either.
 *
 * Parity status: PINNED -- bit-exact against oracle/_ref (the real reference,
 * tests/test_oracle_vs_ref.py) and against tests/golden/ (captured from it).
 */
#include "sigproc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- GSM bit constants: GSM/GSMCommon.cpp:44-57 ------------------------- */
const char so_training_sequence[8][27] = {
  "00100101110000100010010111", "00101101110111100010110111",
  "01000011101110100100001110", "01000111101101000100011110",
  "00011010111001000001101011", "01001110101100000100111010",
  "10100111110110001010011111", "11101111000100101110111100",
};
const char so_dummy_burst[149] =
  "0001111101101110110000010100100111000001001000100000001111100011100010111000"
  "101110001010111010010100011001100111001111010011111000100101111101010000";
const char so_rach_synch[42] = "01001011011111111001100110101010001111000";

/* ---- Complex<float> semantics: Complex.h:77-100,154-160 ------------------- */
static inline so_c32 C(float r, float i) { so_c32 z; z.r = r; z.i = i; return z; }
/* operator*(Complex): (r*a.r-i*a.i, r*a.i+i*a.r), each op rounded (Complex.h:83) */
static inline so_c32 cmul(so_c32 x, so_c32 a) { return C(x.r * a.r - x.i * a.i, x.r * a.i + x.i * a.r); }
static inline so_c32 cmulr(so_c32 x, float a) { return C(x.r * a, x.i * a); }          /* Complex.h:84 */
static inline so_c32 cadd(so_c32 x, so_c32 a) { return C(x.r + a.r, x.i + a.i); }
static inline so_c32 conjc(so_c32 x) { return C(x.r, -x.i); }
static inline float norm2(so_c32 x) { return x.i * x.i + x.r * x.r; }                 /* Complex.h:119 */
static inline so_c32 cinv(so_c32 x) { float n = norm2(x); return C(x.r / n, -x.i / n); } /* Complex.h:154-160 */
static inline so_c32 cdiv(so_c32 x, so_c32 a) { return cmul(x, cinv(a)); }            /* Complex.h:85 */
static inline float cabs_(so_c32 x) { return (float)sqrt((double)norm2(x)); }         /* Complex.h:131 */

static const float M_PI_F = (float)M_PI;                      /* ref:43 */
static const float M_2PI_F = (float)(2.0 * M_PI);             /* ref:44 */
#define M_1_2PI_F (1 / M_2PI_F)                               /* ref:45 */

size_t so_ctx_size(void) { return sizeof(so_ctx); }

/* ---- trig tables and lookups: ref:163-212 ---------------------------------- */
static void init_trig(so_ctx *c)
{
  for (int i = 0; i < SO_TABLESIZE + 1; i++) {                /* ref:207-212 */
    c->cosT[i] = (float)cos(2.0 * M_PI * i / SO_TABLESIZE);
    c->sinT[i] = (float)sin(2.0 * M_PI * i / SO_TABLESIZE);
  }
  /* ref reads table[argI+1] with argI==1024 when arg lands exactly on 1.0
     (one past the array); delta is 0 there, so any finite value gives the
     same result.  Keep a zero guard. */
  c->cosT[SO_TABLESIZE + 1] = 0.0f;
  c->sinT[SO_TABLESIZE + 1] = 0.0f;
}

static inline void lookup_arg(float x, int *argI, float *delta, float *iDelta)
{
  float arg = x * M_1_2PI_F;                                  /* ref:165-167 */
  while (arg > 1.0F) arg -= 1.0F;
  while (arg < 0.0F) arg += 1.0F;
  const float argT = arg * ((float)SO_TABLESIZE);             /* ref:169-172 */
  *argI = (int)argT;
  *delta = argT - *argI;
  *iDelta = 1.0F - *delta;
}
float so_cosLookup(const so_ctx *c, float x)                  /* ref:163-174 */
{
  int k; float d, id; lookup_arg(x, &k, &d, &id);
  return id * c->cosT[k] + d * c->cosT[k + 1];
}
float so_sinLookup(const so_ctx *c, float x)                  /* ref:177-188 */
{
  int k; float d, id; lookup_arg(x, &k, &d, &id);
  return id * c->sinT[k] + d * c->sinT[k + 1];
}
so_c32 so_expjLookup(const so_ctx *c, float x)                /* ref:192-204 */
{
  int k; float d, id; lookup_arg(x, &k, &d, &id);
  return C(id * c->cosT[k] + d * c->cosT[k + 1], id * c->sinT[k] + d * c->sinT[k + 1]);
}
float so_sinc(const so_ctx *c, float x)                       /* ref:567-571 */
{
  if ((x >= 0.01F) || (x <= -0.01F)) return so_sinLookup(c, x) / x;
  return 1.0F;
}

static void init_rotation(so_ctx *c)                          /* ref:214-225 */
{
  float phase = 0.0;
  for (int k = 0; k < 157 * c->sps; k++) {
    c->rot[k] = so_expjLookup(c, phase);
    c->rev[k] = so_expjLookup(c, -phase);
    phase += M_PI_F / 2.0F / (float)c->sps;
  }
}

void so_gmsk_rotate(const so_ctx *c, so_c32 *x, int n, int reverse, int real_only) /* ref:232-264 */
{
  const so_c32 *t = reverse ? c->rev : c->rot;
  if (real_only) for (int k = 0; k < n; k++) x[k] = cmulr(t[k], x[k].r);
  else           for (int k = 0; k < n; k++) x[k] = cmul(t[k], x[k]);
}

/* ---- convolve / correlate: ref:267-408, 474-503; CUSTOM: ref52:270-323 -------- */
int so_convolve(const so_c32 *a, int La, const so_c32 *b, int Lb, so_c32 *out,
                int span, int flags, unsigned startIx, unsigned len)
{
  int startIndex; unsigned outSize;
  switch (span) {
    case SO_FULL_SPAN:    startIndex = 0;  outSize = La + Lb - 1; break;
    case SO_OVERLAP_ONLY: startIndex = La; outSize = abs(La - Lb) + 1; break;
    case SO_START_ONLY:   startIndex = 0;  outSize = La; break;
    case SO_WITH_TAIL:    startIndex = Lb; outSize = La; break;
    case SO_NO_DELAY:     startIndex = (Lb % 2) ? Lb / 2 : Lb / 2 - 1; outSize = La; break;
    case SO_CUSTOM:       startIndex = (int)startIx; outSize = len; break;
    default: return -1;
  }
  const int aReal = flags & 1, bReal = flags & 2;
  int stop = startIndex + (int)outSize;
  so_c32 *cp = out;
  if (flags & 4) {                                            /* b->getSymmetry() == ABSSYM: ref:369-398 */
    /* (the reference pairs a[t-j] with a[t-Lb+j] -- one sample further than the mirror image -- and reads the partner
       without an upper bound check in its fourth arm; partners at or beyond a's end count as zero here) */
    const int half = (Lb % 2) ? (Lb + 1) / 2 : Lb / 2;
    for (int t = startIndex; t < stop; t++) {
      so_c32 sum = C(0, 0);
      int ai = t, as = t - Lb;
      for (int j = 0; j < half; j++, ai--, as++) {
        if (ai < 0) break;
        const so_c32 sv = (as >= 0 && as < La) ? a[as >= 0 && as < La ? as : 0] : C(0, 0);
        if (ai == as) sum = cadd(sum, cmul(a[ai], b[j]));
        else if (ai < La && as >= 0) sum = cadd(sum, cmul(cadd(a[ai], sv), b[j]));
        else if (ai < La) sum = cadd(sum, cmul(a[ai], b[j]));
        else if (as >= 0) sum = cadd(sum, cmul(sv, b[j]));
      }
      *cp++ = sum;
    }
    return (int)outSize;
  }
  for (int t = startIndex; t < stop; t++) {                   /* ref:322-366 */
    int ai = t;
    if (aReal && bReal) {
      float sum = 0.0;
      for (int j = 0; j < Lb; j++, ai--) {
        if (ai < 0) break;
        if (ai < La) sum += a[ai].r * b[j].r;
      }
      *cp++ = C(sum, 0.0f);
    } else if (aReal) {
      so_c32 sum = C(0, 0);
      for (int j = 0; j < Lb; j++, ai--) {
        if (ai < 0) break;
        if (ai < La) sum = cadd(sum, cmulr(b[j], a[ai].r));
      }
      *cp++ = sum;
    } else if (bReal) {
      so_c32 sum = C(0, 0);
      for (int j = 0; j < Lb; j++, ai--) {
        if (ai < 0) break;
        if (ai < La) sum = cadd(sum, cmulr(a[ai], b[j].r));
      }
      *cp++ = sum;
    } else {
      so_c32 sum = C(0, 0);
      for (int j = 0; j < Lb; j++, ai--) {
        if (ai < 0) break;
        if (ai < La) sum = cadd(sum, cmul(a[ai], b[j]));
      }
      *cp++ = sum;
    }
  }
  return (int)outSize;
}

int so_correlate(const so_c32 *a, int La, const so_c32 *b, int Lb, so_c32 *out,
                 int span, int flags)                         /* ref:474-503 */
{
  so_c32 *tmp = (so_c32 *)malloc(sizeof(so_c32) * (size_t)Lb);
  for (int k = 0; k < Lb; k++)
    tmp[Lb - 1 - k] = (flags & 2) ? C(b[k].r, 0.0f) : conjc(b[k]);
  int n = so_convolve(a, La, tmp, Lb, out, span, flags, 0, 0);
  free(tmp);
  return n;
}

void so_scale_vector(so_c32 *x, int n, so_c32 s, int real_only) /* ref:713-730 */
{
  if (!real_only) for (int k = 0; k < n; k++) x[k] = cmul(x[k], s);
  else            for (int k = 0; k < n; k++) x[k] = cmulr(s, x[k].r);
}

static float vector_norm2(const so_c32 *x, int n)             /* ref:146-154 */
{
  float e = 0.0;
  for (int k = 0; k < n; k++) e += norm2(x[k]);
  return e;
}
float so_vector_norm2(const so_c32 *x, int n) { return vector_norm2(x, n); }
float so_vector_power(const so_c32 *x, int n) { return vector_norm2(x, n) / (float)(size_t)n; }   /* ref:157-160 */

float so_dB(float x)                                          /* ref:88-114 */
{
  float arg = 1.0F, dB = 0.0F;
  if (x >= 1.0F) return 0.0F;
  if (x <= 0.0F) return -200.0F;
  float prevArg = arg, prevdB = dB, stepSize = 16.0F, dBstepSize = 12.0F;
  while (stepSize > 1.0F) {
    do {
      prevArg = arg; prevdB = dB;
      arg /= stepSize; dB -= dBstepSize;
    } while (arg > x);
    arg = prevArg; dB = prevdB;
    stepSize *= 0.5F; dBstepSize -= 3.0F;
  }
  return ((arg - x) * (dB - 3.0F) + (x - arg * 0.5F) * dB) / (arg - arg * 0.5F);
}

float so_dBinv(float x)                                       /* ref:117-144 */
{
  float arg = 1.0F, dB = 0.0F;
  if (x >= 0.0F) return 1.0F;
  if (x <= -200.0F) return 0.0F;
  float prevArg = arg, prevdB = dB, stepSize = 16.0F, dBstepSize = 12.0F;
  while (stepSize > 1.0F) {
    do {
      prevArg = arg; prevdB = dB;
      arg /= stepSize; dB -= dBstepSize;
    } while (dB > x);
    arg = prevArg; dB = prevdB;
    stepSize *= 0.5F; dBstepSize -= 3.0F;
  }
  return ((dB - x) * (arg * 0.5F) + (x - (dB - 3.0F)) * (arg)) / 3.0F;
}

/* frequencyShift(y, x, freq, startPhase, &finalPhase): ref:432-471; y may be x */
float so_frequency_shift(const so_ctx *c, const so_c32 *x, int n, float freq, float startPhase, int real_only, so_c32 *y)
{
  float phase = startPhase;
  for (int k = 0; k < n; k++) {
    const so_c32 e = so_expjLookup(c, phase);
    y[k] = real_only ? cmulr(e, x[k].r) : cmul(x[k], e);      /* expjLookup(phase)*real() / (*xP)*expjLookup(phase) */
    phase += freq;
  }
  return phase;
}

/* The channeliser's mixer (include/trxsig_frontend.h): frequencyShift's arithmetic, y[k] = x[k] * expjLookup(phase), with the
   phase of raw sample n = n0 + k formed directly, (float)(t - 2 pi floor(t / 2 pi)), t = (double) n * (double) freq -- i.e.
   frequencyShift (ref:432-471) on the one-sample vector {x[k]} with that startPhase (tests/test_oracle_golden.py checks the
   identity).  Not a function of the reference: a convention of this build, restated here for the checker. */
void so_mix_down(const so_ctx *c, const so_c32 *x, int n, long long n0, float freq, so_c32 *y)
{
  for (int k = 0; k < n; k++) {
    const double t = (double)(n0 + k) * (double)freq;
    const double kk = floor(t * 0.15915494309189535);
    const float phase = (float)(t - kk * 6.283185307179586);
    y[k] = cmul(x[k], so_expjLookup(c, phase));
  }
}

void so_add_vector(so_c32 *x, int nx, const so_c32 *y, int ny) /* ref:746-758 */
{
  for (int k = 0; k < nx && k < ny; k++) x[k] = cadd(x[k], y[k]);
}

void so_offset_vector(so_c32 *x, int n, so_c32 offset, int real_only) /* ref:760-777 */
{
  if (!real_only) for (int k = 0; k < n; k++) x[k] = cadd(x[k], offset);          /* *xP += offset */
  else            for (int k = 0; k < n; k++) x[k] = C(offset.r + x[k].r, offset.i); /* xP->real() + offset (Complex.h:233-236) */
}

/* resampleVector(wVector, expFactor, endPoint): ref:1213-1243 AS IT BEHAVES -- the loop never advances its output
   iterator, so every interpolated value lands in element 0 and the rest of the (zero-initialised) vector stays zero.
   Returns the output length, -1 for expFactor < 1 (NULL in the reference). */
int so_resample_vector(const so_c32 *x, int n, float expFactor, so_c32 endPoint, so_c32 *out)
{
  if (expFactor < 1.0) return -1;
  const int nout = (int)ceilf((float)(size_t)n * expFactor);
  for (int k = 0; k < nout; k++) out[k] = C(0, 0);
  float t = 0.0;
  while (nout > 0) {
    const unsigned tLow = (unsigned)floorf(t);
    const unsigned tHigh = tLow + 1;
    if (tLow > (unsigned)n - 1) break;
    if (tHigh > (unsigned)n) break;
    const so_c32 lowPoint = x[tLow];
    const so_c32 highPoint = (tHigh == (unsigned)n) ? endPoint : x[tHigh];
    const so_c32 a = C((float)tHigh - t, 0.0f);
    const so_c32 b = C(t - (float)tLow, 0.0f);
    out[0] = cadd(cmul(a, lowPoint), cmul(b, highPoint));
    t = (float)((double)t + 1.0 / (double)expFactor);
  }
  return nout;
}

/* gaussianNoise(length, variance, mean): ref:618-637; draws from the C library's rand() as the reference does */
void so_gaussian_noise(int length, float variance, so_c32 mean, so_c32 *out)
{
  const float stddev = sqrtf(variance);
  for (int k = 0; k < length; k++) {
    float u1 = (float)rand() / (float)RAND_MAX;
    while (u1 == 0.0) u1 = (float)rand() / (float)RAND_MAX;
    const float u2 = (float)rand() / (float)RAND_MAX;
    const float arg = (float)(2.0 * M_PI * (double)u2);
    /* C++ overload resolution in the reference: cos / sin / log of a float are the float functions */
    const so_c32 e = C(cosf(arg), sinf(arg));
    const so_c32 v = cmulr(C(e.r * stddev, e.i * stddev), sqrtf((float)(-2.0 * (double)logf(u1))));
    out[k] = cadd(mean, v);
  }
}

/* ---- generateGSMPulse(symbolLength=2, sps): ref:411-430 ------------------------ */
static void gen_pulse(so_ctx *c)
{
  int sps = c->sps, n = sps * 2 + 1, center = (n - 1) / 2;
  so_c32 tmp[2 * SO_MAXSPS + 1];
  for (int i = 0; i < n; i++) {
    float arg = (float)(i - center) / (float)sps;
    tmp[i] = C((float)(0.96 * exp(-1.1380 * arg * arg - 0.527 * arg * arg * arg * arg)), 0.0f);
  }
  float avgAbsval = sqrtf(vector_norm2(tmp, n) / sps);
  for (int i = 0; i < n; i++) c->pulse[i] = tmp[i].r / avgAbsval;
  c->pulse_len = n;
}

/* ---- modulateBurst: ref:521-565 ---------------------------------------------- */
int so_modulate(const so_ctx *c, const char *bits, int nbits, const so_c32 *pulse,
                int pulse_len, int pulse_real, int guard, so_c32 *out)
{
  int sps = c->sps, n = sps * (nbits + guard);
  so_c32 *m = (so_c32 *)calloc((size_t)n, sizeof(so_c32));
  for (int i = 0; i < nbits; i++)                             /* ref:547-550 */
    m[i * sps] = C((float)(2.0 * (bits[i] & 0x01) - 1.0), 0.0f);
  so_gmsk_rotate(c, m, n, 0, 1);                              /* ref:554 (realOnly) */
  so_convolve(m, n, pulse, pulse_len, out, SO_NO_DELAY, pulse_real ? 2 : 0, 0, 0); /* ref:559 */
  free(m);
  return n;
}
int so_modulate_gsm(const so_ctx *c, const char *bits, int nbits, int guard, so_c32 *out)
{
  so_c32 p[2 * SO_MAXSPS + 1];
  for (int i = 0; i < c->pulse_len; i++) p[i] = C(c->pulse[i], 0.0f);
  return so_modulate(c, bits, nbits, p, c->pulse_len, 1, guard, out);
}

/* ---- delayVector: ref:573-616 ------------------------------------------------- */
void so_delay_vector(const so_ctx *c, so_c32 *x, int n, float delay)
{
  int intOffset = (int)floor(delay);
  float fracOffset = delay - intOffset;
  so_c32 *shifted = x;
  if (fabs(fracOffset) > 1e-2) {                              /* double compare, ref:582 */
    so_c32 sincv[21];
    for (int i = 0; i < 21; i++)
      sincv[i] = C(so_sinc(c, M_PI_F * (i - 10 - fracOffset)), 0.0f);
    shifted = (so_c32 *)malloc(sizeof(so_c32) * (size_t)n);
    so_convolve(x, n, sincv, 21, shifted, SO_NO_DELAY, 2, 0, 0);
  }
  if (intOffset < 0) {                                        /* ref:597-605 */
    intOffset = -intOffset;
    int w = 0;
    for (int s = intOffset; s < n; s++) x[w++] = shifted[s];
    while (w < n) x[w++] = C(0, 0);
  } else {                                                    /* ref:606-613 */
    int w = n - 1;
    for (int s = n - 1 - intOffset; s >= 0; s--) x[w--] = shifted[s];
    while (w >= 0) x[w--] = C(0, 0);
  }
  if (shifted != x) free(shifted);
}

/* ---- interpolatePoint: ref:639-659 --------------------------------------------- */
so_c32 so_interpolate_point(const so_ctx *c, const so_c32 *x, int n, float ix, int real_only)
{
  int start = (int)(floor(ix) - 10);
  if (start < 0) start = 0;
  int end = (int)(floor(ix) + 11);
  if ((size_t)(unsigned)end > (size_t)n - 1) end = n - 1;    /* ref:646 (unsigned compare) */
  so_c32 p = C(0, 0);
  if (!real_only)
    for (int i = start; i < end; i++) p = cadd(p, cmulr(x[i], so_sinc(c, M_PI_F * (i - ix))));
  else
    for (int i = start; i < end; i++) p.r += x[i].r * so_sinc(c, M_PI_F * (i - ix));
  return p;
}

/* ---- peakDetect: ref:663-711 ---------------------------------------------------- */
so_c32 so_peak_detect(const so_ctx *c, const so_c32 *x, int n, float *peakIndex, float *avgPwr)
{
  so_c32 maxVal = C(0, 0);
  float maxIndex = -1;
  float sumPower = 0.0;
  for (int i = 0; i < n; i++) {
    float p = norm2(x[i]);
    if (p > maxVal.r) { maxVal = C(p, 0.0f); maxIndex = i; }
    sumPower += p;
  }
  float earlyIndex = maxIndex - 1;
  float lateIndex = maxIndex + 1;
  float incr = 0.5;
  while (incr > 1.0 / 1024.0) {
    so_c32 e = so_interpolate_point(c, x, n, earlyIndex, 0);
    so_c32 l = so_interpolate_point(c, x, n, lateIndex, 0);
    if (norm2(e) < norm2(l)) earlyIndex += incr;            /* Complex.h:109-110 */
    else if (norm2(e) > norm2(l)) earlyIndex -= incr;
    else break;
    incr /= 2.0;
    lateIndex = earlyIndex + 2.0;
  }
  maxIndex = earlyIndex + 1.0;
  maxVal = so_interpolate_point(c, x, n, maxIndex, 0);
  if (peakIndex) *peakIndex = maxIndex;
  if (avgPwr) *avgPwr = (sumPower - norm2(maxVal)) / (size_t)(n - 1);
  return maxVal;
}

/* ---- generateMidamble: ref:779-828 (52M: ref52:804-859) --------------------------- */
static int gen_midamble(so_ctx *c, int tsc)
{
  int sps = c->sps;
  char bits[26];
  for (int i = 0; i < 26; i++) bits[i] = so_training_sequence[tsc][i] == '1';
  so_c32 unit = C(1.0f, 0.0f);                                /* emptyPulse, NOT realOnly */
  so_c32 *middle = c->mid[tsc];
  so_c32 midamble[26 * SO_MAXSPS], autocorr[26 * SO_MAXSPS];
  so_modulate(c, bits + 5, 16, &unit, 1, 0, 0, middle);       /* ref:794-797 */
  so_modulate_gsm(c, bits, 26, 0, midamble);                  /* ref:798-801 */
  so_scale_vector(middle, 16 * sps, C(-1.0f, 0.0f), 0);       /* ref:811 */
  so_scale_vector(midamble, 26 * sps, C(0.0f, 1.0f), 0);      /* ref:812 */
  so_correlate(midamble, 26 * sps, middle, 16 * sps, autocorr, SO_NO_DELAY, 0); /* ref:814 */
  c->mid_gain[tsc] = so_peak_detect(c, autocorr, 26 * sps, &c->mid_toa[tsc], NULL);
  if (!c->variant52m) c->mid_toa[tsc] -= 5 * sps;             /* ref:822; commented out at ref52:854 */
  return 0;
}

/* ---- generateRACHSequence: ref:830-857 ---------------------------------------------- */
static void gen_rach(so_ctx *c)
{
  int sps = c->sps;
  char bits[41];
  for (int i = 0; i < 41; i++) bits[i] = so_rach_synch[i] == '1';
  so_c32 autocorr[41 * SO_MAXSPS];
  so_modulate_gsm(c, bits, 41, 0, c->rach);
  so_correlate(c->rach, 41 * sps, c->rach, 41 * sps, autocorr, SO_NO_DELAY, 0);
  c->rach_gain = so_peak_detect(c, autocorr, 41 * sps, &c->rach_toa, NULL);
}

int so_setup(so_ctx *c, int sps, int variant52m)
{
  if (sps < 1 || sps > SO_MAXSPS) return -1;
  memset(c, 0, sizeof(*c));
  c->sps = sps; c->variant52m = variant52m;
  gen_pulse(c);                                               /* Transceiver.cpp:62 */
  init_trig(c); init_rotation(c);                             /* sigProcLibSetup ref:227-230 */
  gen_rach(c);                                                /* Transceiver.cpp:424 */
  for (int t = 0; t < 8; t++) gen_midamble(c, t);             /* Transceiver.cpp:553 */
  return 0;
}

/* ---- energyDetect: ref:916-932 (52M strides 4: ref52:946-963) ------------------------- */
int so_energy_detect(const so_c32 *x, int n, unsigned win, float thresh, float *avgPwr, int variant52m)
{
  float energy = 0.0;
  if (win > (unsigned)n) win = (unsigned)n;
  int step = variant52m ? 4 : 1;
  for (unsigned i = 0; i < win; i++) energy += norm2(x[(size_t)i * step]);
  if (avgPwr) *avgPwr = energy / win;
  return energy / win > thresh * thresh;
}

/* ---- analyzeTrafficBurst: ref:935-1037; 52M windowed form: ref52:966-1076 --------------- */
int so_analyze_traffic(const so_ctx *c, const so_c32 *x, int n, unsigned tsc, float thresh,
                       unsigned maxTOA, so_c32 *amp, float *toa, int reqChan,
                       so_c32 *chan, int *chan_len, float *chan_off, float *peak_to_mean)
{
  int sps = c->sps;
  (void)n;
  so_c32 corr[36 * SO_MAXSPS > 200 ? 36 * SO_MAXSPS : 200];
  int ncorr;
  if (chan_len) *chan_len = 0;
  if (peak_to_mean) *peak_to_mean = 0.0f;
  if (!c->variant52m) {
    ncorr = 36 * sps;                                         /* ref:951-955 */
    so_correlate(x + sps * 56, 36 * sps, c->mid[tsc], 16 * sps, corr, SO_NO_DELAY, 0);
  } else {
    if (maxTOA < 3 * (unsigned)sps) maxTOA = 3 * sps;         /* ref52:983-1000 */
    unsigned spanTOA = maxTOA;
    if (spanTOA < 5 * (unsigned)sps) spanTOA = 5 * sps;
    unsigned startIx = (66 - spanTOA) * sps;
    unsigned endIx = (66 + 16 + spanTOA) * sps;
    unsigned windowLen = endIx - startIx;
    unsigned corrLen = 2 * maxTOA + 1;
    unsigned expectedTOAPeak = (unsigned)round(c->mid_toa[tsc] + (size_t)((16 * sps - 1) / 2));
    so_c32 rc[16 * SO_MAXSPS];
    for (int k = 0; k < 16 * sps; k++) rc[16 * sps - 1 - k] = conjc(c->mid[tsc][k]);
    ncorr = so_convolve(x + startIx, (int)windowLen, rc, 16 * sps, corr, SO_CUSTOM, 0,
                        expectedTOAPeak - maxTOA, corrLen);
  }
  float meanPower;
  *amp = so_peak_detect(c, corr, ncorr, toa, &meanPower);     /* ref:959 */
  float valleyPower = 0.0;
  int peak = (int)rint(*toa);                                 /* ref:961 */
  if ((*toa < 0.0) || (*toa > (float)(size_t)ncorr)) {        /* ref:964-968 */
    *amp = C(0, 0);
    return 0;
  }
  int numRms = 0;
  for (int i = 2 * sps; i <= 5 * sps; i++) {                  /* ref:970-980 */
    if (peak - i >= 0) { valleyPower += norm2(corr[peak - i]); numRms++; }
    if (peak + i < ncorr) { valleyPower += norm2(corr[peak + i]); numRms++; }
  }
  if (numRms < 2) { *amp = C(0, 0); return 0; }               /* ref:982-987 */
  float RMS = (float)(sqrtf(valleyPower / (float)numRms) + 0.00001); /* ref:989 */
  float peakToMean = cabs_(*amp) / RMS;                       /* ref:990 */
  if (peak_to_mean) *peak_to_mean = peakToMean;
  *amp = cdiv(*amp, c->mid_gain[tsc]);                        /* ref:997 */
  float TOAoffset;
  if (!c->variant52m) {
    *toa = (*toa) - c->mid_toa[tsc];                          /* ref:998 */
    *toa = (*toa) - (66 - 56) * sps;                          /* ref:1000 */
    TOAoffset = c->mid_toa[tsc] + (66 - 56) * sps;            /* ref:1006 */
  } else {
    *toa = (*toa) - (maxTOA);                                 /* ref52:1040 */
    TOAoffset = maxTOA;                                       /* ref52:1047 */
  }
  if (reqChan && (peakToMean > thresh)) {                     /* ref:1005-1031 */
    so_delay_vector(c, corr, ncorr, -(*toa));
    int clen = 6 * sps;
    float maxEnergy = -1.0;
    int maxI = -1;
    for (int i = 0; i < 7; i++) {
      if (TOAoffset + (i - 5) * sps + (float)(size_t)clen > (float)(size_t)ncorr) continue;
      if (TOAoffset + (i - 5) * sps < 0) continue;
      float energy = vector_norm2(corr + (int)floor(TOAoffset + (i - 5) * sps), clen);
      if (energy > 0.95 * maxEnergy) { maxI = i; maxEnergy = energy; }
    }
    int st = (int)floor(TOAoffset + (maxI - 5) * sps);
    for (int k = 0; k < clen; k++) chan[k] = corr[st + k];
    so_scale_vector(chan, clen, cdiv(C(1.0f, 0.0f), c->mid_gain[tsc]), 0); /* ref:1025 */
    if (chan_len) *chan_len = clen;
    if (chan_off) *chan_off = 5 * sps - maxI;                 /* ref:1029 */
  }
  return peakToMean > thresh;
}

/* ---- detectRACHBurst: ref:860-914 ---------------------------------------------------------- */
int so_detect_rach(const so_ctx *c, const so_c32 *x, int n, float thresh, so_c32 *amp,
                   float *toa, float *peak_to_mean)
{
  int sps = c->sps;
  so_c32 *corr = (so_c32 *)malloc(sizeof(so_c32) * (size_t)n);
  so_correlate(x, n, c->rach, 41 * sps, corr, SO_NO_DELAY, 0);
  if (peak_to_mean) *peak_to_mean = 0.0f;
  float meanPower;
  so_c32 peakAmpl = so_peak_detect(c, corr, n, toa, &meanPower);
  float valleyPower = 0.0;
  if ((*toa < 0.0) || (*toa > (float)(size_t)n)) {            /* ref:878-882 */
    free(corr); *amp = C(0, 0); return 0;
  }
  int peak = (int)rint(*toa);
  float numSamples = 0.0;
  for (int i = 57 * sps; i <= 107 * sps; i++) {               /* ref:888-893 */
    if (peak + i >= n) break;
    valleyPower += norm2(corr[peak + i]);
    numSamples++;
  }
  if (numSamples < 2) { free(corr); *amp = C(0, 0); return 0; }
  float RMS = (float)(sqrtf(valleyPower / (float)numSamples) + 0.00001);
  float peakToMean = cabs_(peakAmpl) / RMS;
  if (peak_to_mean) *peak_to_mean = peakToMean;
  *amp = cdiv(peakAmpl, c->rach_gain);                        /* ref:905 */
  *toa = (*toa) - c->rach_toa - 8 * sps;                      /* ref:907 */
  free(corr);
  return peakToMean > thresh;
}

/* ---- decimateVector + vectorSlicer + demodulateBurst: ref:1039-1097, 507-519 ------------------ */
int so_demodulate(const so_ctx *c, const so_c32 *x, int n, so_c32 amp, float toa, float *soft)
{
  int sps = c->sps;
  so_c32 *d = (so_c32 *)malloc(sizeof(so_c32) * (size_t)n);
  memcpy(d, x, sizeof(so_c32) * (size_t)n);
  so_scale_vector(d, n, cdiv(C(1.0f, 0.0f), amp), 0);         /* ref:1066 */
  so_delay_vector(c, d, n, -toa);                             /* ref:1068 */
  so_gmsk_rotate(c, d, n, 1, 0);                              /* ref:1073 */
  int ns = n;
  if (sps > 1) {                                              /* ref:1076-1080, 1045-1050 */
    ns = n / sps;
    for (int k = 0; k < ns; k++) d[k] = d[k * sps];           /* n must be a multiple of sps */
  }
  for (int k = 0; k < ns; k++) {                              /* ref:507-519 */
    float v = (float)(0.5 * (d[k].r + 1.0F));
    if (v > 1.0) v = 1.0f;
    if (v < 0.0) v = 0.0f;
    soft[k] = v;
  }
  free(d);
  return ns;
}

/* ---- createLPF (table normalisation only): ref:1119-1148 --------------------------------------- */
void so_create_lpf(const float *raw, int len, float gainDC, float *out)
{
  double sum = 0.0;
  for (int i = 0; i < len; i++) sum += raw[i];
  float normFactor = (float)(gainDC / sum);
  for (int i = 0; i < len; i++) out[i] = raw[i] * normFactor;
}

/* ---- polyphaseResampleVector: ref:1157-1210 ------------------------------------------------------ */
int so_polyphase_resample(const so_c32 *x, int n, int P, int Q, const float *lpf, int L, so_c32 *out)
{
  int nout = (int)ceil(n * (float)P / (float)Q);              /* ref:1171 */
  int outputIx = (L - 1) / 2 / Q;                             /* ref:1177 */
  for (int o = 0; o < nout; o++, outputIx++) {
    int branch = (outputIx * Q) % P;
    int inOff = (outputIx * Q - branch) / P;
    int fi = branch;
    while (inOff >= n) { inOff--; fi += P; }                  /* ref:1183-1186 */
    so_c32 sum = C(0, 0);
    while ((inOff >= 0) && (fi < L)) {                        /* ref:1196-1200 (real LPF) */
      sum = cadd(sum, cmulr(x[inOff], lpf[fi]));
      inOff--; fi += P;
    }
    out[o] = sum;
  }
  return nout;
}

/* ---- designDFE: ref:1246-1340 ---------------------------------------------------------------------- */
int so_design_dfe(const so_ctx *c, const so_c32 *chan, int nchan, float snr, int Nf, so_c32 *w, so_c32 *b)
{
  enum { MAXNF = 32 };
  if (Nf > MAXNF || nchan > Nf || nchan < 1) return -1;
  so_c32 G0[MAXNF], G1[MAXNF], L[MAXNF][2 * MAXNF];
  memset(G0, 0, sizeof(G0)); memset(G1, 0, sizeof(G1)); memset(L, 0, sizeof(L));
  int nu = nchan - 1;
  G0[0] = C((float)(1.0 / sqrtf(snr)), 0.0f);                 /* ref:1261 */
  for (int j = 0; j <= nu; j++) G1[j] = conjc(chan[j]);
  float d = 0;
  for (int i = 0; i < Nf; i++) {
    d = norm2(G0[0]) + norm2(G1[0]);                          /* ref:1272 */
    for (int k = 0; k < Nf && (i + k) < Nf + nu; k++)         /* ref:1276-1281 */
    {
      so_c32 t = cadd(cmul(G0[k], conjc(G0[0])), cmul(G1[k], conjc(G1[0])));
      L[i][i + k] = C(t.r / d, t.i / d);
    }
    so_c32 kk = cdiv(G1[0], G0[0]);                           /* ref:1282 */
    if (i != Nf - 1) {
      so_c32 G0n[MAXNF], G1n[MAXNF];
      memcpy(G0n, G1, sizeof(so_c32) * (size_t)Nf);
      so_scale_vector(G0n, Nf, conjc(kk), 0);                  /* ref:1286 */
      for (int q = 0; q < Nf; q++) G0n[q] = cadd(G0n[q], G0[q]);
      memcpy(G1n, G0, sizeof(so_c32) * (size_t)Nf);
      so_scale_vector(G1n, Nf, cmulr(kk, -1.0f), 0);           /* ref:1290 */
      for (int q = 0; q < Nf; q++) G1n[q] = cadd(G1n[q], G1[q]);
      so_delay_vector(c, G1n, Nf, -1.0f);                     /* ref:1292 */
      so_c32 s = C((float)(1.0 / sqrtf((float)(1.0 + norm2(kk)))), 0.0f); /* ref:1294-1295 */
      so_scale_vector(G0n, Nf, s, 0);
      so_scale_vector(G1n, Nf, s, 0);
      memcpy(G0, G0n, sizeof(so_c32) * (size_t)Nf);
      memcpy(G1, G1n, sizeof(so_c32) * (size_t)Nf);
    }
  }
  for (int j = 0; j < nu; j++) b[j] = L[Nf - 1][Nf + j];      /* ref:1301-1304 */
  so_scale_vector(b, nu, C(-1.0f, 0.0f), 0);
  for (int j = 0; j < nu; j++) b[j] = conjc(b[j]);
  so_c32 v[MAXNF];
  memset(v, 0, sizeof(v));
  v[Nf - 1] = C(1.0f, 0.0f);
  for (int k = Nf - 2; k >= 0; k--) {                         /* ref:1310-1319 */
    so_c32 vk = C(0, 0);
    for (int j = k + 1; j < Nf; j++) {
      so_c32 p = cmul(v[j], L[k][j]);
      vk.r -= p.r; vk.i -= p.i;
    }
    v[k] = vk;
  }
  for (int i = 0; i < Nf; i++) {                              /* ref:1323-1335 */
    so_c32 wi = C(0, 0);
    int endPt = (nu < (Nf - 1 - i)) ? nu : (Nf - 1 - i);
    for (int k = 0; k < endPt + 1; k++) wi = cadd(wi, cmul(v[i + k], conjc(chan[k])));
    w[i] = C(wi.r / d, wi.i / d);
  }
  return nu;
}

/* ---- equalizeBurst: ref:1343-1399 ---------------------------------------------------------------------- */
int so_equalize(const so_ctx *c, const so_c32 *xin, int n, float toa, const so_c32 *w, int nw,
                const so_c32 *b, int nb, float *soft)
{
  so_c32 *x = (so_c32 *)malloc(sizeof(so_c32) * (size_t)n);
  memcpy(x, xin, sizeof(so_c32) * (size_t)n);
  so_delay_vector(c, x, n, -toa);
  so_c32 *full = (so_c32 *)malloc(sizeof(so_c32) * (size_t)(n + nw - 1));
  so_convolve(x, n, w, nw, full, SO_FULL_SPAN, 0, 0, 0);
  so_c32 *d = full + (nw - 1);                                /* ref:1354-1356 */
  for (int k = 0; k < n; k++) {                               /* ref:1367-1384 */
    for (int j = 0; j < nb && (k - 1 - j) >= 0; j++)
      d[k] = cadd(d[k], cmul(b[j], d[k - 1 - j]));
    d[k] = cmul(d[k], c->rev[k]);
    float re = d[k].r;                                        /* DFE output (pre-decision) */
    d[k] = C((re > 0.0) ? 1.0f : -1.0f, 0.0f);
    d[k] = cmul(d[k], c->rot[k]);
    float v = (float)(0.5 * (re + 1.0F));                     /* vectorSlicer ref:507-519 */
    if (v > 1.0) v = 1.0f;
    if (v < 0.0) v = 0.0f;
    soft[k] = v;
  }
  free(full); free(x);
  return n;
}

/* ---- batched loops (the Transceiver::pullRadioVector per-burst sequence, stateless:
        Transceiver.cpp:298-396 with mEnergyThreshold fixed and demodulateBurst on success) ------- */
int so_normal_batch(const so_ctx *c, const so_c32 *x, const int *off, const int *len, int B,
                    unsigned tsc, float thresh, unsigned char *ok, so_c32 *amp, float *toa,
                    float *soft, int nsoft, int nthreads)
{
  int found = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) reduction(+ : found) schedule(static)
#endif
  for (int i = 0; i < B; i++) {
    float s[160];
    int d = so_analyze_traffic(c, x + off[i], len[i], tsc, thresh, 4, &amp[i], &toa[i], 0,
                               NULL, NULL, NULL, NULL);
    ok[i] = (unsigned char)d;
    if (d) {
      so_demodulate(c, x + off[i], len[i], amp[i], toa[i], s);
      memcpy(soft + (size_t)i * nsoft, s, sizeof(float) * (size_t)nsoft);
      found++;
    } else {
      memset(soft + (size_t)i * nsoft, 0, sizeof(float) * (size_t)nsoft);
    }
  }
  (void)nthreads;
  return found;
}

int so_rach_batch(const so_ctx *c, const so_c32 *x, const int *off, const int *len, int B,
                  float thresh, unsigned char *ok, so_c32 *amp, float *toa,
                  float *soft, int nsoft, int nthreads)
{
  int found = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) reduction(+ : found) schedule(static)
#endif
  for (int i = 0; i < B; i++) {
    float s[160];
    int d = so_detect_rach(c, x + off[i], len[i], thresh, &amp[i], &toa[i], NULL);
    ok[i] = (unsigned char)d;
    if (d) {
      so_demodulate(c, x + off[i], len[i], amp[i], toa[i], s);
      memcpy(soft + (size_t)i * nsoft, s, sizeof(float) * (size_t)nsoft);
      found++;
    } else {
      memset(soft + (size_t)i * nsoft, 0, sizeof(float) * (size_t)nsoft);
    }
  }
  (void)nthreads;
  return found;
}
