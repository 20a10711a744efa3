/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" driver around the *unmodified* reference sigProcLib
 * (compiled in place from /root/reference by oracle/Makefile into
 * oracle/_ref/).  Nothing here restates an algorithm: every entry point
 * marshals plain buffers into the reference's signalVector / BitVector /
 * SoftVector types, calls the reference function named in its comment and
 * copies the result back out.  It exists so that
 *   - oracle/sigproc_oracle.c (the CPU restatement) can be validated
 *     bit-for-bit against the real reference, and
 *   - tests/golden/ fixtures can be generated (oracle/gen_golden.py).
 *
 * Built only in the build container (the reference does not travel to the
 * GPU box).  -DREF_52M selects the Transceiver52M/ signatures.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.
 */
#include "sigProcLib.h"
#include "GSMCommon.h"

#include <cstring>
#include <cstdlib>
#include <cmath>

using namespace GSM;

/* Globals of the reference library (non-static in sigProcLib.cpp:39-59). */
#ifdef REF_52M
typedef struct {
  signalVector *sequence;
  signalVector *sequenceReversedConjugated;
  float TOA;
  complex gain;
} CorrelationSequence;
#else
typedef struct {
  signalVector *sequence;
  float TOA;
  complex gain;
} CorrelationSequence;
#endif
extern CorrelationSequence *gMidambles[];
extern CorrelationSequence *gRACHSequence;
extern signalVector *GMSKRotation;
extern signalVector *GMSKReverseRotation;
extern float cosTable[];
extern float sinTable[];
extern float rcvLPF_651[];
extern float sendLPF_961[];

/* functions defined in sigProcLib.cpp but not declared in sigProcLib.h */
float sinLookup(const float x);
float cosLookup(const float x);
complex expjLookup(float x);
void GMSKRotate(signalVector &x);
void GMSKReverseRotate(signalVector &x);
void conjugateVector(signalVector &x);

static signalVector *gPulse = NULL;
static int gSps = 0;

static void put(const signalVector *v, float *out)
{
  memcpy(out, v->begin(), v->size() * sizeof(complex));
}

static signalVector *mk(const float *in, int n)
{
  signalVector *v = new signalVector(n);
  memcpy(v->begin(), in, n * sizeof(complex));
  return v;
}

extern "C" {

/* sigProcLibSetup + generateGSMPulse(2,sps) + generateRACHSequence +
   generateMidamble(0..7): the init sequence of Transceiver.cpp:62-64,424,553 */
int ref_setup(int sps)
{
  if (gPulse) { delete gPulse; gPulse = NULL; sigProcLibDestroy(); }
  gSps = sps;
  gPulse = generateGSMPulse(2, sps);
  sigProcLibSetup(sps);
  generateRACHSequence(*gPulse, sps);
  for (int t = 0; t < 8; t++)
    if (!generateMidamble(*gPulse, sps, t)) return -1;
  return 0;
}

int ref_sps(void) { return gSps; }

/* --- table dumps ------------------------------------------------------ */
void ref_get_trig_tables(float *cosT, float *sinT) /* 1025 each */
{
  memcpy(cosT, cosTable, 1025 * sizeof(float));
  memcpy(sinT, sinTable, 1025 * sizeof(float));
}
int ref_get_rotation(float *rot, float *rev) /* 157*sps complex each */
{
  put(GMSKRotation, rot);
  put(GMSKReverseRotation, rev);
  return (int)GMSKRotation->size();
}
int ref_get_pulse(float *out)
{
  put(gPulse, out);
  return (int)gPulse->size();
}
int ref_get_midamble(int tsc, float *seq, float *toa, float *gain)
{
  put(gMidambles[tsc]->sequence, seq);
  *toa = gMidambles[tsc]->TOA;
  gain[0] = gMidambles[tsc]->gain.real();
  gain[1] = gMidambles[tsc]->gain.imag();
  return (int)gMidambles[tsc]->sequence->size();
}
int ref_get_rach(float *seq, float *toa, float *gain)
{
  put(gRACHSequence->sequence, seq);
  *toa = gRACHSequence->TOA;
  gain[0] = gRACHSequence->gain.real();
  gain[1] = gRACHSequence->gain.imag();
  return (int)gRACHSequence->sequence->size();
}
void ref_get_gsm_bits(char *tsc /*8*26*/, char *dummy /*148*/, char *rach /*41*/)
{
  for (int t = 0; t < 8; t++)
    for (int i = 0; i < 26; i++) tsc[t * 26 + i] = gTrainingSequence[t][i] & 1;
  for (int i = 0; i < 148; i++) dummy[i] = gDummyBurst[i] & 1;
  for (int i = 0; i < 41; i++) rach[i] = gRACHSynchSequence[i] & 1;
}
/* raw LPF coefficient tables (rcvLPF_651.h / sendLPF_961.h).  n<=651 / n<=960:
   sendLPF_961[] only holds 960 initialisers. */
void ref_get_lpf_raw(float *rcv651, float *send960)
{
  memcpy(rcv651, rcvLPF_651, 651 * sizeof(float));
  memcpy(send960, sendLPF_961, 960 * sizeof(float));
}

/* --- scalar helpers --------------------------------------------------- */
float ref_sinc(float x) { return sinc(x); }
float ref_sinLookup(float x) { return sinLookup(x); }
float ref_cosLookup(float x) { return cosLookup(x); }
void ref_expjLookup(float x, float *out)
{
  complex c = expjLookup(x);
  out[0] = c.real(); out[1] = c.imag();
}

/* --- vector primitives ------------------------------------------------ */
/* convolve(a,b,NULL,span); flags: bit0 a realOnly, bit1 b realOnly */
int ref_convolve(const float *a, int na, const float *b, int nb, int span,
                 int flags, float *out)
{
  signalVector *A = mk(a, na), *B = mk(b, nb);
  A->isRealOnly(flags & 1); B->isRealOnly(flags & 2);
  if (flags & 4) B->setSymmetry(ABSSYM);
  signalVector *c = convolve(A, B, NULL, (ConvType)span);
  int n = -1;
  if (c) { n = c->size(); put(c, out); delete c; }
  delete A; delete B;
  return n;
}
int ref_correlate(const float *a, int na, const float *b, int nb, int span,
                  int flags, float *out)
{
  signalVector *A = mk(a, na), *B = mk(b, nb);
  A->isRealOnly(flags & 1); B->isRealOnly(flags & 2);
  signalVector *c = correlate(A, B, NULL, (ConvType)span);
  int n = -1;
  if (c) { n = c->size(); put(c, out); delete c; }
  delete A; delete B;
  return n;
}
void ref_delay_vector(float *x, int n, float delay)
{
  signalVector *X = mk(x, n);
  delayVector(*X, delay);
  put(X, x);
  delete X;
}
void ref_interpolate_point(const float *x, int n, float ix, float *out)
{
  signalVector *X = mk(x, n);
  complex p = interpolatePoint(*X, ix);
  out[0] = p.real(); out[1] = p.imag();
  delete X;
}
void ref_peak_detect(const float *x, int n, float *peak, float *idx, float *avg)
{
  signalVector *X = mk(x, n);
  complex p = peakDetect(*X, idx, avg);
  peak[0] = p.real(); peak[1] = p.imag();
  delete X;
}
void ref_scale_vector(float *x, int n, float sr, float si)
{
  signalVector *X = mk(x, n);
  scaleVector(*X, complex(sr, si));
  put(X, x);
  delete X;
}
void ref_gmsk_rotate(float *x, int n, int reverse)
{
  signalVector *X = mk(x, n);
  if (reverse) GMSKReverseRotate(*X); else GMSKRotate(*X);
  put(X, x);
  delete X;
}

/* --- burst-level functions ------------------------------------------- */
/* modulateBurst(bits, gsmPulse, guard, sps) */
int ref_modulate(const char *bits, int nbits, int guard, float *out)
{
  BitVector bv(nbits);
  for (int i = 0; i < nbits; i++) bv[i] = bits[i];
  signalVector *m = modulateBurst(bv, *gPulse, guard, gSps);
  int n = m->size();
  put(m, out);
  delete m;
  return n;
}
int ref_energy_detect(const float *x, int n, unsigned win, float thresh, float *avgPwr)
{
  signalVector *X = mk(x, n);
  bool ok = energyDetect(*X, win, thresh, avgPwr);
  delete X;
  return ok;
}
/* analyzeTrafficBurst.  chan must hold 6*sps complex; *chanLen = 0 when the
   reference did not allocate a channel response. */
int ref_analyze_traffic(const float *x, int n, unsigned tsc, float thresh,
                        int maxTOA /* 52M only */, float *amp, float *toa,
                        int reqChan, float *chan, int *chanLen, float *chanOff)
{
  signalVector *X = mk(x, n);
  complex a = 0.0; float t = 0.0; float off = 0.0;
  signalVector *cr = NULL;
#ifdef REF_52M
  bool ok = analyzeTrafficBurst(*X, tsc, thresh, gSps, &a, &t, (unsigned)maxTOA,
                                reqChan, &cr, &off);
#else
  (void)maxTOA;
  bool ok = analyzeTrafficBurst(*X, tsc, thresh, gSps, &a, &t, reqChan, &cr, &off);
#endif
  amp[0] = a.real(); amp[1] = a.imag(); *toa = t;
  if (chanLen) *chanLen = 0;
  if (reqChan && ok && cr) {
    put(cr, chan); *chanLen = cr->size(); *chanOff = off;
    delete cr;
  }
  delete X;
  return ok;
}
int ref_detect_rach(const float *x, int n, float thresh, float *amp, float *toa)
{
  signalVector *X = mk(x, n);
  complex a = 0.0; float t = 0.0;
  bool ok = detectRACHBurst(*X, thresh, gSps, &a, &t);
  amp[0] = a.real(); amp[1] = a.imag(); *toa = t;
  delete X;
  return ok;
}
int ref_demodulate(const float *x, int n, float ar, float ai, float toa, float *soft)
{
  signalVector *X = mk(x, n);
  SoftVector *s = demodulateBurst(*X, *gPulse, gSps, complex(ar, ai), toa);
  int ns = s->size();
  for (int i = 0; i < ns; i++) soft[i] = (*s)[i];
  delete s; delete X;
  return ns;
}

/* --- resampler -------------------------------------------------------- */
/* createLPF(cutoff, 651, gainDC) -- the only OOB-free length */
int ref_create_lpf651(float gainDC, float *out)
{
  signalVector *l = createLPF(0.0, 651, gainDC);
  for (int i = 0; i < 651; i++) out[i] = (*l)[i].real();
  delete l;
  return 651;
}
int ref_polyphase_resample(const float *x, int n, int P, int Q,
                           const float *lpf, int L, float *out)
{
  signalVector *X = mk(x, n);
  signalVector *F = new signalVector(L);
  F->isRealOnly(true);
  for (int i = 0; i < L; i++) (*F)[i] = complex(lpf[i], 0.0);
  signalVector *r = polyphaseResampleVector(*X, P, Q, F);
  int m = r->size();
  put(r, out);
  delete r; delete F; delete X;
  return m;
}

/* --- DFE ---------------------------------------------------------------- */
int ref_design_dfe(const float *chan, int nchan, float snr, int Nf,
                   float *w, float *b)
{
  signalVector *C = mk(chan, nchan);
  signalVector *W = NULL, *B = NULL;
  bool ok = designDFE(*C, snr, Nf, &W, &B);
  if (ok) { put(W, w); put(B, b); }
  int nb = B ? (int)B->size() : 0;
  delete W; delete B; delete C;
  return ok ? nb : -1;
}
int ref_equalize(const float *x, int n, float toa, const float *w, int nw,
                 const float *b, int nb, float *soft)
{
  signalVector *X = mk(x, n), *W = mk(w, nw), *B = mk(b, nb);
  SoftVector *s = equalizeBurst(*X, toa, gSps, *W, *B);
  int ns = s->size();
  for (int i = 0; i < ns; i++) soft[i] = (*s)[i];
  delete s; delete X; delete W; delete B;
  return ns;
}

/* single-thread timing loop used for the cpu_baseline cross-check in the build
   container: analyzeTrafficBurst + demodulateBurst over a packed batch. */
int ref_normal_batch(const float *x, const int *off, const int *len, int B,
                     unsigned tsc, float thresh, unsigned char *ok, float *amp,
                     float *toa, float *soft /* B*148 */)
{
  int found = 0;
  for (int i = 0; i < B; i++) {
    signalVector X((complex *)(x + 2 * (size_t)off[i]), 0, len[i]);
    complex a = 0.0; float t = 0.0;
#ifdef REF_52M
    bool d = analyzeTrafficBurst(X, tsc, thresh, gSps, &a, &t, 4);
#else
    bool d = analyzeTrafficBurst(X, tsc, thresh, gSps, &a, &t);
#endif
    ok[i] = d; amp[2 * i] = a.real(); amp[2 * i + 1] = a.imag(); toa[i] = t;
    if (d) {
#ifdef REF_52M
      signalVector Y(X);
      SoftVector *s = demodulateBurst(Y, *gPulse, gSps, a, t);
#else
      SoftVector *s = demodulateBurst(X, *gPulse, gSps, a, t);
#endif
      for (int k = 0; k < 148; k++) soft[(size_t)i * 148 + k] = (*s)[k];
      delete s;
      found++;
    } else {
      for (int k = 0; k < 148; k++) soft[(size_t)i * 148 + k] = 0.0f;
    }
  }
  return found;
}
/* the equalised receive leg over a packed batch, as Transceiver::pullRadioVector strings the calls together
   (Transceiver.cpp:298, 331-349, 391-396; Transceiver52M: the same with maxTOA): energyDetect -> analyzeTrafficBurst with
   the channel response -> SNR, scaleVector(chan, 1/amp), designDFE(Nf 7) -> scaleVector(burst, 1/amp), equalizeBurst(TOA -
   chanRespOffset).  Timing loop of bench.py's config-5 cpu_baseline; soft: B*157 (zeros where nothing came back). */
int ref_eq_batch(const float *x, const int *off, const int *len, int B, unsigned tsc, float detect_thresh, float energy_thresh,
                 int maxTOA, unsigned char *ok, float *soft)
{
  int found = 0;
  for (int i = 0; i < B; i++) {
    signalVector X((complex *)(x + 2 * (size_t)off[i]), 0, len[i]);
    float *so = soft + (size_t)i * 157;
    for (int k = 0; k < 157; k++) so[k] = 0.0f;
    ok[i] = 0;
    float avgPwr = 0.0f;
    if (!energyDetect(X, 20 * gSps, energy_thresh, &avgPwr)) continue;
    complex a = 0.0; float t = 0.0, choff = 0.0;
    signalVector *cr = NULL;
#ifdef REF_52M
    bool d = analyzeTrafficBurst(X, tsc, detect_thresh, gSps, &a, &t, (unsigned)maxTOA, true, &cr, &choff);
#else
    (void)maxTOA;
    bool d = analyzeTrafficBurst(X, tsc, detect_thresh, gSps, &a, &t, true, &cr, &choff);
#endif
    if (!d || !cr) { delete cr; continue; }
    float snr = (float)((double)a.norm2() / ((double)(energy_thresh * energy_thresh) + 1.0));   /* Transceiver.cpp:340 */
    scaleVector(*cr, complex(1.0, 0.0) / a);
    signalVector *W = NULL, *Bq = NULL;
    designDFE(*cr, snr, 7, &W, &Bq);
    signalVector Y(X);
    scaleVector(Y, complex(1.0, 0.0) / a);
    SoftVector *s = equalizeBurst(Y, t - choff, gSps, *W, *Bq);
    int ns = (int)s->size();
    for (int k = 0; k < ns && k < 157; k++) so[k] = (*s)[k];
    delete s; delete W; delete Bq; delete cr;
    ok[i] = 1;
    found++;
  }
  return found;
}
int ref_rach_batch(const float *x, const int *off, const int *len, int B,
                   float thresh, unsigned char *ok, float *amp, float *toa,
                   float *soft /* B*148 */)
{
  int found = 0;
  for (int i = 0; i < B; i++) {
    signalVector X((complex *)(x + 2 * (size_t)off[i]), 0, len[i]);
    complex a = 0.0; float t = 0.0;
    bool d = detectRACHBurst(X, thresh, gSps, &a, &t);
    ok[i] = d; amp[2 * i] = a.real(); amp[2 * i + 1] = a.imag(); toa[i] = t;
    if (d) {
#ifdef REF_52M
      signalVector Y(X);
      SoftVector *s = demodulateBurst(Y, *gPulse, gSps, a, t);
#else
      SoftVector *s = demodulateBurst(X, *gPulse, gSps, a, t);
#endif
      for (int k = 0; k < 148; k++) soft[(size_t)i * 148 + k] = (*s)[k];
      delete s;
      found++;
    } else {
      for (int k = 0; k < 148; k++) soft[(size_t)i * 148 + k] = 0.0f;
    }
  }
  return found;
}

/* GSM::Time arithmetic and ordering (GSM/GSMCommon.h:327-455, GSMCommon.cpp:161-176): what Transceiver's priority queue,
 * stale-burst test and frame differences rest on.  op: 0 a < b, 1 a > b, 2 a == b, 3 a - b (frames), 4 FNDelta(fn1, fn2),
 * 5 a.incTN(step), 6 a.decTN(step), 7 a + step (frames).  Result in *out (ops 0-4) or *out_fn / *out_tn (ops 5-7). */
/* --- the rest of sigProcLib.h's surface (sigProcLib.h:101-111, 149-153, 177, 184-190, 225-226, 352-354) --------- */
float ref_dB(float x) { return dB(x); }
float ref_dBinv(float x) { return dBinv(x); }
float ref_vector_norm2(const float *x, int n) { signalVector *X = mk(x, n); float e = vectorNorm2(*X); delete X; return e; }
float ref_vector_power(const float *x, int n) { signalVector *X = mk(x, n); float e = vectorPower(*X); delete X; return e; }
float ref_frequency_shift(const float *x, int n, float freq, float startPhase, int real_only, float *y)
{
  signalVector *X = mk(x, n);
  X->isRealOnly(real_only);
  float fin = 0;
  signalVector *Y = frequencyShift(NULL, X, freq, startPhase, &fin);
  put(Y, y);
  delete X; delete Y;
  return fin;
}
void ref_add_vector(float *x, int nx, const float *y, int ny)
{
  signalVector *X = mk(x, nx), *Y = mk(y, ny);
  addVector(*X, *Y);
  put(X, x);
  delete X; delete Y;
}
void ref_offset_vector(float *x, int n, float orr, float oi, int real_only)
{
  signalVector *X = mk(x, n);
  X->isRealOnly(real_only);
  offsetVector(*X, complex(orr, oi));
  put(X, x);
  delete X;
}
int ref_resample_vector(const float *x, int n, float expFactor, float er, float ei, float *out)
{
  signalVector *X = mk(x, n);
  signalVector *Y = resampleVector(*X, expFactor, complex(er, ei));
  int m = -1;
  if (Y) { m = Y->size(); put(Y, out); delete Y; }
  delete X;
  return m;
}
void ref_gaussian_noise(unsigned seed, int length, float variance, float mr, float mi, float *out)
{
  srand(seed);
  signalVector *N = gaussianNoise(length, variance, complex(mr, mi));
  put(N, out);
  delete N;
}

int ref_gsm_time(int op, int fn1, int tn1, int fn2, int tn2, int step, int *out, int *out_fn, int *out_tn) {
  GSM::Time a(fn1, tn1), b(fn2, tn2);
  switch (op) {
    case 0: *out = a < b; return 0;
    case 1: *out = a > b; return 0;
    case 2: *out = a == b; return 0;
    case 3: *out = a - b; return 0;
    case 4: *out = GSM::FNDelta(fn1, fn2); return 0;
    case 5: a.incTN((unsigned)step); break;
    case 6: a.decTN((unsigned)step); break;
    case 7: a += step; break;
    case 8: a = a + b; break;                               // Time::operator+(const Time&) (GSMCommon.h:405-410)
    default: return -1;
  }
  *out_fn = a.FN(); *out_tn = (int)a.TN();
  return 0;
}

} /* extern "C" */
