// trxsig_tablegen.cpp -- init-time construction of the constant tables (host, once per context).
//
// The reference builds these at start-up on the CPU as well (sigProcLibSetup, generateGSMPulse,
// generateMidamble, generateRACHSequence: Transceiver/sigProcLib.cpp:207-230, 411-430, 779-857,
// called from Transceiver.cpp:62-64, 424, 553).  They are not on the per-burst path.  To make
// the device results bit-identical to the reference the tables must hold the reference's exact
// float32 values (SURVEY 8a' item 1: the rotation table is NOT i^k, sinc is NOT sin(x)/x), so the
// construction below follows the reference's evaluation order operation by operation.  Build
// with -ffp-contract=off.
#include "trxsig_tablegen.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef trx_c32 cx;
typedef std::vector<cx> cvec;

inline cx mk(float r, float i) { cx z; z.r = r; z.i = i; return z; }
inline cx mul(cx x, cx a) { return mk(x.r * a.r - x.i * a.i, x.r * a.i + x.i * a.r); }   // Complex.h:83
inline cx mulr(cx x, float a) { return mk(x.r * a, x.i * a); }                            // Complex.h:84
inline cx add(cx x, cx a) { return mk(x.r + a.r, x.i + a.i); }
inline cx conj(cx x) { return mk(x.r, -x.i); }
inline float norm2(cx x) { return x.i * x.i + x.r * x.r; }                                // Complex.h:119

const float kPiF = (float)M_PI;                  // sigProcLib.cpp:43
const float k2PiF = (float)(2.0 * M_PI);         // :44
const float kInv2PiF = 1 / k2PiF;                // :45

struct Trig {
  const float *cosT, *sinT;
  // table index + interpolation weights of x: sigProcLib.cpp:165-172
  static void split(float x, int &k, float &d, float &id) {
    float arg = x * kInv2PiF;
    while (arg > 1.0F) arg -= 1.0F;
    while (arg < 0.0F) arg += 1.0F;
    const float argT = arg * (float)TRX_TABLESIZE;
    k = (int)argT;
    d = argT - k;
    id = 1.0F - d;
  }
  float sin(float x) const { int k; float d, id; split(x, k, d, id); return id * sinT[k] + d * sinT[k + 1]; }
  cx expj(float x) const {                                                                 // :192-204
    int k; float d, id; split(x, k, d, id);
    return mk(id * cosT[k] + d * cosT[k + 1], id * sinT[k] + d * sinT[k + 1]);
  }
  float sinc(float x) const {                                                              // :567-571
    if ((x >= 0.01F) || (x <= -0.01F)) return sin(x) / x;
    return 1.0F;
  }
};

// NO_DELAY convolution c[t] = sum_j b[j] a[t+s-j], sequential in j (sigProcLib.cpp:295-300,322-366)
cvec convolveNoDelay(const cvec &a, const cvec &b, bool bReal) {
  const int La = (int)a.size(), Lb = (int)b.size();
  const int s = (Lb % 2) ? Lb / 2 : Lb / 2 - 1;
  cvec c(La);
  for (int t = 0; t < La; t++) {
    cx sum = mk(0, 0);
    for (int j = 0, ai = t + s; j < Lb; j++, ai--) {
      if (ai < 0) break;
      if (ai < La) sum = add(sum, bReal ? mulr(a[ai], b[j].r) : mul(a[ai], b[j]));
    }
    c[t] = sum;
  }
  return c;
}

cvec correlateNoDelay(const cvec &a, const cvec &b) {                                      // :474-503
  cvec t(b.size());
  for (size_t k = 0; k < b.size(); k++) t[b.size() - 1 - k] = conj(b[k]);
  return convolveNoDelay(a, t, false);
}

// modulateBurst (sigProcLib.cpp:521-565) with guard 0
cvec modulate(const TrxTables &T, const char *bits, int nbits, const cvec &pulse, bool pulseReal) {
  const int sps = (int)T.sps, n = sps * nbits;
  cvec m(n, mk(0, 0));
  for (int i = 0; i < nbits; i++) m[i * sps] = mk((float)(2.0 * (bits[i] & 1) - 1.0), 0.0f);
  for (int k = 0; k < n; k++) m[k] = mulr(T.rot[k], m[k].r);            // GMSKRotate, realOnly (:235-239)
  return convolveNoDelay(m, pulse, pulseReal);
}

cx interpolate(const Trig &tr, const cvec &x, float ix) {                                  // :639-659
  int start = (int)(std::floor(ix) - 10);
  if (start < 0) start = 0;
  int end = (int)(std::floor(ix) + 11);
  if ((size_t)(unsigned)end > x.size() - 1) end = (int)x.size() - 1;
  cx p = mk(0, 0);
  for (int i = start; i < end; i++) p = add(p, mulr(x[i], tr.sinc(kPiF * (i - ix))));
  return p;
}

cx peakDetect(const Trig &tr, const cvec &x, float *peakIndex) {                           // :663-711
  float maxP = 0.0f, maxIndex = -1;
  for (size_t i = 0; i < x.size(); i++) {
    float p = norm2(x[i]);
    if (p > maxP) { maxP = p; maxIndex = (float)i; }
  }
  float early = maxIndex - 1, late = maxIndex + 1, incr = 0.5;
  while (incr > 1.0 / 1024.0) {
    cx e = interpolate(tr, x, early), l = interpolate(tr, x, late);
    if (norm2(e) < norm2(l)) early += incr;
    else if (norm2(e) > norm2(l)) early -= incr;
    else break;
    incr /= 2.0;
    late = early + 2.0;
  }
  maxIndex = early + 1.0;
  *peakIndex = maxIndex;
  return interpolate(tr, x, maxIndex);
}

const char *kTSC[8] = {                                                  // GSM/GSMCommon.cpp:44-53
  "00100101110000100010010111", "00101101110111100010110111", "01000011101110100100001110",
  "01000111101101000100011110", "00011010111001000001101011", "01001110101100000100111010",
  "10100111110110001010011111", "11101111000100101110111100" };
const char *kRACH = "01001011011111111001100110101010001111000";       // GSM/GSMCommon.cpp:57

uint32_t fnv1a(const unsigned char *p, size_t n) {
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}

}  // namespace

uint32_t trx_tables_checksum(const TrxTables *T) {
  const unsigned char *p = (const unsigned char *)T;
  const size_t skip = offsetof(TrxTables, pad0);
  return fnv1a(p + skip, sizeof(TrxTables) - skip);
}

bool trx_tables_valid(const TrxTables *T) {
  return T->magic == TRX_MAGIC && T->version == TRX_BLOB_VERSION && T->bytes == sizeof(TrxTables) &&
         (T->sps == 1 || T->sps == 2 || T->sps == 4) && T->checksum == trx_tables_checksum(T);
}

int trx_build_tables(TrxTables *T, int sps) {
  if (!(sps == 1 || sps == 2 || sps == 4)) return -1;
  std::memset(T, 0, sizeof(*T));
  T->magic = TRX_MAGIC; T->version = TRX_BLOB_VERSION; T->sps = (uint32_t)sps; T->bytes = sizeof(TrxTables);

  for (int i = 0; i < TRX_TABLESIZE + 1; i++) {                          // initTrigTables :207-212
    T->cosT[i] = (float)std::cos(2.0 * M_PI * i / TRX_TABLESIZE);
    T->sinT[i] = (float)std::sin(2.0 * M_PI * i / TRX_TABLESIZE);
  }
  Trig tr = { T->cosT, T->sinT };

  float phase = 0.0;                                                     // initGMSKRotationTables :214-225
  for (int k = 0; k < 157 * sps; k++) {
    T->rot[k] = tr.expj(phase);
    T->rev[k] = tr.expj(-phase);
    phase += kPiF / 2.0F / (float)sps;
  }

  {                                                                      // generateGSMPulse(2,sps) :411-430
    const int n = sps * 2 + 1, center = (n - 1) / 2;
    float v[2 * TRX_MAXSPS + 1], e = 0.0;
    for (int i = 0; i < n; i++) {
      float arg = (float)(i - center) / (float)sps;
      v[i] = (float)(0.96 * std::exp(-1.1380 * arg * arg - 0.527 * arg * arg * arg * arg));
    }
    for (int i = 0; i < n; i++) e += 0.0f * 0.0f + v[i] * v[i];            // norm2 of (v,0): i*i + r*r
    float avgAbsval = sqrtf(e / sps);
    for (int i = 0; i < n; i++) T->pulse[i] = v[i] / avgAbsval;
  }
  cvec pulse(2 * sps + 1), unit(1, mk(1.0f, 0.0f));
  for (int i = 0; i < 2 * sps + 1; i++) pulse[i] = mk(T->pulse[i], 0.0f);

  {                                                                      // generateRACHSequence :830-857
    cvec seq = modulate(*T, kRACH, 41, pulse, true);
    cvec ac = correlateNoDelay(seq, seq);
    T->rach_gain = peakDetect(tr, ac, &T->rach_toa);
    std::memcpy(T->rach, seq.data(), seq.size() * sizeof(cx));
  }
  for (int t = 0; t < 8; t++) {                                          // generateMidamble :779-828
    cvec middle = modulate(*T, kTSC[t] + 5, 16, unit, false);
    cvec full = modulate(*T, kTSC[t], 26, pulse, true);
    for (auto &z : middle) z = mul(z, mk(-1.0f, 0.0f));
    for (auto &z : full) z = mul(z, mk(0.0f, 1.0f));
    cvec ac = correlateNoDelay(full, middle);
    T->mid_gain[t] = peakDetect(tr, ac, &T->mid_toa[t]);
    T->mid_toa[t] -= 5 * sps;
    std::memcpy(T->mid[t], middle.data(), middle.size() * sizeof(cx));
    for (int k = 0; k < 16; k++) T->mid_ctap[t][k] = conj(middle[(size_t)sps * k]);
  }
  for (int f = 0; f < 512; f++)
    for (int j = 0; j < 21; j++) {
      // (i - ix) with ix = I + f/512, i = I - 10 + j: exact in float32
      float d = (float)(j - 10) - (float)f / 512.0f;
      T->sinc_grid[f][j] = tr.sinc(kPiF * d);
    }
  T->checksum = trx_tables_checksum(T);
  return 0;
}

// the 26 training-sequence bits of TSC 0..7 as '0'/'1' characters (GSM 05.02 5.2.3; GSM/GSMCommon.cpp:44-53)
const char *trx_training_sequence(int tsc) { return (tsc >= 0 && tsc < 8) ? kTSC[tsc] : nullptr; }
