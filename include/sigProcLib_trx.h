// sigProcLib_trx.h -- source-compatible C++ facade over libtrxsig for code written against the
// reference's Transceiver/sigProcLib.h (OpenBTS).  Header only; link with -ltrxsig.
//
// It re-creates the names of that header -- `complex`, `Vector`, `signalVector` (owning or aliasing, with the
// concatenating constructor, segment(), copyToSegment(), ...), `BitVector`, `SoftVector`, `Symmetry`, `ConvType`, and the
// free functions sigProcLibSetup / sigProcLibDestroy, convolve, correlate, generateGSMPulse, vectorSlicer, modulateBurst,
// delayVector, interpolatePoint, peakDetect, scaleVector, generateMidamble, generateRACHSequence, energyDetect,
// detectRACHBurst, analyzeTrafficBurst, decimateVector, demodulateBurst, createLPF (with setLPFTables),
// polyphaseResampleVector, designDFE, equalizeBurst, plus GMSKRotate / GMSKReverseRotate (sigProcLib.cpp:232-264) -- with
// the reference's argument meaning, ownership (functions returning a pointer allocate with `new`, the caller deletes:
// Transceiver.cpp:112,407,672) and error behaviour (NULL / false, amplitude set to 0 on a "bogus result":
// sigProcLib.cpp:878-882, 964-968).  The line cited at each function is its declaration in Transceiver/sigProcLib.h.
// Define TRXFACADE_52M before including this header for the Transceiver52M/sigProcLib.h signatures instead
// (analyzeTrafficBurst with maxTOA :295-305, the in-place demodulateBurst :324-328; symbol-rate samples).
//
// Every call that processes samples runs on the GPU through the C-ABI's host-buffer entry points (one PCIe round trip
// per call); nothing here computes on the CPU.  A real deployment batches instead -- see INTEGRATION.md.
//
// The functions no caller on the path uses are here too, for a complete surface: dB, dBinv, vectorNorm2, vectorPower,
// frequencyShift, sinc, addVector, gaussianNoise (the C library's rand(), in the reference's draw order), offsetVector,
// resampleVector (as the reference behaves) and convolve with an ABSSYM filter.
// Accepted ranges narrower than the reference's: modulateBurst takes 148-bit bursts with guard 0..9 and the library's
// own pulse; designDFE wants Nf = 7 and a 6-tap channel, equalizeBurst 7 + 5 taps;
// demodulateBurst wants 92..157 symbols (a multiple of sps samples) and |TOA| <= 4096 -- outside them the call returns
// NULL / false instead of a value.
#ifndef SIGPROCLIB_TRX_H
#define SIGPROCLIB_TRX_H

#include <cstddef>
#include <cstring>
#include <vector>

#include "trxsig.h"
#include "trxsig_transceiver.h"   // trxsig_create_lpf_host

namespace trxfacade {

// Complex<float> as the path uses it (Transceiver/Complex.h:39-151): storage + accessors only.
struct complex {
  float r, i;
  complex(float re = 0.0f, float im = 0.0f) : r(re), i(im) {}
  float real() const { return r; }
  float imag() const { return i; }
  float norm2() const { return i * i + r * r; }
  complex conj() const { return complex(r, -i); }
  complex inv() const { const float n = norm2(); return complex(r / n, -i / n); }             // Complex.h:154-160
  complex operator*(const complex &a) const { return complex(r * a.r - i * a.i, r * a.i + i * a.r); }   // Complex.h:83
  complex operator*(float a) const { return complex(r * a, i * a); }                           // Complex.h:84
  complex operator/(const complex &a) const { return (*this) * a.inv(); }                      // Complex.h:85
  complex operator+(const complex &a) const { return complex(r + a.r, i + a.i); }              // Complex.h:79
  complex operator-(const complex &a) const { return complex(r - a.r, i - a.i); }              // Complex.h:81
  complex &operator+=(const complex &a) { r += a.r; i += a.i; return *this; }                  // Complex.h:162-167
};

// Vector<T> (CommonLibs/Vector.h:42-252): a contiguous block that is either owned (freed on destruction) or an alias
// of somebody else's storage (segment(), the pointer constructors).  As in the reference, copying from a NON-const
// Vector moves the block (the source becomes an alias of it), copying from a const one clones.
template <class T>
class Vector {
 public:
  typedef T *iterator;
  typedef const T *const_iterator;
  Vector(size_t n = 0) : own_(NULL), b_(NULL), e_(NULL) { resize(n); }
  Vector(Vector<T> &o) : own_(o.own_), b_(o.b_), e_(o.e_) { o.own_ = NULL; }                 // Vector.h:98-100
  Vector(const Vector<T> &o) : own_(NULL), b_(NULL), e_(NULL) { clone(o); }                   // :103
  Vector(T *owned, T *first, T *last) : own_(owned), b_(first), e_(last) {}                   // :106-108
  Vector(T *first, size_t span) : own_(NULL), b_(first), e_(first + span) {}                  // :111-113 (not deleted)
  Vector(const Vector<T> &x, const Vector<T> &y) : own_(NULL), b_(NULL), e_(NULL) {           // :116-122 concatenation
    resize(x.size() + y.size());
    for (size_t k = 0; k < x.size(); k++) b_[k] = x.b_[k];
    for (size_t k = 0; k < y.size(); k++) b_[x.size() + k] = y.b_[k];
  }
  ~Vector() { clear(); }
  void operator=(Vector<T> &o) { clear(); own_ = o.own_; b_ = o.b_; e_ = o.e_; o.own_ = NULL; }   // :131-138
  void operator=(const Vector<T> &o) { clone(o); }                                            // :141
  size_t size() const { return (size_t)(e_ - b_); }
  size_t bytes() const { return size() * sizeof(T); }
  void resize(size_t n) {                                                                      // :66-73 (content discarded)
    delete[] own_;
    own_ = n ? new T[n]() : NULL;
    b_ = own_; e_ = b_ + n;
  }
  void clear() { resize(0); }
  void clone(const Vector<T> &o) { resize(o.size()); for (size_t k = 0; k < o.size(); k++) b_[k] = o.b_[k]; }
  Vector<T> segment(size_t start, size_t span) { return Vector<T>(NULL, b_ + start, b_ + start + span); }       // :147-153
  const Vector<T> segment(size_t start, size_t span) const { return Vector<T>(NULL, b_ + start, b_ + start + span); }
  Vector<T> head(size_t span) { return segment(0, span); }
  const Vector<T> head(size_t span) const { return segment(0, span); }
  Vector<T> tail(size_t start) { return segment(start, size() - start); }
  const Vector<T> tail(size_t start) const { return segment(start, size() - start); }
  void copyToSegment(Vector<T> &o, size_t start, size_t span) const { for (size_t k = 0; k < span; k++) o.b_[start + k] = b_[k]; }
  void copyToSegment(Vector<T> &o, size_t start = 0) const { copyToSegment(o, start, size()); }
  void copyTo(Vector<T> &o) const { copyToSegment(o, 0, size()); }
  void segmentCopyTo(Vector<T> &o, size_t start, size_t span) const { for (size_t k = 0; k < span; k++) o.b_[k] = b_[start + k]; }
  void fill(const T &v) { for (T *p = b_; p < e_; p++) *p = v; }
  void fill(const T &v, unsigned start, unsigned length) { for (unsigned k = 0; k < length; k++) b_[start + k] = v; }
  T *begin() { return b_; }
  const T *begin() const { return b_; }
  T *end() { return e_; }
  const T *end() const { return e_; }
  T &operator[](size_t k) { return b_[k]; }
  const T &operator[](size_t k) const { return b_[k]; }
 protected:
  T *own_;   // allocated block, if this Vector owns one
  T *b_;     // first useful element
  T *e_;     // one past the last
};

enum Symmetry { NONE = 0, ABSSYM = 1 };                                                        // sigProcLib.h:34-37
enum ConvType { FULL_SPAN = 0, OVERLAP_ONLY = 1, START_ONLY = 2, WITH_TAIL = 3, NO_DELAY = 4, UNDEFINED = 255 };   // :41-48

// signalVector (sigProcLib.h:51-99)
class signalVector : public Vector<complex> {
 public:
  signalVector(int dSize = 0, Symmetry wSymmetry = NONE) : Vector<complex>((size_t)dSize), symmetry_(wSymmetry), realOnly_(false) {}
  signalVector(complex *wData, size_t start, size_t span, Symmetry wSymmetry = NONE)            // :65-71 an alias
      : Vector<complex>(NULL, wData + start, wData + start + span), symmetry_(wSymmetry), realOnly_(false) {}
  signalVector(const signalVector &vec1, const signalVector &vec2)                              // :73-79 concatenation
      : Vector<complex>(vec1, vec2), symmetry_(vec1.symmetry_), realOnly_(false) {}
  signalVector(const signalVector &wVector)                                                     // :81-88 a copy
      : Vector<complex>(wVector.size()), symmetry_(wVector.symmetry_), realOnly_(false) { wVector.copyTo(*this); }
  Symmetry getSymmetry() const { return symmetry_; }
  void setSymmetry(Symmetry wSymmetry) { symmetry_ = wSymmetry; }
  bool isRealOnly() const { return realOnly_; }
  void isRealOnly(bool v) { realOnly_ = v; }
 private:
  Symmetry symmetry_;
  bool realOnly_;
};

// BitVector: one bit per char, consumers mask with 0x01 (CommonLibs/BitVector.cpp:54-63)
class BitVector : public Vector<char> {
 public:
  BitVector(size_t n = 0) : Vector<char>(n) {}
  BitVector(const char *s) : Vector<char>(std::strlen(s)) {
    for (size_t k = 0; k < size(); k++) (*this)[k] = (s[k] == '1');
  }
  bool bit(size_t k) const { return (*this)[k] & 0x01; }
};

// SoftVector: float 0..1, bit() = > 0.5F (CommonLibs/BitVector.h:415-420)
class SoftVector : public Vector<float> {
 public:
  SoftVector(size_t n = 0) : Vector<float>(n) {}
  bool bit(size_t k) const { return (*this)[k] > 0.5F; }
};

// ---- library state (the reference keeps process globals, sigProcLib.cpp:39-59) ----------------
struct State {
  trxsig_ctx *ctx = nullptr;
  int sps = 0;
  int device = 0;
};
inline State &state() { static State s; return s; }

inline void sigProcLibDestroy(void) {                         // sigProcLib.h:117
  State &s = state();
  if (s.ctx) { trxsig_destroy(s.ctx); s.ctx = nullptr; }
}
// sigProcLibSetup (sigProcLib.h:114; void, as there): also builds the pulse, the 8 midambles and the RACH sequence,
// which the reference creates through separate calls at start-up (Transceiver.cpp:62-64,424,553).  There is no CPU
// fallback: without a gfx950 device the library stays unset, sigProcLibReady() says so and every function below
// returns NULL / false.  sigProcLibSetDevice picks the GPU of the next sigProcLibSetup (default 0).
inline void sigProcLibSetDevice(int device) { state().device = device; }
inline void sigProcLibSetup(int samplesPerSymbol) {
  State &s = state();
  sigProcLibDestroy();
  s.sps = samplesPerSymbol;
  if (trxsig_abi_version() != TRXSIG_ABI_VERSION) { s.ctx = nullptr; return; }   // header and library must agree (trxsig.h)
  if (trxsig_create(&s.ctx, s.device, samplesPerSymbol) != TRXSIG_OK) s.ctx = nullptr;
}
inline bool sigProcLibReady() { return state().ctx != nullptr; }

// generateGSMPulse(2, sps) (sigProcLib.h:137-138; the definition takes (symbolLength, samplesPerSymbol),
// sigProcLib.cpp:411): a copy of the table built at set-up
inline signalVector *generateGSMPulse(int symbolLength, int samplesPerSymbol) {
  State &s = state();
  if (!s.ctx || symbolLength != 2 || samplesPerSymbol != s.sps) return NULL;
  trxsig_tables_view v;
  if (trxsig_tables_view_get(s.ctx, &v) != TRXSIG_OK) return NULL;
  signalVector *p = new signalVector(2 * s.sps + 1);
  for (int k = 0; k < 2 * s.sps + 1; k++) (*p)[k] = complex(v.gsm_pulse[k], 0.0f);
  p->isRealOnly(true);
  return p;
}
// generateMidamble / generateRACHSequence (sigProcLib.h:235-245): built at set-up; these only check.
inline bool generateMidamble(signalVector &, int samplesPerSymbol, int TSC) {
  return state().ctx && samplesPerSymbol == state().sps && TSC >= 0 && TSC <= 7;
}
inline bool generateRACHSequence(signalVector &, int samplesPerSymbol) {
  return state().ctx && samplesPerSymbol == state().sps;
}

// modulateBurst (sigProcLib.h:171-174).  gsmPulse must be the library's pulse (it is the only one
// Transceiver ever passes, Transceiver.cpp:68,105); 148-bit bursts, guard 0..9.
inline signalVector *modulateBurst(const BitVector &wBurst, const signalVector &, int guardPeriodLength,
                                   int samplesPerSymbol) {
  State &s = state();
  if (!s.ctx || samplesPerSymbol != s.sps || wBurst.size() != 148 || guardPeriodLength < 0 || guardPeriodLength > 9)
    return NULL;
  uint8_t bits[148];
  for (int k = 0; k < 148; k++) bits[k] = (uint8_t)wBurst[k];
  const int32_t guard = guardPeriodLength, off = 0;
  const int n = s.sps * (148 + guardPeriodLength);
  signalVector *out = new signalVector(n);
  if (trxsig_modulate_host(s.ctx, bits, &guard, NULL, 1, (trxsig_c32 *)out->begin(), &off, n) != TRXSIG_OK) {
    delete out;
    return NULL;
  }
  return out;
}

namespace detail {
inline bool detect(bool rach, signalVector &rxBurst, unsigned TSC, float thresh, float energyThresh,
                   complex *amplitude, float *TOA, float *avgPwr, bool *energyOk) {
  State &s = state();
  if (!s.ctx) return false;
  const int32_t off = 0, len = (int32_t)rxBurst.size();
  uint8_t flags = 0; trxsig_c32 amp = {0, 0}; float toa = 0, pwr = 0;
  int rc = rach ? trxsig_detect_demod_rach_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), &off, &len, 1, thresh,
                                                energyThresh, &flags, &amp, &toa, &pwr, NULL, 0, 0)
                : trxsig_detect_demod_normal_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), &off, &len, 1, (int)TSC,
                                                  thresh, energyThresh, &flags, &amp, &toa, &pwr, NULL, 0, 0);
  if (rc != TRXSIG_OK) return false;
  if (amplitude) *amplitude = complex(amp.re, amp.im);
  if (TOA) *TOA = toa;
  if (avgPwr) *avgPwr = pwr;
  if (energyOk) *energyOk = (flags & TRXSIG_F_ENERGY) != 0;
  return (flags & TRXSIG_F_DETECT) != 0;
}
}  // namespace detail

// energyDetect (sigProcLib.h:255-258), any window length; with TRXFACADE_52M the 52 MHz variant's stride of four samples
// (Transceiver52M/sigProcLib.cpp:946-963)
inline bool energyDetect(signalVector &rxBurst, unsigned windowLength, float detectThreshold, float *avgPwr = NULL) {
  State &s = state();
  if (!s.ctx || rxBurst.size() == 0) return false;
#ifdef TRXFACADE_52M
  const int step = 4;
#else
  const int step = 1;
#endif
  return trxsig_energy_detect_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), (int)rxBurst.size(), windowLength, step, detectThreshold,
                                   avgPwr) == 1;
}
namespace detail {
// analyzeTrafficBurst with a channel estimate, either variant (trxsig_channel_estimate_host)
inline bool analyze_chan(signalVector &rxBurst, unsigned TSC, float detectThreshold, int variant52m, int maxTOA, complex *amplitude,
                         float *TOA, bool wantChannel, signalVector **channelResponse, float *channelResponseOffset) {
  State &s = state();
  if (!s.ctx || s.sps != 1 || (wantChannel && !channelResponse)) return false;
  uint8_t flags = 0; trxsig_c32 amp = {0, 0}, chan[6]; float toa = 0, choff = 0;
  if (trxsig_channel_estimate_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), (int)rxBurst.size(), (int)TSC, detectThreshold,
                                   variant52m, maxTOA, &flags, &amp, &toa, &choff, chan) != TRXSIG_OK)
    return false;
  if (amplitude) *amplitude = complex(amp.re, amp.im);
  if (TOA) *TOA = toa;
  if (!(flags & TRXSIG_F_DETECT)) return false;
  if (wantChannel) {
    *channelResponse = new signalVector(6);
    for (int k = 0; k < 6; k++) (**channelResponse)[k] = complex(chan[k].re, chan[k].im);
    if (channelResponseOffset) *channelResponseOffset = choff;
  }
  return true;
}
}  // namespace detail

#ifndef TRXFACADE_52M
// analyzeTrafficBurst (sigProcLib.h:288-296).  requestChannel (symbol-rate samples only, as the reference's equaliser):
// *channelResponse = new signalVector(6) (caller deletes) and *channelResponseOffset are set when the burst is detected
// (sigProcLib.cpp:1005-1031).
inline bool analyzeTrafficBurst(signalVector &rxBurst, unsigned TSC, float detectThreshold, int samplesPerSymbol,
                                complex *amplitude, float *TOA, bool requestChannel = false,
                                signalVector **channelResponse = NULL, float *channelResponseOffset = NULL) {
  if (samplesPerSymbol != state().sps || TSC > 7) return false;
  if (!requestChannel) return detail::detect(false, rxBurst, TSC, detectThreshold, -1.0f, amplitude, TOA, NULL, NULL);
  return detail::analyze_chan(rxBurst, TSC, detectThreshold, 0, 0, amplitude, TOA, true, channelResponse, channelResponseOffset);
}
#else
// analyzeTrafficBurst of the 52 MHz transceiver (Transceiver52M/sigProcLib.h:295-305): the correlation only covers the
// 2*maxTOA + 1 lags round the expected peak (Transceiver52M/sigProcLib.cpp:966-1076); symbol-rate samples only.
inline bool analyzeTrafficBurst(signalVector &rxBurst, unsigned TSC, float detectThreshold, int samplesPerSymbol,
                                complex *amplitude, float *TOA, unsigned maxTOA, bool requestChannel = false,
                                signalVector **channelResponse = NULL, float *channelResponseOffset = NULL) {
  if (samplesPerSymbol != state().sps || TSC > 7) return false;
  return detail::analyze_chan(rxBurst, TSC, detectThreshold, 1, (int)maxTOA, amplitude, TOA, requestChannel, channelResponse,
                              channelResponseOffset);
}
#endif
// scaleVector (sigProcLib.h:217-218): x[k] = x[k] * scale, or x[k].real() * scale for a real-only vector (sigProcLib.cpp:713-730)
inline void scaleVector(signalVector &x, complex scale) {
  State &s = state();
  if (!s.ctx || x.size() == 0) return;
  const trxsig_c32 sc = {scale.r, scale.i};
  (void)trxsig_elementwise_host(s.ctx, 0, (trxsig_c32 *)x.begin(), (int)x.size(), sc, x.isRealOnly());
}
// designDFE (sigProcLib.h:365-369): Nf = 7 and a 6-tap channel, as the Transceiver uses it (Transceiver.cpp:347);
// *feedForwardFilter (7 taps) and *feedbackFilter (5 taps) are allocated with new (the caller deletes)
inline bool designDFE(signalVector &channelResponse, float SNRestimate, int Nf, signalVector **feedForwardFilter,
                      signalVector **feedbackFilter) {
  State &s = state();
  if (!s.ctx || Nf != 7 || channelResponse.size() != 6 || !feedForwardFilter || !feedbackFilter) return false;
  trxsig_c32 chan[6], w[7], b[5];
  for (int k = 0; k < 6; k++) { chan[k].re = channelResponse[k].r; chan[k].im = channelResponse[k].i; }
  if (trxsig_design_dfe_host(s.ctx, chan, SNRestimate, w, b) != TRXSIG_OK) return false;
  *feedForwardFilter = new signalVector(7);
  *feedbackFilter = new signalVector(5);
  for (int k = 0; k < 7; k++) (**feedForwardFilter)[k] = complex(w[k].re, w[k].im);
  for (int k = 0; k < 5; k++) (**feedbackFilter)[k] = complex(b[k].re, b[k].im);
  return true;
}
// equalizeBurst (sigProcLib.h:380-384): the burst is expected scaled by 1/amplitude already (Transceiver.cpp:391);
// returns one soft bit per sample; caller deletes
inline SoftVector *equalizeBurst(signalVector &rxBurst, float TOA, int samplesPerSymbol, signalVector &w, signalVector &b) {
  State &s = state();
  if (!s.ctx || samplesPerSymbol != 1 || s.sps != 1 || w.size() != 7 || b.size() != 5 || rxBurst.size() > 157) return NULL;
  trxsig_c32 wt[7], bt[5];
  for (int k = 0; k < 7; k++) { wt[k].re = w[k].r; wt[k].im = w[k].i; }
  for (int k = 0; k < 5; k++) { bt[k].re = b[k].r; bt[k].im = b[k].i; }
  const int ns = (int)rxBurst.size();
  SoftVector *out = new SoftVector(ns);
  const trxsig_c32 one = {1.0f, 0.0f};
  if (trxsig_equalize_taps_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), ns, one, TOA, wt, bt, out->begin(), ns) != TRXSIG_OK) {
    delete out;
    return NULL;
  }
  return out;
}
// detectRACHBurst (sigProcLib.h:269-273)
inline bool detectRACHBurst(signalVector &rxBurst, float detectThreshold, int samplesPerSymbol, complex *amplitude,
                            float *TOA) {
  if (samplesPerSymbol != state().sps) return false;
  return detail::detect(true, rxBurst, 0, detectThreshold, -1.0f, amplitude, TOA, NULL, NULL);
}
namespace detail {
inline SoftVector *demod(const signalVector &rxBurst, int samplesPerSymbol, complex channel, float TOA) {
  State &s = state();
  if (!s.ctx || samplesPerSymbol != s.sps) return NULL;
  const int ns = (int)(rxBurst.size() / (size_t)s.sps);
  SoftVector *out = new SoftVector(ns);
  trxsig_c32 a = {channel.r, channel.i};
  if (trxsig_demodulate_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), (int)rxBurst.size(), a, TOA, out->begin(),
                             ns) != TRXSIG_OK) {
    delete out;
    return NULL;
  }
  return out;
}
}  // namespace detail

// vectorSlicer (sigProcLib.h:168), delayVector (:180-181), GMSKRotate / GMSKReverseRotate (sigProcLib.cpp:232-264): in place
inline bool vectorSlicer(signalVector *x) {
  State &s = state();
  const trxsig_c32 one = {1.0f, 0.0f};
  return s.ctx && x && x->size() > 0 &&
         trxsig_elementwise_host(s.ctx, 3, (trxsig_c32 *)x->begin(), (int)x->size(), one, 0) == TRXSIG_OK;
}
// (a delay / index beyond +-TRXSIG_MAX_INDEX, infinite or NaN -- the reference's sinc range reduction never returns on those --
//  is refused by the library: delayVector leaves the vector as it is, interpolatePoint answers (0, 0))
inline void delayVector(signalVector &wBurst, float delay) {
  State &s = state();
  if (s.ctx && wBurst.size() > 0)
    (void)trxsig_delay_vector_host(s.ctx, (trxsig_c32 *)wBurst.begin(), (int)wBurst.size(), delay, wBurst.isRealOnly());
}
inline void GMSKRotate(signalVector &x) {
  State &s = state();
  const trxsig_c32 one = {1.0f, 0.0f};
  if (s.ctx && x.size() > 0) (void)trxsig_elementwise_host(s.ctx, 1, (trxsig_c32 *)x.begin(), (int)x.size(), one, x.isRealOnly());
}
inline void GMSKReverseRotate(signalVector &x) {
  State &s = state();
  const trxsig_c32 one = {1.0f, 0.0f};
  if (s.ctx && x.size() > 0) (void)trxsig_elementwise_host(s.ctx, 2, (trxsig_c32 *)x.begin(), (int)x.size(), one, x.isRealOnly());
}

#ifndef TRXFACADE_52M
// demodulateBurst (sigProcLib.h:316-320): returns N/sps soft bits; caller deletes
inline SoftVector *demodulateBurst(const signalVector &rxBurst, const signalVector &, int samplesPerSymbol,
                                   complex channel, float TOA) {
  return detail::demod(rxBurst, samplesPerSymbol, channel, TOA);
}
#else
// demodulateBurst of the 52 MHz transceiver (Transceiver52M/sigProcLib.h:324-328): the burst itself is scaled, delayed and
// de-rotated in place (Transceiver52M/sigProcLib.cpp:1095-1135), and that is what the caller finds in it afterwards
inline SoftVector *demodulateBurst(signalVector &rxBurst, const signalVector &, int samplesPerSymbol, complex channel,
                                   float TOA) {
  SoftVector *out = detail::demod(rxBurst, samplesPerSymbol, channel, TOA);
  if (!out) return NULL;
  State &s = state();
  const complex inv = complex(1.0f, 0.0f) / channel;       // ((complex) 1.0)/channel
  const trxsig_c32 sc = {inv.r, inv.i};
  (void)trxsig_elementwise_host(s.ctx, 0, (trxsig_c32 *)rxBurst.begin(), (int)rxBurst.size(), sc, rxBurst.isRealOnly());
  delayVector(rxBurst, -TOA);
  GMSKReverseRotate(rxBurst);
  return out;
}
#endif

// convolve (sigProcLib.h:126-129) and correlate (:162-165): c == NULL allocates the result (caller deletes); a
// preallocated c must have exactly the output size, else NULL (sigProcLib.cpp:307-310).  b->getSymmetry() == ABSSYM takes
// convolve's symmetric-filter branch (:369-398).
namespace detail {
inline signalVector *conv(const signalVector *a, const signalVector *b, signalVector *c, ConvType spanType, int correlate) {
  State &s = state();
  if (!s.ctx || a == NULL || b == NULL) return NULL;
  const int n = trxsig_convolve_out_len((int)a->size(), (int)b->size(), (int)spanType, 0);
  if (n <= 0 || spanType > NO_DELAY) return NULL;
  const bool mine = (c == NULL);
  if (mine) c = new signalVector(n);
  else if ((int)c->size() != n) return NULL;
  // (correlate's reversed copy of b never carries a symmetry, sigProcLib.cpp:480-481)
  const int flags = (a->isRealOnly() ? 1 : 0) | (b->isRealOnly() ? 2 : 0) | ((!correlate && b->getSymmetry() == ABSSYM) ? 4 : 0);
  if (trxsig_convolve_host(s.ctx, (const trxsig_c32 *)a->begin(), (int)a->size(), (const trxsig_c32 *)b->begin(), (int)b->size(),
                           (int)spanType, flags, correlate, 0, 0, (trxsig_c32 *)c->begin(), n) != n) {
    if (mine) delete c;
    return NULL;
  }
  return c;
}
}  // namespace detail
inline signalVector *convolve(const signalVector *a, const signalVector *b, signalVector *c, ConvType spanType) {
  return detail::conv(a, b, c, spanType, 0);
}
inline signalVector *correlate(signalVector *a, signalVector *b, signalVector *c, ConvType spanType) {
  return detail::conv(a, b, c, spanType, 1);
}
// interpolatePoint (sigProcLib.h:198-199) and peakDetect (:208-210)
inline complex interpolatePoint(const signalVector &inSig, float ix) {
  State &s = state();
  trxsig_c32 r = {0.0f, 0.0f};
  if (s.ctx && inSig.size() > 0)
    (void)trxsig_interpolate_point_host(s.ctx, (const trxsig_c32 *)inSig.begin(), (int)inSig.size(), ix, inSig.isRealOnly(), &r);
  return complex(r.re, r.im);
}
inline complex peakDetect(const signalVector &rxBurst, float *peakIndex, float *avgPwr) {
  State &s = state();
  trxsig_c32 r = {0.0f, 0.0f};
  if (s.ctx && rxBurst.size() > 0)
    (void)trxsig_peak_detect_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), (int)rxBurst.size(), &r, peakIndex, avgPwr);
  return complex(r.re, r.im);
}
// decimateVector (sigProcLib.h:304-305): NULL for a factor <= 1 (sigProcLib.cpp:1043); caller deletes
inline signalVector *decimateVector(signalVector &wVector, int decimationFactor) {
  State &s = state();
  if (!s.ctx || decimationFactor <= 1 || wVector.size() < (size_t)decimationFactor) return NULL;
  signalVector *d = new signalVector((int)(wVector.size() / (size_t)decimationFactor));
  d->isRealOnly(wVector.isRealOnly());
  if (trxsig_decimate_host(s.ctx, (const trxsig_c32 *)wVector.begin(), (int)wVector.size(), decimationFactor,
                           (trxsig_c32 *)d->begin()) != (int)d->size()) {
    delete d;
    return NULL;
  }
  return d;
}

// ---- the functions no caller on the burst path uses (sigProcLib.h:101-111, 149-153, 177, 184-190, 225-226, 352-354) ----
inline float dB(float x) { return trxsig_db(x); }                                              // :102
inline float dBinv(float x) { return trxsig_dbinv(x); }                                        // :105
inline float vectorNorm2(const signalVector &x) {                                              // :108
  State &s = state();
  float e = 0.0f;
  if (s.ctx && x.size() > 0) (void)trxsig_vector_norm2_host(s.ctx, (const trxsig_c32 *)x.begin(), (int)x.size(), &e, NULL);
  return e;
}
inline float vectorPower(const signalVector &x) {                                              // :111
  State &s = state();
  float p = 0.0f;
  if (s.ctx && x.size() > 0) (void)trxsig_vector_norm2_host(s.ctx, (const trxsig_c32 *)x.begin(), (int)x.size(), NULL, &p);
  return p;
}
// frequencyShift (:149-153): y == NULL allocates the result (caller deletes); y may be x; NULL if y is shorter than x
inline signalVector *frequencyShift(signalVector *y, signalVector *x, float freq = 0.0, float startPhase = 0.0, float *finalPhase = NULL) {
  State &s = state();
  if (!s.ctx || !x) return NULL;
  const bool mine = (y == NULL);
  if (mine) { y = new signalVector((int)x->size()); y->isRealOnly(x->isRealOnly()); }
  if (y->size() < x->size()) return NULL;
  float fin = startPhase;
  if (x->size() > 0 && trxsig_frequency_shift_host(s.ctx, (const trxsig_c32 *)x->begin(), (int)x->size(), freq, startPhase, x->isRealOnly(),
                                                   (trxsig_c32 *)y->begin(), &fin) != TRXSIG_OK) {
    if (mine) delete y;
    return NULL;
  }
  if (finalPhase) *finalPhase = fin;
  return y;
}
inline float sinc(float x) {                                                                   // :177
  float v = 1.0f;
  if (state().ctx) (void)trxsig_sinc_host(state().ctx, x, &v);
  return v;
}
inline bool addVector(signalVector &x, signalVector &y) {                                      // :184-185, in place on x
  State &s = state();
  if (!s.ctx) return false;
  if (x.size() == 0 || y.size() == 0) return true;
  return trxsig_add_vector_host(s.ctx, (trxsig_c32 *)x.begin(), (int)x.size(), (const trxsig_c32 *)y.begin(), (int)y.size()) == TRXSIG_OK;
}
// gaussianNoise (:188-190): the C library's rand() in the reference's draw order -- seed it with srand(); caller deletes
inline signalVector *gaussianNoise(int length, float variance = 1.0, complex mean = complex(0.0)) {
  if (length < 0) return NULL;
  signalVector *noise = new signalVector(length);
  const trxsig_c32 m = {mean.r, mean.i};
  if (trxsig_gaussian_noise_host(length, variance, m, (trxsig_c32 *)noise->begin()) != TRXSIG_OK) { delete noise; return NULL; }
  return noise;
}
inline void offsetVector(signalVector &x, complex offset) {                                    // :225-226
  State &s = state();
  if (!s.ctx || x.size() == 0) return;
  const trxsig_c32 o = {offset.r, offset.i};
  (void)trxsig_elementwise_host(s.ctx, 4, (trxsig_c32 *)x.begin(), (int)x.size(), o, x.isRealOnly());
}
// resampleVector (:352-354) as the reference behaves (its loop never advances the output iterator: element 0 takes every
// interpolated value, the rest stays zero); NULL for expFactor < 1; caller deletes
inline signalVector *resampleVector(signalVector &wVector, float expFactor, complex endPoint) {
  State &s = state();
  const int n = trxsig_resample_linear_out_len((int)wVector.size(), expFactor);
  if (!s.ctx || n < 0) return NULL;
  signalVector *out = new signalVector(n);
  const trxsig_c32 e = {endPoint.r, endPoint.i};
  if (wVector.size() > 0 && trxsig_resample_linear_host(s.ctx, (const trxsig_c32 *)wVector.begin(), (int)wVector.size(), expFactor, e,
                                                        (trxsig_c32 *)out->begin(), n) != n) {
    delete out;
    return NULL;
  }
  return out;
}

// createLPF (sigProcLib.h:329-331).  The reference ignores the cutoff and loads one of its two coefficient
// tables (rcvLPF_651.h for filterLen 651, else sendLPF_961.h; sigProcLib.cpp:1119-1139); those tables are
// reference data, so the caller registers them once (the arrays of the reference's own headers will do).
inline const float *&lpfTable(int which) { static const float *t[2] = {NULL, NULL}; return t[which]; }
inline void setLPFTables(const float *rcvLPF_651, const float *sendLPF_961) { lpfTable(0) = rcvLPF_651; lpfTable(1) = sendLPF_961; }
inline signalVector *createLPF(float /*cutoffFreq*/, int filterLen, float gainDC = 1.0F) {
  const float *raw = lpfTable(filterLen == 651 ? 0 : 1);
  const int len = filterLen == 651 ? 651 : 961;
  if (!raw) return NULL;
  std::vector<float> taps(len);
  if (trxsig_create_lpf_host(raw, len, gainDC, taps.data()) != TRXSIG_OK) return NULL;
  signalVector *lpf = new signalVector(len);
  for (int k = 0; k < len; k++) (*lpf)[k] = complex(taps[k], 0.0f);
  lpf->isRealOnly(true);
  return lpf;
}
// polyphaseResampleVector (sigProcLib.h:341-343) with a real-only LPF (the only kind createLPF makes)
inline signalVector *polyphaseResampleVector(signalVector &wVector, int P, int Q, signalVector *LPF) {
  State &s = state();
  if (!s.ctx || !LPF || !LPF->isRealOnly() || wVector.size() == 0) return NULL;
  std::vector<float> taps(LPF->size());
  for (size_t k = 0; k < LPF->size(); k++) taps[k] = (*LPF)[k].r;
  const int nout = trxsig_resample_out_len((int)wVector.size(), P, Q);
  signalVector *out = new signalVector(nout);
  if (trxsig_resample_host(s.ctx, (const trxsig_c32 *)wVector.begin(), (int)wVector.size(), P, Q, taps.data(),
                           (int)taps.size(), (trxsig_c32 *)out->begin(), nout) != nout) {
    delete out;
    return NULL;
  }
  return out;
}

}  // namespace trxfacade

#ifndef TRXFACADE_NO_GLOBAL_NAMES   /* the reference's names are namespace-less (sigProcLib.h) */
using trxfacade::analyzeTrafficBurst;
using trxfacade::dB;
using trxfacade::dBinv;
using trxfacade::vectorNorm2;
using trxfacade::vectorPower;
using trxfacade::frequencyShift;
using trxfacade::sinc;
using trxfacade::addVector;
using trxfacade::gaussianNoise;
using trxfacade::offsetVector;
using trxfacade::resampleVector;
using trxfacade::BitVector;
using trxfacade::createLPF;
using trxfacade::polyphaseResampleVector;
using trxfacade::setLPFTables;
using trxfacade::complex;
using trxfacade::Vector;
using trxfacade::Symmetry;
using trxfacade::NONE;
using trxfacade::ABSSYM;
using trxfacade::ConvType;
using trxfacade::FULL_SPAN;
using trxfacade::OVERLAP_ONLY;
using trxfacade::START_ONLY;
using trxfacade::WITH_TAIL;
using trxfacade::NO_DELAY;
using trxfacade::UNDEFINED;
using trxfacade::convolve;
using trxfacade::correlate;
using trxfacade::vectorSlicer;
using trxfacade::delayVector;
using trxfacade::GMSKRotate;
using trxfacade::GMSKReverseRotate;
using trxfacade::interpolatePoint;
using trxfacade::peakDetect;
using trxfacade::decimateVector;
using trxfacade::sigProcLibReady;
using trxfacade::sigProcLibSetDevice;
using trxfacade::demodulateBurst;
using trxfacade::designDFE;
using trxfacade::equalizeBurst;
using trxfacade::scaleVector;
using trxfacade::detectRACHBurst;
using trxfacade::energyDetect;
using trxfacade::generateGSMPulse;
using trxfacade::generateMidamble;
using trxfacade::generateRACHSequence;
using trxfacade::modulateBurst;
using trxfacade::signalVector;
using trxfacade::sigProcLibDestroy;
using trxfacade::sigProcLibSetup;
using trxfacade::SoftVector;
#endif

#endif  // SIGPROCLIB_TRX_H
