// sigProcLib_trx.h -- source-compatible C++ facade over libtrxsig for code written against the
// reference's Transceiver/sigProcLib.h (OpenBTS).  Header only; link with -ltrxsig.
//
// It re-creates the names a Transceiver / RadioInterface translation unit uses -- `complex`,
// `signalVector`, `BitVector`, `SoftVector`, sigProcLibSetup, generateGSMPulse, modulateBurst,
// generateMidamble, generateRACHSequence, energyDetect, analyzeTrafficBurst, detectRACHBurst,
// demodulateBurst, scaleVector, designDFE, equalizeBurst, polyphaseResampleVector, createLPF (with setLPFTables) --
// i.e. every sigProcLib function Transceiver.cpp and radioInterface.cpp call -- with the reference's argument meaning,
// ownership (functions returning a pointer allocate with `new`, the caller deletes:
// Transceiver.cpp:112,407,672) and error behaviour (NULL / false, amplitude set to 0 on a "bogus
// result": sigProcLib.cpp:878-882, 964-968).  Every call that processes samples runs on the GPU
// through the C-ABI's host-buffer entry points (one PCIe round trip per call); nothing here computes
// on the CPU except scaleVector's element-wise complex multiply.  A real deployment batches instead -- see INTEGRATION.md.
//
// Not provided: the free-standing vector primitives (convolve, correlate, delayVector, peakDetect,
// interpolatePoint, ...).  Transceiver.cpp and radioInterface.cpp never call them (SURVEY 8b lists
// the call sites); inside the library they only exist fused into the burst-level kernels.
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
};

// Vector<T> subset (CommonLibs/Vector.h:42-252): contiguous, owning.
template <class T>
class Vector {
 public:
  typedef T *iterator;
  typedef const T *const_iterator;
  explicit Vector(size_t n = 0) : d_(n) {}
  size_t size() const { return d_.size(); }
  T *begin() { return d_.data(); }
  const T *begin() const { return d_.data(); }
  T *end() { return d_.data() + d_.size(); }
  const T *end() const { return d_.data() + d_.size(); }
  T &operator[](size_t k) { return d_[k]; }
  const T &operator[](size_t k) const { return d_[k]; }
  void fill(const T &v) { for (auto &e : d_) e = v; }
 private:
  std::vector<T> d_;
};

// signalVector (sigProcLib.h:51-99)
class signalVector : public Vector<complex> {
 public:
  explicit signalVector(size_t n = 0) : Vector<complex>(n), realOnly_(false) {}
  bool isRealOnly() const { return realOnly_; }
  void isRealOnly(bool v) { realOnly_ = v; }
 private:
  bool realOnly_;
};

// BitVector: one bit per char, consumers mask with 0x01 (CommonLibs/BitVector.cpp:54-63)
class BitVector : public Vector<char> {
 public:
  explicit BitVector(size_t n = 0) : Vector<char>(n) {}
  explicit BitVector(const char *s) : Vector<char>(std::strlen(s)) {
    for (size_t k = 0; k < size(); k++) (*this)[k] = (s[k] == '1');
  }
  bool bit(size_t k) const { return (*this)[k] & 0x01; }
};

// SoftVector: float 0..1, bit() = > 0.5F (CommonLibs/BitVector.h:415-420)
class SoftVector : public Vector<float> {
 public:
  explicit SoftVector(size_t n = 0) : Vector<float>(n) {}
  bool bit(size_t k) const { return (*this)[k] > 0.5F; }
};

// ---- library state (the reference keeps process globals, sigProcLib.cpp:39-59) ----------------
struct State {
  trxsig_ctx *ctx = nullptr;
  int sps = 0;
  int device = 0;
};
inline State &state() { static State s; return s; }

inline void sigProcLibDestroy(void) {                         // sigProcLib.h:113
  State &s = state();
  if (s.ctx) { trxsig_destroy(s.ctx); s.ctx = nullptr; }
}
// sigProcLibSetup (sigProcLib.h:110): also builds the pulse, the 8 midambles and the RACH sequence,
// which the reference creates through separate calls at start-up (Transceiver.cpp:62-64,424,553).
inline bool sigProcLibSetup(int samplesPerSymbol, int device = 0) {
  State &s = state();
  sigProcLibDestroy();
  s.sps = samplesPerSymbol; s.device = device;
  return trxsig_create(&s.ctx, device, samplesPerSymbol) == TRXSIG_OK;
}

// generateGSMPulse(2, sps) (sigProcLib.h:137-138): a copy of the table built at set-up
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
// generateMidamble / generateRACHSequence (sigProcLib.h:227-239): built at set-up; these only check.
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

// energyDetect (sigProcLib.h:246-249); windowLength must be 20*sps, the only value the Transceiver
// uses (Transceiver.cpp:298)
inline bool energyDetect(signalVector &rxBurst, unsigned windowLength, float detectThreshold, float *avgPwr = NULL) {
  if (windowLength != 20u * (unsigned)state().sps) return false;
  bool ok = false;
  detail::detect(false, rxBurst, 0, 1e30f, detectThreshold, NULL, NULL, avgPwr, &ok);
  return ok;
}
// analyzeTrafficBurst (sigProcLib.h:277-285).  requestChannel (symbol-rate samples only, as the reference's equaliser):
// *channelResponse = new signalVector(6) (caller deletes) and *channelResponseOffset are set when the burst is detected
// (sigProcLib.cpp:1005-1031).
inline bool analyzeTrafficBurst(signalVector &rxBurst, unsigned TSC, float detectThreshold, int samplesPerSymbol,
                                complex *amplitude, float *TOA, bool requestChannel = false,
                                signalVector **channelResponse = NULL, float *channelResponseOffset = NULL) {
  if (samplesPerSymbol != state().sps || TSC > 7) return false;
  if (!requestChannel) return detail::detect(false, rxBurst, TSC, detectThreshold, -1.0f, amplitude, TOA, NULL, NULL);
  State &s = state();
  if (!s.ctx || s.sps != 1 || !channelResponse) return false;
  uint8_t flags = 0; trxsig_c32 amp = {0, 0}, chan[6]; float toa = 0, choff = 0;
  if (trxsig_channel_estimate_host(s.ctx, (const trxsig_c32 *)rxBurst.begin(), (int)rxBurst.size(), (int)TSC, detectThreshold,
                                   0, 0, &flags, &amp, &toa, &choff, chan) != TRXSIG_OK)
    return false;
  if (amplitude) *amplitude = complex(amp.re, amp.im);
  if (TOA) *TOA = toa;
  if (!(flags & TRXSIG_F_DETECT)) return false;
  *channelResponse = new signalVector(6);
  for (int k = 0; k < 6; k++) (**channelResponse)[k] = complex(chan[k].re, chan[k].im);
  if (channelResponseOffset) *channelResponseOffset = choff;
  return true;
}
// scaleVector (sigProcLib.h:182-183): x[k] = x[k] * scale (sigProcLib.cpp:713-723)
inline void scaleVector(signalVector &x, complex scale) {
  for (size_t k = 0; k < x.size(); k++) x[k] = x[k] * scale;
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
// detectRACHBurst (sigProcLib.h:263-267)
inline bool detectRACHBurst(signalVector &rxBurst, float detectThreshold, int samplesPerSymbol, complex *amplitude,
                            float *TOA) {
  if (samplesPerSymbol != state().sps) return false;
  return detail::detect(true, rxBurst, 0, detectThreshold, -1.0f, amplitude, TOA, NULL, NULL);
}
// demodulateBurst (sigProcLib.h:316-320): returns N/sps soft bits; caller deletes
inline SoftVector *demodulateBurst(const signalVector &rxBurst, const signalVector &, int samplesPerSymbol,
                                   complex channel, float TOA) {
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

// createLPF (sigProcLib.h:340-343).  The reference ignores the cutoff and loads one of its two coefficient
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
// polyphaseResampleVector (sigProcLib.h:352-354) with a real-only LPF (the only kind createLPF makes)
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
using trxfacade::BitVector;
using trxfacade::createLPF;
using trxfacade::polyphaseResampleVector;
using trxfacade::setLPFTables;
using trxfacade::complex;
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
