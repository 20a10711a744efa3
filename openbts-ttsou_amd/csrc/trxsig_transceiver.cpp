// trxsig_transceiver.cpp -- the host-side Transceiver orchestration (include/trxsig_transceiver.h): the state
// machine of Transceiver/Transceiver.cpp around libtrxsig's GPU calls.  All signal processing happens in the
// kernels; what is computed here is exactly what the reference computes on the host between its sigProcLib
// calls (thresholds, SNR, RSSI, timing offset, time arithmetic), in the reference's types (double where it
// uses double).  Line references: Transceiver/Transceiver.cpp unless another file is named.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <queue>
#include <string>
#include <vector>

#include "trxsig_transceiver.h"
#include "trxsig_trxstate.h"

namespace {

constexpr int kHyperframe = 2048 * 26 * 51;                 // GSM/GSMCommon.h:306

// GSM::Time (GSM/GSMCommon.h:327-455) and FNDelta / FNCompare (GSM/GSMCommon.cpp:161-176)
struct Time {
  int fn = 0, tn = 0;
};
int fn_delta(int v1, int v2) {
  const int half = kHyperframe / 2;
  int d = v1 - v2;
  if (d >= half) d -= kHyperframe;
  else if (d < -half) d += kHyperframe;
  return d;
}
int fn_compare(int v1, int v2) { const int d = fn_delta(v1, v2); return d > 0 ? 1 : (d < 0 ? -1 : 0); }
bool time_less(const Time &a, const Time &b) { return a.fn == b.fn ? a.tn < b.tn : fn_compare(a.fn, b.fn) < 0; }
bool time_equal(const Time &a, const Time &b) { return a.fn == b.fn && a.tn == b.tn; }
int time_minus(const Time &a, const Time &b) { return fn_delta(a.fn, b.fn); }        // operator-(Time): frames

// the dummy burst of GSM 05.02 5.2.6 (gDummyBurst, GSM/GSMCommon.cpp)
const char kDummyBurst[149] =
    "0001111101101110110000010100100111000001001000100000001111100011100010111000101110001010111010010100"
    "011001100111001111010011111000100101111101010000";

bool time_greater(const Time &a, const Time &b) { return a.fn == b.fn ? a.tn > b.tn : fn_compare(a.fn, b.fn) > 0; }   // GSMCommon.h:431-435
Time time_plus(const Time &a, const Time &b) {               // Time::operator+(const Time&) (GSMCommon.h:405-410)
  Time r;
  r.tn = (a.tn + b.tn) % 8;
  r.fn = (int)(((long long)a.fn + b.fn + (a.tn + b.tn) / 8) % kHyperframe);
  return r;
}
void time_inc_tn(Time &t) {                                  // incTN() (GSMCommon.h:375-386)
  t.tn += 1;
  if (t.tn > 7) { t.tn -= 8; t.fn = (t.fn + 1) % kHyperframe; }
}
void time_dec_tn(Time &t) {                                  // decTN() (GSMCommon.h:363-373)
  t.tn -= 1;
  if (t.tn < 0) { t.tn += 8; t.fn -= 1; if (t.fn < 0) t.fn += kHyperframe; }
}

struct Queued {
  Time time;
  std::vector<trxsig_c32> samples;
};
// VectorQueue (radioInterface.h:64-72) = InterthreadPriorityQueue<radioVector> (Interthread.h:432-528): the standard library's
// priority_queue of POINTERS ordered by `*v1 > *v2` on the timestamps.  The same container with the same comparator here, so
// that bursts with equal timestamps leave in the order they would leave the reference's queue (it depends on the heap's shape).
struct QueuedLater {
  bool operator()(const Queued *a, const Queued *b) const { return time_greater(a->time, b->time); }
};
typedef std::priority_queue<Queued *, std::vector<Queued *>, QueuedLater> TxQueue;

}  // namespace

struct trxsig_trx {
  trxsig_ctx *ctx = nullptr;
  int sps = 1;
  std::string err;
  TrxControl ctl;                                           // control state (:83-91, 439-580)
  int tscLeg = TRXSIG_TSCLEG_EQUALIZE;                      // how the TSC leg ends (trxsig_trx_set_tsc_leg)
  // receive state
  double energyThreshold = 250.0;                           // :88
  Time prevFalseDetectionTime;
  bool haveChan[8];
  float chanRespOffset[8];
  float SNRestimate[8];
  Time channelEstimateTime[8];
  trxsig_c32 dfeW[8][7], dfeB[8][5];
  // transmit state (fillerModulus lives in ctl)
  std::vector<trxsig_c32> fillerTable[102][8];
  TxQueue queue;                                            // earliest time on top (VectorQueue)
  // device scratch for the single-burst calls
  char *d = nullptr;
  char *hpin = nullptr;              // pinned host mirror of the first part of d (samples + scalars + taps): one DMA per step
  size_t d_bytes = 0;
};

namespace {

int fail(trxsig_trx *t, int code, const std::string &what) {
  if (t) t->err = what;
  return code;
}
#define TRX_HIP(t, call)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) return fail(t, TRXSIG_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define TRX_LIB(t, call)                                                                  \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != TRXSIG_OK) return fail(t, rc_, std::string(#call) + ": " + trxsig_last_error((t)->ctx)); \
  } while (0)

// modulateBurst(bits, pulse, 8 + (TN % 4 == 0), sps) [+ scaleVector(gain)] on the GPU
int modulate(trxsig_trx *t, const uint8_t *bits, int tn, const float *gain, std::vector<trxsig_c32> &out) {
  const int32_t guard = 8 + ((tn % 4) == 0), off = 0;
  out.assign((size_t)t->sps * (148 + guard), trxsig_c32{0.0f, 0.0f});
  TRX_LIB(t, trxsig_modulate_host(t->ctx, bits, &guard, gain, 1, out.data(), &off, (int64_t)out.size()));
  return TRXSIG_OK;
}

}  // namespace

// ---- TrxControl (trxsig_trxstate.h) ----
void TrxControl::setModulus(int ts) {                       // setModulus (:183-204)
  switch (chanType[ts]) {
    case TRXSIG_CHAN_NONE: case TRXSIG_CHAN_I: case TRXSIG_CHAN_II: case TRXSIG_CHAN_III: fillerModulus[ts] = 26; break;
    case TRXSIG_CHAN_IV: case TRXSIG_CHAN_VI: case TRXSIG_CHAN_V: fillerModulus[ts] = 51; break;
    case TRXSIG_CHAN_VII: fillerModulus[ts] = 102; break;
    default: break;
  }
}

int TrxControl::corrType(int chanType, int fn) {            // expectedCorrType (:207-269)
  const unsigned burstFN = (unsigned)fn;
  switch (chanType) {
    case TRXSIG_CHAN_NONE: return TRXSIG_CORR_OFF;
    case TRXSIG_CHAN_I: return TRXSIG_CORR_TSC;
    case TRXSIG_CHAN_II: return (burstFN % 2 == 1) ? TRXSIG_CORR_IDLE : TRXSIG_CORR_TSC;
    case TRXSIG_CHAN_III: return TRXSIG_CORR_TSC;
    case TRXSIG_CHAN_IV:
    case TRXSIG_CHAN_VI: return ((burstFN % 51) % 10 < 2) ? TRXSIG_CORR_RACH : TRXSIG_CORR_OFF;
    case TRXSIG_CHAN_V: {
      const int mod51 = burstFN % 51;
      if (mod51 <= 36 && mod51 >= 14) return TRXSIG_CORR_RACH;
      if (mod51 == 4 || mod51 == 5) return TRXSIG_CORR_RACH;
      if (mod51 == 45 || mod51 == 46) return TRXSIG_CORR_RACH;
      return TRXSIG_CORR_TSC;
    }
    case TRXSIG_CHAN_VII:
      if (burstFN % 51 == 12 || burstFN % 51 == 13 || burstFN % 51 == 14) return TRXSIG_CORR_IDLE;
      return TRXSIG_CORR_TSC;
    case TRXSIG_CHAN_LOOPBACK:
      return (burstFN % 51 <= 50 && burstFN % 51 >= 48) ? TRXSIG_CORR_IDLE : TRXSIG_CORR_TSC;
    default: return TRXSIG_CORR_OFF;
  }
}
int TrxControl::expectedCorrType(int tn, int fn) const { return corrType(chanType[tn & 7], fn); }

// ---- driveControl (Transceiver.cpp:439-580; SETMAXDELAY: Transceiver52M/Transceiver.cpp:476-486) as a table ----
// A datagram is `CMD <verb> [int [int]]`; the answer is `RSP <verb> <status> [ints echoed]` (README.TRXManager).  One row per
// verb: how many integers it takes and a rule that updates the state and says what goes back.  What is on the wire -- the
// verbs, the status digit, which integers are echoed -- is the contract with TRXManager and follows the reference verb for
// verb; an integer that is missing from the datagram reads as 0 here (the reference leaves its local uninitialised).
namespace {
struct Reply { int status; int n_echo; int echo[2]; bool silent; };
typedef Reply (*VerbRule)(TrxControl &, const int *);
Reply rule_poweroff(TrxControl &, const int *) { return {0, 0, {0, 0}, false}; }          // answers 0 and leaves the radio as it is (:472-475)
Reply rule_poweron(TrxControl &c, const int *) {                                           // refused until both frequencies are set (:476-488)
  if (!c.txFreq || !c.rxFreq) return {1, 0, {0, 0}, false};
  if (!c.on) { c.power = -20; c.on = true; }
  return {0, 0, {0, 0}, false};
}
Reply rule_setpower(TrxControl &c, const int *a) {                                        // only while on (:489-500)
  if (!c.on) return {1, 1, {a[0], 0}, false};
  c.power = a[0];
  return {0, 1, {a[0], 0}, false};
}
Reply rule_adjpower(TrxControl &c, const int *a) {                                        // echoes the resulting power (:501-512)
  if (!c.on) return {1, 1, {c.power, 0}, false};
  c.power += a[0];
  return {0, 1, {c.power, 0}, false};
}
Reply rule_rxtune(TrxControl &c, const int *a) {                                          // only while off (:513-528)
  if (c.on) return {1, 1, {a[0], 0}, false};
  c.rxFreq = a[0] * 1.0e3;
  return {0, 1, {a[0], 0}, false};
}
Reply rule_txtune(TrxControl &c, const int *a) {                                          // (:529-544)
  if (c.on) return {1, 1, {a[0], 0}, false};
  c.txFreq = a[0] * 1.0e3;
  return {0, 1, {a[0], 0}, false};
}
Reply rule_settsc(TrxControl &c, const int *a) {
  // only while off (:545-556).  A training sequence outside 0..7 is refused (status 1): the reference stores it and then
  // indexes gMidambles[] with it at the next normal burst (sigProcLib.cpp:946) -- undefined behaviour, nothing to reproduce
  if (c.on || a[0] < 0 || a[0] > 7) return {1, 1, {a[0], 0}, false};
  c.tsc = (unsigned)a[0]; c.epoch++;
  return {0, 1, {a[0], 0}, false};
}
Reply rule_setslot(TrxControl &c, const int *a) {                                         // (:557-570): a timeslot outside 0..7 gets NO answer
  if (a[0] < 0 || a[0] > 7) return {0, 0, {0, 0}, true};
  c.chanType[a[0]] = a[1];
  c.setModulus(a[0]);
  c.epoch++;
  return {0, 2, {a[0], a[1]}, false};
}
Reply rule_setmaxdelay(TrxControl &c, const int *a) {                                     // 52M: only while on
  if (!c.on) return {1, 1, {a[0], 0}, false};
  c.maxDelay = a[0];
  return {0, 1, {a[0], 0}, false};
}
const struct { const char *verb; int n_args; VerbRule rule; } kVerbs[] = {
  {"POWEROFF", 0, rule_poweroff}, {"POWERON", 0, rule_poweron},   {"SETPOWER", 1, rule_setpower},
  {"ADJPOWER", 1, rule_adjpower}, {"RXTUNE", 1, rule_rxtune},     {"TXTUNE", 1, rule_txtune},
  {"SETTSC", 1, rule_settsc},     {"SETSLOT", 2, rule_setslot},   {"SETMAXDELAY", 1, rule_setmaxdelay},
};
}  // namespace

int TrxControl::command(const char *buffer, char *response) {
  char tag[4] = {0}, verb[100] = {0};
  int args[2] = {0, 0};
  response[0] = 0;
  std::sscanf(buffer, "%3s %99s %d %d", tag, verb, &args[0], &args[1]);
  if (std::strcmp(tag, "CMD") != 0) return 0;                 // "bogus message": no response (:466-470)
  for (const auto &v : kVerbs) {
    if (std::strcmp(verb, v.verb) != 0) continue;
    if (v.n_args < 2) args[1] = 0;
    if (v.n_args < 1) args[0] = 0;
    const Reply r = v.rule(*this, args);
    if (r.silent) return 0;
    int n = std::sprintf(response, "RSP %s %d", v.verb, r.status);
    for (int k = 0; k < r.n_echo; k++) n += std::sprintf(response + n, " %d", r.echo[k]);
    return 1;
  }
  return 1;                                                   // unknown verb: the empty response buffer is sent (:571-575)
}

extern "C" {

int trxsig_trx_create(trxsig_trx **out, int device, int sps, int start_fn, int start_tn) {
  if (!out) return TRXSIG_EINVAL;
  *out = nullptr;
  trxsig_trx *t = new trxsig_trx();
  t->sps = sps;
  int rc = trxsig_create(&t->ctx, device, sps);             // sigProcLibSetup + pulse + midambles + RACH sequence (:62-64, 424, 553)
  if (rc != TRXSIG_OK) { delete t; return rc; }
  const Time start{start_fn, start_tn};
  t->prevFalseDetectionTime = start;
  uint8_t dummy[148];
  for (int i = 0; i < 148; i++) dummy[i] = kDummyBurst[i] == '1';
  for (int i = 0; i < 8; i++) {                             // :68-85
    std::vector<trxsig_c32> mod;
    rc = modulate(t, dummy, i, nullptr, mod);
    if (rc != TRXSIG_OK) { trxsig_destroy(t->ctx); delete t; return rc; }
    t->ctl.fillerModulus[i] = 26;
    for (int j = 0; j < 102; j++) t->fillerTable[j][i] = mod;
    t->ctl.chanType[i] = TRXSIG_CHAN_NONE;
    t->haveChan[i] = false;
    t->chanRespOffset[i] = 0.0f;
    t->SNRestimate[i] = 0.0f;
    t->channelEstimateTime[i] = start;
  }
  // device scratch: one burst (<= 157*sps samples) + offsets + results + taps
  t->d_bytes = 8 * (size_t)157 * sps + 4096;
  if (hipMalloc((void **)&t->d, t->d_bytes) != hipSuccess) { trxsig_destroy(t->ctx); delete t; return TRXSIG_EHIP; }
  if (hipHostMalloc((void **)&t->hpin, t->d_bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipFree(t->d); trxsig_destroy(t->ctx); delete t; return TRXSIG_EHIP;
  }
  *out = t;
  return TRXSIG_OK;
}

void trxsig_trx_destroy(trxsig_trx *t) {
  if (!t) return;
  while (!t->queue.empty()) { delete t->queue.top(); t->queue.pop(); }
  if (t->d) (void)hipFree(t->d);
  if (t->hpin) (void)hipHostFree(t->hpin);
  if (t->ctx) trxsig_destroy(t->ctx);
  delete t;
}

const char *trxsig_trx_last_error(const trxsig_trx *t) { return t ? t->err.c_str() : "null transceiver"; }
trxsig_ctx *trxsig_trx_context(trxsig_trx *t) { return t ? t->ctx : nullptr; }
double trxsig_trx_energy_threshold(const trxsig_trx *t) { return t ? t->energyThreshold : 0.0; }
int trxsig_trx_filler_modulus(const trxsig_trx *t, int tn) { return (t && tn >= 0 && tn < 8) ? t->ctl.fillerModulus[tn] : -1; }
int trxsig_trx_queue_size(const trxsig_trx *t) { return t ? (int)t->queue.size() : -1; }
int trxsig_trx_expected_corr_type(const trxsig_trx *t, int tn, int fn) {
  return (t && tn >= 0 && tn < 8) ? t->ctl.expectedCorrType(tn, fn) : TRXSIG_CORR_OFF;
}

int trxsig_trx_control(trxsig_trx *t, const char *buffer, char *response_out, int cap) {
  if (!t || !buffer || !response_out || cap < 1) return TRXSIG_EINVAL;
  char response[100] = {0};
  if (std::strlen(buffer) >= 100) return fail(t, TRXSIG_EINVAL, "control message longer than MAX_PACKET_LENGTH");
  if (!t->ctl.command(buffer, response)) { response_out[0] = 0; return 0; }
  const int n = (int)std::strlen(response);
  if (n + 1 > cap) return fail(t, TRXSIG_EINVAL, "response buffer too small");
  std::memcpy(response_out, response, (size_t)n + 1);
  return n;
}

int trxsig_trx_set_tsc_leg(trxsig_trx *t, int leg) {
  if (!t || (leg != TRXSIG_TSCLEG_EQUALIZE && leg != TRXSIG_TSCLEG_DEMOD)) return TRXSIG_EINVAL;
  t->tscLeg = leg;
  return TRXSIG_OK;
}

int trxsig_trx_pull_radio_vector(trxsig_trx *t, const trxsig_c32 *h_burst, int n, int tn, int fn, float *h_soft,
                                 int *n_soft, int *rssi_out, int *toa_out) {
  if (!t || !h_burst || !h_soft || !n_soft || !rssi_out || !toa_out || tn < 0 || tn > 7 || n <= 0 || n > 157 * t->sps ||
      fn < 0 || fn >= kHyperframe)
    return fail(t, TRXSIG_EINVAL, "trxsig_trx_pull_radio_vector: bad argument");
  const Time now{fn, tn};
  const int corrType = t->ctl.expectedCorrType(tn, fn);
  if (corrType == TRXSIG_CORR_OFF || corrType == TRXSIG_CORR_IDLE) return 0;      // :288-291
  // TRXSIG_TSCLEG_DEMOD: the TSC leg ends in demodulateBurst and keeps no channel cache, as Transceiver52M/Transceiver.cpp
  // does while mMaxExpectedDelay <= 1 (needDFE false: :272, 322, 382)
  const bool needDFE = t->tscLeg == TRXSIG_TSCLEG_EQUALIZE;
  if (corrType == TRXSIG_CORR_TSC && needDFE && t->sps != 1)
    return fail(t, TRXSIG_EINVAL, "the equalising TSC leg (Transceiver.cpp:391-396) needs sps == 1 (see trxsig_trx_set_tsc_leg)");

  // ---- the burst's detection numbers from the GPU: avgPwr of energyDetect, and analyzeTrafficBurst /
  //      detectRACHBurst (stateless, so running them ahead of the energy decision changes nothing) ----
  const int32_t off = 0, len = n;
  uint8_t flags = 0;
  trxsig_c32 amplitude{0.0f, 0.0f};
  float TOA = 0.0f, avgPwr = 0.0f;
  const int nsoft = n / t->sps;
  if (corrType == TRXSIG_CORR_TSC && needDFE)
    TRX_LIB(t, trxsig_detect_demod_normal_host(t->ctx, h_burst, &off, &len, 1, (int)t->ctl.tsc, 3.0f, -1.0f, &flags, &amplitude,
                                               &TOA, &avgPwr, nullptr, 0, 0));
  else if (corrType == TRXSIG_CORR_TSC)
    TRX_LIB(t, trxsig_detect_demod_normal_host(t->ctx, h_burst, &off, &len, 1, (int)t->ctl.tsc, 3.0f, -1.0f, &flags, &amplitude,
                                               &TOA, &avgPwr, h_soft, nsoft, nsoft));     // demodulateBurst(amp, TOA) rides along
  else
    TRX_LIB(t, trxsig_detect_demod_rach_host(t->ctx, h_burst, &off, &len, 1, 5.0f, -1.0f, &flags, &amplitude, &TOA, &avgPwr,
                                             h_soft, nsoft, nsoft));     // demodulateBurst(amp, TOA) rides along
  if (flags & TRXSIG_F_BADLEN) return fail(t, TRXSIG_EINVAL, "burst length not accepted by the detector");

  // ---- energyDetect's decision against the adaptive threshold (:298-306; sigProcLib.cpp:929-931) ----
  const float thrF = (float)t->energyThreshold;
  if (!(avgPwr > thrF * thrF)) {
    const double framesElapsed = time_minus(now, t->prevFalseDetectionTime);
    if (framesElapsed > 50) { t->energyThreshold -= 10.0; t->prevFalseDetectionTime = now; }
    return 0;
  }

  const bool success = (flags & TRXSIG_F_DETECT) != 0;
  if (corrType == TRXSIG_CORR_TSC) {
    // per-timeslot channel / DFE cache (:313-325)
    const double sinceEstimate = time_minus(now, t->channelEstimateTime[tn]);
    bool estimateChannel = false;
    if (sinceEstimate > 50 || !t->haveChan[tn]) { t->haveChan[tn] = false; estimateChannel = true; }
    if (!needDFE) estimateChannel = false;                 // Transceiver52M/Transceiver.cpp:322
    if (success) {
      t->energyThreshold -= 1.0F;                          // :338-339
      if (t->energyThreshold < 0.0) t->energyThreshold = 0.0;
      const float n2 = amplitude.im * amplitude.im + amplitude.re * amplitude.re;          // Complex::norm2 (Complex.h:119)
      t->SNRestimate[tn] = (float)(n2 / (t->energyThreshold * t->energyThreshold + 1.0));  // :340
      if (estimateChannel) {                               // :341-349, on the GPU with this SNR
        char *d = t->d;
        hipStream_t st = (hipStream_t)trxsig_get_stream(t->ctx);
        trxsig_c32 *d_x = (trxsig_c32 *)d;
        char *p = d + 8 * (size_t)157 * t->sps;
        int32_t *d_off = (int32_t *)p, *d_len = (int32_t *)(p + 16);
        uint8_t *d_fl = (uint8_t *)(p + 32);
        trxsig_c32 *d_amp = (trxsig_c32 *)(p + 64), *d_w = (trxsig_c32 *)(p + 128), *d_b = (trxsig_c32 *)(p + 256);
        float *d_toa = (float *)(p + 96), *d_co = (float *)(p + 112);
        TRX_HIP(t, hipMemcpyAsync(d_x, h_burst, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        TRX_HIP(t, hipMemcpyAsync(d_off, &off, 4, hipMemcpyHostToDevice, st));
        TRX_HIP(t, hipMemcpyAsync(d_len, &len, 4, hipMemcpyHostToDevice, st));
        // the SNR estimate goes in as formed above (the reference squares its DOUBLE threshold; the batch
        // kernel's own formula squares a float one)
        if (!(t->SNRestimate[tn] > 0.0f)) return fail(t, TRXSIG_EINVAL, "SNR estimate is not positive");
        TRX_LIB(t, trxsig_estimate_dfe_batch(t->ctx, d_x, d_off, d_len, 1, (int)t->ctl.tsc, 3.0f, -1.0f, t->SNRestimate[tn], 0, 0,
                                             d_fl, d_amp, d_toa, d_co, d_w, d_b));
        uint8_t efl = 0;
        TRX_HIP(t, hipMemcpyAsync(&efl, d_fl, 1, hipMemcpyDeviceToHost, st));
        TRX_HIP(t, hipMemcpyAsync(&t->chanRespOffset[tn], d_co, 4, hipMemcpyDeviceToHost, st));
        TRX_HIP(t, hipMemcpyAsync(t->dfeW[tn], d_w, 8 * 7, hipMemcpyDeviceToHost, st));
        TRX_HIP(t, hipMemcpyAsync(t->dfeB[tn], d_b, 8 * 5, hipMemcpyDeviceToHost, st));
        TRX_HIP(t, hipStreamSynchronize(st));
        if (!(efl & TRXSIG_F_DETECT)) return fail(t, TRXSIG_EHIP, "channel estimate disagrees with the detector");
        t->haveChan[tn] = true;
        t->channelEstimateTime[tn] = now;
      }
    } else {
      const double framesElapsed = time_minus(now, t->prevFalseDetectionTime);          // :352-357
      t->energyThreshold += 10.0F * std::exp(-framesElapsed);
      t->prevFalseDetectionTime = now;
      t->haveChan[tn] = false;
    }
  } else {
    if (success) {                                         // :367-371
      t->energyThreshold -= 1.0F;
      if (t->energyThreshold < 0.0) t->energyThreshold = 0.0;
      t->haveChan[tn] = false;
    } else {                                               // :372-376
      const double framesElapsed = time_minus(now, t->prevFalseDetectionTime);
      t->energyThreshold += 10.0F * std::exp(-framesElapsed);
      t->prevFalseDetectionTime = now;
    }
  }
  if (!success) return 0;

  if (corrType == TRXSIG_CORR_TSC && needDFE) {
    // scaleVector(burst, 1/amp); equalizeBurst(burst, TOA - chanRespOffset[ts], sps, w[ts], b[ts]) (:391-396)
    char *d = t->d;
    hipStream_t st = (hipStream_t)trxsig_get_stream(t->ctx);
    trxsig_c32 *d_x = (trxsig_c32 *)d;
    char *p = d + 8 * (size_t)157 * t->sps;
    int32_t *d_off = (int32_t *)p, *d_len = (int32_t *)(p + 16);
    uint8_t *d_fl = (uint8_t *)(p + 32);
    trxsig_c32 *d_amp = (trxsig_c32 *)(p + 64), *d_w = (trxsig_c32 *)(p + 128), *d_b = (trxsig_c32 *)(p + 256);
    float *d_toa = (float *)(p + 96), *d_soft = (float *)(p + 512);
    const float toa_eq = TOA - t->chanRespOffset[tn];
    const uint8_t en = TRXSIG_F_DETECT;
    // everything the step needs in ONE host-to-device copy from the pinned mirror (same layout as d), the soft bits back
    // in one (134 -> 94 us per call)
    {
      char *m = t->hpin, *mp = m + (p - d);
      std::memcpy(m, h_burst, 8 * (size_t)n);
      std::memcpy(mp, &off, 4); std::memcpy(mp + 16, &len, 4); std::memcpy(mp + 32, &en, 1);
      std::memcpy(mp + 64, &amplitude, 8); std::memcpy(mp + 96, &toa_eq, 4);
      std::memcpy(mp + 128, t->dfeW[tn], 56); std::memcpy(mp + 256, t->dfeB[tn], 40);
      TRX_HIP(t, hipMemcpyAsync(d, m, (size_t)(p - d) + 512, hipMemcpyHostToDevice, st));
    }
    TRX_LIB(t, trxsig_equalize_taps_batch(t->ctx, d_x, d_off, d_len, 1, d_amp, d_toa, d_fl, d_w, d_b, d_soft, nullptr, nsoft, 160));
    TRX_HIP(t, hipMemcpyAsync(t->hpin + (p - d) + 512, d_soft, 4 * (size_t)nsoft, hipMemcpyDeviceToHost, st));
    TRX_HIP(t, hipStreamSynchronize(st));
    std::memcpy(h_soft, t->hpin + (p - d) + 512, 4 * (size_t)nsoft);
  }
  *n_soft = nsoft;
  // :400-402 -- Complex::abs() is (float)sqrt((double)norm2) (Complex.h:131)
  const float n2 = amplitude.im * amplitude.im + amplitude.re * amplitude.re;
  const float absA = (float)std::sqrt((double)n2);
  *rssi_out = (int)std::floor(20.0 * std::log10(9450.0 / absA));
  *toa_out = (int)std::round(TOA * 256.0 / t->sps);
  return 1;
}

int trxsig_trx_encode_rx_datagram(int tn, int fn, int rssi, int toa, const float *soft, int n_soft, uint8_t *out) {
  if (!soft || !out || n_soft < 148) return TRXSIG_EINVAL;
  out[0] = (uint8_t)tn;                                     // :658-672
  for (int i = 0; i < 4; i++) out[1 + i] = (uint8_t)((fn >> ((3 - i) * 8)) & 0xff);
  out[5] = (uint8_t)rssi;
  out[6] = (uint8_t)((toa >> 8) & 0xff);
  out[7] = (uint8_t)(toa & 0xff);
  for (int i = 0; i < 148; i++) out[8 + i] = (uint8_t)(int)std::round((double)soft[i] * 255.0);   // (char) round(x*255.0)
  out[156] = 0;                                             // burstString[gSlotLen+9] = '\0'; [gSlotLen+8] is never written:
  out[157] = 0;                                             // sent as 0 here
  return TRXSIG_OK;
}

int trxsig_trx_decode_tx_datagram(const uint8_t *in, int len, int *tn, int *fn, int *rssi, uint8_t *bits) {
  if (!in || !tn || !fn || !rssi || !bits) return TRXSIG_EINVAL;
  if (len != TRXSIG_TX_DATAGRAM_BYTES) return TRXSIG_EINVAL;    // "badly formatted packet" (:590-593)
  *tn = (int)(signed char)in[0];
  unsigned long long frameNum = 0;
  for (int i = 0; i < 4; i++) frameNum = (frameNum << 8) | (0x0ff & in[i + 1]);
  // A frame number lives in [0, gHyperframe) (GSM/GSMCommon.h:306).  The reference takes whatever 32-bit value the
  // datagram holds (Transceiver.cpp:597-600) and would index its filler table with it; behind a UDP socket that is
  // an out-of-bounds write, so a datagram with such a value is "badly formatted" here.
  if (frameNum >= (unsigned long long)kHyperframe) return TRXSIG_EINVAL;
  *fn = (int)frameNum;
  *rssi = (int)(signed char)in[5];                          // (int) buffer[5], char buffer
  std::memcpy(bits, in + 6, 148);
  return TRXSIG_OK;
}

int trxsig_trx_add_radio_vector(trxsig_trx *t, const uint8_t *bits, int RSSI, int tn, int fn) {
  if (!t || !bits || tn < 0 || tn > 7 || fn < 0 || fn >= kHyperframe)
    return fail(t, TRXSIG_EINVAL, "trxsig_trx_add_radio_vector: bad argument (tn 0..7, fn 0..gHyperframe-1)");
  // scaleVector(*modBurst, pow(10,-RSSI/10)): integer division, double pow, complex(float) scale (:108)
  const float gain = (float)std::pow(10, -RSSI / 10);
  Queued *q = new (std::nothrow) Queued;
  if (!q) return fail(t, TRXSIG_ENOMEM, "trxsig_trx_add_radio_vector: out of memory");
  q->time = Time{fn, tn};
  int rc = modulate(t, bits, tn, &gain, q->samples);
  if (rc != TRXSIG_OK) { delete q; return rc; }
  t->queue.push(q);                                         // mTransmitPriorityQueue.write(newVec) (:109)
  return TRXSIG_OK;
}

int trxsig_trx_push_radio_vector(trxsig_trx *t, int tn, int fn, trxsig_c32 *h_out, int *n_out, int *from_queue) {
  if (!t || !h_out || !n_out || tn < 0 || tn > 7 || fn < 0 || fn >= kHyperframe)
    return fail(t, TRXSIG_EINVAL, "trxsig_trx_push_radio_vector: bad argument (tn 0..7, fn 0..gHyperframe-1)");
  const Time now{fn, tn};
  // dump stale bursts into the filler table (:142-153)
  while (!t->queue.empty() && time_less(t->queue.top()->time, now)) {
    Queued *q = t->queue.top();
    t->queue.pop();
    const int modFN = q->time.fn % t->ctl.fillerModulus[q->time.tn];
    t->fillerTable[modFN][q->time.tn] = std::move(q->samples);
    delete q;
  }
  const int modFN = fn % t->ctl.fillerModulus[tn];
  int fq = 0;
  if (!t->queue.empty() && time_equal(t->queue.top()->time, now)) {         // :159-173
    Queued *q = t->queue.top();
    t->queue.pop();
    t->fillerTable[modFN][tn] = std::move(q->samples);
    delete q;
    fq = 1;
  }
  const std::vector<trxsig_c32> &v = t->fillerTable[modFN][tn];             // :175-177 (or the burst just stored)
  std::memcpy(h_out, v.data(), 8 * v.size());
  *n_out = (int)v.size();
  if (from_queue) *from_queue = fq;
  return TRXSIG_OK;
}

// ---- driveTransmitFIFO's controller (:679-729) ----

void trxsig_txclock_init(trxsig_txclock *c, int start_fn, int start_tn, int latency_fn, int latency_tn) {
  if (!c) return;
  c->deadline_fn = c->latency_update_fn = c->last_clock_fn = start_fn;
  c->deadline_tn = c->latency_update_tn = c->last_clock_tn = start_tn;
  c->latency_fn = latency_fn; c->latency_tn = latency_tn;
}

int trxsig_txclock_advance(trxsig_txclock *c, int radio_fn, int radio_tn, int *underrun, int max_slots, int *push_fn, int *push_tn) {
  if (!c || !push_fn || !push_tn || max_slots <= 0) return TRXSIG_EINVAL;
  const Time radio{radio_fn, radio_tn};
  Time deadline{c->deadline_fn, c->deadline_tn}, latency{c->latency_fn, c->latency_tn}, upd{c->latency_update_fn, c->latency_update_tn};
  *push_fn = deadline.fn; *push_tn = deadline.tn;
  int n = 0;
  while (n < max_slots && time_greater(time_plus(radio, latency), deadline)) {
    if (underrun && *underrun) {                             // isUnderrun() reads and clears the flag (radioInterface.h:172)
      *underrun = 0;
      // only do latency update every 10 frames, so we don't over update (:697-703)
      if (time_greater(radio, time_plus(upd, Time{10, 0}))) { latency = time_plus(latency, Time{1, 0}); upd = radio; }
    } else if (time_greater(latency, Time{1, 1})) {          // no under-run for a second (216 frames): one timeslot less (:705-714)
      if (time_greater(radio, time_plus(upd, Time{216, 0}))) { time_dec_tn(latency); upd = radio; }
    }
    time_inc_tn(deadline);                                   // pushRadioVector(mTransmitDeadlineClock); mTransmitDeadlineClock.incTN() (:716-717)
    n++;
  }
  c->deadline_fn = deadline.fn; c->deadline_tn = deadline.tn;
  c->latency_fn = latency.fn; c->latency_tn = latency.tn;
  c->latency_update_fn = upd.fn; c->latency_update_tn = upd.tn;
  return n;
}

int trxsig_txclock_indication_due(const trxsig_txclock *c) {
  if (!c) return 0;
  return time_greater(Time{c->deadline_fn, c->deadline_tn}, time_plus(Time{c->last_clock_fn, c->last_clock_tn}, Time{216, 0})) ? 1 : 0;
}

int trxsig_txclock_indication(trxsig_txclock *c, char *msg, int cap) {
  if (!c || !msg) return TRXSIG_EINVAL;
  char tmp[50];
  const int n = std::snprintf(tmp, sizeof tmp, "IND CLOCK %llu", (unsigned long long)(c->deadline_fn + 20));
  if (n + 1 > cap) return TRXSIG_EINVAL;
  std::memcpy(msg, tmp, (size_t)n + 1);
  c->last_clock_fn = c->deadline_fn; c->last_clock_tn = c->deadline_tn;
  return n;
}

int trxsig_create_lpf_host(const float *raw, int len, float gainDC, float *out) {
  if (!raw || !out || len <= 0) return TRXSIG_EINVAL;
  double sum = 0.0;                                         // sigProcLib.cpp:1119-1139
  for (int i = 0; i < len; i++) sum += raw[i];
  const float normFactor = (float)(gainDC / sum);           // :1141
  for (int i = 0; i < len; i++) out[i] = raw[i] * normFactor;
  return TRXSIG_OK;
}

}  // extern "C"
