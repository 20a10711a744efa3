// trxsig_api.cpp -- the C-ABI of libtrxsig (include/trxsig.h): context, tables, workspace and the
// batch entry points that enqueue the gfx950 kernels.  No signal processing happens on the host
// here except the init-time table construction (trxsig_tablegen.cpp); there is no CPU fallback.
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <string>
#include <vector>

#include "trxsig.h"
#include "trxsig_ctx.h"
#include "trxsig_launch.h"
#include "trxsig_tablegen.h"

struct EventProfiler : TrxProfiler {
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void begin(int id, hipStream_t st) override {
    Rec r = { id, get(), get() };
    if (r.a) (void)hipEventRecord(r.a, st);
    recs.push_back(r);
  }
  void end(int id, hipStream_t st) override {
    if (!recs.empty() && recs.back().id == id && recs.back().b) (void)hipEventRecord(recs.back().b, st);
  }
  void collect(float *ms, int *n) {
    for (Rec &r : recs) {
      float t = 0;
      if (r.a && r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
        ms[r.id] += t; n[r.id]++;
      }
      if (r.a) pool.push_back(r.a);
      if (r.b) pool.push_back(r.b);
    }
    recs.clear();
  }
  ~EventProfiler() override {
    float ms[TRXSIG_K_COUNT] = {0}; int n[TRXSIG_K_COUNT] = {0};
    collect(ms, n);
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  }
};

struct trxsig_ctx {
  EventProfiler *prof = nullptr;
  int device = -1;
  int sps = 0;
  hipStream_t stream = nullptr;
  TrxTables *h_tables = nullptr;     // host copy
  TrxTables *d_tables = nullptr;     // device blob
  // workspace (device)
  int cap_bursts = 0;
  trx_c32 *d_rec = nullptr;          // [slots][cap_bursts] detect -> peak records
  trx_c32 *d_rec2 = nullptr;         // a second one (cap2_bursts): the Transceiver group's access-burst class beside its normal-burst classes
  int cap2_bursts = 0;
  // equaliser scratch: toa_eq [B], w [B*7], b [B*5], xd [B*160]
  int eq_cap = 0;
  char *d_eq = nullptr;
  // staging for the *_host wrappers
  size_t stage_bytes = 0;
  void *d_stage = nullptr;
  size_t pin_bytes = 0;              // pinned host mirror of the staging area (small host calls: one DMA each way)
  void *h_pin = nullptr;
  // TRXSIG_TUNE_DEMOD_BESIDE: the demodulator on a side stream, from one of two private copies of (flags, amp, TOA)
  int children = 0;                  // front ends, back ends and groups living on this context (trx_ctx_retain / _release)
  bool zombie = false;               // trxsig_destroy came while some were alive: the last one to go destroys the context
  int demod_beside = 0;
  int soft_mode = 0;                 // trxsig_set_soft_mode: TRXSIG_SOFT_EXACT (0) / TRXSIG_SOFT_TOLERANCE (1), demodulateBurst's arithmetic
  int det_cus = 0;                   // TRXSIG_TUNE_BESIDE_DET_CUS: > 0 = the detectors on their own stream masked to that many CUs, the
                                     // demodulator's side stream masked to the others (hipExtStreamCreateWithCUMask); 0 = no masks
  int cu_layout = 0;                 // TRXSIG_TUNE_CU_LAYOUT: which bits of the mask the two sets take (see cu_mask())
  hipStream_t det = nullptr;         // the masked detect stream (det_cus > 0)
  hipEvent_t ev_in = nullptr;        // "the caller's inputs are ready" (context's stream -> detect stream)
  bool det_in_flight = false;        // detectors issued on `det` since the last join
  int side_prio = 0;                 // TRXSIG_TUNE_BESIDE_PRIORITY: 0 = the side stream at normal priority, 1 = highest, 2 = lowest
  int beside_nodeps = 0;             // tuning build, TIMING EXPERIMENT ONLY (key 10; results are racy): 1 = no "inputs ready" wait,
                                     // 2 = no cross-stream waits at all -- what the dependencies themselves cost (tools/cu_split.py)
  hipStream_t side = nullptr;
  hipEvent_t ev_pk[2] = {nullptr, nullptr}, ev_dm[2] = {nullptr, nullptr};
  bool dm_in_flight[2] = {false, false};
  uint8_t *pb_flags[2] = {nullptr, nullptr};
  trx_c32 *pb_amp[2] = {nullptr, nullptr};
  float *pb_toa[2] = {nullptr, nullptr};
  int pb_cap = 0, pb_k = 0;
  int rach_variant = 2;              // 2 = k_rach_front + k_rach_peak2 (approximate-then-exact, bisection in its own kernel), 1 = k_rach_fast alone, 0 = exact at every lag
  int variant = 0;                   // normal-burst path (TRXSIG_TUNE_NORMAL_PATH / env TRXSIG_TSC_VARIANT)
  int spec_peak = 0;                 // peak kernel of path 0: 0 = k_tsc_peak2 (2 lanes per burst), 1 = k_tsc_peak8 (8, speculated), 2 = k_tsc_peak (1)
  int generic_taps = 0;              // 1: correlators without the tap-class specialisation (TRXSIG_TUNE_GENERIC_TAPS)
  // single-launch chain (trxsig_chain.hip): per-burst hand-over granules (all tags clear between launches) and a
  // host-visible word that a demodulator raises when its bounded wait runs out
  int det_cap = 0;
  void *d_det = nullptr;
  unsigned *h_chain_status = nullptr; // pinned, mapped
  unsigned *d_chain_status = nullptr; // the same word as the device sees it
  int chain_lag = 48;                 // tiles (per stream) between a tile's detect and demodulate workgroups
  unsigned chain_spin = 200000;       // polls (x ~0.25 us) before a demodulator gives up
  float rach_amp_err = 0.0f;          // trx_rach_amp_err(h_tables): error bar of k_rach_*'s approximate correlation
  int chain_dbg = 0;                  // timing experiments (env TRXSIG_CHAIN_DBG): 1 = no detect role, 2 = no demodulate role
  bool chain_broken = false;          // a wait ran out once: three launches from then on
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint8_t *d_tsc = nullptr;          // 8 x 26 training-sequence bits (XCCH encoder), uploaded on first use
  std::string err;
};

namespace {

int fail(trxsig_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
  }
  return code;
}

}  // namespace
// for the other host translation units of the library (trxsig_frontend.cpp)
int trx_ctx_fail(trxsig_ctx *c, int code, const char *what, hipError_t e) { return fail(c, code, what, e); }
TrxProfiler *trx_ctx_profiler(trxsig_ctx *c) { return c ? c->prof : nullptr; }
namespace {

#define HIPCHK(c, call)                                          \
  do {                                                           \
    hipError_t e_ = (call);                                      \
    if (e_ != hipSuccess) return fail((c), TRXSIG_EHIP, #call, e_); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int check_device(int device, std::string &why) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { why = "no HIP device visible (libtrxsig has no CPU fallback)"; return TRXSIG_ENODEV; }
  if (device < 0 || device >= n) { why = "device ordinal out of range"; return TRXSIG_ENODEV; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) { why = "hipGetDeviceProperties failed"; return TRXSIG_ENODEV; }
  if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    why = std::string("device is ") + p.gcnArchName + ", libtrxsig is built for gfx950 only";
    return TRXSIG_ENODEV;
  }
  return TRXSIG_OK;
}

int finish_create(trxsig_ctx *c) {
#ifdef TRX_TUNING_BUILD
  if (const char *v = std::getenv("TRXSIG_TSC_VARIANT")) c->variant = std::atoi(v);
  if (const char *v = std::getenv("TRXSIG_CHAIN_DBG")) c->chain_dbg = std::atoi(v);
#endif
  if (const char *v = std::getenv("TRXSIG_RACH_VARIANT")) {
    c->rach_variant = std::atoi(v);
#ifndef TRX_TUNING_BUILD
    if (c->rach_variant < 1 || c->rach_variant > 2) c->rach_variant = 2;
#endif
  }
  c->rach_amp_err = trx_rach_amp_err(c->h_tables);
  HIPCHK(c, hipEventCreate(&c->ev0));
  HIPCHK(c, hipEventCreate(&c->ev1));
  return TRXSIG_OK;
}

int ensure_ws(trxsig_ctx *c, int B) {
  if (B <= c->cap_bursts) return TRXSIG_OK;
  int cap = (B + 255) & ~255;
  if (c->d_rec) { HIPCHK(c, hipFree(c->d_rec)); c->d_rec = nullptr; c->cap_bursts = 0; }
  size_t per_burst = sizeof(trx_c32) * (size_t)trx_rec_slots(c->sps);
  const size_t rach = sizeof(float) * (size_t)trx_rach_rec_floats(c->sps);
  if (rach > per_burst) per_burst = rach;
  HIPCHK(c, hipMalloc((void **)&c->d_rec, per_burst * cap));
  c->cap_bursts = cap;
  return TRXSIG_OK;
}

int ensure_ws2(trxsig_ctx *c, int B) {
  if (B <= c->cap2_bursts) return TRXSIG_OK;
  const int cap = (B + 255) & ~255;
  if (c->d_rec2) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(c->d_rec2)); c->d_rec2 = nullptr; c->cap2_bursts = 0; }
  size_t per_burst = sizeof(trx_c32) * (size_t)trx_rec_slots(c->sps);
  const size_t rach = sizeof(float) * (size_t)trx_rach_rec_floats(c->sps);
  if (rach > per_burst) per_burst = rach;
  HIPCHK(c, hipMalloc((void **)&c->d_rec2, per_burst * cap));
  c->cap2_bursts = cap;
  return TRXSIG_OK;
}

// the context's stream waits for every demodulator still running on the side stream (TRXSIG_TUNE_DEMOD_BESIDE)
int join_demod(trxsig_ctx *c) {
  if (c->det_in_flight) {                                   // (flags / amp / TOA of the last call are written on the detect stream)
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_pk[c->pb_k], 0));
    c->det_in_flight = false;
  }
  for (int k = 0; k < 2; k++) {
    if (!c->dm_in_flight[k]) continue;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_dm[k], 0));
    c->dm_in_flight[k] = false;
  }
  return TRXSIG_OK;
}
// CU masks of the two streams.  `n_cu` CUs in all; the detect set takes `det` of them, the demodulate set the rest.
// layout 0: the detect set is mask bits 0 .. det-1 (the driver deals mask bits out to the XCDs in turn, so a run of low bits is
//           spread evenly over the eight XCDs); layout 1: bit i belongs to the detect set when i mod 8 < det / (n_cu / 8), i.e.
//           if the bits were XCD-major instead, the same even spread -- the measurement says which reading holds
void cu_mask(int n_cu, int det, int layout, bool detect_set, uint32_t *mask, int words) {
  for (int w = 0; w < words; w++) mask[w] = 0;
  for (int i = 0; i < n_cu && i < 32 * words; i++) {
    bool in_det;
    if (layout == 0) in_det = i < det;
    else { const int per = n_cu / 8; in_det = (i % per) < det / 8; }
    if (in_det == detect_set) mask[i >> 5] |= 1u << (i & 31);
  }
}
int drop_beside_streams(trxsig_ctx *c) {
  if (c->side) { HIPCHK(c, hipStreamSynchronize(c->side)); HIPCHK(c, hipStreamDestroy(c->side)); c->side = nullptr; }
  if (c->det) { HIPCHK(c, hipStreamSynchronize(c->det)); HIPCHK(c, hipStreamDestroy(c->det)); c->det = nullptr; }
  c->dm_in_flight[0] = c->dm_in_flight[1] = false;
  c->det_in_flight = false;
  return TRXSIG_OK;
}
int ensure_beside(trxsig_ctx *c, int B) {
  if (!c->side) {
    if (c->det_cus > 0) {
      hipDeviceProp_t p;
      HIPCHK(c, hipGetDeviceProperties(&p, c->device));
      const int n_cu = p.multiProcessorCount;
      if (c->det_cus >= n_cu || n_cu > 512) return fail(c, TRXSIG_EINVAL, "TRXSIG_TUNE_BESIDE_DET_CUS: more CUs than the device has");
      uint32_t m[16];
      const int words = (n_cu + 31) / 32;
      cu_mask(n_cu, c->det_cus, c->cu_layout, true, m, words);
      HIPCHK(c, hipExtStreamCreateWithCUMask(&c->det, (uint32_t)words, m));
      cu_mask(n_cu, c->det_cus, c->cu_layout, false, m, words);
      HIPCHK(c, hipExtStreamCreateWithCUMask(&c->side, (uint32_t)words, m));
    } else if (c->side_prio) {
      int lo = 0, hi = 0;                                   // (numerically: `hi` is the greatest priority, the smaller number)
      HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIPCHK(c, hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, c->side_prio == 1 ? hi : lo));
    } else {
      HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    }
    if (!c->ev_in) {
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
      for (int k = 0; k < 2; k++) {
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_pk[k], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_dm[k], hipEventDisableTiming));
      }
    }
  }
  if (B <= c->pb_cap) return TRXSIG_OK;
  HIPCHK(c, hipStreamSynchronize(c->side));
  if (c->det) HIPCHK(c, hipStreamSynchronize(c->det));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int cap = (B + 255) & ~255;
  for (int k = 0; k < 2; k++) {
    if (c->pb_flags[k]) { HIPCHK(c, hipFree(c->pb_flags[k])); c->pb_flags[k] = nullptr; }
    if (c->pb_amp[k]) { HIPCHK(c, hipFree(c->pb_amp[k])); c->pb_amp[k] = nullptr; }
    if (c->pb_toa[k]) { HIPCHK(c, hipFree(c->pb_toa[k])); c->pb_toa[k] = nullptr; }
    c->dm_in_flight[k] = false;
  }
  c->pb_cap = 0;
  for (int k = 0; k < 2; k++) {
    HIPCHK(c, hipMalloc((void **)&c->pb_flags[k], (size_t)cap));
    HIPCHK(c, hipMalloc((void **)&c->pb_amp[k], sizeof(trx_c32) * (size_t)cap));
    HIPCHK(c, hipMalloc((void **)&c->pb_toa[k], sizeof(float) * (size_t)cap));
  }
  c->pb_cap = cap;
  return TRXSIG_OK;
}

#ifdef TRX_TUNING_BUILD
int ensure_chain(trxsig_ctx *c, int B) {
  if (!c->h_chain_status) {
    HIPCHK(c, hipHostMalloc((void **)&c->h_chain_status, 64, hipHostMallocMapped));
    *c->h_chain_status = 0;
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_chain_status, c->h_chain_status, 0));
  }
  if (B <= c->det_cap) return TRXSIG_OK;
  const int cap = (B + 255) & ~255;
  if (c->d_det) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->d_det)); c->d_det = nullptr; c->det_cap = 0; }
  HIPCHK(c, hipMalloc(&c->d_det, trx_chain_ws_bytes(cap)));
  HIPCHK(c, hipMemsetAsync(c->d_det, 0, trx_chain_ws_bytes(cap), c->stream));
  c->det_cap = cap;
  return TRXSIG_OK;
}

// a demodulator of an earlier chain launch gave up its wait (never observed; HIP does not promise the dispatch
// order the chain relies on for progress): that call's soft bits are incomplete.  Report it once, clear the
// hand-over words, and use the three-launch path from here on.
#endif

int chain_check(trxsig_ctx *c) {
  if (!c->h_chain_status || !*c->h_chain_status) return TRXSIG_OK;
#ifndef TRX_TUNING_BUILD
  return TRXSIG_OK;
#else
  (void)hipStreamSynchronize(c->stream);
  *c->h_chain_status = 0;
  c->chain_broken = true;
  if (c->d_det) (void)hipMemsetAsync(c->d_det, 0, trx_chain_ws_bytes(c->det_cap), c->stream);
  return fail(c, TRXSIG_EHIP, "an earlier trxsig_detect_demod_normal_batch (single-launch path) timed out waiting for its "
                              "detect workgroups; its soft bits are incomplete.  Falling back to the three-launch path");
#endif
}

int ensure_stage(trxsig_ctx *c, size_t bytes) {
  if (bytes <= c->stage_bytes) return TRXSIG_OK;
  if (c->d_stage) { HIPCHK(c, hipFree(c->d_stage)); c->d_stage = nullptr; c->stage_bytes = 0; }
  bytes = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;
  HIPCHK(c, hipMalloc(&c->d_stage, bytes));
  c->stage_bytes = bytes;
  return TRXSIG_OK;
}

int ensure_pin(trxsig_ctx *c, size_t bytes) {
  if (bytes <= c->pin_bytes) return TRXSIG_OK;
  if (c->h_pin) { HIPCHK(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->pin_bytes = 0; }
  bytes = (bytes + 0xFFFF) & ~(size_t)0xFFFF;
  HIPCHK(c, hipHostMalloc(&c->h_pin, bytes, hipHostMallocDefault));
  c->pin_bytes = bytes;
  return TRXSIG_OK;
}

bool bad_batch(const void *s, const void *o, const void *l, int B) { return B < 0 || (B > 0 && (!s || !o || !l)); }

}  // namespace

extern "C" {

int trxsig_abi_version(void) { return TRXSIG_ABI_VERSION; }

int trxsig_create(trxsig_ctx **out, int device, int sps) {
  if (!out) return TRXSIG_EINVAL;
  *out = nullptr;
  if (!(sps == 1 || sps == 2 || sps == 4)) return TRXSIG_EINVAL;
  std::string why;
  int rc = check_device(device, why);
  if (rc != TRXSIG_OK) { std::fprintf(stderr, "trxsig_create: %s\n", why.c_str()); return rc; }
  trxsig_ctx *c = new (std::nothrow) trxsig_ctx;
  if (!c) return TRXSIG_ENOMEM;
  c->device = device; c->sps = sps;
  DeviceGuard g(device);
  c->h_tables = (TrxTables *)std::malloc(sizeof(TrxTables));
  if (!c->h_tables || trx_build_tables(c->h_tables, sps) != 0) { trxsig_destroy(c); return TRXSIG_ENOMEM; }
  if (hipMalloc((void **)&c->d_tables, sizeof(TrxTables)) != hipSuccess ||
      hipMemcpy(c->d_tables, c->h_tables, sizeof(TrxTables), hipMemcpyHostToDevice) != hipSuccess ||
      finish_create(c) != TRXSIG_OK) {
    std::fprintf(stderr, "trxsig_create: device allocation/upload failed\n");
    trxsig_destroy(c);
    return TRXSIG_EHIP;
  }
  *out = c;
  return TRXSIG_OK;
}

int trxsig_create_from_tables(trxsig_ctx **out, int device, const void *d_blob, size_t bytes) {
  if (!out || !d_blob || bytes != sizeof(TrxTables)) return TRXSIG_EINVAL;
  *out = nullptr;
  std::string why;
  int rc = check_device(device, why);
  if (rc != TRXSIG_OK) { std::fprintf(stderr, "trxsig_create_from_tables: %s\n", why.c_str()); return rc; }
  trxsig_ctx *c = new (std::nothrow) trxsig_ctx;
  if (!c) return TRXSIG_ENOMEM;
  c->device = device;
  DeviceGuard g(device);
  c->h_tables = (TrxTables *)std::malloc(sizeof(TrxTables));
  if (!c->h_tables) { trxsig_destroy(c); return TRXSIG_ENOMEM; }
  if (hipMalloc((void **)&c->d_tables, sizeof(TrxTables)) != hipSuccess ||
      hipMemcpy(c->d_tables, d_blob, sizeof(TrxTables), hipMemcpyDeviceToDevice) != hipSuccess ||
      hipMemcpy(c->h_tables, c->d_tables, sizeof(TrxTables), hipMemcpyDeviceToHost) != hipSuccess) {
    trxsig_destroy(c);
    return TRXSIG_EHIP;
  }
  if (!trx_tables_valid(c->h_tables)) {
    std::fprintf(stderr, "trxsig_create_from_tables: blob failed validation (magic/version/checksum)\n");
    trxsig_destroy(c);
    return TRXSIG_EINVAL;
  }
  c->sps = (int)c->h_tables->sps;
  if (finish_create(c) != TRXSIG_OK) { trxsig_destroy(c); return TRXSIG_EHIP; }
  *out = c;
  return TRXSIG_OK;
}

static void destroy_now(trxsig_ctx *c);
extern "C++" {
void trx_ctx_retain(trxsig_ctx *c) { if (c) c->children++; }
void trx_ctx_release(trxsig_ctx *c) {
  if (c && --c->children == 0 && c->zombie) destroy_now(c);
}
}
void trxsig_destroy(trxsig_ctx *c) {
  if (!c) return;
  if (c->children > 0) { c->zombie = true; return; }        // objects created on it are still alive: they keep it until they go
  destroy_now(c);
}
static void destroy_now(trxsig_ctx *c) {
  {
    DeviceGuard g(c->device);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_rec) (void)hipFree(c->d_rec);
    if (c->d_rec2) (void)hipFree(c->d_rec2);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_eq) (void)hipFree(c->d_eq);
    if (c->d_det) (void)hipFree(c->d_det);
    if (c->h_chain_status) (void)hipHostFree(c->h_chain_status);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->d_tsc) (void)hipFree(c->d_tsc);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->det) { (void)hipStreamSynchronize(c->det); (void)hipStreamDestroy(c->det); }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (int k = 0; k < 2; k++) {
      if (c->ev_pk[k]) (void)hipEventDestroy(c->ev_pk[k]);
      if (c->ev_dm[k]) (void)hipEventDestroy(c->ev_dm[k]);
      if (c->pb_flags[k]) (void)hipFree(c->pb_flags[k]);
      if (c->pb_amp[k]) (void)hipFree(c->pb_amp[k]);
      if (c->pb_toa[k]) (void)hipFree(c->pb_toa[k]);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c->prof;
  }
  std::free(c->h_tables);
  delete c;
}

int trxsig_sps(const trxsig_ctx *c) { return c ? c->sps : TRXSIG_EINVAL; }
int trxsig_device(const trxsig_ctx *c) { return c ? c->device : TRXSIG_EINVAL; }
int trxsig_live_children(const trxsig_ctx *c) { return c ? c->children : TRXSIG_EINVAL; }
int trxsig_set_stream(trxsig_ctx *c, void *s) { if (!c) return TRXSIG_EINVAL; c->stream = (hipStream_t)s; return TRXSIG_OK; }
void *trxsig_get_stream(trxsig_ctx *c) { return c ? (void *)c->stream : nullptr; }
int trxsig_get_device(trxsig_ctx *c) { return c ? c->device : -1; }
int trxsig_synchronize(trxsig_ctx *c) {
  if (!c) return TRXSIG_EINVAL;
  {
    DeviceGuard g(c->device);
    int rc = join_demod(c);
    if (rc != TRXSIG_OK) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return chain_check(c);
}
const char *trxsig_last_error(const trxsig_ctx *c) { return c ? c->err.c_str() : "null context"; }

int trxsig_reserve(trxsig_ctx *c, int max_bursts) {
  if (!c || max_bursts < 0) return TRXSIG_EINVAL;
  DeviceGuard g(c->device);
  return ensure_ws(c, max_bursts);
}

size_t trxsig_tables_bytes(int sps) { (void)sps; return sizeof(TrxTables); }
int trxsig_tables_build_host(int sps, void *h_buf, size_t cap) {
  if (!h_buf || cap < sizeof(TrxTables)) return TRXSIG_EINVAL;
  return trx_build_tables((TrxTables *)h_buf, sps) == 0 ? TRXSIG_OK : TRXSIG_EINVAL;
}
void *trxsig_tables_device(trxsig_ctx *c) { return c ? (void *)c->d_tables : nullptr; }
int trxsig_tables_export(trxsig_ctx *c, void *h_buf, size_t cap) {
  if (!c || !h_buf || cap < sizeof(TrxTables)) return TRXSIG_EINVAL;
  DeviceGuard g(c->device);
  HIPCHK(c, hipMemcpy(h_buf, c->d_tables, sizeof(TrxTables), hipMemcpyDeviceToHost));
  return TRXSIG_OK;
}

int trxsig_tables_view_get(const trxsig_ctx *c, trxsig_tables_view *v) {
  if (!c || !v) return TRXSIG_EINVAL;
  const TrxTables *T = c->h_tables;
  v->sps = c->sps;
  v->cos_table = T->cosT; v->sin_table = T->sinT;
  v->gmsk_rotation = (const trxsig_c32 *)T->rot; v->gmsk_reverse = (const trxsig_c32 *)T->rev;
  v->gsm_pulse = T->pulse;
  for (int t = 0; t < 8; t++) {
    v->midamble[t] = (const trxsig_c32 *)T->mid[t];
    v->midamble_toa[t] = T->mid_toa[t];
    v->midamble_gain[t].re = T->mid_gain[t].r; v->midamble_gain[t].im = T->mid_gain[t].i;
  }
  v->rach = (const trxsig_c32 *)T->rach;
  v->rach_toa = T->rach_toa;
  v->rach_gain.re = T->rach_gain.r; v->rach_gain.im = T->rach_gain.i;
  return TRXSIG_OK;
}

// ---- RX hot path ---------------------------------------------------------------------------------
int trxsig_detect_demod_normal_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset,
                                     const int32_t *d_length, int B, int tsc, float detect_thresh,
                                     float energy_thresh, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                     float *d_avgpwr, float *d_soft, uint8_t *d_hard, int nsoft,
                                     int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_samples, d_offset, d_length, B) || tsc < 0 || tsc > 7 || nsoft < 0 || nsoft > 157 ||
      soft_stride < nsoft || (B > 0 && (!d_flags || !d_amp || !d_toa || (nsoft > 0 && !d_soft))))
    return fail(c, TRXSIG_EINVAL, "trxsig_detect_demod_normal_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = chain_check(c);
  if (rc != TRXSIG_OK) return rc;
#ifdef TRX_TUNING_BUILD
  if (c->variant == 5 && !c->chain_broken && nsoft > 0 && nsoft <= 148) {
    // one launch: detect workgroups hand over to demodulate workgroups inside it (trxsig_chain.hip)
    rc = ensure_chain(c, B);
    if (rc != TRXSIG_OK) return rc;
    HIPCHK(c, trx_launch_normal_chain(c->stream, c->sps, c->d_tables, c->h_tables, (const trx_c32 *)d_samples, d_offset,
                                      d_length, B, tsc, detect_thresh, energy_thresh, d_flags, (trx_c32 *)d_amp, d_toa,
                                      d_avgpwr, d_soft, d_hard, nsoft, soft_stride, c->d_det, c->d_chain_status,
                                      c->chain_lag, c->chain_spin, c->generic_taps, c->prof, c->chain_dbg, c->soft_mode));
    return TRXSIG_OK;
  }
#endif
  rc = ensure_ws(c, B);
  if (rc != TRXSIG_OK) return rc;
#ifdef TRX_TUNING_BUILD
  if (c->variant == 4) {
    // detection (correlation + speculative bisection, four bursts per wave) in one kernel, then k_demod
    HIPCHK(c, trx_launch_normal_fused(c->stream, c->sps, 16, c->d_tables, c->h_tables, (const trx_c32 *)d_samples,
                                      d_offset, d_length, B, tsc, detect_thresh, energy_thresh, d_flags,
                                      (trx_c32 *)d_amp, d_toa, d_avgpwr, nullptr, nullptr, 0, 0, c->generic_taps, c->prof));
    if (nsoft > 0)
      HIPCHK(c, trx_launch_demod(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B,
                                 (const trx_c32 *)d_amp, d_toa, d_flags, TRXSIG_F_DETECT, d_soft, d_hard, nsoft,
                                 soft_stride, c->prof, c->soft_mode));
    return TRXSIG_OK;
  }
  if (c->variant >= 1 && c->variant <= 3 && nsoft <= 148) {
    // one kernel for the whole leg: every burst crosses HBM once (k_normal_fused)
    HIPCHK(c, trx_launch_normal_fused(c->stream, c->sps, c->variant == 1 ? 64 : (c->variant == 2 ? 32 : 16), c->d_tables, c->h_tables,
                                      (const trx_c32 *)d_samples, d_offset, d_length, B, tsc, detect_thresh,
                                      energy_thresh, d_flags, (trx_c32 *)d_amp, d_toa, d_avgpwr, d_soft, d_hard, nsoft,
                                      soft_stride, c->generic_taps, c->prof, c->soft_mode));
    return TRXSIG_OK;
  }
#endif
  const bool beside = c->demod_beside && nsoft > 0;
  int k = 0;
  hipStream_t sdet = c->stream;
  if (beside) {
    rc = ensure_beside(c, B);
    if (rc != TRXSIG_OK) return rc;
    k = c->pb_k ^= 1;
    if (c->det) {                                           // masked detect stream: it starts when the caller's inputs are ready
      sdet = c->det;
      if (c->beside_nodeps < 1) {
        HIPCHK(c, hipEventRecord(c->ev_in, c->stream));
        HIPCHK(c, hipStreamWaitEvent(sdet, c->ev_in, 0));
      }
    }
    if (c->dm_in_flight[k]) {                               // the demodulator that read this copy two calls ago
      if (c->beside_nodeps < 2) HIPCHK(c, hipStreamWaitEvent(sdet, c->ev_dm[k], 0));
      c->dm_in_flight[k] = false;
    }
  } else {
    rc = join_demod(c);                                     // (outputs of this call must not be overtaken by an older demodulator)
    if (rc != TRXSIG_OK) return rc;
  }
  HIPCHK(c, trx_launch_tsc_detect(sdet, c->sps, c->d_tables, c->h_tables, (const trx_c32 *)d_samples, d_offset, d_length,
                                  B, tsc, detect_thresh, energy_thresh, c->d_rec, c->cap_bursts, d_flags,
                                  (trx_c32 *)d_amp, d_toa, d_avgpwr, c->generic_taps | (c->spec_peak == 1 ? 2 : 0) | (c->spec_peak == 2 ? 4 : 0), c->prof));
  if (beside) {
    // the demodulator reads its own copy of the verdict: the next call's k_tsc_peak2 overwrites d_flags / d_amp / d_toa beside it
    HIPCHK(c, trx_launch_copy_verdict(sdet, d_flags, (const trx_c32 *)d_amp, d_toa, B, c->pb_flags[k], c->pb_amp[k], c->pb_toa[k]));
    HIPCHK(c, hipEventRecord(c->ev_pk[k], sdet));
    if (c->det) c->det_in_flight = true;
    if (c->beside_nodeps < 2) HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_pk[k], 0));
    HIPCHK(c, trx_launch_demod(c->side, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B, c->pb_amp[k], c->pb_toa[k],
                               c->pb_flags[k], TRXSIG_F_DETECT, d_soft, d_hard, nsoft, soft_stride, c->prof, c->soft_mode));
    HIPCHK(c, hipEventRecord(c->ev_dm[k], c->side));
    c->dm_in_flight[k] = true;
    return TRXSIG_OK;
  }
  if (nsoft > 0)
    HIPCHK(c, trx_launch_demod(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B,
                               (const trx_c32 *)d_amp, d_toa, d_flags, TRXSIG_F_DETECT, d_soft, d_hard, nsoft,
                               soft_stride, c->prof, c->soft_mode));
  return TRXSIG_OK;
}

}  // extern "C"
// the normal-burst leg on bursts that are computed from the raw int16 stream (trxsig_rxfe_push_detect_demod_normal,
// trxsig_frontend.cpp): same kernels, same scratch, no complex float32 stream in HBM
int trx_ctx_rx_normal(trxsig_ctx *c, const TrxRxGen &gen, int B, int tsc, float detect_thresh, float energy_thresh, uint8_t *d_flags,
                      trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr, float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride,
                      hipStream_t on) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 4) return fail(c, TRXSIG_EINVAL, "the fused receive front end needs sps == 4");
  if (tsc < 0 || tsc > 7 || nsoft < 0 || nsoft > 148 || soft_stride < nsoft || B < 0 ||
      (B > 0 && (!d_flags || !d_amp || !d_toa || (nsoft > 0 && !d_soft))))
    return fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push_detect_demod_normal: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_ws(c, B);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, trx_launch_rx_normal(on ? on : c->stream, c->d_tables, c->h_tables, gen, B, tsc, detect_thresh, energy_thresh, c->d_rec, c->cap_bursts,
                                 d_flags, (trx_c32 *)d_amp, d_toa, d_avgpwr, d_soft, d_hard, nsoft, soft_stride, c->generic_taps, c->prof,
                                 c->soft_mode));
  return TRXSIG_OK;
}
int trx_ctx_rx_rach(trxsig_ctx *c, const TrxRxGen &gen, const int32_t *d_len, int B, float detect_thresh, float energy_thresh,
                    uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr, int own_records) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 4) return fail(c, TRXSIG_EINVAL, "the fused receive front end needs sps == 4");
  if (B < 0 || (B > 0 && (!d_len || !d_flags || !d_amp || !d_toa))) return fail(c, TRXSIG_EINVAL, "trx_ctx_rx_rach: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = own_records ? ensure_ws2(c, B) : ensure_ws(c, B);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, trx_launch_rx_rach(c->stream, c->d_tables, gen, d_len, B, detect_thresh, energy_thresh, c->rach_amp_err,
                               (float *)(own_records ? c->d_rec2 : c->d_rec), own_records ? c->cap2_bursts : c->cap_bursts, d_flags,
                               (trx_c32 *)d_amp, d_toa, d_avgpwr, c->prof));
  return TRXSIG_OK;
}
int trx_ctx_rx_demod(trxsig_ctx *c, const TrxRxGen &gen, int B, const trxsig_c32 *d_amp, const float *d_toa, const uint8_t *d_enable,
                     int need_mask, float *d_soft, int nsoft, int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 4) return fail(c, TRXSIG_EINVAL, "the fused receive front end needs sps == 4");
  if (B < 0 || nsoft < 0 || nsoft > 148 || soft_stride < nsoft || (B > 0 && (!d_amp || !d_toa || (nsoft > 0 && !d_soft))))
    return fail(c, TRXSIG_EINVAL, "trx_ctx_rx_demod: bad argument");
  if (B == 0 || nsoft == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_rx_demod(c->stream, c->d_tables, gen, B, (const trx_c32 *)d_amp, d_toa, d_enable, need_mask, d_soft, nullptr, nsoft,
                                soft_stride, c->prof, c->soft_mode));
  return TRXSIG_OK;
}
// trxsig_demodulate_batch with the enable test spelt out: a burst is demodulated when (d_enable[b] & need_mask) == need_mask
int trx_ctx_demod_masked(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length, int B,
                         const trxsig_c32 *d_amp, const float *d_toa, const uint8_t *d_enable, int need_mask, float *d_soft, int nsoft,
                         int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_samples, d_offset, d_length, B) || nsoft < 0 || nsoft > 157 || soft_stride < nsoft ||
      (B > 0 && (!d_amp || !d_toa || !d_soft || !d_enable)))
    return fail(c, TRXSIG_EINVAL, "trx_ctx_demod_masked: bad argument");
  if (B == 0 || nsoft == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_demod(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B,
                             (const trx_c32 *)d_amp, d_toa, d_enable, need_mask, d_soft, nullptr, nsoft, soft_stride, c->prof, c->soft_mode));
  return TRXSIG_OK;
}
extern "C" {

int trxsig_detect_demod_rach_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset,
                                   const int32_t *d_length, int B, float detect_thresh, float energy_thresh,
                                   uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr,
                                   float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_samples, d_offset, d_length, B) || nsoft < 0 || nsoft > 157 || soft_stride < nsoft ||
      (B > 0 && (!d_flags || !d_amp || !d_toa || (nsoft > 0 && !d_soft))))
    return fail(c, TRXSIG_EINVAL, "trxsig_detect_demod_rach_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_ws(c, B);
  if (rc != TRXSIG_OK) return rc;
  if (c->rach_variant >= 1)
    HIPCHK(c, trx_launch_rach_fast(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B,
                                   detect_thresh, energy_thresh, c->rach_amp_err, (float *)c->d_rec, c->cap_bursts, c->rach_variant == 2,
                                   d_flags, (trx_c32 *)d_amp, d_toa, d_avgpwr, c->prof));
  else
    HIPCHK(c, trx_launch_rach_detect(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length,
                                     B, detect_thresh, energy_thresh, (float *)c->d_rec, c->cap_bursts, d_flags,
                                     (trx_c32 *)d_amp, d_toa, d_avgpwr, c->prof));
  if (nsoft > 0)
    HIPCHK(c, trx_launch_demod(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B,
                               (const trx_c32 *)d_amp, d_toa, d_flags, TRXSIG_F_DETECT, d_soft, d_hard, nsoft,
                               soft_stride, c->prof, c->soft_mode));
  return TRXSIG_OK;
}

int trxsig_demodulate_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset,
                            const int32_t *d_length, int B, const trxsig_c32 *d_amp, const float *d_toa,
                            const uint8_t *d_enable, float *d_soft, uint8_t *d_hard, int nsoft,
                            int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_samples, d_offset, d_length, B) || nsoft < 0 || nsoft > 157 || soft_stride < nsoft ||
      (B > 0 && (!d_amp || !d_toa || !d_soft)))
    return fail(c, TRXSIG_EINVAL, "trxsig_demodulate_batch: bad argument");
  if (B == 0 || nsoft == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_demod(c->stream, c->sps, c->d_tables, (const trx_c32 *)d_samples, d_offset, d_length, B,
                             (const trx_c32 *)d_amp, d_toa, d_enable, 0, d_soft, d_hard, nsoft, soft_stride, c->prof, c->soft_mode));
  return TRXSIG_OK;
}

// ---- equaliser path (sps = 1) ------------------------------------------------------------------------
// equaliser scratch: per burst toa_eq (4 B), 7 + 5 complex taps, EQ_XS complex delayed samples
static constexpr int EQ_XS = 160;
static int ensure_eq(trxsig_ctx *c, int B) {
  if (B > c->eq_cap) {
    const int cap = (B + 255) & ~255;
    if (c->d_eq) { HIPCHK(c, hipFree(c->d_eq)); c->d_eq = nullptr; c->eq_cap = 0; }
    HIPCHK(c, hipMalloc((void **)&c->d_eq, (size_t)cap * (4 + 8 * 7 + 8 * 5 + 8 * EQ_XS)));
    c->eq_cap = cap;
  }
  return TRXSIG_OK;
}

int trxsig_estimate_dfe_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length,
                              int B, int tsc, float detect_thresh, float snr_thresh, float snr_value, int variant52m,
                              int max_toa, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_chan_off, trxsig_c32 *d_w,
                              trxsig_c32 *d_b) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 1) return fail(c, TRXSIG_EINVAL, "trxsig_estimate_dfe_batch: the channel estimate / DFE path needs sps == 1");
  if (bad_batch(d_samples, d_offset, d_length, B) || tsc < 0 || tsc > 7 || max_toa < 0 || max_toa > 17 || (snr_thresh < 0.0f && !(snr_value > 0.0f)) ||
      (B > 0 && (!d_flags || !d_amp || !d_toa || !d_chan_off || !d_w || !d_b)))
    return fail(c, TRXSIG_EINVAL, "trxsig_estimate_dfe_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_eq(c, B);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, trx_launch_estimate_dfe(c->stream, c->d_tables, d_samples, TRXSIG_SAMPLES_C32, d_offset, d_length, B, tsc,
                                    detect_thresh, snr_thresh, snr_value, variant52m, max_toa, d_flags, (trx_c32 *)d_amp, d_toa,
                                    (float *)c->d_eq, d_chan_off, (trx_c32 *)d_w, (trx_c32 *)d_b, nullptr, c->prof, nullptr, nullptr,
                                    trx_eq52_geometry(c->h_tables, tsc)));
  return TRXSIG_OK;
}

int trxsig_channel_estimate_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length,
                                  int B, int tsc, float detect_thresh, int variant52m, int max_toa, uint8_t *d_flags,
                                  trxsig_c32 *d_amp, float *d_toa, float *d_chan_off, trxsig_c32 *d_chan) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 1) return fail(c, TRXSIG_EINVAL, "trxsig_channel_estimate_batch: the channel estimate needs sps == 1");
  if (bad_batch(d_samples, d_offset, d_length, B) || tsc < 0 || tsc > 7 || max_toa < 0 || max_toa > 17 ||
      (B > 0 && (!d_flags || !d_amp || !d_toa || !d_chan_off || !d_chan)))
    return fail(c, TRXSIG_EINVAL, "trxsig_channel_estimate_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_eq(c, B);
  if (rc != TRXSIG_OK) return rc;
  // (the kernel also designs a DFE for a nominal SNR into scratch: toa_eq [B], w [7 B], b [5 B] of the equaliser workspace)
  float *toa_eq = (float *)c->d_eq;
  trx_c32 *w = (trx_c32 *)(c->d_eq + sizeof(float) * (size_t)c->eq_cap);
  trx_c32 *bq = w + (size_t)7 * c->eq_cap;
  HIPCHK(c, trx_launch_estimate_dfe(c->stream, c->d_tables, d_samples, TRXSIG_SAMPLES_C32, d_offset, d_length, B, tsc,
                                    detect_thresh, -1.0f, 1.0f, variant52m, max_toa, d_flags, (trx_c32 *)d_amp, d_toa, toa_eq,
                                    d_chan_off, w, bq, (trx_c32 *)d_chan, c->prof, nullptr, nullptr, trx_eq52_geometry(c->h_tables, tsc)));
  return TRXSIG_OK;
}

int trxsig_design_dfe_batch(trxsig_ctx *c, const trxsig_c32 *d_chan, const trxsig_c32 *d_amp, const float *d_snr, int B,
                            trxsig_c32 *d_w, trxsig_c32 *d_b) {
  if (!c) return TRXSIG_EINVAL;
  if (B < 0 || (B > 0 && (!d_chan || !d_snr || !d_w || !d_b))) return fail(c, TRXSIG_EINVAL, "trxsig_design_dfe_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_design_dfe(c->stream, (const trx_c32 *)d_chan, (const trx_c32 *)d_amp, d_snr, B, (trx_c32 *)d_w,
                                  (trx_c32 *)d_b, c->prof));
  return TRXSIG_OK;
}

int trxsig_equalize_taps_batch_fmt(trxsig_ctx *c, const void *d_samples, int sample_format, const int32_t *d_offset,
                                   const int32_t *d_length, int B, const trxsig_c32 *d_amp, const float *d_toa_eq,
                                   const uint8_t *d_enable, const trxsig_c32 *d_w, const trxsig_c32 *d_b, float *d_soft,
                                   uint8_t *d_hard, int nsoft, int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 1) return fail(c, TRXSIG_EINVAL, "trxsig_equalize_taps_batch: equalizeBurst needs sps == 1");
  if (bad_batch(d_samples, d_offset, d_length, B) || nsoft < 0 || nsoft > 157 || soft_stride < nsoft ||
      (sample_format != TRXSIG_SAMPLES_C32 && sample_format != TRXSIG_SAMPLES_F16) ||
      (B > 0 && (!d_amp || !d_toa_eq || !d_enable || !d_w || !d_b || (nsoft > 0 && !d_soft))))
    return fail(c, TRXSIG_EINVAL, "trxsig_equalize_taps_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_eq(c, B);
  if (rc != TRXSIG_OK) return rc;
  trx_c32 *xd = (trx_c32 *)(c->d_eq + (size_t)c->eq_cap * (4 + 56 + 40));
  HIPCHK(c, trx_launch_equalize_taps(c->stream, c->d_tables, d_samples, sample_format, d_offset, d_length, B,
                                     (const trx_c32 *)d_amp, d_toa_eq, d_enable, (const trx_c32 *)d_w, (const trx_c32 *)d_b,
                                     xd, EQ_XS, d_soft, d_hard, nsoft, soft_stride, c->prof));
  return TRXSIG_OK;
}
int trxsig_equalize_taps_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length,
                               int B, const trxsig_c32 *d_amp, const float *d_toa_eq, const uint8_t *d_enable,
                               const trxsig_c32 *d_w, const trxsig_c32 *d_b, float *d_soft, uint8_t *d_hard, int nsoft,
                               int soft_stride) {
  return trxsig_equalize_taps_batch_fmt(c, d_samples, TRXSIG_SAMPLES_C32, d_offset, d_length, B, d_amp, d_toa_eq, d_enable, d_w, d_b,
                                        d_soft, d_hard, nsoft, soft_stride);
}

int trxsig_equalize_normal_batch(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset,
                                 const int32_t *d_length, int B, int tsc, float detect_thresh, float energy_thresh,
                                 int variant52m, int max_toa, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                 trxsig_c32 *d_w, trxsig_c32 *d_b, float *d_soft, uint8_t *d_hard, int nsoft,
                                 int soft_stride) {
  return trxsig_equalize_normal_batch_fmt(c, d_samples, TRXSIG_SAMPLES_C32, d_offset, d_length, B, tsc, detect_thresh, energy_thresh,
                                          variant52m, max_toa, d_flags, d_amp, d_toa, d_w, d_b, d_soft, d_hard, nsoft, soft_stride);
}

int trxsig_equalize_normal_batch_fmt(trxsig_ctx *c, const void *d_samples, int sample_format, const int32_t *d_offset,
                                     const int32_t *d_length, int B, int tsc, float detect_thresh, float energy_thresh,
                                     int variant52m, int max_toa, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                     trxsig_c32 *d_w, trxsig_c32 *d_b, float *d_soft, uint8_t *d_hard, int nsoft,
                                     int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 1) return fail(c, TRXSIG_EINVAL, "trxsig_equalize_normal_batch: equalizeBurst needs sps == 1");
  if (sample_format != TRXSIG_SAMPLES_C32 && sample_format != TRXSIG_SAMPLES_F16)
    return fail(c, TRXSIG_EINVAL, "trxsig_equalize_normal_batch_fmt: unknown sample format");
  if (bad_batch(d_samples, d_offset, d_length, B) || tsc < 0 || tsc > 7 || nsoft < 0 || nsoft > 157 ||
      soft_stride < nsoft || max_toa < 0 || max_toa > 17 ||
      (B > 0 && (!d_flags || !d_amp || !d_toa || (nsoft > 0 && !d_soft))))
    return fail(c, TRXSIG_EINVAL, "trxsig_equalize_normal_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  constexpr int XS = EQ_XS;
  int rc = ensure_eq(c, B);
  if (rc != TRXSIG_OK) return rc;
  const size_t cap = (size_t)c->eq_cap;
  float *toa_eq = (float *)c->d_eq;
  trx_c32 *w = (trx_c32 *)(c->d_eq + cap * 4);
  trx_c32 *bq = (trx_c32 *)(c->d_eq + cap * (4 + 56));
  trx_c32 *xd = (trx_c32 *)(c->d_eq + cap * (4 + 56 + 40));
  if (d_w) w = (trx_c32 *)d_w;
  if (d_b) bq = (trx_c32 *)d_b;
  HIPCHK(c, trx_launch_equalize(c->stream, c->d_tables, d_samples, sample_format, d_offset, d_length, B, tsc,
                                detect_thresh, energy_thresh, variant52m, max_toa, d_flags, (trx_c32 *)d_amp, d_toa,
                                toa_eq, w, bq, xd, XS, d_soft, d_hard, nsoft, soft_stride, c->prof, trx_eq52_geometry(c->h_tables, tsc)));
  return TRXSIG_OK;
}

}  // extern "C"
// the equalising TSC leg of the Transceiver group (trxsig_ctx.h)
int trx_ctx_group_estimate(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length, int B, int tsc,
                           const uint8_t *d_enable, const float *d_snr, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                           float *d_toa_eq, float *d_chan_off, trxsig_c32 *d_w, trxsig_c32 *d_b, int32_t *d_listed) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 1) return fail(c, TRXSIG_EINVAL, "the channel estimate / DFE path needs sps == 1");
  if (B <= 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_eq(c, B);
  if (rc != TRXSIG_OK) return rc;
  // (the list of the marked bursts goes where trx_ctx_group_equalize will afterwards put the delayed bursts)
  int32_t *work = (int32_t *)(c->d_eq + (size_t)c->eq_cap * (4 + 56 + 40));
  HIPCHK(c, trx_launch_estimate_dfe(c->stream, c->d_tables, d_samples, TRXSIG_SAMPLES_C32, d_offset, d_length, B, tsc, 3.0f, -1.0f, 1.0f,
                                    0, 0, d_flags, (trx_c32 *)d_amp, d_toa, d_toa_eq, d_chan_off, (trx_c32 *)d_w, (trx_c32 *)d_b,
                                    nullptr, c->prof, d_enable, d_snr, trx_eq52_geometry(c->h_tables, tsc), d_enable ? (d_listed ? d_listed : work) : nullptr,
                                    d_enable && d_listed));
  return TRXSIG_OK;
}
int trx_ctx_group_equalize(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length, int B,
                           const trxsig_c32 *d_amp, const float *d_toa_eq, const uint8_t *d_gate, const trxsig_c32 *d_w_tab,
                           const trxsig_c32 *d_b_tab, const int32_t *d_tap_ix, float *d_soft, int nsoft, int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (c->sps != 1) return fail(c, TRXSIG_EINVAL, "equalizeBurst needs sps == 1");
  if (B <= 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  int rc = ensure_eq(c, B);
  if (rc != TRXSIG_OK) return rc;
  trx_c32 *xd = (trx_c32 *)(c->d_eq + (size_t)c->eq_cap * (4 + 56 + 40));
  HIPCHK(c, trx_launch_equalize_taps(c->stream, c->d_tables, d_samples, TRXSIG_SAMPLES_C32, d_offset, d_length, B, (const trx_c32 *)d_amp,
                                     d_toa_eq, d_gate, (const trx_c32 *)d_w_tab, (const trx_c32 *)d_b_tab, xd, EQ_XS, d_soft, nullptr,
                                     nsoft, soft_stride, c->prof, d_tap_ix));
  return TRXSIG_OK;
}
extern "C" {

// ---- TX path, rate conversion, sample format ---------------------------------------------------------
int trxsig_modulate_batch(trxsig_ctx *c, const uint8_t *d_bits, const int32_t *d_guard, const float *d_gain, int B,
                          trxsig_c32 *d_out, const int32_t *d_out_offset) {
  if (!c) return TRXSIG_EINVAL;
  if (B < 0 || (B > 0 && (!d_bits || !d_guard || !d_out || !d_out_offset)))
    return fail(c, TRXSIG_EINVAL, "trxsig_modulate_batch: bad argument");
  if (B == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_modulate(c->stream, c->sps, c->d_tables, d_bits, d_guard, d_gain, B, (trx_c32 *)d_out,
                                d_out_offset, c->prof));
  return TRXSIG_OK;
}

int trxsig_resample_out_len(int n_in, int P, int Q) {
  if (n_in < 0 || P <= 0 || Q <= 0) return TRXSIG_EINVAL;
  return (int)std::ceil(n_in * (float)P / (float)Q);       // sigProcLib.cpp:1171
}

int trxsig_resample_batch(trxsig_ctx *c, const trxsig_c32 *d_in, int n_in, int64_t in_stride, int S, int P, int Q,
                          const float *d_lpf, int L, trxsig_c32 *d_out, int64_t out_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (n_in < 0 || S < 0 || P <= 0 || Q <= 0 || L <= 0 || (S > 0 && (!d_in || !d_out || !d_lpf)) ||
      (int64_t)n_in * Q > (int64_t)1 << 40)
    return fail(c, TRXSIG_EINVAL, "trxsig_resample_batch: bad argument");
  const int nout = trxsig_resample_out_len(n_in, P, Q);
  if (S == 0 || nout == 0) return TRXSIG_OK;
  if (S > 65535) return fail(c, TRXSIG_EINVAL, "trxsig_resample_batch: more than 65535 streams");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_resample(c->stream, (const trx_c32 *)d_in, n_in, in_stride, S, P, Q, d_lpf, L,
                                (trx_c32 *)d_out, out_stride, nout, c->prof));
  return TRXSIG_OK;
}

int trxsig_resample_host(trxsig_ctx *c, const trxsig_c32 *h_in, int n_in, int P, int Q, const float *h_lpf, int L,
                         trxsig_c32 *h_out, int out_cap) {
  if (!c) return TRXSIG_EINVAL;
  if (n_in <= 0 || P <= 0 || Q <= 0 || L <= 0 || !h_in || !h_lpf || !h_out)
    return fail(c, TRXSIG_EINVAL, "trxsig_resample_host: bad argument");
  const int nout = trxsig_resample_out_len(n_in, P, Q);
  if (nout > out_cap) return fail(c, TRXSIG_EINVAL, "trxsig_resample_host: output buffer too small");
  DeviceGuard g(c->device);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_in = 0, o_lpf = up(8 * (size_t)n_in), o_out = o_lpf + up(4 * (size_t)L), end = o_out + up(8 * (size_t)nout);
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  HIPCHK(c, hipMemcpyAsync(d + o_in, h_in, 8 * (size_t)n_in, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_lpf, h_lpf, 4 * (size_t)L, hipMemcpyHostToDevice, c->stream));
  rc = trxsig_resample_batch(c, (const trxsig_c32 *)(d + o_in), n_in, n_in, 1, P, Q, (const float *)(d + o_lpf), L,
                             (trxsig_c32 *)(d + o_out), nout);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_out, 8 * (size_t)nout, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return nout;
}

int trxsig_unpack_int16(trxsig_ctx *c, const int16_t *d_iq, int64_t n, int swap_iq, trxsig_c32 *d_out) {
  if (!c) return TRXSIG_EINVAL;
  if (n < 0 || (n > 0 && (!d_iq || !d_out))) return fail(c, TRXSIG_EINVAL, "trxsig_unpack_int16: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_convert(c->stream, 0, d_iq, n, swap_iq, d_out, c->prof));
  return TRXSIG_OK;
}
int trxsig_pack_int16_scaled(trxsig_ctx *c, const trxsig_c32 *d_in, int64_t n, float gain, int16_t *d_iq) {
  if (!c) return TRXSIG_EINVAL;
  if (n < 0 || (n > 0 && (!d_iq || !d_in))) return fail(c, TRXSIG_EINVAL, "trxsig_pack_int16_scaled: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_convert(c->stream, 1, d_in, n, 0, d_iq, c->prof, gain));
  return TRXSIG_OK;
}
int trxsig_unpack_half(trxsig_ctx *c, const uint16_t *d_iq, int64_t n, trxsig_c32 *d_out) {
  if (!c) return TRXSIG_EINVAL;
  if (n < 0 || (n > 0 && (!d_iq || !d_out))) return fail(c, TRXSIG_EINVAL, "trxsig_unpack_half: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_convert(c->stream, 2, d_iq, n, 0, d_out, c->prof));
  return TRXSIG_OK;
}
int trxsig_pack_int16(trxsig_ctx *c, const trxsig_c32 *d_in, int64_t n, int16_t *d_iq) {
  if (!c) return TRXSIG_EINVAL;
  if (n < 0 || (n > 0 && (!d_iq || !d_in))) return fail(c, TRXSIG_EINVAL, "trxsig_pack_int16: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_convert(c->stream, 1, d_in, n, 0, d_iq, c->prof));
  return TRXSIG_OK;
}

int trxsig_modulate_host(trxsig_ctx *c, const uint8_t *h_bits, const int32_t *h_guard, const float *h_gain, int B,
                         trxsig_c32 *h_out, const int32_t *h_out_offset, int64_t out_samples) {
  if (!c) return TRXSIG_EINVAL;
  if (B < 0 || out_samples < 0 || (B > 0 && (!h_bits || !h_guard || !h_out || !h_out_offset)))
    return fail(c, TRXSIG_EINVAL, "trxsig_modulate_host: bad argument");
  if (B == 0) return TRXSIG_OK;
  for (int b = 0; b < B; b++)
    if (h_guard[b] < 0 || h_guard[b] > 9 || h_out_offset[b] < 0 ||
        (int64_t)h_out_offset[b] + (int64_t)c->sps * (148 + h_guard[b]) > out_samples)
      return fail(c, TRXSIG_EINVAL, "trxsig_modulate_host: guard/offset out of range");
  DeviceGuard g(c->device);
  auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
  const size_t o_bits = 0, o_g = up((size_t)B * 148), o_off = o_g + up(4 * (size_t)B), o_gain = o_off + up(4 * (size_t)B),
               o_out = o_gain + up(4 * (size_t)B), end = o_out + up(8 * (size_t)out_samples);
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  HIPCHK(c, hipMemcpyAsync(d + o_bits, h_bits, (size_t)B * 148, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_g, h_guard, 4 * (size_t)B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_off, h_out_offset, 4 * (size_t)B, hipMemcpyHostToDevice, c->stream));
  if (h_gain) HIPCHK(c, hipMemcpyAsync(d + o_gain, h_gain, 4 * (size_t)B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(d + o_out, 0, 8 * (size_t)out_samples, c->stream));
  rc = trxsig_modulate_batch(c, (const uint8_t *)(d + o_bits), (const int32_t *)(d + o_g),
                             h_gain ? (const float *)(d + o_gain) : nullptr, B, (trxsig_c32 *)(d + o_out),
                             (const int32_t *)(d + o_off));
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_out, 8 * (size_t)out_samples, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

// ---- host-buffer wrappers (PCIe-inclusive convenience; never the timed path) ------------------------
static int detect_demod_host(trxsig_ctx *c, bool rach, const trxsig_c32 *h_samples, const int32_t *h_offset,
                             const int32_t *h_length, int B, int tsc, float detect_thresh, float energy_thresh,
                             uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa, float *h_avgpwr, float *h_soft,
                             int nsoft, int soft_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(h_samples, h_offset, h_length, B) || (B > 0 && (!h_flags || !h_amp || !h_toa)) || nsoft < 0 ||
      nsoft > 157 || soft_stride < nsoft || (nsoft > 0 && B > 0 && !h_soft))
    return fail(c, TRXSIG_EINVAL, "trxsig_detect_demod_*_host: bad argument");
  if (B == 0) return TRXSIG_OK;
  int64_t total = 0;
  for (int b = 0; b < B; b++) {
    if (h_offset[b] < 0 || h_length[b] < 0) return fail(c, TRXSIG_EINVAL, "negative offset/length");
    int64_t e = (int64_t)h_offset[b] + h_length[b];
    if (e > total) total = e;
  }
  DeviceGuard g(c->device);
  auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
  const size_t o_s = 0, o_off = o_s + up(sizeof(trx_c32) * (size_t)total), o_len = o_off + up(4 * (size_t)B),
               o_fl = o_len + up(4 * (size_t)B), o_amp = o_fl + up((size_t)B), o_toa = o_amp + up(8 * (size_t)B),
               o_pwr = o_toa + up(4 * (size_t)B), o_soft = o_pwr + up(4 * (size_t)B),
               end = o_soft + up(4 * (size_t)B * soft_stride);
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  // A handful of bursts (the drop-in form: one per call): inputs and outputs go through a pinned mirror of the staging
  // area, one DMA each way instead of eight pageable copies (115 -> 38 us per call, tools/host_path_bench.py)
  const bool small = end <= (size_t)256 * 1024;
  char *m = nullptr;
  if (small) {
    rc = ensure_pin(c, end);
    if (rc != TRXSIG_OK) return rc;
    m = (char *)c->h_pin;
    std::memcpy(m + o_s, h_samples, sizeof(trx_c32) * (size_t)total);
    std::memcpy(m + o_off, h_offset, 4 * (size_t)B);
    std::memcpy(m + o_len, h_length, 4 * (size_t)B);
    HIPCHK(c, hipMemcpyAsync(d, m, o_fl, hipMemcpyHostToDevice, c->stream));
  } else {
    HIPCHK(c, hipMemcpyAsync(d + o_s, h_samples, sizeof(trx_c32) * (size_t)total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + o_off, h_offset, 4 * (size_t)B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + o_len, h_length, 4 * (size_t)B, hipMemcpyHostToDevice, c->stream));
  }
  if (rach)
    rc = trxsig_detect_demod_rach_batch(c, (trxsig_c32 *)(d + o_s), (int32_t *)(d + o_off), (int32_t *)(d + o_len), B,
                                        detect_thresh, energy_thresh, (uint8_t *)(d + o_fl),
                                        (trxsig_c32 *)(d + o_amp), (float *)(d + o_toa), (float *)(d + o_pwr),
                                        (float *)(d + o_soft), nullptr, nsoft, soft_stride);
  else
    rc = trxsig_detect_demod_normal_batch(c, (trxsig_c32 *)(d + o_s), (int32_t *)(d + o_off), (int32_t *)(d + o_len),
                                          B, tsc, detect_thresh, energy_thresh, (uint8_t *)(d + o_fl),
                                          (trxsig_c32 *)(d + o_amp), (float *)(d + o_toa), (float *)(d + o_pwr),
                                          (float *)(d + o_soft), nullptr, nsoft, soft_stride);
  if (rc != TRXSIG_OK) return rc;
  if (small) {
    const size_t out_end = nsoft > 0 ? end : o_soft;
    HIPCHK(c, hipMemcpyAsync(m + o_fl, d + o_fl, out_end - o_fl, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_flags, m + o_fl, (size_t)B);
    std::memcpy(h_amp, m + o_amp, 8 * (size_t)B);
    std::memcpy(h_toa, m + o_toa, 4 * (size_t)B);
    if (h_avgpwr) std::memcpy(h_avgpwr, m + o_pwr, 4 * (size_t)B);
    if (nsoft > 0) std::memcpy(h_soft, m + o_soft, 4 * (size_t)B * soft_stride);
    return TRXSIG_OK;
  }
  HIPCHK(c, hipMemcpyAsync(h_flags, d + o_fl, (size_t)B, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_amp, d + o_amp, 8 * (size_t)B, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_toa, d + o_toa, 4 * (size_t)B, hipMemcpyDeviceToHost, c->stream));
  if (h_avgpwr) HIPCHK(c, hipMemcpyAsync(h_avgpwr, d + o_pwr, 4 * (size_t)B, hipMemcpyDeviceToHost, c->stream));
  if (nsoft > 0)
    HIPCHK(c, hipMemcpyAsync(h_soft, d + o_soft, 4 * (size_t)B * soft_stride, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_detect_demod_normal_host(trxsig_ctx *c, const trxsig_c32 *h_samples, const int32_t *h_offset,
                                    const int32_t *h_length, int B, int tsc, float detect_thresh,
                                    float energy_thresh, uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa,
                                    float *h_avgpwr, float *h_soft, int nsoft, int soft_stride) {
  if (tsc < 0 || tsc > 7) return TRXSIG_EINVAL;
  return detect_demod_host(c, false, h_samples, h_offset, h_length, B, tsc, detect_thresh, energy_thresh, h_flags,
                           h_amp, h_toa, h_avgpwr, h_soft, nsoft, soft_stride);
}
int trxsig_detect_demod_rach_host(trxsig_ctx *c, const trxsig_c32 *h_samples, const int32_t *h_offset,
                                  const int32_t *h_length, int B, float detect_thresh, float energy_thresh,
                                  uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa, float *h_avgpwr,
                                  float *h_soft, int nsoft, int soft_stride) {
  return detect_demod_host(c, true, h_samples, h_offset, h_length, B, 0, detect_thresh, energy_thresh, h_flags, h_amp,
                           h_toa, h_avgpwr, h_soft, nsoft, soft_stride);
}

int trxsig_demodulate_host(trxsig_ctx *c, const trxsig_c32 *h_samples, int n, trxsig_c32 amp, float toa,
                           float *h_soft, int nsoft) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_samples || !h_soft || n <= 0 || nsoft < 0 || nsoft > 157)
    return fail(c, TRXSIG_EINVAL, "trxsig_demodulate_host: bad argument");
  // what the kernel would silently answer with all-zero soft bits (trxsig.h: accepted burst geometry) is an error for
  // the one-burst form: the reference's demodulateBurst has no such limits, so a drop-in caller must hear about it
  if (n < 92 * c->sps || n > 157 * c->sps || n % c->sps != 0 || !(std::fabs(toa) <= 4096.0f))
    return fail(c, TRXSIG_EINVAL, "trxsig_demodulate_host: burst must be 92..157 symbols (a multiple of sps samples) and |TOA| <= 4096");
  DeviceGuard g(c->device);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_s = 0, o_off = up(8 * (size_t)n), o_len = o_off + 256, o_amp = o_len + 256, o_toa = o_amp + 256,
               o_soft = o_toa + 256, end = o_soft + up(4 * 160);
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  const int32_t zero = 0, len = n;
  HIPCHK(c, hipMemcpyAsync(d + o_s, h_samples, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_off, &zero, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_len, &len, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_amp, &amp, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_toa, &toa, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));              // the scalars above live on this stack frame
  rc = trxsig_demodulate_batch(c, (trxsig_c32 *)(d + o_s), (int32_t *)(d + o_off), (int32_t *)(d + o_len), 1,
                               (trxsig_c32 *)(d + o_amp), (float *)(d + o_toa), nullptr, (float *)(d + o_soft),
                               nullptr, nsoft, 160);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_soft, d + o_soft, 4 * (size_t)nsoft, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

// ---- single-burst host forms of the equaliser steps (the source-compatible facade; one PCIe round trip each) ----
int trxsig_channel_estimate_host(trxsig_ctx *c, const trxsig_c32 *h_samples, int n, int tsc, float detect_thresh, int variant52m,
                                 int max_toa, uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa, float *h_chan_off,
                                 trxsig_c32 h_chan[6]) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_samples || n <= 0 || !h_flags || !h_amp || !h_toa || !h_chan_off || !h_chan)
    return fail(c, TRXSIG_EINVAL, "trxsig_channel_estimate_host: bad argument");
  DeviceGuard g(c->device);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_s = 0, o_off = up(8 * (size_t)n), o_len = o_off + 256, o_fl = o_len + 256, o_amp = o_fl + 256, o_toa = o_amp + 256,
               o_co = o_toa + 256, o_ch = o_co + 256, end = o_ch + 256;
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  const int32_t zero = 0, len = n;
  HIPCHK(c, hipMemcpyAsync(d + o_s, h_samples, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_off, &zero, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_len, &len, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));              // the scalars above live on this stack frame
  rc = trxsig_channel_estimate_batch(c, (trxsig_c32 *)(d + o_s), (int32_t *)(d + o_off), (int32_t *)(d + o_len), 1, tsc, detect_thresh,
                                     variant52m, max_toa, (uint8_t *)(d + o_fl), (trxsig_c32 *)(d + o_amp), (float *)(d + o_toa),
                                     (float *)(d + o_co), (trxsig_c32 *)(d + o_ch));
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_flags, d + o_fl, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_amp, d + o_amp, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_toa, d + o_toa, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_chan_off, d + o_co, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_chan, d + o_ch, 48, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_design_dfe_host(trxsig_ctx *c, const trxsig_c32 h_chan[6], float snr, trxsig_c32 h_w[7], trxsig_c32 h_b[5]) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_chan || !h_w || !h_b) return fail(c, TRXSIG_EINVAL, "trxsig_design_dfe_host: bad argument");
  DeviceGuard g(c->device);
  const size_t o_ch = 0, o_snr = 256, o_w = 512, o_b = 768, end = 1024;
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  HIPCHK(c, hipMemcpyAsync(d + o_ch, h_chan, 48, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_snr, &snr, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_design_dfe_batch(c, (trxsig_c32 *)(d + o_ch), nullptr, (float *)(d + o_snr), 1, (trxsig_c32 *)(d + o_w), (trxsig_c32 *)(d + o_b));
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_w, d + o_w, 56, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_b, d + o_b, 40, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_equalize_taps_host(trxsig_ctx *c, const trxsig_c32 *h_samples, int n, trxsig_c32 amp, float toa_eq,
                              const trxsig_c32 h_w[7], const trxsig_c32 h_b[5], float *h_soft, int nsoft) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_samples || n <= 0 || !h_w || !h_b || !h_soft || nsoft < 0 || nsoft > 157)
    return fail(c, TRXSIG_EINVAL, "trxsig_equalize_taps_host: bad argument");
  DeviceGuard g(c->device);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_s = 0, o_off = up(8 * (size_t)n), o_len = o_off + 256, o_fl = o_len + 256, o_amp = o_fl + 256, o_toa = o_amp + 256,
               o_w = o_toa + 256, o_b = o_w + 256, o_soft = o_b + 256, end = o_soft + up(4 * 160);
  int rc = ensure_stage(c, end);
  if (rc != TRXSIG_OK) return rc;
  char *d = (char *)c->d_stage;
  const int32_t zero = 0, len = n;
  const uint8_t fl = TRXSIG_F_ENERGY | TRXSIG_F_DETECT;
  HIPCHK(c, hipMemcpyAsync(d + o_s, h_samples, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_off, &zero, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_len, &len, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_fl, &fl, 1, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_amp, &amp, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_toa, &toa_eq, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_w, h_w, 56, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_b, h_b, 40, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_equalize_taps_batch(c, (trxsig_c32 *)(d + o_s), (int32_t *)(d + o_off), (int32_t *)(d + o_len), 1, (trxsig_c32 *)(d + o_amp),
                                  (float *)(d + o_toa), (uint8_t *)(d + o_fl), (trxsig_c32 *)(d + o_w), (trxsig_c32 *)(d + o_b),
                                  (float *)(d + o_soft), nullptr, nsoft, 160);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_soft, d + o_soft, 4 * (size_t)nsoft, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

// ---- measurement helpers ------------------------------------------------------------------------------
int trxsig_timer_start(trxsig_ctx *c) {
  if (!c) return TRXSIG_EINVAL;
  DeviceGuard g(c->device);
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  return TRXSIG_OK;
}
int trxsig_timer_stop(trxsig_ctx *c, float *ms) {
  if (!c || !ms) return TRXSIG_EINVAL;
  DeviceGuard g(c->device);
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return TRXSIG_OK;
}

const char *trxsig_kernel_name(int id) {
  static const char *names[TRXSIG_K_COUNT] = { "k_tsc_corr", "k_tsc_peak", "k_demod", "k_rach_corr", "k_rach_peak",
                                               "k_modulate", "k_resample", "k_eq_detect", "k_convert", "k_normal_fused", "k_fec_viterbi",
                                               "k_normal_chain", "k_eq_delay", "k_eq_dfe", "k_group_replay" };
  return (id >= 0 && id < TRXSIG_K_COUNT) ? names[id] : "?";
}
int trxsig_fec_xcch_decode_batch(trxsig_ctx *c, const float *d_soft, int soft_stride, int n_blocks, int wire,
                                 uint8_t *d_frames, uint8_t *d_ok) {
  if (!c) return TRXSIG_EINVAL;
  if (n_blocks < 0 || soft_stride < 148 || (n_blocks > 0 && (!d_soft || !d_frames || !d_ok)))
    return fail(c, TRXSIG_EINVAL, "trxsig_fec_xcch_decode_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_fec(c->stream, 1, d_soft, soft_stride, 456, 228, n_blocks, wire, d_frames, d_ok, nullptr, 0, c->prof));
  return TRXSIG_OK;
}
int trxsig_fec_rach_decode_batch(trxsig_ctx *c, const float *d_soft, int soft_stride, int n_bursts, int wire,
                                 uint8_t *d_tail_ok, uint8_t *d_bsic, uint8_t *d_ra) {
  if (!c) return TRXSIG_EINVAL;
  if (n_bursts < 0 || soft_stride < 85 || (n_bursts > 0 && (!d_soft || !d_tail_ok || !d_bsic || !d_ra)))
    return fail(c, TRXSIG_EINVAL, "trxsig_fec_rach_decode_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_fec(c->stream, 2, d_soft, soft_stride, 36, 18, n_bursts, wire, d_tail_ok, d_bsic, d_ra, 0, c->prof));
  return TRXSIG_OK;
}
int trxsig_fec_xcch_encode_batch(trxsig_ctx *c, const uint8_t *d_frames, int n_blocks, int tsc, uint8_t *d_bits) {
  if (!c) return TRXSIG_EINVAL;
  if (n_blocks < 0 || tsc < 0 || tsc > 7 || (n_blocks > 0 && (!d_frames || !d_bits)))
    return fail(c, TRXSIG_EINVAL, "trxsig_fec_xcch_encode_batch: bad argument");
  if (n_blocks == 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  if (!c->d_tsc) {
    uint8_t h[8 * 26];
    for (int t = 0; t < 8; t++)
      for (int k = 0; k < 26; k++) h[26 * t + k] = trx_training_sequence(t)[k] == '1';
    HIPCHK(c, hipMalloc((void **)&c->d_tsc, sizeof h));
    HIPCHK(c, hipMemcpy(c->d_tsc, h, sizeof h, hipMemcpyHostToDevice));
  }
  HIPCHK(c, trx_launch_fec_xcch_encode(c->stream, d_frames, n_blocks, c->d_tsc + 26 * tsc, d_bits, c->prof));
  return TRXSIG_OK;
}
int trxsig_fec_tch_decode_batch(trxsig_ctx *c, const float *d_soft, int soft_stride, int n_bursts, int wire,
                                uint8_t *d_tch, uint8_t *d_tch_good, uint8_t *d_facch, uint8_t *d_facch_ok,
                                uint8_t *d_stolen) {
  if (!c) return TRXSIG_EINVAL;
  const int nblk = n_bursts / 4 - 1;
  if (n_bursts < 0 || soft_stride < 148 || (nblk > 0 && (!d_soft || !d_tch || !d_tch_good || !d_stolen)) ||
      ((d_facch == nullptr) != (d_facch_ok == nullptr)))
    return fail(c, TRXSIG_EINVAL, "trxsig_fec_tch_decode_batch: bad argument");
  if (nblk <= 0) return TRXSIG_OK;
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_fec(c->stream, 3, d_soft, soft_stride, 378, 189, nblk, wire, d_tch, d_tch_good, d_stolen, 0, c->prof));
  if (d_facch)
    HIPCHK(c, trx_launch_fec(c->stream, 1, d_soft, soft_stride, 456, 228, nblk, wire, d_facch, d_facch_ok, nullptr, 0,
                             c->prof, 1));
  return TRXSIG_OK;
}
int trxsig_fec_viterbi_batch(trxsig_ctx *c, const float *d_soft, int n_soft, int64_t in_stride, int n_blocks,
                             uint8_t *d_bits, int64_t out_stride) {
  if (!c) return TRXSIG_EINVAL;
  if (n_blocks < 0 || n_soft < 2 || n_soft > 1024 || (n_soft & 1) || in_stride < n_soft || out_stride < n_soft / 2 ||
      (n_blocks > 0 && (!d_soft || !d_bits)))
    return fail(c, TRXSIG_EINVAL, "trxsig_fec_viterbi_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_fec(c->stream, 0, d_soft, in_stride, n_soft, n_soft / 2, n_blocks, 0, d_bits, nullptr, nullptr,
                           out_stride, c->prof));
  return TRXSIG_OK;
}

int trxsig_tuning_build(void) {
#ifdef TRX_TUNING_BUILD
  return 1;
#else
  return 0;
#endif
}

// library-wide implementation knobs (trxsig_launch.h): plain atomics, defaults here
static std::atomic<int> g_knob[TRX_KNOB_COUNT] = {{1}, {4096}, {0}, {1}, {0}, {0}};
extern "C++" {
int trx_knob(int id) { return (id >= 0 && id < TRX_KNOB_COUNT) ? g_knob[id].load(std::memory_order_relaxed) : 0; }
void trx_knob_set(int id, int value) { if (id >= 0 && id < TRX_KNOB_COUNT) g_knob[id].store(value, std::memory_order_relaxed); }
}

int trxsig_set_soft_mode(trxsig_ctx *c, int mode) {
  if (!c) return TRXSIG_EINVAL;
  if (mode != TRXSIG_SOFT_EXACT && mode != TRXSIG_SOFT_TOLERANCE) return fail(c, TRXSIG_EINVAL, "trxsig_set_soft_mode: unknown mode");
  c->soft_mode = mode;                                      // (read at the next launch: stream-ordered like every other call)
  return TRXSIG_OK;
}
int trxsig_get_soft_mode(const trxsig_ctx *c) { return c ? c->soft_mode : TRXSIG_EINVAL; }

int trxsig_set_tuning(trxsig_ctx *c, int key, int value) {
  if (!c) return TRXSIG_EINVAL;
#ifndef TRX_TUNING_BUILD
  // the product library carries the default implementations only (normal path 0 with the two-lane peak kernel, RACH paths 1
  // and 2); the alternates that measured slower live in libtrxsig_tune.so (make -C csrc tune)
  if ((key == TRXSIG_TUNE_NORMAL_PATH && value != 0) || (key == TRXSIG_TUNE_RACH_PATH && value == 0) ||
      (key == TRXSIG_TUNE_SPECULATIVE_PEAK && value != 0) || key == TRXSIG_TUNE_CHAIN_LAG || key == TRXSIG_TUNE_CHAIN_SPIN || key == 6 || key == 10)
    return fail(c, TRXSIG_EINVAL, "trxsig_set_tuning: this implementation is only in the tuning build (libtrxsig_tune.so)");
#endif
  if (key == TRXSIG_TUNE_NORMAL_PATH && value >= 0 && value <= 5) { c->variant = value; return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_CHAIN_LAG && value >= 1) { c->chain_lag = value; return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_CHAIN_SPIN && value >= 0) { c->chain_spin = (unsigned)value; return TRXSIG_OK; }
  if (key == 6 && value >= 0 && value <= 3) { c->chain_dbg = value; return TRXSIG_OK; }   // timing experiments (tools/chain_roles.py)
  if (key == 10 && value >= 0 && value <= 2) { c->beside_nodeps = value; return TRXSIG_OK; }   // timing experiment (tools/cu_split.py)
  if (key == TRXSIG_TUNE_RACH_PATH && value >= 0 && value <= 2) { c->rach_variant = value; return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_GENERIC_TAPS && value >= 0 && value <= 1) { c->generic_taps = value; return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_DEMOD_BESIDE && value >= 0 && value <= 1) {
    DeviceGuard g(c->device);
    int rc = join_demod(c);
    if (rc != TRXSIG_OK) return rc;
    c->demod_beside = value;
    return TRXSIG_OK;
  }
  if (key == TRXSIG_TUNE_SPECULATIVE_PEAK && value >= 0 && value <= 2) { c->spec_peak = value; return TRXSIG_OK; }
  // library-wide knobs (every context of the process; read by the launchers)
  if (key == TRXSIG_TUNE_EQ_TAIL && (value == 1 || value == 2)) { trx_knob_set(TRX_KNOB_EQ_TAIL, value); return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_EQ_DENSE && value >= 0) { trx_knob_set(TRX_KNOB_EQ_DENSE, value); return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_RXRES_WPB && value >= 0 && value <= 64) { trx_knob_set(TRX_KNOB_RXRES_WPB, value); return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_RXRES_ROWS && value >= 0 && value <= 1) { trx_knob_set(TRX_KNOB_RXRES_ROWS, value); return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_CHAN_TPW && value >= 0 && value <= 64) { trx_knob_set(TRX_KNOB_CHAN_TPW, value); return TRXSIG_OK; }
  if (key == TRXSIG_TUNE_GROUP_REPLAY && value >= 0 && value <= 1) { trx_knob_set(TRX_KNOB_GROUP_REPLAY, value); return TRXSIG_OK; }
  if ((key == TRXSIG_TUNE_BESIDE_DET_CUS && value >= 0 && value <= 504) || (key == TRXSIG_TUNE_CU_LAYOUT && value >= 0 && value <= 1) ||
      (key == TRXSIG_TUNE_BESIDE_PRIORITY && value >= 0 && value <= 2)) {
    DeviceGuard g(c->device);
    int rc = join_demod(c);
    if (rc != TRXSIG_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = drop_beside_streams(c);                            // re-created with the new masks at the next call
    if (rc != TRXSIG_OK) return rc;
    if (key == TRXSIG_TUNE_BESIDE_DET_CUS) c->det_cus = value;
    else if (key == TRXSIG_TUNE_CU_LAYOUT) c->cu_layout = value;
    else c->side_prio = value;
    return TRXSIG_OK;
  }
  return fail(c, TRXSIG_EINVAL, "trxsig_set_tuning: unknown key or value");
}
int trxsig_profile_enable(trxsig_ctx *c, int on) {
  if (!c) return TRXSIG_EINVAL;
  DeviceGuard g(c->device);
  if (on && !c->prof) c->prof = new (std::nothrow) EventProfiler;
  if (!on && c->prof) { delete c->prof; c->prof = nullptr; }
  return TRXSIG_OK;
}
int trxsig_profile_collect(trxsig_ctx *c, float total_ms[TRXSIG_K_COUNT], int launches[TRXSIG_K_COUNT]) {
  if (!c || !total_ms || !launches) return TRXSIG_EINVAL;
  for (int i = 0; i < TRXSIG_K_COUNT; i++) { total_ms[i] = 0; launches[i] = 0; }
  if (!c->prof) return TRXSIG_OK;
  DeviceGuard g(c->device);
  c->prof->collect(total_ms, launches);
  return TRXSIG_OK;
}
int trxsig_kernel_count(void) { return TRXSIG_K_COUNT; }
int trxsig_profile_collect_n(trxsig_ctx *c, int cap, float *total_ms, int *launches) {
  if (!c || cap < 0 || (cap > 0 && (!total_ms || !launches))) return TRXSIG_EINVAL;
  float ms[TRXSIG_K_COUNT]; int n[TRXSIG_K_COUNT];
  const int rc = trxsig_profile_collect(c, ms, n);
  if (rc != TRXSIG_OK) return rc;
  for (int i = 0; i < cap && i < TRXSIG_K_COUNT; i++) { total_ms[i] = ms[i]; launches[i] = n[i]; }
  return TRXSIG_K_COUNT;
}
int trxsig_tables_validate_host(const void *h_blob, size_t bytes) {
  if (!h_blob || bytes != sizeof(TrxTables)) return TRXSIG_EINVAL;
  return trx_tables_valid((const TrxTables *)h_blob) ? TRXSIG_OK : TRXSIG_EINVAL;
}

// ---- RCCL broadcast of the table blob for hosts that are not Python (the only collective of the path) ----
// librccl is loaded on first use; libtrxsig does not link it (a single-GPU host never needs it).
int trxsig_tables_broadcast(void *nccl_comm, void *d_blob, size_t bytes, int root, void *hip_stream) {
  if (!nccl_comm || !d_blob || bytes != sizeof(TrxTables) || root < 0) return TRXSIG_EINVAL;
  typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);   // ncclBroadcast (rccl.h:591)
  static bcast_fn fn = nullptr;
  if (!fn) {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { std::fprintf(stderr, "trxsig_tables_broadcast: cannot load librccl.so (%s)\n", dlerror()); return TRXSIG_ENODEV; }
    fn = (bcast_fn)dlsym(h, "ncclBroadcast");
    if (!fn) { std::fprintf(stderr, "trxsig_tables_broadcast: librccl.so has no ncclBroadcast\n"); return TRXSIG_ENODEV; }
  }
  const int nccl_uint8 = 1;                                 // ncclUint8 (rccl.h:460)
  const int rc = fn(d_blob, d_blob, bytes, nccl_uint8, root, nccl_comm, (hipStream_t)hip_stream);   // in place on every rank
  if (rc != 0) { std::fprintf(stderr, "trxsig_tables_broadcast: ncclBroadcast failed (%d)\n", rc); return TRXSIG_EHIP; }
  return TRXSIG_OK;
}

int trxsig_tables_rach_error_bound(const void *h_blob, size_t bytes, float *bound, float *seq_norm) {
  if (!h_blob || bytes != sizeof(TrxTables) || !bound || !trx_tables_valid((const TrxTables *)h_blob)) return TRXSIG_EINVAL;
  const TrxTables *T = (const TrxTables *)h_blob;
  *bound = trx_rach_amp_err(T);
  if (seq_norm) {
    double n2 = 0.0;
    for (int m = 0; m < 41 * (int)T->sps; m++) n2 += (double)T->rach[m].r * T->rach[m].r + (double)T->rach[m].i * T->rach[m].i;
    *seq_norm = (float)std::sqrt(n2);
  }
  return TRXSIG_OK;
}

// ---- the free-standing vector primitives of sigProcLib.h (trxsig_prim.hip) ----------------------------------------
int trxsig_convolve_out_len(int La, int Lb, int span, int cust_len) {
  if (La <= 0 || Lb <= 0) return -1;
  return trx_convolve_out_len(La, Lb, span, cust_len);
}

int trxsig_convolve_batch(trxsig_ctx *c, const trxsig_c32 *d_a, const int32_t *d_a_off, const int32_t *d_a_len, int B,
                          int max_len, const trxsig_c32 *d_b, int Lb, int span, int flags, int correlate, int cust_start,
                          int cust_len, trxsig_c32 *d_out, const int32_t *d_out_off) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_a, d_a_off, d_a_len, B) || max_len <= 0 || !d_b || Lb <= 0 || span < 0 || span > TRXSIG_CUSTOM ||
      (flags & ~7) || ((flags & 4) && correlate) || (B > 0 && (!d_out || !d_out_off)) || (span == TRXSIG_CUSTOM && (cust_start < 0 || cust_len <= 0)))
    return fail(c, TRXSIG_EINVAL, "trxsig_convolve_batch: bad argument");
  DeviceGuard g(c->device);
  const int max_out = trx_convolve_out_len(max_len, Lb, span, cust_len);
  // (OVERLAP_ONLY's length is |La - Lb| + 1: the longest output may belong to the shortest vector)
  const int grid_out = span == TRXSIG_OVERLAP_ONLY ? (max_len > Lb ? max_len : Lb) + 1 : max_out;
  HIPCHK(c, trx_launch_convolve(c->stream, (const trx_c32 *)d_a, d_a_off, d_a_len, B, grid_out, (const trx_c32 *)d_b, Lb, span,
                                flags, correlate != 0, cust_start, cust_len, (trx_c32 *)d_out, d_out_off));
  return TRXSIG_OK;
}

namespace {
// staging-area layout helper for the single-vector host forms: 256-byte aligned regions handed out in order
struct Stager {
  trxsig_ctx *c;
  size_t used = 0;
  explicit Stager(trxsig_ctx *ctx) : c(ctx) {}
  size_t take(size_t bytes) { const size_t o = used; used += (bytes + 255) & ~(size_t)255; return o; }
  char *base() const { return (char *)c->d_stage; }
};
}  // namespace

int trxsig_convolve_host(trxsig_ctx *c, const trxsig_c32 *h_a, int La, const trxsig_c32 *h_b, int Lb, int span, int flags,
                         int correlate, int cust_start, int cust_len, trxsig_c32 *h_out, int out_cap) {
  if (!c) return TRXSIG_EINVAL;
  const int nout = (La > 0 && Lb > 0) ? trx_convolve_out_len(La, Lb, span, cust_len) : -1;
  if (!h_a || !h_b || !h_out || nout <= 0 || out_cap < nout) return fail(c, TRXSIG_EINVAL, "trxsig_convolve_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_a = s.take(8 * (size_t)La), o_b = s.take(8 * (size_t)Lb), o_m = s.take(16), o_out = s.take(8 * (size_t)nout);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  const int32_t meta[3] = {0, La, 0};                      // a_off, a_len, out_off
  HIPCHK(c, hipMemcpyAsync(d + o_a, h_a, 8 * (size_t)La, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_b, h_b, 8 * (size_t)Lb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));              // meta lives on this stack frame
  rc = trxsig_convolve_batch(c, (trxsig_c32 *)(d + o_a), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1, La,
                             (trxsig_c32 *)(d + o_b), Lb, span, flags, correlate, cust_start, cust_len, (trxsig_c32 *)(d + o_out),
                             (int32_t *)(d + o_m) + 2);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_out, 8 * (size_t)nout, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return nout;
}

int trxsig_delay_vector_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                              const float *d_delay, int real_only, trxsig_c32 *d_out) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || (B > 0 && (!d_delay || !d_out || d_out == d_in)))
    return fail(c, TRXSIG_EINVAL, "trxsig_delay_vector_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_delay_vector(c->stream, c->d_tables, (const trx_c32 *)d_in, d_off, d_len, B, d_delay, real_only != 0,
                                    (trx_c32 *)d_out));
  return TRXSIG_OK;
}

int trxsig_delay_vector_host(trxsig_ctx *c, trxsig_c32 *h_x, int n, float delay, int real_only) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0) return fail(c, TRXSIG_EINVAL, "trxsig_delay_vector_host: bad argument");
  if (!(std::fabs(delay) <= TRXSIG_MAX_INDEX))
    return fail(c, TRXSIG_EINVAL, "trxsig_delay_vector_host: delay beyond +-2^24 (or not finite): the reference's sinc range reduction would not end");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_y = s.take(8 * (size_t)n), o_m = s.take(16);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  int32_t meta[3] = {0, n, 0};
  std::memcpy(&meta[2], &delay, 4);
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_delay_vector_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1,
                                 (float *)(d + o_m) + 2, real_only, (trxsig_c32 *)(d + o_y));
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_x, d + o_y, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_interpolate_point_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                                   const float *d_ix, int real_only, trxsig_c32 *d_out) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || (B > 0 && (!d_ix || !d_out)))
    return fail(c, TRXSIG_EINVAL, "trxsig_interpolate_point_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_interpolate_point(c->stream, c->d_tables, (const trx_c32 *)d_in, d_off, d_len, B, d_ix, real_only != 0,
                                         (trx_c32 *)d_out));
  return TRXSIG_OK;
}

int trxsig_interpolate_point_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, float ix, int real_only, trxsig_c32 *h_out) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0 || !h_out) return fail(c, TRXSIG_EINVAL, "trxsig_interpolate_point_host: bad argument");
  if (!(std::fabs(ix) <= TRXSIG_MAX_INDEX))
    return fail(c, TRXSIG_EINVAL, "trxsig_interpolate_point_host: index beyond +-2^24 (or not finite): the reference's sinc range reduction would not end");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(16), o_out = s.take(8);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  int32_t meta[3] = {0, n, 0};
  std::memcpy(&meta[2], &ix, 4);
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_interpolate_point_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1,
                                      (float *)(d + o_m) + 2, real_only, (trxsig_c32 *)(d + o_out));
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_out, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_peak_detect_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                             trxsig_c32 *d_peak, float *d_index, float *d_avgpwr) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || (B > 0 && !d_peak)) return fail(c, TRXSIG_EINVAL, "trxsig_peak_detect_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_peak_detect(c->stream, c->d_tables, (const trx_c32 *)d_in, d_off, d_len, B, (trx_c32 *)d_peak, d_index,
                                   d_avgpwr));
  return TRXSIG_OK;
}

int trxsig_peak_detect_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, trxsig_c32 *h_peak, float *h_index, float *h_avgpwr) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0 || !h_peak) return fail(c, TRXSIG_EINVAL, "trxsig_peak_detect_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(16), o_out = s.take(16);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  const int32_t meta[2] = {0, n};
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_peak_detect_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1,
                                (trxsig_c32 *)(d + o_out), (float *)(d + o_out) + 2, (float *)(d + o_out) + 3);
  if (rc != TRXSIG_OK) return rc;
  float res[4];
  HIPCHK(c, hipMemcpyAsync(res, d + o_out, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h_peak->re = res[0]; h_peak->im = res[1];
  if (h_index) *h_index = res[2];
  if (h_avgpwr) *h_avgpwr = res[3];
  return TRXSIG_OK;
}

int trxsig_energy_detect_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                               unsigned window, int sample_step, float thresh, float *d_avgpwr, uint8_t *d_ok) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || (sample_step != 1 && sample_step != 4))
    return fail(c, TRXSIG_EINVAL, "trxsig_energy_detect_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_energy_detect(c->stream, (const trx_c32 *)d_in, d_off, d_len, B, window, sample_step, thresh, d_avgpwr, d_ok));
  return TRXSIG_OK;
}

int trxsig_energy_detect_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, unsigned window, int sample_step, float thresh,
                              float *h_avgpwr) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0) return fail(c, TRXSIG_EINVAL, "trxsig_energy_detect_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(16), o_out = s.take(16);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  const int32_t meta[2] = {0, n};
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_energy_detect_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1, window, sample_step,
                                  thresh, (float *)(d + o_out), (uint8_t *)(d + o_out) + 8);
  if (rc != TRXSIG_OK) return rc;
  unsigned char res[16];
  HIPCHK(c, hipMemcpyAsync(res, d + o_out, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h_avgpwr) std::memcpy(h_avgpwr, res, 4);
  return res[8] ? 1 : 0;
}

namespace {
int elementwise_batch(trxsig_ctx *c, const char *who, int op, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B,
                      int max_len, const trxsig_c32 *d_scale, int real_only) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_x, d_off, d_len, B) || max_len <= 0 || (op == 0 && B > 0 && !d_scale)) return fail(c, TRXSIG_EINVAL, who);
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_elementwise(c->stream, op, c->d_tables, (trx_c32 *)d_x, d_off, d_len, B, max_len, (const trx_c32 *)d_scale,
                                   real_only != 0));
  return TRXSIG_OK;
}
}  // namespace

int trxsig_scale_vector_batch(trxsig_ctx *c, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                              const trxsig_c32 *d_scale, int real_only) {
  return elementwise_batch(c, "trxsig_scale_vector_batch: bad argument", 0, d_x, d_off, d_len, B, max_len, d_scale, real_only);
}
int trxsig_gmsk_rotate_batch(trxsig_ctx *c, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                             int reverse, int real_only) {
  return elementwise_batch(c, "trxsig_gmsk_rotate_batch: bad argument", reverse ? 2 : 1, d_x, d_off, d_len, B, max_len, nullptr,
                           real_only);
}
int trxsig_vector_slicer_batch(trxsig_ctx *c, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len) {
  return elementwise_batch(c, "trxsig_vector_slicer_batch: bad argument", 3, d_x, d_off, d_len, B, max_len, nullptr, 0);
}

int trxsig_offset_vector_batch(trxsig_ctx *c, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                               const trxsig_c32 *d_offset, int real_only) {
  if (c && B > 0 && !d_offset) return fail(c, TRXSIG_EINVAL, "trxsig_offset_vector_batch: bad argument");
  return elementwise_batch(c, "trxsig_offset_vector_batch: bad argument", 4, d_x, d_off, d_len, B, max_len, d_offset, real_only);
}

// ---- the rest of sigProcLib.h (host scalars in the reference's float arithmetic; this file is built with -ffp-contract=off) ----
float trxsig_db(float x) {                                 // sigProcLib.cpp:88-114
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
float trxsig_dbinv(float x) {                              // sigProcLib.cpp:117-144
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
int trxsig_sinc_host(const trxsig_ctx *c, float x, float *out) {   // sinc (:567-571) on sinLookup (:177-188)
  if (!c || !out) return TRXSIG_EINVAL;
  if (!((x >= 0.01F) || (x <= -0.01F))) { *out = 1.0F; return TRXSIG_OK; }
  if (!(std::fabs(x) <= 1.0e7f)) return TRXSIG_EINVAL;     // (the reference's subtract-one range reduction would not end)
  const float M_1_2PI_F = 1 / 6.28318548202514648f;
  float arg = x * M_1_2PI_F;
  while (arg > 1.0F) arg -= 1.0F;
  while (arg < 0.0F) arg += 1.0F;
  const float argT = arg * ((float)TRX_TABLESIZE);
  const int argI = (int)argT;
  const float delta = argT - argI;
  const float iDelta = 1.0F - delta;
  const float sn = iDelta * c->h_tables->sinT[argI] + delta * c->h_tables->sinT[argI + 1];
  *out = sn / x;
  return TRXSIG_OK;
}
int trxsig_gaussian_noise_host(int length, float variance, trxsig_c32 mean, trxsig_c32 *h_out) {   // :618-637
  if (length < 0 || (length > 0 && !h_out)) return TRXSIG_EINVAL;
  const float stddev = sqrtf(variance);
  for (int k = 0; k < length; k++) {
    float u1 = (float)rand() / (float)RAND_MAX;
    while (u1 == 0.0) u1 = (float)rand() / (float)RAND_MAX;
    const float u2 = (float)rand() / (float)RAND_MAX;
    const float arg = 2.0 * M_PI * u2;
    // mean + stddev*complex(cos(arg),sin(arg))*sqrtf(-2.0*log(u1)): cos / sin / log of a float are the float overloads;
    // Real * Complex scales both parts (Complex.h:226-229), Complex * Real likewise (:84), then the complex sum
    const float er = std::cos(arg) * stddev, ei = std::sin(arg) * stddev;
    const float m = sqrtf(-2.0 * std::log(u1));
    h_out[k].re = mean.re + er * m;
    h_out[k].im = mean.im + ei * m;
  }
  return TRXSIG_OK;
}

int trxsig_vector_norm2_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B, float *d_norm2,
                              float *d_power) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B)) return fail(c, TRXSIG_EINVAL, "trxsig_vector_norm2_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_vector_norm2(c->stream, (const trx_c32 *)d_in, d_off, d_len, B, d_norm2, d_power));
  return TRXSIG_OK;
}
int trxsig_vector_norm2_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, float *norm2, float *power) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0) return fail(c, TRXSIG_EINVAL, "trxsig_vector_norm2_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(16), o_out = s.take(16);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  const int32_t meta[2] = {0, n};
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_vector_norm2_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1, (float *)(d + o_out),
                                 (float *)(d + o_out) + 1);
  if (rc != TRXSIG_OK) return rc;
  float res[2];
  HIPCHK(c, hipMemcpyAsync(res, d + o_out, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (norm2) *norm2 = res[0];
  if (power) *power = res[1];
  return TRXSIG_OK;
}

int trxsig_frequency_shift_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                                 const float *d_freq, const float *d_start_phase, int real_only, trxsig_c32 *d_out, float *d_final_phase) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || (B > 0 && (!d_freq || !d_start_phase || !d_out)))
    return fail(c, TRXSIG_EINVAL, "trxsig_frequency_shift_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_frequency_shift(c->stream, c->d_tables, (const trx_c32 *)d_in, d_off, d_len, B, d_freq, d_start_phase, real_only != 0,
                                       (trx_c32 *)d_out, d_final_phase));
  return TRXSIG_OK;
}
int trxsig_frequency_shift_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, float freq, float start_phase, int real_only,
                                trxsig_c32 *h_out, float *final_phase) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0 || !h_out) return fail(c, TRXSIG_EINVAL, "trxsig_frequency_shift_host: bad argument");
  if (!(std::fabs((double)start_phase) + (double)n * std::fabs((double)freq) <= (double)trx_frequency_shift_max_phase()))
    return fail(c, TRXSIG_EINVAL, "trxsig_frequency_shift_host: phase beyond +-25000 rad (the reference's range reduction would not return)");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(32);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  int32_t meta[5] = {0, n, 0, 0, 0};
  std::memcpy(&meta[2], &freq, 4); std::memcpy(&meta[3], &start_phase, 4);
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 20, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_frequency_shift_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1, (float *)(d + o_m) + 2,
                                    (float *)(d + o_m) + 3, real_only, (trxsig_c32 *)(d + o_x), (float *)(d + o_m) + 4);
  if (rc != TRXSIG_OK) return rc;
  float fin = 0.0f;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_x, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&fin, (float *)(d + o_m) + 4, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (final_phase) *final_phase = fin;
  return TRXSIG_OK;
}

int trxsig_add_vector_batch(trxsig_ctx *c, trxsig_c32 *d_x, const int32_t *d_xoff, const int32_t *d_xlen, const trxsig_c32 *d_y,
                            const int32_t *d_yoff, const int32_t *d_ylen, int B, int max_len) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_x, d_xoff, d_xlen, B) || bad_batch(d_y, d_yoff, d_ylen, B) || max_len <= 0)
    return fail(c, TRXSIG_EINVAL, "trxsig_add_vector_batch: bad argument");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_add_vector(c->stream, (trx_c32 *)d_x, d_xoff, d_xlen, (const trx_c32 *)d_y, d_yoff, d_ylen, B, max_len));
  return TRXSIG_OK;
}
int trxsig_add_vector_host(trxsig_ctx *c, trxsig_c32 *h_x, int nx, const trxsig_c32 *h_y, int ny) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || !h_y || nx <= 0 || ny <= 0) return fail(c, TRXSIG_EINVAL, "trxsig_add_vector_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)nx), o_y = s.take(8 * (size_t)ny), o_m = s.take(16);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  const int32_t meta[3] = {0, nx, ny};
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)nx, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_y, h_y, 8 * (size_t)ny, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_add_vector_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, (trxsig_c32 *)(d + o_y),
                               (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 2, 1, nx < ny ? nx : ny);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_x, d + o_x, 8 * (size_t)nx, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_resample_linear_out_len(int n, float exp_factor) {
  if (n < 0 || !(exp_factor >= 1.0f)) return -1;
  return (int)std::ceil((float)(size_t)n * exp_factor);    // (int) ceil(wVector.size()*expFactor) (:1222)
}
int trxsig_resample_linear_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                                 float exp_factor, const trxsig_c32 *d_end_point, trxsig_c32 *d_out, const int32_t *d_out_off) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || !(exp_factor >= 1.0f) || !(exp_factor <= 65536.0f) || (B > 0 && (!d_end_point || !d_out || !d_out_off)))
    return fail(c, TRXSIG_EINVAL, "trxsig_resample_linear_batch: bad argument (resampleVector returns NULL for expFactor < 1)");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_resample_linear(c->stream, (const trx_c32 *)d_in, d_off, d_len, B, exp_factor, (const trx_c32 *)d_end_point,
                                       (trx_c32 *)d_out, d_out_off));
  return TRXSIG_OK;
}
int trxsig_resample_linear_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, float exp_factor, trxsig_c32 end_point, trxsig_c32 *h_out,
                                int out_cap) {
  if (!c) return TRXSIG_EINVAL;
  const int nout = trxsig_resample_linear_out_len(n, exp_factor);
  if (!h_x || n <= 0 || !h_out || nout < 0 || nout > out_cap) return fail(c, TRXSIG_EINVAL, "trxsig_resample_linear_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(32), o_out = s.take(8 * (size_t)(nout > 0 ? nout : 1));
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  int32_t meta[6] = {0, n, 0, 0, 0, 0};
  std::memcpy(&meta[4], &end_point, 8);
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 24, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_resample_linear_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1, exp_factor,
                                    (trxsig_c32 *)((int32_t *)(d + o_m) + 4), (trxsig_c32 *)(d + o_out), (int32_t *)(d + o_m) + 2);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_out, 8 * (size_t)nout, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return nout;
}

int trxsig_decimate_batch(trxsig_ctx *c, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                          int factor, trxsig_c32 *d_out, const int32_t *d_out_off) {
  if (!c) return TRXSIG_EINVAL;
  if (bad_batch(d_in, d_off, d_len, B) || max_len <= 0 || factor <= 1 || (B > 0 && (!d_out || !d_out_off)))
    return fail(c, TRXSIG_EINVAL, "trxsig_decimate_batch: bad argument (decimateVector returns NULL for a factor <= 1)");
  DeviceGuard g(c->device);
  HIPCHK(c, trx_launch_decimate(c->stream, (const trx_c32 *)d_in, d_off, d_len, B, max_len, factor, (trx_c32 *)d_out, d_out_off));
  return TRXSIG_OK;
}

int trxsig_elementwise_host(trxsig_ctx *c, int op, trxsig_c32 *h_x, int n, trxsig_c32 scale, int real_only) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0 || op < 0 || op > 4) return fail(c, TRXSIG_EINVAL, "trxsig_elementwise_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(16);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  int32_t meta[4] = {0, n, 0, 0};
  std::memcpy(&meta[2], &scale, 8);
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 16, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = elementwise_batch(c, "trxsig_elementwise_host: bad argument", op, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m),
                         (int32_t *)(d + o_m) + 1, 1, n, (trxsig_c32 *)((int32_t *)(d + o_m) + 2), real_only);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_x, d + o_x, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TRXSIG_OK;
}

int trxsig_decimate_host(trxsig_ctx *c, const trxsig_c32 *h_x, int n, int factor, trxsig_c32 *h_out) {
  if (!c) return TRXSIG_EINVAL;
  if (!h_x || n <= 0 || factor <= 1 || !h_out || n / factor <= 0) return fail(c, TRXSIG_EINVAL, "trxsig_decimate_host: bad argument");
  DeviceGuard g(c->device);
  Stager s(c);
  const int nout = n / factor;
  const size_t o_x = s.take(8 * (size_t)n), o_m = s.take(16), o_out = s.take(8 * (size_t)nout);
  int rc = ensure_stage(c, s.used);
  if (rc != TRXSIG_OK) return rc;
  char *d = s.base();
  const int32_t meta[3] = {0, n, 0};
  HIPCHK(c, hipMemcpyAsync(d + o_x, h_x, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + o_m, meta, 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  rc = trxsig_decimate_batch(c, (trxsig_c32 *)(d + o_x), (int32_t *)(d + o_m), (int32_t *)(d + o_m) + 1, 1, n, factor,
                             (trxsig_c32 *)(d + o_out), (int32_t *)(d + o_m) + 2);
  if (rc != TRXSIG_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(h_out, d + o_out, 8 * (size_t)nout, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return nout;
}

}  // extern "C"
