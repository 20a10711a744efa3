// trxsig_trxgroup.cpp -- include/trxsig_trxgroup.h: S Transceivers' receive side (Transceiver/Transceiver.cpp:207-410) per
// call for n_slots timeslots, the per-ARFCN state machine replayed on the device (trxsig_group.hip).  The host only
// classifies slots (expectedCorrType) and enqueues; line references: Transceiver/Transceiver.cpp unless a file is named.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "trxsig_ctx.h"
#include "trxsig_group.h"
#include "trxsig_trxgroup.h"
#include "trxsig_trxstate.h"
#include "trxsig_txq_lds.h"

namespace {
constexpr int kHyperframe = 2048 * 26 * 51;                 // GSM/GSMCommon.h:306
constexpr int kSoft = 148;                                  // soft values kept per burst (gSlotLen: what the datagram carries, :658-672)

struct Guard {
  int prev = -1;
  explicit Guard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
  ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// one column of a timeslot's segment table: the ARFCNs that share (channel combination, TSC) on that timeslot take the
// same path at every frame number
struct Column { int chanType, tsc, count; };

template <typename T>
struct DevBuf {                                             // grow-only device array
  T *p = nullptr;
  size_t cap = 0;
  // keep: leading elements to carry over a reallocation; st: the stream whose queued work may still use the old array
  hipError_t need(size_t n, hipStream_t st, size_t keep = 0) {
    if (n <= cap) return hipSuccess;
    const size_t ncap = n + n / 4 + 256;
    T *q = nullptr;
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    e = hipMalloc((void **)&q, sizeof(T) * ncap);
    if (e != hipSuccess) return e;
    if (p && keep) e = hipMemcpy(q, p, sizeof(T) * (keep < cap ? keep : cap), hipMemcpyDeviceToDevice);
    if (p) (void)hipFree(p);
    p = q; cap = ncap;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
}  // namespace

struct trxsig_trxgroup {
  trxsig_ctx *c = nullptr;
  int S = 0, sps = 1, leg = TRXSIG_TSCLEG_EQUALIZE;
  std::vector<TrxControl> ctl;
  // segment tables, re-derived when a SETSLOT / SETTSC changed something
  bool dirty = true;
  unsigned epoch_sum = 0;
  std::vector<Column> cols[8];
  int G = 1;
  uint16_t *d_gid = nullptr;                                // [8][S]
  int32_t *d_pos = nullptr;                                 // [8][S]
  TrxGroupArfcn *d_state = nullptr;
  double *d_exp = nullptr;
  int *d_err = nullptr;                                     // TrxGroupReplay::err
  // the demodulating leg runs the state machine (two waves, latency-bound) BESIDE demodulateBurst: a side stream, forked
  // from and joined back into the context's stream inside every pull
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // per-call workspace: two sets, so that in pipelined mode (trxsig_trxgroup_set_pipelined) the state machine can still be
  // replaying call i's rows on the side stream while call i+1's detectors fill the other set
  struct Work {
    DevBuf<int32_t> rowmap, seg, off, len, tap_ix, tix_g;
    DevBuf<float4> packed;
    DevBuf<uint8_t> flags, gate, ev, ev_flags, succ_g;
    DevBuf<trx_c32> amp, ev_amp;
    DevBuf<float> toa, avgpwr, toa_eq, snr, ev_toa, ev_toaeq, soft;
    DevBuf<double> thr_after, thr_g;
    hipEvent_t done = nullptr;                              // recorded on the side stream behind the set's last replay
    bool in_flight = false;                                 // ... which the context's stream has not waited for yet
    void release() {
      rowmap.release(); seg.release(); off.release(); len.release(); tap_ix.release(); packed.release(); flags.release(); gate.release();
      ev.release(); ev_flags.release(); succ_g.release(); amp.release(); ev_amp.release(); toa.release(); avgpwr.release(); toa_eq.release();
      snr.release(); ev_toa.release(); ev_toaeq.release(); soft.release(); thr_after.release(); thr_g.release(); tix_g.release();
    }
  } wk[2];
  int cur = 0;                                              // the set of the last pull
  bool pipelined = false;
  int split_rows = 0;                // trxsig_trxgroup_set_split_rows: fused pulls with at least this many rows detect their access bursts beside the normal bursts (0, the default: never)
  int beside_rows = 0x7fffffff;      // trxsig_trxgroup_set_beside_rows: calls with at least this many rows replay on the side stream (kBesideRows)
  DevBuf<trx_c32> w_tab, b_tab, in;
  DevBuf<float> chan_off;
  std::vector<int32_t> h_seg;
  TrxPinRing seg_up;                 // h_seg's way up (pinned: trxsig_ctx.h)
  // the last pull
  int n_slots = 0, n_rows = 0, n_tsc_rows = 0;
  bool have = false;
  // ---- transmit half (created on first use): queue, filler table and payload pool on the device (trxsig_grouptx.hip) ----
  bool tx_ready = false;
  bool fmod_dirty = true;                                   // a SETSLOT changed some fillerModulus (setModulus, :183-204)
  TrxGroupTx tx = {};
  uint32_t *d_dummy = nullptr;
  // the datagrams of an add call as they arrived (k_group_tx_ingest parses and sorts them) and their ARFCN ids: two device sets in
  // turn.  Three streams besides the context's:
  //   tx_up  nothing but the uploads (a copy enqueued behind a kernel or an event wait was seen to hold the HOST until that had run);
  //   tx_q   everything that touches the queues, in call order: ingest, walk, gather, (rarely) the filler moduli's upload -- the chain
  //          std::priority_queue's moves make serial runs without a cross-stream hop inside it;
  //   the context's stream: what the CALLER does with a push's output (the transmit back end: ring store, modulate + resample) --
  //          batch i's back end runs beside batch i + 1's ingest and walk.
  // Events: the ingest waits for its upload; the context's stream waits for the gather whose output the call returns; a push writes
  // one of TWO output sets in turn and waits for whatever the context's stream had been given when the push before it was called
  // (the output's contract: valid until the next push); a staging set is handed out again when its upload and the ingest that read
  // it have run (the host waits there: that is where it is held back when the device is more than a batch behind).
  static constexpr int kTxSets = 3;  // staging sets in turn: the host fills set i + 2 while set i's ingest runs (with two, the upload + arrival chain of a
                                     // batch -- ~100 us from the staging call to the arrival kernel's end -- could only start when the batch two before it was through)
  DevBuf<int32_t> tx_arfcn[kTxSets], tx_alf[kTxSets], tx_alk[kTxSets], tx_atot[kTxSets];   // (+ what k_group_tx_arrive leaves for k_group_tx_ingest, per set)
  DevBuf<uint8_t> tx_dgram[kTxSets];
  hipStream_t tx_up = nullptr, tx_q = nullptr;
  hipEvent_t tx_q_ev = nullptr, tx_out_ev[2] = {nullptr, nullptr}, tx_read_ev[kTxSets] = {};
  bool tx_q_armed = false, tx_out_armed[2] = {false, false}, tx_read_armed[kTxSets] = {};
  hipEvent_t tx_q_last = nullptr;    // the event behind the queues' stream's latest work (an ingest's tx_read_ev or a walk's tx_q_ev): one record per kernel
  unsigned tx_pushes = 0;
  // an add call's ingest is left PENDING (its upload and arrival kernel are under way): the push that usually follows takes it into
  // its own launch (trx_launch_group_tx_both); anything else that needs the queues -- another add, a queue-size query, the end --
  // launches it on its own first (tx_flush_pending)
  bool tx_pend = false;
  int tx_pend_k = 0, tx_pend_n = 0, tx_pend_ref = 0, tx_pend_far = 0;
  uint8_t *tx_pin[kTxSets] = {};     // pinned staging blocks the caller receives into (trxsig_trxgroup_tx_staging), in turn
  int tx_pin_cap[kTxSets] = {};
  bool tx_stage_held = false;        // the current set has been handed out and not yet added
  DevBuf<uint8_t> tx_bits[2], tx_fq[2];
  DevBuf<float> tx_gain[2];
  // host staging of an add call: kTxSets sets in turn, each guarded by an event recorded behind its uploads -- a set is refilled only
  // when the copies that read it have run (the library does not rely on pageable hipMemcpyAsync being synchronous)
  std::vector<uint8_t> h_fmod;
  hipEvent_t tx_fm_ev = nullptr;
  bool tx_fm_armed = false;
  hipEvent_t tx_ev[kTxSets] = {};
  bool tx_ev_armed[kTxSets] = {};
  int tx_set = 0;
  float gain_tab[26] = {0};                                 // pow(10, q), q = -12..13: every value -RSSI/10 of a signed char can take
};

namespace {
// Calls with at least this many rows replay on the group's side stream, beside the demodulator.  Round 3's default was 24,576: the
// replay was a chain of ~0.1 us per timeslot and hid under the demodulator.  Since round 4 a long call's replay runs parallel in
// time (k_group_replay_seg: 15 us instead of 88 for 468 slots) and, started beside a kernel that fills the machine, its few
// workgroups wait for that kernel to drain (80 us): everything on ONE stream is faster (309 against 279 Mbursts/s on bench.py
// --workload config4).  The side-stream arrangement stays selectable (trxsig_trxgroup_set_beside_rows) and tested.
constexpr int kBesideRows = 0x7fffffff;
#define G_HIP(g, call)                                                          \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) return trx_ctx_fail((g)->c, TRXSIG_EHIP, #call, e_);  \
  } while (0)
#define G_LIB(call)                       \
  do {                                    \
    int rc_ = (call);                     \
    if (rc_ != TRXSIG_OK) return rc_;     \
  } while (0)

// which ARFCNs travel together on each timeslot
int derive_tables(trxsig_trxgroup *g) {
  const int S = g->S;
  std::vector<uint16_t> gid((size_t)8 * S);
  std::vector<int32_t> pos((size_t)8 * S);
  int G = 1;
  for (int tn = 0; tn < 8; tn++) {
    std::vector<Column> &cs = g->cols[tn];
    cs.clear();
    for (int a = 0; a < S; a++) {
      const int ct = g->ctl[a].chanType[tn], tsc = (int)g->ctl[a].tsc;
      size_t k = 0;
      while (k < cs.size() && !(cs[k].chanType == ct && cs[k].tsc == tsc)) k++;
      if (k == cs.size()) cs.push_back(Column{ct, tsc, 0});
      gid[(size_t)tn * S + a] = (uint16_t)k;
      pos[(size_t)tn * S + a] = cs[k].count++;
    }
    if ((int)cs.size() > G) G = (int)cs.size();
  }
  g->G = G;
  G_HIP(g, hipMemcpy(g->d_gid, gid.data(), sizeof(uint16_t) * gid.size(), hipMemcpyHostToDevice));
  G_HIP(g, hipMemcpy(g->d_pos, pos.data(), sizeof(int32_t) * pos.size(), hipMemcpyHostToDevice));
  g->dirty = false;
  return TRXSIG_OK;
}

inline int row_class(const Column &c, int fn) {             // -1: OFF / IDLE (pullRadioVector returns NULL at once, :288-291)
  const int t = TrxControl::corrType(c.chanType, fn);
  return t == TRXSIG_CORR_TSC ? c.tsc : (t == TRXSIG_CORR_RACH ? TRXG_CLASS_RACH : -1);
}
}  // namespace

extern "C" {

int trxsig_trxgroup_create(trxsig_trxgroup **out, trxsig_ctx *c, int n_arfcn, int tsc_leg, int start_fn, int start_tn) {
  if (!out) return TRXSIG_EINVAL;
  *out = nullptr;
  if (!c) return TRXSIG_EINVAL;
  if (n_arfcn <= 0 || n_arfcn > 65535 || (tsc_leg != TRXSIG_TSCLEG_EQUALIZE && tsc_leg != TRXSIG_TSCLEG_DEMOD) || start_fn < 0 ||
      start_fn >= kHyperframe || start_tn < 0 || start_tn > 7)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_create: bad argument", hipSuccess);
  if (tsc_leg == TRXSIG_TSCLEG_EQUALIZE && trxsig_sps(c) != 1)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_create: the equalising TSC leg (Transceiver.cpp:391-396) needs sps == 1", hipSuccess);
  trxsig_trxgroup *g = new (std::nothrow) trxsig_trxgroup;
  if (!g) return TRXSIG_ENOMEM;
  g->c = c; trx_ctx_retain(c); g->S = n_arfcn; g->sps = trxsig_sps(c); g->leg = tsc_leg;
  g->ctl.resize((size_t)n_arfcn);
  Guard gd(trxsig_device(c));
  const int S = n_arfcn;
  std::vector<TrxGroupArfcn> st((size_t)S);
  for (TrxGroupArfcn &a : st) {                             // Transceiver::Transceiver (:58-92)
    a.thr = 250.0; a.prev_false_fn = start_fn; a.pad = 0;
    for (int k = 0; k < 8; k++) { a.est_fn[k] = start_fn; a.tap_src[k] = -1; }
  }
  std::vector<double> ex(TRXG_EXP_N);
  for (int k = -TRXG_EXP_LO; k <= TRXG_EXP_HI; k++) ex[(size_t)(k + TRXG_EXP_LO)] = std::exp(-(double)k);   // exp(-framesElapsed) (:355)
  const size_t S8 = (size_t)S * 8;
  if (hipMalloc((void **)&g->d_gid, sizeof(uint16_t) * S8) != hipSuccess || hipMalloc((void **)&g->d_pos, sizeof(int32_t) * S8) != hipSuccess ||
      hipMalloc((void **)&g->d_state, sizeof(TrxGroupArfcn) * (size_t)S) != hipSuccess ||
      hipMalloc((void **)&g->d_exp, sizeof(double) * TRXG_EXP_N) != hipSuccess ||
      hipMalloc((void **)&g->d_err, sizeof(int)) != hipSuccess || hipMemset(g->d_err, 0, sizeof(int)) != hipSuccess ||
      hipMemcpy(g->d_state, st.data(), sizeof(TrxGroupArfcn) * (size_t)S, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(g->d_exp, ex.data(), sizeof(double) * TRXG_EXP_N, hipMemcpyHostToDevice) != hipSuccess ||
      g->w_tab.need(S8 * 7, nullptr) != hipSuccess || g->b_tab.need(S8 * 5, nullptr) != hipSuccess || g->chan_off.need(S8, nullptr) != hipSuccess ||
      hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g->wk[0].done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g->wk[1].done, hipEventDisableTiming) != hipSuccess) {
    const int rc = trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_trxgroup_create: device allocation failed", hipSuccess);   // (first the error text: the release below may be the context's end)
    trxsig_trxgroup_destroy(g);
    return rc;
  }
  *out = g;
  return TRXSIG_OK;
}

void trxsig_trxgroup_destroy(trxsig_trxgroup *g) {
  if (!g) return;
  {
    Guard gd(trxsig_device(g->c));
    (void)hipStreamSynchronize((hipStream_t)trxsig_get_stream(g->c));
    if (g->side) { (void)hipStreamSynchronize(g->side); (void)hipStreamDestroy(g->side); }
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    (void)hipFree(g->d_gid); (void)hipFree(g->d_pos); (void)hipFree(g->d_state); (void)hipFree(g->d_exp); (void)hipFree(g->d_err);
    for (int k = 0; k < 2; k++) { g->wk[k].release(); if (g->wk[k].done) (void)hipEventDestroy(g->wk[k].done); }
    g->w_tab.release(); g->b_tab.release(); g->in.release(); g->chan_off.release(); g->seg_up.release();
    if (g->tx_ready) {
      (void)hipFree(g->tx.q_fn); (void)hipFree(g->tx.q_key); (void)hipFree(g->tx.q_n); (void)hipFree(g->tx.free_stack);
      (void)hipFree(g->tx.free_n); (void)hipFree(g->tx.filler); (void)hipFree(g->tx.fmod); (void)hipFree(g->tx.pool);
      (void)hipFree(g->tx.status); (void)hipFree(g->d_dummy);
    }
    for (int k = 0; k < trxsig_trxgroup::kTxSets; k++) if (g->tx_ev[k]) (void)hipEventDestroy(g->tx_ev[k]);
    if (g->tx_fm_ev) (void)hipEventDestroy(g->tx_fm_ev);
    if (g->tx_up) { (void)hipStreamSynchronize(g->tx_up); (void)hipStreamDestroy(g->tx_up); }
    if (g->tx_q) { (void)hipStreamSynchronize(g->tx_q); (void)hipStreamDestroy(g->tx_q); }
    for (int k = 0; k < trxsig_trxgroup::kTxSets; k++) {
      g->tx_arfcn[k].release(); g->tx_dgram[k].release(); g->tx_alf[k].release(); g->tx_alk[k].release(); g->tx_atot[k].release();
      if (g->tx_read_ev[k]) (void)hipEventDestroy(g->tx_read_ev[k]);
    }
    for (int k = 0; k < 2; k++) if (g->tx_out_ev[k]) (void)hipEventDestroy(g->tx_out_ev[k]);
    if (g->tx_q_ev) (void)hipEventDestroy(g->tx_q_ev);
    for (int k = 0; k < trxsig_trxgroup::kTxSets; k++) if (g->tx_pin[k]) (void)hipHostFree(g->tx_pin[k]);
    for (int k = 0; k < 2; k++) { g->tx_bits[k].release(); g->tx_fq[k].release(); g->tx_gain[k].release(); }
  }
  trx_ctx_release(g->c);
  delete g;
}

int trxsig_trxgroup_arfcns(const trxsig_trxgroup *g) { return g ? g->S : TRXSIG_EINVAL; }

int trxsig_trxgroup_control(trxsig_trxgroup *g, int arfcn, const char *command, char *response_out, int cap) {
  if (!g) return TRXSIG_EINVAL;
  if (arfcn < 0 || arfcn >= g->S || !command || !response_out || cap < 1 || std::strlen(command) >= 100)
    return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_control: bad argument", hipSuccess);
  char response[100] = {0};
  TrxControl &ctl = g->ctl[(size_t)arfcn];
  const unsigned before = ctl.epoch;
  const int answered = ctl.command(command, response);
  if (ctl.epoch != before) { g->dirty = true; g->fmod_dirty = true; }
  if (!answered) { response_out[0] = 0; return 0; }
  const int n = (int)std::strlen(response);
  if (n + 1 > cap) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_control: response buffer too small", hipSuccess);
  std::memcpy(response_out, response, (size_t)n + 1);
  return n;
}

int trxsig_trxgroup_expected_corr_type(const trxsig_trxgroup *g, int arfcn, int tn, int fn) {
  return (g && arfcn >= 0 && arfcn < g->S && tn >= 0 && tn < 8) ? g->ctl[(size_t)arfcn].expectedCorrType(tn, fn) : TRXSIG_CORR_OFF;
}

}  // extern "C"

namespace {
// where a pull's bursts are: packed complex float32 in device memory (burst (t, a) at t*slot_stride + a*arfcn_stride), or
// computed by the detectors from the raw int16 stream of a receive front end (gen != NULL; burst (t, a) = its a*nb + t)
struct PullSource {
  const trxsig_c32 *d_samples = nullptr;
  int64_t slot_stride = 0, arfcn_stride = 0;
  int burst_len = 0;
  const TrxRxGen *gen = nullptr;
  const int32_t *d_off = nullptr, *d_len = nullptr;        // listed bursts: burst t of ARFCN a = entry a*n_slots + t
};

// every replay still in flight on the side stream (pipelined mode) is waited for by the context's stream
int join_side(trxsig_trxgroup *g, hipStream_t st) {
  for (int k = 0; k < 2; k++) {
    if (!g->wk[k].in_flight) continue;
    G_HIP(g, hipStreamWaitEvent(st, g->wk[k].done, 0));
    g->wk[k].in_flight = false;
  }
  return TRXSIG_OK;
}

int pull_core(trxsig_trxgroup *g, const PullSource &src, int fn, int tn, int n_slots, trxsig_trxgroup_result *res) {
  trxsig_ctx *c = g->c;
  const int S = g->S, sps = g->sps;
  const trxsig_c32 *d_samples = src.d_samples;
  const int burst_len = src.burst_len;
  const long long cells = (long long)n_slots * S;
  if (cells > std::numeric_limits<int32_t>::max() / 2)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull: too many bursts in one call (split it)", hipSuccess);
  Guard gd(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  g->have = false;
  if (g->dirty) G_LIB(derive_tables(g));
  // the workspace set of this call: the other one than the last call's while that call may still be replaying (pipelined mode);
  // whoever used this set last (two calls ago) is waited for before anything of it is overwritten
  if (g->wk[g->cur].in_flight) g->cur ^= 1;
  trxsig_trxgroup::Work &W = g->wk[g->cur];
  if (W.in_flight) {
    G_HIP(g, hipStreamWaitEvent(st, W.done, 0));
    W.in_flight = false;
  }

  // ---- expectedCorrType for every (slot, column): rows by class ----
  const int G = g->G;
  g->h_seg.assign((size_t)n_slots * G, -1);
  int count[TRXG_NCLASS] = {0}, base[TRXG_NCLASS + 1] = {0};
  {
    int t_tn = tn, t_fn = fn;
    for (int t = 0; t < n_slots; t++) {
      const std::vector<Column> &cs = g->cols[t_tn];
      for (size_t k = 0; k < cs.size(); k++) {
        const int cl = row_class(cs[k], t_fn);
        if (cl >= 0) count[cl] += cs[k].count;
      }
      if (++t_tn == 8) { t_tn = 0; if (++t_fn == kHyperframe) t_fn = 0; }
    }
    for (int k = 0; k < TRXG_NCLASS; k++) base[k + 1] = base[k] + count[k];
    int run[TRXG_NCLASS];
    for (int k = 0; k < TRXG_NCLASS; k++) run[k] = base[k];
    t_tn = tn; t_fn = fn;
    for (int t = 0; t < n_slots; t++) {
      const std::vector<Column> &cs = g->cols[t_tn];
      for (size_t k = 0; k < cs.size(); k++) {
        const int cl = row_class(cs[k], t_fn);
        if (cl >= 0) { g->h_seg[(size_t)t * G + k] = run[cl]; run[cl] += cs[k].count; }
      }
      if (++t_tn == 8) { t_tn = 0; if (++t_fn == kHyperframe) t_fn = 0; }
    }
  }
  const int n_rows = base[TRXG_NCLASS], n_tsc = base[TRXG_CLASS_RACH];
  const size_t R = (size_t)(n_rows > 0 ? n_rows : 1), S8 = (size_t)S * 8;

  // ---- workspace (grow-only; an array that grows waits for the stream first) ----
  G_HIP(g, W.rowmap.need((size_t)cells, st)); G_HIP(g, W.packed.need((size_t)cells, st)); G_HIP(g, W.seg.need((size_t)n_slots * G, st));
  G_HIP(g, W.off.need(R, st)); G_HIP(g, W.len.need(R, st)); G_HIP(g, W.tap_ix.need(R, st));
  G_HIP(g, W.flags.need(R, st)); G_HIP(g, W.gate.need(R, st)); G_HIP(g, W.ev.need(R, st)); G_HIP(g, W.ev_flags.need(R, st));
  G_HIP(g, W.amp.need(R, st)); G_HIP(g, W.ev_amp.need(R, st));
  G_HIP(g, W.toa.need(R, st)); G_HIP(g, W.avgpwr.need(R, st)); G_HIP(g, W.toa_eq.need(R, st)); G_HIP(g, W.snr.need(R, st));
  G_HIP(g, W.ev_toa.need(R, st)); G_HIP(g, W.ev_toaeq.need(R, st)); G_HIP(g, W.thr_after.need(R, st));
  G_HIP(g, W.soft.need(R * kSoft, st));
  {
    const size_t ns = trx_group_replay_scratch(S, n_slots);
    G_HIP(g, W.thr_g.need(ns, st)); G_HIP(g, W.succ_g.need(ns, st));
    if (g->leg == TRXSIG_TSCLEG_EQUALIZE) G_HIP(g, W.tix_g.need(ns > R + 16 ? ns : R + 16, st));
  }
  G_HIP(g, g->w_tab.need((S8 + R) * 7, st, S8 * 7)); G_HIP(g, g->b_tab.need((S8 + R) * 5, st, S8 * 5));
  G_HIP(g, g->chan_off.need(S8 + R, st, S8));

  {
    void *up = nullptr;
    int slot = 0;
    const size_t bytes = sizeof(int32_t) * g->h_seg.size();
    G_HIP(g, g->seg_up.take(bytes, &up, &slot));
    std::memcpy(up, g->h_seg.data(), bytes);
    G_HIP(g, g->seg_up.upload(slot, W.seg.p, bytes, st));
  }
  TrxGroupExpand ex = {};
  ex.S = S; ex.n_slots = n_slots; ex.tn0 = tn; ex.sps = sps; ex.fixed_len = burst_len; ex.G = G;
  ex.slot_stride = src.slot_stride; ex.arfcn_stride = src.arfcn_stride; ex.base = 0; ex.rx_nb = src.gen ? src.gen->nb : 0;
  ex.src_off = src.d_off; ex.src_len = src.d_len; ex.src_nb = n_slots;
  ex.gid = g->d_gid; ex.pos = g->d_pos; ex.seg_base = W.seg.p; ex.rowmap = W.rowmap.p; ex.off = W.off.p; ex.len = W.len.p;
  G_HIP(g, trx_launch_group_expand(st, ex));

  // ---- the stateless detectors, a launch per class in use; thresholds 3.0 / 5.0 (:331, 363), energy gate off ----
  struct SideGuard {                                        // (an error return between a fork and its join must not leave the side stream working on this call's arrays)
    hipStream_t s = nullptr;
    ~SideGuard() { if (s) (void)hipStreamSynchronize(s); }
  } side_guard;
  // Selectable (trxsig_trxgroup_set_split_rows; OFF by default): a fused pull with both kinds of burst in it detects them side by side --
  // the access-burst class goes FIRST on the context's stream, so that its ~500 waves are resident before the correlator fills the
  // machine, the normal-burst classes follow on the side stream and are joined before the state machine.  Measured (round 5, config 4):
  // 195 us per step against 191 in series -- the 39 us of access-burst kernels are hidden, but two cross-stream hops (~20 us) and
  // the kernels slowing each other (correlator 43 -> 46, access-burst chain 40 -> 51) take it back.  (The other way round -- the access
  // bursts on the side stream, arriving behind a correlator that has filled every SIMD -- loses more: they run at half speed and
  // become the critical path, profiles/r05_config4_ab_rach_beside.txt.)
  const bool split = src.gen && g->split_rows > 0 && n_rows >= g->split_rows && count[TRXG_CLASS_RACH] > 0 && n_tsc > 0;
  if (split) {
    G_HIP(g, hipEventRecord(g->ev_fork, st));
    G_HIP(g, hipStreamWaitEvent(g->side, g->ev_fork, 0));
    side_guard.s = g->side;
    {
      const int k = TRXG_CLASS_RACH, b0 = base[k];
      TrxRxGen gen = *src.gen;
      gen.sel = W.off.p + b0;
      G_LIB(trx_ctx_rx_rach(c, gen, W.len.p + b0, count[k], 5.0f, -1.0f, W.flags.p + b0, (trxsig_c32 *)W.amp.p + b0, W.toa.p + b0,
                            W.avgpwr.p + b0, 1));
    }
    for (int k = 0; k < TRXG_CLASS_RACH; k++) {
      if (!count[k]) continue;
      const int b0 = base[k];
      TrxRxGen gen = *src.gen;
      gen.sel = W.off.p + b0;
      G_LIB(trx_ctx_rx_normal(c, gen, count[k], k, 3.0f, -1.0f, W.flags.p + b0, (trxsig_c32 *)W.amp.p + b0, W.toa.p + b0, W.avgpwr.p + b0,
                              nullptr, nullptr, 0, 0, g->side));
    }
    G_HIP(g, hipEventRecord(g->ev_join, g->side));
    G_HIP(g, hipStreamWaitEvent(st, g->ev_join, 0));
    side_guard.s = nullptr;
  }
  for (int k = 0; k < TRXG_NCLASS && !split; k++) {
    if (!count[k]) continue;
    const int b0 = base[k];
    if (src.gen) {                                          // the detectors compute their samples from the raw stream; off = the selection
      TrxRxGen gen = *src.gen;
      gen.sel = W.off.p + b0;
      if (k < TRXG_CLASS_RACH)
        G_LIB(trx_ctx_rx_normal(c, gen, count[k], k, 3.0f, -1.0f, W.flags.p + b0, (trxsig_c32 *)W.amp.p + b0, W.toa.p + b0,
                                W.avgpwr.p + b0, nullptr, nullptr, 0, 0));
      else
        G_LIB(trx_ctx_rx_rach(c, gen, W.len.p + b0, count[k], 5.0f, -1.0f, W.flags.p + b0, (trxsig_c32 *)W.amp.p + b0, W.toa.p + b0,
                              W.avgpwr.p + b0));
    } else if (k < TRXG_CLASS_RACH)
      G_LIB(trxsig_detect_demod_normal_batch(c, d_samples, W.off.p + b0, W.len.p + b0, count[k], k, 3.0f, -1.0f, W.flags.p + b0,
                                             (trxsig_c32 *)W.amp.p + b0, W.toa.p + b0, W.avgpwr.p + b0, nullptr, nullptr, 0, 0));
    else
      G_LIB(trxsig_detect_demod_rach_batch(c, d_samples, W.off.p + b0, W.len.p + b0, count[k], 5.0f, -1.0f, W.flags.p + b0,
                                           (trxsig_c32 *)W.amp.p + b0, W.toa.p + b0, W.avgpwr.p + b0, nullptr, nullptr, 0, 0));
  }

  // ---- the state machine, a lane per ARFCN ----
  const bool equalize = g->leg == TRXSIG_TSCLEG_EQUALIZE;
  TrxGroupReplay rp = {};
  rp.S = S; rp.n_slots = n_slots; rp.fn0 = fn; rp.tn0 = tn; rp.equalize = equalize; rp.n_tsc_rows = n_tsc;
  rp.form = trx_group_replay_form(n_slots);
  // the wave form's cache walk lists the estimating bursts itself (a class's count and rows at tix_g + base[k] + k: n_rows + 8 ints at most)
  rp.ev_list = (equalize && rp.form == 0) ? W.tix_g.p : nullptr;
  for (int k = 0; k <= TRXG_NCLASS; k++) rp.class_base[k] = base[k];
  rp.rowmap = W.rowmap.p; rp.flags = W.flags.p; rp.amp = W.amp.p; rp.avgpwr = W.avgpwr.p; rp.exp_tab = g->d_exp; rp.state = g->d_state;
  rp.gate = W.gate.p; rp.ev = W.ev.p; rp.tap_ix = W.tap_ix.p; rp.snr = W.snr.p; rp.thr_after = W.thr_after.p; rp.err = g->d_err;
  // Demodulating leg: demodulateBurst needs nothing the state machine decides except WHETHER a burst is handed up, and every
  // burst the machine accepts is one the stateless detector flagged -- so the rows the detectors flagged are demodulated on
  // the context's stream while the machine replays on the side stream (two waves for ~0.1 us per slot: it fills no CU), and
  // d_valid (the machine's verdict) says which soft vectors count.  The equalising leg needs the machine's events first.
  // (a cross-stream dependency costs ~10 us each way: a small call keeps everything on one stream -- 1,024 bursts took 77
  // instead of 48 us with the fork, 65,536 take 229 instead of 281)
  const bool lean = !equalize;
  const bool beside = lean && n_rows >= g->beside_rows;   // (trxsig_trxgroup_set_beside_rows: A/B and the tests of the side-stream arrangement)
  const bool piped = beside && g->pipelined;                // the join is left to the next call but one / trxsig_trxgroup_sync
  if (!piped) G_LIB(join_side(g, st));                      // (state order: nothing replays on this stream before the side stream is done)
  G_HIP(g, trx_launch_group_pack(st, rp, W.packed.p));
  if (beside) {
    G_HIP(g, hipEventRecord(g->ev_fork, st));
    G_HIP(g, hipStreamWaitEvent(g->side, g->ev_fork, 0));
    side_guard.s = g->side;
  }
  G_HIP(g, trx_launch_group_replay(beside ? g->side : st, rp, W.packed.p, W.thr_g.p, W.succ_g.p, equalize ? W.tix_g.p : nullptr,
                                   trx_ctx_profiler(c)));
  if (beside) G_HIP(g, hipEventRecord(piped ? W.done : g->ev_join, g->side));
  const uint8_t *demod_gate = beside ? W.flags.p : W.gate.p;   // (gate holds TRXSIG_F_DETECT or 0: the same mask serves both)

  // ---- what comes back as a SoftVector ----
  if (n_tsc > 0) {
    if (equalize) {
      for (int k = 0; k < TRXG_CLASS_RACH; k++) {           // :341-349 for the rows the replay marked
        if (!count[k]) continue;
        const int b0 = base[k];
        G_LIB(trx_ctx_group_estimate(c, d_samples, W.off.p + b0, W.len.p + b0, count[k], k, W.ev.p + b0, W.snr.p + b0, W.ev_flags.p + b0,
                                     (trxsig_c32 *)W.ev_amp.p + b0, W.ev_toa.p + b0, W.ev_toaeq.p + b0, g->chan_off.p + S8 + b0,
                                     (trxsig_c32 *)g->w_tab.p + (S8 + b0) * 7, (trxsig_c32 *)g->b_tab.p + (S8 + b0) * 5,
                                     rp.ev_list ? rp.ev_list + b0 + k : nullptr));
      }
      G_HIP(g, trx_launch_group_toa_eq(st, n_tsc, W.gate.p, W.toa.p, W.tap_ix.p, g->chan_off.p, W.toa_eq.p));
      G_LIB(trx_ctx_group_equalize(c, d_samples, W.off.p, W.len.p, n_tsc, (const trxsig_c32 *)W.amp.p, W.toa_eq.p, W.gate.p,
                                   (const trxsig_c32 *)g->w_tab.p, (const trxsig_c32 *)g->b_tab.p, W.tap_ix.p, W.soft.p, kSoft, kSoft));
    } else if (!src.gen) {
      G_LIB(trx_ctx_demod_masked(c, d_samples, W.off.p, W.len.p, n_tsc, (const trxsig_c32 *)W.amp.p, W.toa.p, demod_gate, TRXSIG_F_DETECT,
                                 W.soft.p, kSoft, kSoft));
    }
  }
  if (src.gen) {                                            // normal and access bursts alike: demodulateBurst on every gated row
    TrxRxGen gen = *src.gen;
    gen.sel = W.off.p;
    G_LIB(trx_ctx_rx_demod(c, gen, n_rows, (const trxsig_c32 *)W.amp.p, W.toa.p, demod_gate, TRXSIG_F_DETECT, W.soft.p, kSoft, kSoft));
  } else if (n_rows > n_tsc)                                // :385-388
    G_LIB(trx_ctx_demod_masked(c, d_samples, W.off.p + n_tsc, W.len.p + n_tsc, n_rows - n_tsc, (const trxsig_c32 *)W.amp.p + n_tsc,
                               W.toa.p + n_tsc, demod_gate + n_tsc, TRXSIG_F_DETECT, W.soft.p + (size_t)n_tsc * kSoft, kSoft, kSoft));
  if (piped) {
    W.in_flight = true;                                     // d_valid / d_threshold / the state: complete behind W.done
    side_guard.s = nullptr;
  } else if (beside) {
    G_HIP(g, hipStreamWaitEvent(st, g->ev_join, 0));
    side_guard.s = nullptr;                                 // joined: ordered on the context's stream from here on
  }
  if (equalize) G_HIP(g, trx_launch_group_commit(st, S, g->d_state, g->w_tab.p, g->b_tab.p, g->chan_off.p));

  g->n_slots = n_slots; g->n_rows = n_rows; g->n_tsc_rows = n_tsc; g->have = true;
  if (res) {
    res->n_slots = n_slots; res->n_arfcn = S; res->n_rows = n_rows;
    res->d_row = W.rowmap.p; res->d_valid = W.gate.p; res->d_flags = W.flags.p; res->d_amp = (const trxsig_c32 *)W.amp.p;
    res->d_toa = W.toa.p; res->d_avgpwr = W.avgpwr.p; res->d_threshold = W.thr_after.p; res->d_soft = W.soft.p; res->soft_stride = kSoft;
  }
  return TRXSIG_OK;
}
}  // namespace

extern "C" {

int trxsig_trxgroup_pull(trxsig_trxgroup *g, const trxsig_c32 *d_samples, int64_t slot_stride, int64_t arfcn_stride, int burst_len,
                         int fn, int tn, int n_slots, trxsig_trxgroup_result *res) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  const int S = g->S, sps = g->sps;
  if (!d_samples || n_slots <= 0 || fn < 0 || fn >= kHyperframe || tn < 0 || tn > 7 || slot_stride < 0 || arfcn_stride < 0 || burst_len < 0 ||
      (burst_len > 0 && (burst_len % sps != 0 || burst_len < 92 * sps || burst_len > 157 * sps)))
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull: bad argument", hipSuccess);
  const long long last = (long long)(n_slots - 1) * slot_stride + (long long)(S - 1) * arfcn_stride + 157LL * sps;
  if (last > std::numeric_limits<int32_t>::max())
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull: the batch's sample offsets must stay below 2^31 (split the call)", hipSuccess);
  PullSource src;
  src.d_samples = d_samples; src.slot_stride = slot_stride; src.arfcn_stride = arfcn_stride; src.burst_len = burst_len;
  return pull_core(g, src, fn, tn, n_slots, res);
}

int trxsig_trxgroup_pull_bursts(trxsig_trxgroup *g, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length,
                                int n_per_arfcn, int fn, int tn, trxsig_trxgroup_result *res) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  if (!d_samples || !d_offset || !d_length || n_per_arfcn <= 0 || fn < 0 || fn >= kHyperframe || tn < 0 || tn > 7)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_bursts: bad argument", hipSuccess);
  PullSource src;
  src.d_samples = d_samples; src.d_off = d_offset; src.d_len = d_length;
  return pull_core(g, src, fn, tn, n_per_arfcn, res);
}

int trxsig_trxgroup_pull_rxfe(trxsig_trxgroup *g, trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, int fn, int *n_slots,
                              trxsig_trxgroup_result *res) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  if (!fe || !n_slots || fn < 0 || fn >= kHyperframe) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_rxfe: bad argument", hipSuccess);
  if (trx_rxfe_ctx(fe) != c) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_rxfe: the front end lives on another context", hipSuccess);
  if (g->leg != TRXSIG_TSCLEG_DEMOD || g->sps != 4) {
    // The fused front end is the 260 : 96 resampler feeding the demodulating leg.  Everything else -- the equalising leg at one
    // sample per symbol, the reference's own configuration -- goes through the resampled stream: push, pop, and the group on the
    // bursts the pop lists (same results as the fused form where both exist: tests/test_gpu_trxgroup.py).
    if (trx_rxfe_streams(fe) != g->S) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_rxfe: one stream per ARFCN, please", hipSuccess);
    G_LIB(trxsig_rxfe_push(fe, d_iq, n_chunks));
    const trxsig_c32 *xs = nullptr;
    const int32_t *off = nullptr, *len = nullptr;
    const int tn0 = trx_rxfe_next_tn(fe);
    int nb = 0;
    G_LIB(trxsig_rxfe_pop(fe, &xs, &off, &len, nullptr, 0, &nb));
    *n_slots = nb;
    if (nb <= 0) {
      g->have = false;
      if (res) std::memset(res, 0, sizeof *res);
      return TRXSIG_OK;
    }
    return trxsig_trxgroup_pull_bursts(g, xs, off, len, nb, fn, tn0, res);
  }
  TrxRxfePush p;
  G_LIB(trx_rxfe_fused_begin(fe, d_iq, n_chunks, &p));
  if (p.n_streams != g->S) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_rxfe: one stream per ARFCN, please", hipSuccess);
  *n_slots = p.nb;
  if (p.nb > 0) {
    PullSource src;
    src.gen = &p.gen;
    G_LIB(pull_core(g, src, fn, p.tn0, p.nb, res));
  } else {
    g->have = false;
    if (res) std::memset(res, 0, sizeof *res);
  }
  return trx_rxfe_fused_end(fe, d_iq, n_chunks, p);
}

int trxsig_trxgroup_collect(trxsig_trxgroup *g, uint8_t *h_valid, float *h_soft, int *h_rssi, int *h_timing, double *h_threshold) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  if (!g->have) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_collect: no pull to collect", hipSuccess);
  if (!h_valid || !h_rssi || !h_timing) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_collect: bad argument", hipSuccess);
  Guard gd(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  G_LIB(join_side(g, st));
  trxsig_trxgroup::Work &W = g->wk[g->cur];
  const size_t cells = (size_t)g->n_slots * g->S, R = (size_t)g->n_rows;
  std::vector<int32_t> row(cells);
  std::vector<uint8_t> gate(R);
  std::vector<trx_c32> amp(R);
  std::vector<float> toa(R), soft(h_soft ? R * kSoft : 0);
  std::vector<double> thr(h_threshold ? R : 0);
  int replay_err = 0;
  G_HIP(g, hipMemcpyAsync(&replay_err, g->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  G_HIP(g, hipMemcpyAsync(row.data(), W.rowmap.p, sizeof(int32_t) * cells, hipMemcpyDeviceToHost, st));
  if (R) {
    G_HIP(g, hipMemcpyAsync(gate.data(), W.gate.p, R, hipMemcpyDeviceToHost, st));
    G_HIP(g, hipMemcpyAsync(amp.data(), W.amp.p, sizeof(trx_c32) * R, hipMemcpyDeviceToHost, st));
    G_HIP(g, hipMemcpyAsync(toa.data(), W.toa.p, sizeof(float) * R, hipMemcpyDeviceToHost, st));
    if (h_soft) G_HIP(g, hipMemcpyAsync(soft.data(), W.soft.p, sizeof(float) * R * kSoft, hipMemcpyDeviceToHost, st));
    if (h_threshold) G_HIP(g, hipMemcpyAsync(thr.data(), W.thr_after.p, sizeof(double) * R, hipMemcpyDeviceToHost, st));
  }
  G_HIP(g, hipStreamSynchronize(st));
  if (replay_err) {                                         // (k_group_replay_seg's round bound: the thresholds of that pull are not validated)
    (void)hipMemsetAsync(g->d_err, 0, sizeof(int), st);
    return trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_trxgroup: the time-parallel replay of the state machine did not converge within its round bound", hipSuccess);
  }
  for (size_t i = 0; i < cells; i++) {
    const int r = row[i];
    const bool ok = r >= 0 && (gate[(size_t)r] & TRXSIG_F_DETECT);
    h_valid[i] = ok ? 1 : 0;
    h_rssi[i] = 0; h_timing[i] = 0;
    if (h_threshold) h_threshold[i] = r >= 0 ? thr[(size_t)r] : std::numeric_limits<double>::quiet_NaN();
    if (!ok) continue;
    // :400-402 -- Complex::abs() is (float)sqrt((double)norm2) (Complex.h:131); round() = half away from zero
    const trx_c32 a = amp[(size_t)r];
    const float n2 = a.i * a.i + a.r * a.r;
    const float absA = (float)std::sqrt((double)n2);
    h_rssi[i] = (int)std::floor(20.0 * std::log10(9450.0 / absA));
    h_timing[i] = (int)std::round(toa[(size_t)r] * 256.0 / g->sps);
    if (h_soft) std::memcpy(h_soft + i * kSoft, soft.data() + (size_t)r * kSoft, sizeof(float) * kSoft);
  }
  return TRXSIG_OK;
}

int trxsig_trxgroup_pull_host(trxsig_trxgroup *g, const trxsig_c32 *h_samples, int64_t slot_stride, int64_t arfcn_stride, int burst_len,
                              int fn, int tn, int n_slots) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  if (!h_samples || n_slots <= 0 || slot_stride < 0 || arfcn_stride < 0)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_host: bad argument", hipSuccess);
  const long long span = (long long)(n_slots - 1) * slot_stride + (long long)(g->S - 1) * arfcn_stride +
                         (burst_len > 0 ? burst_len : 157LL * g->sps);
  if (span > std::numeric_limits<int32_t>::max())
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_pull_host: the batch's sample offsets must stay below 2^31", hipSuccess);
  Guard gd(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  G_HIP(g, g->in.need((size_t)span, st));
  G_HIP(g, hipMemcpyAsync(g->in.p, h_samples, sizeof(trx_c32) * (size_t)span, hipMemcpyHostToDevice, st));
  return trxsig_trxgroup_pull(g, (const trxsig_c32 *)g->in.p, slot_stride, arfcn_stride, burst_len, fn, tn, n_slots, nullptr);
}

int trxsig_trxgroup_set_pipelined(trxsig_trxgroup *g, int on) {
  if (!g) return TRXSIG_EINVAL;
  if (g->leg != TRXSIG_TSCLEG_DEMOD && on)
    return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_set_pipelined: the demodulating TSC leg only (the equalising leg needs the machine's events inside the call)", hipSuccess);
  Guard gd(trxsig_device(g->c));
  G_LIB(join_side(g, (hipStream_t)trxsig_get_stream(g->c)));
  g->pipelined = on != 0;
  return TRXSIG_OK;
}

int trxsig_trxgroup_set_split_rows(trxsig_trxgroup *g, int rows) {
  if (!g) return TRXSIG_EINVAL;
  if (rows < 0) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_set_split_rows: negative threshold", hipSuccess);
  g->split_rows = rows;
  return TRXSIG_OK;
}

int trxsig_trxgroup_set_beside_rows(trxsig_trxgroup *g, int rows) {
  if (!g) return TRXSIG_EINVAL;
  if (rows < 0) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_set_beside_rows: negative threshold", hipSuccess);
  Guard gd(trxsig_device(g->c));
  G_LIB(join_side(g, (hipStream_t)trxsig_get_stream(g->c)));
  g->beside_rows = rows == 0 ? kBesideRows : rows;
  return TRXSIG_OK;
}

int trxsig_trxgroup_sync(trxsig_trxgroup *g) {
  if (!g) return TRXSIG_EINVAL;
  Guard gd(trxsig_device(g->c));
  return join_side(g, (hipStream_t)trxsig_get_stream(g->c));
}

int trxsig_trxgroup_energy_threshold(trxsig_trxgroup *g, int arfcn, double *thr) {
  if (!g) return TRXSIG_EINVAL;
  if (arfcn < 0 || arfcn >= g->S || !thr) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_energy_threshold: bad argument", hipSuccess);
  Guard gd(trxsig_device(g->c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(g->c);
  TrxGroupArfcn a;
  G_LIB(join_side(g, st));
  G_HIP(g, hipMemcpyAsync(&a, g->d_state + arfcn, sizeof a, hipMemcpyDeviceToHost, st));
  G_HIP(g, hipStreamSynchronize(st));
  *thr = a.thr;
  return TRXSIG_OK;
}

}  // extern "C"

// ======================================================================================================================
// The transmit half: addRadioVector (:100-113) and pushRadioVector (:138-181) for every ARFCN of the group
// ======================================================================================================================
namespace {
const char kDummyBits[149] =                                // gDummyBurst (GSM/GSMCommon.cpp:52-55, GSM 05.02 5.2.6)
    "0001111101101110110000010100100111000001001000100000001111100011100010111000101110001010111010010100"
    "011001100111001111010011111000100101111101010000";
constexpr int kTxQueueCap = 256;                            // queued bursts per ARFCN (32 frames' worth of all eight timeslots)

int tx_setup(trxsig_trxgroup *g) {
  if (g->tx_ready) return TRXSIG_OK;
  const int S = g->S;
  TrxGroupTx &x = g->tx;
  x.S = S; x.qcap = kTxQueueCap; x.npool = kTxQueueCap + 102 * 8;
  const size_t nq = (size_t)x.qcap * S, np = (size_t)x.npool * S;
  G_HIP(g, hipMalloc((void **)&x.q_fn, 4 * nq));
  G_HIP(g, hipMalloc((void **)&x.q_key, 4 * nq));
  G_HIP(g, hipMalloc((void **)&x.q_n, 4 * (size_t)S));
  G_HIP(g, hipMalloc((void **)&x.free_stack, 2 * np));
  G_HIP(g, hipMalloc((void **)&x.free_n, 4 * (size_t)S));
  G_HIP(g, hipMalloc((void **)&x.filler, 2 * (size_t)102 * 8 * S));
  G_HIP(g, hipMalloc((void **)&x.fmod, (size_t)8 * S));
  G_HIP(g, hipMalloc((void **)&x.pool, 4 * np * TRXG_PAYLOAD_WORDS));
  G_HIP(g, hipMalloc((void **)&x.status, 4 * (size_t)S));
  G_HIP(g, hipMalloc((void **)&g->d_dummy, 4 * TRXG_PAYLOAD_WORDS));
  // Transceiver::Transceiver (:66-75): every filler entry is the dummy burst, unscaled (a gain of 1 is the same samples)
  std::vector<int16_t> fs(np), fill((size_t)102 * 8 * S, (int16_t)-1);
  for (int k = 0; k < x.npool; k++)
    for (int a = 0; a < S; a++) fs[(size_t)k * S + a] = (int16_t)(x.npool - 1 - k);   // slot 0 is handed out first
  std::vector<int32_t> cnt((size_t)S, x.npool);
  uint32_t dummy[TRXG_PAYLOAD_WORDS];
  uint8_t *db = (uint8_t *)dummy;
  for (int i = 0; i < 148; i++) db[i] = kDummyBits[i] == '1';
  const float one = 1.0f;
  std::memcpy(db + 148, &one, 4);
  G_HIP(g, hipMemcpy(x.free_stack, fs.data(), 2 * np, hipMemcpyHostToDevice));
  G_HIP(g, hipMemcpy(x.filler, fill.data(), 2 * fill.size(), hipMemcpyHostToDevice));
  G_HIP(g, hipMemcpy(x.free_n, cnt.data(), 4 * (size_t)S, hipMemcpyHostToDevice));
  G_HIP(g, hipMemset(x.q_n, 0, 4 * (size_t)S));
  G_HIP(g, hipMemset(x.status, 0, 4 * (size_t)S));
  G_HIP(g, hipMemcpy(g->d_dummy, dummy, sizeof dummy, hipMemcpyHostToDevice));
  x.dummy = g->d_dummy;
  // scaleVector(*modBurst, pow(10, -RSSI/10)) (:108): integer division, pow in double, the scale a Complex<float>
  for (int q = -12; q <= 13; q++) g->gain_tab[q + 12] = (float)std::pow(10, q);
  for (int k = 0; k < trxsig_trxgroup::kTxSets; k++) G_HIP(g, hipEventCreateWithFlags(&g->tx_ev[k], hipEventDisableTiming));
  G_HIP(g, hipEventCreateWithFlags(&g->tx_fm_ev, hipEventDisableTiming));

  g->tx_ready = true;
  g->fmod_dirty = true;
  return TRXSIG_OK;
}

// the next host staging set, free to be refilled (its previous uploads have run)
int tx_take_set(trxsig_trxgroup *g, int *k) {
  *k = g->tx_set = (g->tx_set + 1) % trxsig_trxgroup::kTxSets;
  if (g->tx_ev_armed[*k]) { G_HIP(g, hipEventSynchronize(g->tx_ev[*k])); g->tx_ev_armed[*k] = false; }
  // ... and its device arrays: free when the ingest that read them two calls ago has run (here the host is held back when the
  // device is more than a batch behind)
  if (g->tx_read_armed[*k]) { G_HIP(g, hipEventSynchronize(g->tx_read_ev[*k])); g->tx_read_armed[*k] = false; }
  return TRXSIG_OK;
}
int tx_seal_set(trxsig_trxgroup *g, int k, hipStream_t st) {
  G_HIP(g, hipEventRecord(g->tx_ev[k], st));
  g->tx_ev_armed[k] = true;
  return TRXSIG_OK;
}
// the queues' stream (created on first use, with the events), and the context's stream behind everything it has been given
int tx_streams(trxsig_trxgroup *g) {
  if (g->tx_q) return TRXSIG_OK;
  G_HIP(g, hipStreamCreateWithFlags(&g->tx_up, hipStreamNonBlocking));
  G_HIP(g, hipStreamCreateWithFlags(&g->tx_q, hipStreamNonBlocking));
  G_HIP(g, hipEventCreateWithFlags(&g->tx_q_ev, hipEventDisableTiming));
  for (int j = 0; j < trxsig_trxgroup::kTxSets; j++) G_HIP(g, hipEventCreateWithFlags(&g->tx_read_ev[j], hipEventDisableTiming));
  for (int j = 0; j < 2; j++) G_HIP(g, hipEventCreateWithFlags(&g->tx_out_ev[j], hipEventDisableTiming));
  return TRXSIG_OK;
}
// "the queues' stream has run up to here": ONE event record behind every kernel of that stream (each record, each wait is a packet
// the stream's dependent launches queue up behind: ~3 us apiece on the serial chain)
int tx_q_mark(trxsig_trxgroup *g, hipEvent_t ev) {
  G_HIP(g, hipEventRecord(ev, g->tx_q));
  g->tx_q_last = ev;
  g->tx_q_armed = true;
  return TRXSIG_OK;
}
int tx_join(trxsig_trxgroup *g, hipStream_t st) {
  if (g->tx_q_armed) { G_HIP(g, hipStreamWaitEvent(st, g->tx_q_last, 0)); g->tx_q_armed = false; }
  return TRXSIG_OK;
}
// the queues' stream behind set k's upload and arrival kernel (spared when they have run)
int tx_wait_arrival(trxsig_trxgroup *g, int k) {
  const bool ran = hipEventQuery(g->tx_ev[k]) == hipSuccess;
  (void)hipGetLastError();                                  // (hipErrorNotReady is an answer, not an error to be found by the next launch check)
  if (!ran) G_HIP(g, hipStreamWaitEvent(g->tx_q, g->tx_ev[k], 0));
  return TRXSIG_OK;
}
// the pending ingest as a launch of its own
int tx_flush_pending(trxsig_trxgroup *g) {
  if (!g->tx_pend) return TRXSIG_OK;
  const int k = g->tx_pend_k;
  g->tx_pend = false;
  G_LIB(tx_wait_arrival(g, k));
  G_HIP(g, trx_launch_group_tx_ingest(g->tx_q, g->tx, g->tx_pend_n, g->tx_dgram[k].p, g->tx_alf[k].p, g->tx_alk[k].p, g->tx_atot[k].p, g->gain_tab,
                                      g->tx_pend_ref, g->tx_pend_far));
  G_LIB(tx_q_mark(g, g->tx_read_ev[k]));                   // (the set's device arrays are free behind it; the context's stream joins on it)
  g->tx_read_armed[k] = true;
  return TRXSIG_OK;
}
// fillerModulus[TN] of every ARFCN (setModulus, :183-204) after a SETSLOT (rare: its own host staging vector, waited for before it is refilled)
int tx_sync_modulus(trxsig_trxgroup *g, hipStream_t st) {
  if (!g->fmod_dirty) return TRXSIG_OK;
  if (g->tx_fm_armed) { G_HIP(g, hipEventSynchronize(g->tx_fm_ev)); g->tx_fm_armed = false; }
  std::vector<uint8_t> &fm = g->h_fmod;
  fm.assign((size_t)8 * g->S, 0);
  for (int tn = 0; tn < 8; tn++)
    for (int a = 0; a < g->S; a++) fm[(size_t)tn * g->S + a] = (uint8_t)g->ctl[(size_t)a].fillerModulus[tn];
  G_HIP(g, hipMemcpyAsync(g->tx.fmod, fm.data(), fm.size(), hipMemcpyHostToDevice, st));
  G_HIP(g, hipEventRecord(g->tx_fm_ev, st));
  g->tx_fm_armed = true;
  g->fmod_dirty = false;
  return TRXSIG_OK;
}
}  // namespace

extern "C" {

// the pinned staging block of set k, grown to hold n_max datagrams + their ARFCN ids ([ids: 4 n_max bytes][datagrams: 154 n_max])
static int tx_stage_need(trxsig_trxgroup *g, int k, int n_max) {
  if (n_max <= g->tx_pin_cap[k]) return TRXSIG_OK;
  if (g->tx_ev_armed[k]) { G_HIP(g, hipEventSynchronize(g->tx_ev[k])); g->tx_ev_armed[k] = false; }
  const int cap = n_max + n_max / 4 + 256;
  void *q = nullptr;
  G_HIP(g, hipHostMalloc(&q, (size_t)cap * (TRXSIG_TX_DATAGRAM_BYTES + 4), hipHostMallocDefault));
  if (g->tx_pin[k]) (void)hipHostFree(g->tx_pin[k]);
  g->tx_pin[k] = (uint8_t *)q;
  g->tx_pin_cap[k] = cap;
  return TRXSIG_OK;
}

int trxsig_trxgroup_tx_staging(trxsig_trxgroup *g, int n_max, uint8_t **h_datagrams, int32_t **h_arfcn) {
  if (!g) return TRXSIG_EINVAL;
  if (n_max <= 0 || !h_datagrams || !h_arfcn) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_tx_staging: bad argument", hipSuccess);
  Guard gd(trxsig_device(g->c));
  G_LIB(tx_setup(g));
  if (!g->tx_stage_held) {                                  // the next set, free to be refilled (its previous upload has run)
    int set = 0;
    G_LIB(tx_take_set(g, &set));
    g->tx_stage_held = true;
  }
  const int k = g->tx_set;
  G_LIB(tx_stage_need(g, k, n_max));
  *h_arfcn = (int32_t *)g->tx_pin[k];
  *h_datagrams = g->tx_pin[k] + (size_t)4 * g->tx_pin_cap[k];
  return TRXSIG_OK;
}

// driveTransmitPriorityQueue's checks on the headers (:596-620), then ONE upload of the block as it arrived and one launch
static int tx_add_staged(trxsig_trxgroup *g, int n) {
  trxsig_ctx *c = g->c;
  const int S = g->S, k = g->tx_set;
  const int32_t *h_arfcn = (const int32_t *)g->tx_pin[k];
  const uint8_t *h_d = g->tx_pin[k] + (size_t)4 * g->tx_pin_cap[k];
  int ref_fn = 0, far = 0;                                  // (the kernel's packed queue entries are relative to the first datagram's frame)
  for (int i = 0; i < n; i++) {
    const uint8_t *d = h_d + (size_t)i * TRXSIG_TX_DATAGRAM_BYTES;
    const int a = h_arfcn[i], tn = (int)(int8_t)d[0];
    const uint32_t fn = ((uint32_t)d[1] << 24) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 8) | d[4];
    if (a < 0 || a >= S || tn < 0 || tn > 7 || fn >= (uint32_t)kHyperframe)
      return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_add_bursts: ARFCN, timeslot or frame number out of range (nothing was queued)", hipSuccess);
    if (i == 0) ref_fn = (int)fn;
    const int32_t dd = trxq_fn_delta((int32_t)fn, ref_fn);  // (half the packed entries' window: the push that takes this ingest into its launch
    far |= dd < -TRXQ_PK_WIN / 2 || dd >= TRXQ_PK_WIN / 2;  //  may start up to the other half away from ref_fn)
  }
  G_LIB(tx_streams(g));                                     // (the filler moduli are the walk's business: trxsig_trxgroup_push uploads them)
  hipStream_t up = g->tx_up, q = g->tx_q;
  // (set k's device arrays are free: tx_take_set has waited for the ingest that read them two calls ago)
  G_HIP(g, g->tx_dgram[k].need((size_t)n * TRXSIG_TX_DATAGRAM_BYTES + 8, up));   // (+ 8: the ingest kernel reads whole aligned words round the last payload)
  G_HIP(g, g->tx_arfcn[k].need((size_t)n, up));
  size_t tot_ints = 0;
  const size_t list_ints = trx_group_tx_arrive_ints(S, n, &tot_ints);
  G_HIP(g, g->tx_alf[k].need(list_ints, up)); G_HIP(g, g->tx_alk[k].need(list_ints, up)); G_HIP(g, g->tx_atot[k].need(tot_ints, up));
  G_HIP(g, hipMemcpyAsync(g->tx_arfcn[k].p, h_arfcn, 4 * (size_t)n, hipMemcpyHostToDevice, up));
  G_HIP(g, hipMemcpyAsync(g->tx_dgram[k].p, h_d, (size_t)n * TRXSIG_TX_DATAGRAM_BYTES, hipMemcpyHostToDevice, up));
  // the arrival half -- parse, sort by ARFCN -- needs nothing of the queues: it runs here, behind its upload, beside the previous batch's walk
  G_HIP(g, trx_launch_group_tx_arrive(up, S, n, g->tx_dgram[k].p, g->tx_arfcn[k].p, g->tx_alf[k].p, g->tx_alk[k].p, g->tx_atot[k].p));
  G_LIB(tx_seal_set(g, k, up));                             // (the pinned set is the DMA's until this event has passed: the next staging call takes the other)
  g->tx_stage_held = false;
  (void)q;
  G_LIB(tx_flush_pending(g));                               // (an add behind an add: the earlier one's ingest goes first)
  g->tx_pend = true; g->tx_pend_k = k; g->tx_pend_n = n; g->tx_pend_ref = ref_fn; g->tx_pend_far = far;
  return TRXSIG_OK;
}

int trxsig_trxgroup_add_staged(trxsig_trxgroup *g, int n) {
  if (!g) return TRXSIG_EINVAL;
  if (!g->tx_stage_held || n < 0 || n > g->tx_pin_cap[g->tx_set])
    return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_add_staged: no staging block is held (trxsig_trxgroup_tx_staging first) or n exceeds it", hipSuccess);
  if (n == 0) return TRXSIG_OK;
  Guard gd(trxsig_device(g->c));
  return tx_add_staged(g, n);
}

int trxsig_trxgroup_add_bursts(trxsig_trxgroup *g, const uint8_t *h_datagrams, const int32_t *h_arfcn, int n) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  if (n < 0 || (n > 0 && (!h_datagrams || !h_arfcn))) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_add_bursts: bad argument", hipSuccess);
  if (n == 0) return TRXSIG_OK;
  // the caller's (pageable) arrays into the pinned staging block, then as trxsig_trxgroup_add_staged
  uint8_t *pd = nullptr;
  int32_t *pa = nullptr;
  G_LIB(trxsig_trxgroup_tx_staging(g, n, &pd, &pa));
  std::memcpy(pa, h_arfcn, 4 * (size_t)n);
  std::memcpy(pd, h_datagrams, (size_t)n * TRXSIG_TX_DATAGRAM_BYTES);
  Guard gd(trxsig_device(c));
  return tx_add_staged(g, n);
}

int trxsig_trxgroup_push(trxsig_trxgroup *g, int fn, int tn, int n_slots, const uint8_t **d_bits, const float **d_gain,
                         const uint8_t **d_from_queue) {
  if (!g) return TRXSIG_EINVAL;
  trxsig_ctx *c = g->c;
  // (n_slots < 8 * gHyperframe: k_group_tx_push wraps a slot's frame number with at most two subtractions, trxsig_grouptx.hip)
  if (fn < 0 || fn >= kHyperframe || tn < 0 || tn > 7 || n_slots <= 0 || (long long)n_slots * g->S > (1LL << 28) ||
      (long long)n_slots >= 8LL * kHyperframe)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_trxgroup_push: bad argument", hipSuccess);
  Guard gd(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  G_LIB(tx_setup(g));
  G_LIB(tx_streams(g));
  hipStream_t q = g->tx_q;
  // what the context's stream has been given so far is what may still read the output of the push before this one: the push after
  // this one (same output set) waits for it
  const int o = (int)(g->tx_pushes & 1u);
  G_HIP(g, hipEventRecord(g->tx_out_ev[o ^ 1], st));
  g->tx_out_armed[o ^ 1] = true;
  if (g->tx_out_armed[o]) {                                 // (a step later that work has normally run: then the queues' stream is spared the wait)
    const bool ran = hipEventQuery(g->tx_out_ev[o]) == hipSuccess;
    (void)hipGetLastError();                                // (hipErrorNotReady is an answer, not an error to be found by the next launch check)
    if (!ran) G_HIP(g, hipStreamWaitEvent(q, g->tx_out_ev[o], 0));
    g->tx_out_armed[o] = false;
  }
  g->tx_pushes++;
  G_LIB(tx_sync_modulus(g, q));
  const size_t cells = (size_t)n_slots * g->S;
  G_HIP(g, g->tx_bits[o].need(cells * 148, q));
  G_HIP(g, g->tx_gain[o].need(cells, q)); G_HIP(g, g->tx_fq[o].need(cells, q));
  // an add call's ingest still pending, and its datagrams near this push's start (all of them then lie inside the packed entries'
  // window round fn): ingest and walk are ONE launch; else the ingest goes first on its own
  const int32_t dref = g->tx_pend ? trxq_fn_delta(g->tx_pend_ref, fn) : 0;
  if (g->tx_pend && dref >= -TRXQ_PK_WIN / 2 && dref < TRXQ_PK_WIN / 2) {
    const int k = g->tx_pend_k;
    g->tx_pend = false;
    G_LIB(tx_wait_arrival(g, k));
    G_HIP(g, trx_launch_group_tx_both(q, g->tx, g->tx_pend_n, g->tx_dgram[k].p, g->tx_alf[k].p, g->tx_alk[k].p, g->tx_atot[k].p, g->gain_tab,
                                      g->tx_pend_far, fn, tn, n_slots, g->tx_bits[o].p, g->tx_gain[o].p, g->tx_fq[o].p));
    G_LIB(tx_q_mark(g, g->tx_read_ev[k]));
    g->tx_read_armed[k] = true;
  } else {
    G_LIB(tx_flush_pending(g));
    G_HIP(g, trx_launch_group_tx_push(q, g->tx, fn, tn, n_slots, g->tx_bits[o].p, g->tx_gain[o].p, g->tx_fq[o].p));
    G_LIB(tx_q_mark(g, g->tx_q_ev));
  }
  G_LIB(tx_join(g, st));                                    // the caller reads the output on the context's stream
  if (d_bits) *d_bits = g->tx_bits[o].p;
  if (d_gain) *d_gain = g->tx_gain[o].p;
  if (d_from_queue) *d_from_queue = g->tx_fq[o].p;
  return TRXSIG_OK;
}

int trxsig_trxgroup_push_txbe(trxsig_trxgroup *g, trxsig_txbe *be, int fn, int tn, int n_slots) {
  if (!g) return TRXSIG_EINVAL;
  if (!be) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_push_txbe: no back end", hipSuccess);
  // everything the back end could refuse is settled BEFORE the queue is popped: a refused call leaves queue, filler table and the
  // caller's deadline clock where they were (the bursts of those slots are not consumed)
  if (trx_txbe_context(be) != g->c)
    return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_push_txbe: the back end lives on another context (another stream: its reads would race the gather)", hipSuccess);
  if (trxsig_txbe_streams(be) != g->S)
    return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_push_txbe: the back end's stream count is not the group's ARFCN count", hipSuccess);
  if (n_slots <= 0 || tn < 0 || tn > 7) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_push_txbe: bad argument", hipSuccess);
  std::vector<int32_t> guard((size_t)n_slots);
  for (int t = 0; t < n_slots; t++) guard[(size_t)t] = 8 + ((((tn + t) & 7) % 4) == 0);   // modulateBurst(..., 8 + (TN % 4 == 0), ...) (:105)
  G_LIB(trxsig_txbe_can_push(be, guard.data(), n_slots));
  const uint8_t *bits = nullptr;
  const float *gain = nullptr;
  G_LIB(trxsig_trxgroup_push(g, fn, tn, n_slots, &bits, &gain, nullptr));
  return trxsig_txbe_push_bursts(be, bits, guard.data(), gain, n_slots);
}

int trxsig_trxgroup_tx_queue_size(trxsig_trxgroup *g, int arfcn, int *dropped) {
  if (!g) return TRXSIG_EINVAL;
  if (arfcn < 0 || arfcn >= g->S) return trx_ctx_fail(g->c, TRXSIG_EINVAL, "trxsig_trxgroup_tx_queue_size: bad argument", hipSuccess);
  if (!g->tx_ready) { if (dropped) *dropped = 0; return 0; }
  Guard gd(trxsig_device(g->c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(g->c);
  int32_t n = 0; uint32_t stt = 0;
  G_LIB(tx_flush_pending(g));
  G_LIB(tx_join(g, st));
  G_HIP(g, hipMemcpyAsync(&n, g->tx.q_n + arfcn, 4, hipMemcpyDeviceToHost, st));
  G_HIP(g, hipMemcpyAsync(&stt, g->tx.status + arfcn, 4, hipMemcpyDeviceToHost, st));
  G_HIP(g, hipStreamSynchronize(st));
  if (dropped) *dropped = (int)(stt & 1u);
  return n;
}

}  // extern "C"
