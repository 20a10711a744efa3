// trxsig_group.hip -- the kernels of the Transceiver group (include/trxsig_trxgroup.h, csrc/trxsig_trxgroup.cpp):
//   k_group_expand : (slot, ARFCN) -> row of its correlator class, and the row's burst offset / length;
//   k_group_pack   : the stateless detectors' answers gathered into (slot, ARFCN) order for the replay;
//   k_group_replay : class Transceiver's receive state machine (Transceiver/Transceiver.cpp:288-376) for S ARFCNs, a
//                    lane per ARFCN walking its bursts in time order;
//   k_group_toa_eq, k_group_commit : what the equaliser needs from the replay, and the per-slot DFE cache update.
// The arithmetic of the state machine is the reference's host arithmetic, type by type (double where it uses double);
// built with -ffp-contract=off like the rest of the library.  exp() comes from a table the host fills with ITS libm
// (trxsig_group.h), so the threshold follows the reference to the last bit.
#include "trxsig_dev.h"
#include "trxsig_group.h"

namespace {

constexpr int kHyperframe = 2048 * 26 * 51;                 // GSM/GSMCommon.h:306

__device__ __forceinline__ int fn_delta(int v1, int v2) {   // GSM::FNDelta (GSM/GSMCommon.cpp:161-168)
  const int half = kHyperframe / 2;
  int d = v1 - v2;
  if (d >= half) d -= kHyperframe;
  else if (d < -half) d += kHyperframe;
  return d;
}

__global__ __launch_bounds__(256) void k_group_expand(TrxGroupExpand a) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)a.n_slots * a.S) return;
  const int t = (int)(g / a.S), s = (int)(g - (long long)t * a.S);
  const int tn = (a.tn0 + t) & 7;
  const int col = a.gid[tn * a.S + s];
  const int base = a.seg_base[(size_t)t * a.G + col];
  int row = -1;
  if (base >= 0) {
    row = base + a.pos[tn * a.S + s];
    if (a.src_off) {                                        // listed bursts (a front end's pop)
      a.off[row] = a.src_off[(size_t)s * a.src_nb + t];
      a.len[row] = a.src_len[(size_t)s * a.src_nb + t];
    } else {
      a.off[row] = a.rx_nb > 0 ? s * a.rx_nb + t : (int32_t)(a.base + (long long)t * a.slot_stride + (long long)s * a.arfcn_stride);
      a.len[row] = a.fixed_len > 0 ? a.fixed_len : (156 + ((tn & 3) == 0)) * a.sps;   // radioInterface.cpp:370-378
    }
  }
  a.rowmap[g] = row;
}

// packed[g] = {code, avgPwr, amp.re, amp.im} of (slot, ARFCN) g's row; code: bit 0 = the burst reaches the state machine
// (its slot expects a correlation and its length was accepted), bit 1 = the correlator detected, bit 2 = normal burst
enum { RP_ACT = 1, RP_DET = 2, RP_TSC = 4 };
__global__ __launch_bounds__(256) void k_group_pack(long long n, int n_tsc_rows, const int32_t *__restrict__ rowmap,
                                                    const uint8_t *__restrict__ flags, const float *__restrict__ avgpwr,
                                                    const cx *__restrict__ amp, float4 *__restrict__ packed) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const int row = rowmap[g];
  float4 v = make_float4(0, 0, 0, 0);
  if (row >= 0) {
    const cx a = amp[row];
    const int fl = flags[row];
    const int code = ((fl & TRXSIG_F_BADLEN) ? 0 : RP_ACT) | ((fl & TRXSIG_F_DETECT) ? RP_DET : 0) | (row < n_tsc_rows ? RP_TSC : 0);
    v = make_float4(__int_as_float(code), avgpwr[row], a.r, a.i);
  }
  packed[g] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_group_replay: pullRadioVector's bookkeeping between its sigProcLib calls (Transceiver.cpp:288-376), for one ARFCN per
// lane.  The detectors ran statelessly (energy gate off) on every burst whose slot expects a correlation; here the
// burst's avgPwr meets the adaptive threshold (energyDetect's decision, sigProcLib.cpp:929-931, as the reference passes
// its double threshold through a float parameter), the threshold moves (-10 after 50 quiet frames, -1 per success floored
// at 0, +10 exp(-frames) per false detection), and on the equalising leg (EQ) the per-timeslot channel cache decides
// whether this burst estimates (first burst of a slot, after 50 frames, after a miss) and which taps equalise it.
// One predicated pass instead of the reference's nest of branches (a wave's lanes would walk every arm in turn):
//   pass  energyDetect's decision (:298);  succ / fail  the correlator's answer once the energy gate is open;
//   qdec  50 quiet frames (:300-304).
// The serial part is stripped to what the recurrence needs (a step of the first version was 137 instructions, 0.12 - 0.29 us
// per slot for 128 ARFCNs): a slot's common cases -- every lane's burst idle, accepted, or below the energy threshold --
// cost a float conversion, a product, two comparisons and guarded updates; the frame difference to prevFalseDetectionTime is
// CARRIED (it grows by one per frame, with the hyperframe wrap, and restarts at a false detection / quiet decrement) instead of
// being formed per slot, and exp(-frames) is looked up only on a slot where some lane's correlator missed behind an open
// gate.  Verdicts, thresholds (and the channel cache's events) leave in (slot, ARFCN) order -- one coalesced store each, no
// predication: the scratch is padded to whole groups and whole waves -- and k_group_scatter puts them where the rows are
// (and forms SNRestimate, :340, a double division, for the bursts that estimate).  Per-timeslot cache state sits in registers
// ROTATED so that entry r belongs to timeslot (tn0 + r) & 7: slot i of a 16-slot group always uses entry i & 7.
// A step touches only this lane's state; the inputs of a group are loaded ahead of the 16 steps before them.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kReplayDepth = 16;                            // two frames: a group starts on the call's first timeslot number
constexpr int kExpLds = 64;                                 // exp(-k), k = 0 .. 63, sits in LDS (false detections come in runs)
enum { RV_SUCC = TRXSIG_F_DETECT, RV_EVT = 0x40 };          // the verdict byte: the gate's value, + "this burst estimates"

template <bool EQ>
__global__ __launch_bounds__(64) void k_group_replay(TrxGroupReplay a, const float4 *__restrict__ packed, double *__restrict__ thr_g,
                                                     uint8_t *__restrict__ verdict_g, int32_t *__restrict__ tix_g, int Spad) {
  __shared__ int rot_i[2][8][64];
  __shared__ double exp_s[kExpLds];
  const int lane = threadIdx.x;
  exp_s[lane] = a.exp_tab[TRXG_EXP_LO + lane];
  const int col = blockIdx.x * 64 + lane;
  const bool mine = col < a.S;
  const int sc = mine ? col : a.S - 1;                      // (a spare lane shadows the last ARFCN; its state is not stored)
  TrxGroupArfcn st = a.state[sc];
  int est[8], src[8];
  if (EQ) {
#pragma unroll
    for (int k = 0; k < 8; k++) { rot_i[0][k][lane] = st.est_fn[k]; rot_i[1][k][lane] = st.tap_src[k]; }
#pragma unroll
    for (int r = 0; r < 8; r++) { est[r] = rot_i[0][(a.tn0 + r) & 7][lane]; src[r] = rot_i[1][(a.tn0 + r) & 7][lane]; }
  }
  __syncthreads();                                          // exp_s complete (the only barrier; every lane reaches it)
  double thr = st.thr;
  int prev_false = st.prev_false_fn;
  const int S8 = a.S * 8;
  int fnA = a.fn0;
  // dcur = rxBurst->time() - prevFalseDetectionTime in frames (fn_delta), kept current: +1 (with the hyperframe wrap) at
  // every slot that starts a frame -- the call's very first slot included when it is a timeslot 0, hence the -1 here
  constexpr int half = kHyperframe / 2;
  int dcur = fn_delta(a.fn0, prev_false) - ((a.tn0 & 7) == 0 ? 1 : 0);
  float2 cur[kReplayDepth];
  int crow[kReplayDepth];
  auto fetch = [&](int t0, float2 (&v)[kReplayDepth], int (&r)[kReplayDepth]) {
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      const int t = t0 + i;
      const bool in = t < a.n_slots;                        // (uniform)
      const size_t g = (size_t)(in ? t : 0) * a.S + sc;
      const float2 w = *reinterpret_cast<const float2 *>(packed + g);   // {code, avgPwr}; the amplitude is k_group_scatter's business
      v[i] = in ? w : make_float2(0.0f, 0.0f);
      if (EQ) r[i] = in ? a.rowmap[g] : -1;                 // (the cache remembers WHICH row's taps a slot uses)
    }
  };
  fetch(0, cur, crow);
  for (int t0 = 0; t0 < a.n_slots; t0 += kReplayDepth) {
    float2 nxt[kReplayDepth];
    int nrow[kReplayDepth];
    fetch(t0 + kReplayDepth, nxt, nrow);
    double *__restrict__ thr_row = thr_g + (size_t)t0 * Spad;
    uint8_t *__restrict__ v_row = verdict_g + (size_t)t0 * Spad;
    int32_t *__restrict__ tix_row = EQ ? tix_g + (size_t)t0 * Spad : nullptr;
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      const int r = i & 7;
      int fn = fnA + ((a.tn0 + i) >> 3);                     // (uniform)
      fn -= fn >= kHyperframe ? kHyperframe : 0;
      if (((a.tn0 + i) & 7) == 0) {                          // a new frame (uniform branch)
        dcur += 1;
        dcur -= dcur >= half ? kHyperframe : 0;
      }
      const int code = __float_as_int(cur[i].x);
      const bool act = (code & RP_ACT) != 0;                 // OFF / IDLE slots never reach the state (:288-291)
      const bool det = (code & RP_DET) != 0;
      const float thrF = (float)thr;
      const bool pass = act && (cur[i].y > thrF * thrF);
      const bool succ = pass && det, fail = pass && !det;
      const bool qdec = act && !pass && dcur > 50;           // ((double)d > 50 of an integer d)
      bool evt = false;
      int tix = 0;
      if (EQ) {                                              // the per-timeslot channel cache (:313-325, 341-349, 357, 370)
        const bool is_tsc = (code & RP_TSC) != 0;
        const bool stale = pass && is_tsc && ((double)fn_delta(fn, est[r]) > 50 || src[r] < 0);
        int sr = stale ? -1 : src[r];
        evt = succ && stale;                                 // this burst estimates the channel
        sr = evt ? S8 + crow[i] : sr;
        est[r] = evt ? fn : est[r];
        tix = (succ && is_tsc) ? sr : 0;
        sr = ((fail && is_tsc) || (succ && !is_tsc)) ? -1 : sr;   // a missed normal burst / a detected access burst drop it
        src[r] = sr;
      }
      double t1 = thr - 1.0;                                 // mEnergyThreshold -= 1.0F; floor 0 (:338-339, 368-369)
      t1 = t1 < 0.0 ? 0.0 : t1;
      double tn_ = succ ? t1 : (qdec ? thr - 10.0 : thr);
      if (__any(fail)) {                                     // exp(-framesElapsed) (:355, 374)
        const int d = dcur;
        const bool near = (unsigned)d < (unsigned)kExpLds;
        double e = exp_s[near ? d : 0];
        if (__any(fail && !near)) {
          const int k = d < -TRXG_EXP_LO ? -TRXG_EXP_LO : (d > TRXG_EXP_HI ? TRXG_EXP_HI : d);
          const double eg = a.exp_tab[k + TRXG_EXP_LO];
          e = near ? e : eg;
        }
        tn_ = fail ? thr + 10.0 * e : tn_;                   // 10.0F*exp(...): float * double
      }
      thr = tn_;
      const bool mark = fail || qdec;                        // prevFalseDetectionTime = this burst's time
      prev_false = mark ? fn : prev_false;
      dcur = mark ? 0 : dcur;
      thr_row[(size_t)i * Spad + col] = thr;
      v_row[(size_t)i * Spad + col] = (uint8_t)((succ ? RV_SUCC : 0) | (evt ? RV_EVT : 0));
      if (EQ) tix_row[(size_t)i * Spad + col] = tix;
    }
    fnA += kReplayDepth / 8;
    fnA -= fnA >= kHyperframe ? kHyperframe : 0;
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) { cur[i] = nxt[i]; crow[i] = nrow[i]; }
  }
  if (!mine) return;
  st.thr = thr;
  st.prev_false_fn = prev_false;
  if (EQ) {
#pragma unroll
    for (int r = 0; r < 8; r++) { rot_i[0][(a.tn0 + r) & 7][lane] = est[r]; rot_i[1][(a.tn0 + r) & 7][lane] = src[r]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { st.est_fn[k] = rot_i[0][k][lane]; st.tap_src[k] = rot_i[1][k][lane]; }
  }
  a.state[col] = st;
}

// (slot, ARFCN) order -> rows: gate, the threshold after the burst and, on the equalising leg, the estimation events, the tap
// index and SNRestimate = |amp|^2 / (thr^2 + 1) in double with the threshold AFTER its decrement (:340)
__global__ __launch_bounds__(256) void k_group_scatter(TrxGroupReplay a, int Spad, const double *__restrict__ thr_g,
                                                       const uint8_t *__restrict__ verdict_g, const int32_t *__restrict__ tix_g) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)a.n_slots * a.S) return;
  const int row = a.rowmap[g];
  if (row < 0) return;
  const int t = (int)(g / a.S), s = (int)(g - (long long)t * a.S);
  const size_t q = (size_t)t * Spad + s;
  const int v = verdict_g[q];
  const double thr = thr_g[q];
  a.gate[row] = (uint8_t)(v & RV_SUCC);
  a.thr_after[row] = thr;
  if (a.equalize) {
    a.ev[row] = (v & RV_EVT) ? 1 : 0;
    a.tap_ix[row] = tix_g[q];
    if (v & RV_EVT) {
      const trx_c32 am = a.amp[row];
      const float n2 = am.i * am.i + am.r * am.r;           // Complex::norm2 (Complex.h:119)
      a.snr[row] = (float)((double)n2 / (thr * thr + 1.0));
    }
  }
}

__global__ __launch_bounds__(256) void k_group_toa_eq(int n, const uint8_t *__restrict__ gate, const float *__restrict__ toa,
                                                      const int32_t *__restrict__ tap_ix, const float *__restrict__ chan_off,
                                                      float *__restrict__ toa_eq) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  toa_eq[row] = (gate[row] & TRXSIG_F_DETECT) ? toa[row] - chan_off[tap_ix[row]] : 0.0f;   // TOA - chanRespOffset[timeslot] (:393)
}

__global__ __launch_bounds__(256) void k_group_commit(int S, TrxGroupArfcn *__restrict__ state, cx *__restrict__ w_tab,
                                                      cx *__restrict__ b_tab, float *__restrict__ chan_off) {
  const int i = blockIdx.x * 256 + threadIdx.x;             // (ARFCN, timeslot)
  if (i >= S * 8) return;
  const int s = i >> 3, tn = i & 7;
  const int src = state[s].tap_src[tn];
  if (src < S * 8) return;                                  // empty, or already the slot's own cache entry
#pragma unroll
  for (int j = 0; j < 7; j++) w_tab[(size_t)i * 7 + j] = w_tab[(size_t)src * 7 + j];
#pragma unroll
  for (int j = 0; j < 5; j++) b_tab[(size_t)i * 5 + j] = b_tab[(size_t)src * 5 + j];
  chan_off[i] = chan_off[src];
  state[s].tap_src[tn] = i;
}

}  // namespace

hipError_t trx_launch_group_expand(hipStream_t st, const TrxGroupExpand &a) {
  const long long n = (long long)a.n_slots * a.S;
  if (n <= 0) return hipSuccess;
  k_group_expand<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(a);
  return hipGetLastError();
}

size_t trx_group_replay_scratch(int S, int n_slots) {         // entries of thr_g / verdict_g / tix_g
  return (size_t)((n_slots + kReplayDepth - 1) / kReplayDepth * kReplayDepth) * (size_t)((S + 63) / 64 * 64);
}

hipError_t trx_launch_group_replay(hipStream_t st, const TrxGroupReplay &a, float4 *packed, double *thr_g, uint8_t *verdict_g, int32_t *tix_g,
                                   TrxProfiler *prof) {
  const long long n = (long long)a.n_slots * a.S;
  if (n <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_GROUP, st);
  k_group_pack<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, a.n_tsc_rows, a.rowmap, a.flags, a.avgpwr, a.amp, packed);
  const int Spad = (a.S + 63) / 64 * 64;
  if (a.equalize) k_group_replay<true><<<dim3(Spad / 64), dim3(64), 0, st>>>(a, packed, thr_g, verdict_g, tix_g, Spad);
  else k_group_replay<false><<<dim3(Spad / 64), dim3(64), 0, st>>>(a, packed, thr_g, verdict_g, tix_g, Spad);
  k_group_scatter<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(a, Spad, thr_g, verdict_g, tix_g);
  if (prof) prof->end(TRXSIG_K_GROUP, st);
  return hipGetLastError();
}

hipError_t trx_launch_group_toa_eq(hipStream_t st, int n_rows, const uint8_t *gate, const float *toa, const int32_t *tap_ix,
                                   const float *chan_off_tab, float *toa_eq) {
  if (n_rows <= 0) return hipSuccess;
  k_group_toa_eq<<<dim3((n_rows + 255) / 256), dim3(256), 0, st>>>(n_rows, gate, toa, tap_ix, chan_off_tab, toa_eq);
  return hipGetLastError();
}

hipError_t trx_launch_group_commit(hipStream_t st, int S, TrxGroupArfcn *state, trx_c32 *w_tab, trx_c32 *b_tab, float *chan_off_tab) {
  k_group_commit<<<dim3((S * 8 + 255) / 256), dim3(256), 0, st>>>(S, state, w_tab, b_tab, chan_off_tab);
  return hipGetLastError();
}
