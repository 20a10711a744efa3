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
    a.off[row] = a.rx_nb > 0 ? s * a.rx_nb + t : (int32_t)(a.base + (long long)t * a.slot_stride + (long long)s * a.arfcn_stride);
    a.len[row] = a.fixed_len > 0 ? a.fixed_len : (156 + ((tn & 3) == 0)) * a.sps;   // radioInterface.cpp:370-378
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
// at 0, +10 exp(-frames) per false detection), and on the TSC leg the per-timeslot channel cache decides whether this
// burst estimates (first burst of a slot, after 50 frames, after a miss) and which taps equalise it.
// A step touches only this lane's state; the inputs of D steps are loaded ahead of the D steps before them.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kReplayDepth = 16;                            // two frames: a group starts on the call's first timeslot number
constexpr int kExpLds = 64;                                 // exp(-k), k = 0 .. 63, sits in LDS (false detections come in runs)

template <bool EQ>                                          // EQ: the equalising TSC leg (its channel cache is part of the state)
__global__ __launch_bounds__(64) void k_group_replay(TrxGroupReplay a, const float4 *__restrict__ packed) {
  __shared__ int rot_i[2][8][64];
  __shared__ double exp_s[kExpLds];
  const int lane = threadIdx.x;
  exp_s[lane] = a.exp_tab[TRXG_EXP_LO + lane];
  const int s = blockIdx.x * 64 + lane;
  const bool mine = s < a.S;
  const int sc = mine ? s : a.S - 1;                        // (a spare lane shadows the last ARFCN and stores nothing)
  TrxGroupArfcn st = a.state[sc];
  // per-timeslot state in registers, ROTATED so that entry r belongs to timeslot (tn0 + r) & 7: slot t0 + i of a group
  // (t0 a multiple of 16) then always uses entry i & 7 -- a compile-time index
  int est[8], src[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { rot_i[0][k][lane] = st.est_fn[k]; rot_i[1][k][lane] = st.tap_src[k]; }
#pragma unroll
  for (int r = 0; r < 8; r++) { est[r] = rot_i[0][(a.tn0 + r) & 7][lane]; src[r] = rot_i[1][(a.tn0 + r) & 7][lane]; }
  __syncthreads();                                          // exp_s complete (the only barrier; every lane reaches it)
  double thr = st.thr;
  int prev_false = st.prev_false_fn;
  const int S8 = a.S * 8;
  int fnA = a.fn0;                                          // frame number of the group's first slot

  float4 cur[kReplayDepth];
  int crow[kReplayDepth];
  auto fetch = [&](int t0, float4 (&v)[kReplayDepth], int (&r)[kReplayDepth]) {
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      const int t = t0 + i;
      const bool in = t < a.n_slots;
      const size_t g = (size_t)(in ? t : 0) * a.S + sc;
      r[i] = in ? a.rowmap[g] : -1;
      v[i] = in ? packed[g] : make_float4(0, 0, 0, 0);
    }
  };
  fetch(0, cur, crow);
  for (int t0 = 0; t0 < a.n_slots; t0 += kReplayDepth) {
    float4 nxt[kReplayDepth];
    int nrow[kReplayDepth];
    fetch(t0 + kReplayDepth, nxt, nrow);
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      const int r = i & 7;
      int fn = fnA + ((a.tn0 + i) >> 3);
      fn -= fn >= kHyperframe ? kHyperframe : 0;
      const int row = crow[i];
      const int code = __float_as_int(cur[i].x);
      // One predicated pass instead of the reference's nest of branches (a wave's lanes would walk every arm in turn):
      //   pass  energyDetect's decision (:298; sigProcLib.cpp:929-931, the double threshold through a float parameter)
      //   succ / fail  the correlator's answer once the energy gate is open; qdec  50 quiet frames (:300-304)
      const bool act = (code & RP_ACT) != 0;                 // OFF / IDLE slots never reach the state (:288-291)
      const float thrF = (float)thr;
      const bool pass = act && (cur[i].y > thrF * thrF);
      const bool det = (code & RP_DET) != 0;
      const bool is_tsc = (code & RP_TSC) != 0;
      const int d = fn_delta(fn, prev_false);                // rxBurst->time() - prevFalseDetectionTime, in frames
      const bool succ = pass && det, fail = pass && !det, qdec = act && !pass && ((double)d > 50);
      bool evt = false;
      int tix = 0;
      if (EQ) {                                              // the per-timeslot channel cache (:313-325, 341-349, 357, 370)
        const bool stale = pass && is_tsc && ((double)fn_delta(fn, est[r]) > 50 || src[r] < 0);
        int sr = stale ? -1 : src[r];
        evt = succ && stale;                                 // this burst estimates the channel
        sr = evt ? S8 + row : sr;
        est[r] = evt ? fn : est[r];
        tix = sr;
        sr = ((fail && is_tsc) || (succ && !is_tsc)) ? -1 : sr;   // a missed normal burst / a detected access burst drop it
        src[r] = sr;
      }
      double e = 0.0;                                        // exp(-framesElapsed) (:355, 374)
      if (__any(fail)) {
        const bool near = (unsigned)d < (unsigned)kExpLds;
        e = exp_s[near ? d : 0];
        if (__any(fail && !near)) {
          const int k = d < -TRXG_EXP_LO ? -TRXG_EXP_LO : (d > TRXG_EXP_HI ? TRXG_EXP_HI : d);
          const double eg = a.exp_tab[k + TRXG_EXP_LO];
          e = near ? e : eg;
        }
      }
      double t1 = thr - 1.0;                                 // mEnergyThreshold -= 1.0F; floor 0 (:338-339, 368-369)
      t1 = t1 < 0.0 ? 0.0 : t1;
      thr = succ ? t1 : (fail ? thr + 10.0 * e : (qdec ? thr - 10.0 : thr));   // 10.0F*exp(...): float * double
      prev_false = (fail || qdec) ? fn : prev_false;
      if (mine && row >= 0) {
        const unsigned ur = (unsigned)row;
        a.gate[ur] = succ ? (uint8_t)TRXSIG_F_DETECT : (uint8_t)0;
        a.thr_after[ur] = thr;
        if (EQ) {
          a.ev[ur] = evt ? 1 : 0;
          a.tap_ix[ur] = (succ && is_tsc) ? tix : 0;
        }
      }
      if (EQ && __any(evt)) {
        const float n2 = cur[i].w * cur[i].w + cur[i].z * cur[i].z;   // Complex::norm2 (Complex.h:119)
        const float snr = (float)((double)n2 / (thr * thr + 1.0));   // SNRestimate (:340), after the -= 1
        if (mine && evt) a.snr[(unsigned)row] = snr;
      }
    }
    fnA += kReplayDepth / 8;
    fnA -= fnA >= kHyperframe ? kHyperframe : 0;
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) { cur[i] = nxt[i]; crow[i] = nrow[i]; }
  }
  if (!mine) return;
  st.thr = thr;
  st.prev_false_fn = prev_false;
#pragma unroll
  for (int r = 0; r < 8; r++) { rot_i[0][(a.tn0 + r) & 7][lane] = est[r]; rot_i[1][(a.tn0 + r) & 7][lane] = src[r]; }
#pragma unroll
  for (int k = 0; k < 8; k++) { st.est_fn[k] = rot_i[0][k][lane]; st.tap_src[k] = rot_i[1][k][lane]; }
  a.state[s] = st;
}

// The demodulating leg's replay (no channel cache): the same machine with the serial part stripped to what the recurrence
// needs.  A slot's common cases -- every lane's burst idle, accepted, or below the energy threshold -- cost a float
// conversion, a product, two comparisons and guarded updates; the frame difference to prevFalseDetectionTime is carried
// along (it grows by one per frame and restarts at a false detection / quiet decrement) instead of being formed per slot,
// and exp(-frames) is looked up only on a slot where some lane's correlator missed behind an open energy gate.  Verdicts
// and thresholds leave in (slot, ARFCN) order (one coalesced store each, no predication: the scratch is padded to whole
// groups and whole waves); k_group_scatter puts them where the rows are.  (0.12 - 0.29 us per slot before: 137 instructions.)
__global__ __launch_bounds__(64) void k_group_replay_lean(TrxGroupReplay a, const float4 *__restrict__ packed, double *__restrict__ thr_g,
                                                          uint8_t *__restrict__ succ_g, int Spad) {
  __shared__ double exp_s[kExpLds];
  const int lane = threadIdx.x;
  exp_s[lane] = a.exp_tab[TRXG_EXP_LO + lane];
  const int col = blockIdx.x * 64 + lane;
  const bool mine = col < a.S;
  const int sc = mine ? col : a.S - 1;                      // (a spare lane shadows the last ARFCN; its state is not stored)
  double thr = a.state[sc].thr;
  int prev_false = a.state[sc].prev_false_fn;
  __syncthreads();
  int fnA = a.fn0;
  // dcur = rxBurst->time() - prevFalseDetectionTime in frames (fn_delta), kept current: +1 (with the hyperframe wrap) at
  // every slot that starts a frame -- the call's very first slot included when it is a timeslot 0, hence the -1 here
  constexpr int half = kHyperframe / 2;
  int dcur = fn_delta(a.fn0, prev_false) - ((a.tn0 & 7) == 0 ? 1 : 0);
  float2 cur[kReplayDepth];
  auto fetch = [&](int t0, float2 (&v)[kReplayDepth]) {
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      const int t = t0 + i;
      const bool in = t < a.n_slots;                        // (uniform)
      const float4 *p = packed + (size_t)(in ? t : 0) * a.S + sc;
      const float2 w = *reinterpret_cast<const float2 *>(p);
      v[i] = in ? w : make_float2(0.0f, 0.0f);
    }
  };
  fetch(0, cur);
  for (int t0 = 0; t0 < a.n_slots; t0 += kReplayDepth) {
    float2 nxt[kReplayDepth];
    fetch(t0 + kReplayDepth, nxt);
    double *__restrict__ thr_row = thr_g + (size_t)t0 * Spad;
    uint8_t *__restrict__ succ_row = succ_g + (size_t)t0 * Spad;
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      int fn = fnA + ((a.tn0 + i) >> 3);                     // (uniform)
      fn -= fn >= kHyperframe ? kHyperframe : 0;
      if (((a.tn0 + i) & 7) == 0) {                          // a new frame (uniform branch)
        dcur += 1;
        dcur -= dcur >= half ? kHyperframe : 0;
      }
      const int code = __float_as_int(cur[i].x);
      const bool act = (code & RP_ACT) != 0;
      const bool det = (code & RP_DET) != 0;
      const float thrF = (float)thr;
      const bool pass = act && (cur[i].y > thrF * thrF);
      const bool succ = pass && det, fail = pass && !det;
      const bool qdec = act && !pass && dcur > 50;           // ((double)d > 50 of an integer d)
      double t1 = thr - 1.0;
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
        tn_ = fail ? thr + 10.0 * e : tn_;
      }
      thr = tn_;
      const bool mark = fail || qdec;                        // prevFalseDetectionTime = this burst's time
      prev_false = mark ? fn : prev_false;
      dcur = mark ? 0 : dcur;
      thr_row[(size_t)i * Spad + col] = thr;
      succ_row[(size_t)i * Spad + col] = succ ? (uint8_t)TRXSIG_F_DETECT : (uint8_t)0;
    }
    fnA += kReplayDepth / 8;
    fnA -= fnA >= kHyperframe ? kHyperframe : 0;
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) cur[i] = nxt[i];
  }
  if (!mine) return;
  a.state[col].thr = thr;
  a.state[col].prev_false_fn = prev_false;
}

__global__ __launch_bounds__(256) void k_group_scatter(int n_slots, int S, int Spad, const int32_t *__restrict__ rowmap,
                                                       const double *__restrict__ thr_g, const uint8_t *__restrict__ succ_g,
                                                       uint8_t *__restrict__ gate, double *__restrict__ thr_after) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)n_slots * S) return;
  const int row = rowmap[g];
  if (row < 0) return;
  const int t = (int)(g / S), s = (int)(g - (long long)t * S);
  gate[row] = succ_g[(size_t)t * Spad + s];
  thr_after[row] = thr_g[(size_t)t * Spad + s];
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

size_t trx_group_replay_scratch(int S, int n_slots) {         // entries of thr_g / succ_g
  return (size_t)((n_slots + kReplayDepth - 1) / kReplayDepth * kReplayDepth) * (size_t)((S + 63) / 64 * 64);
}

hipError_t trx_launch_group_replay(hipStream_t st, const TrxGroupReplay &a, float4 *packed, double *thr_g, uint8_t *succ_g, TrxProfiler *prof) {
  const long long n = (long long)a.n_slots * a.S;
  if (n <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_GROUP, st);
  k_group_pack<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, a.n_tsc_rows, a.rowmap, a.flags, a.avgpwr, a.amp, packed);
  if (a.equalize) k_group_replay<true><<<dim3((a.S + 63) / 64), dim3(64), 0, st>>>(a, packed);
  else if (!thr_g) k_group_replay<false><<<dim3((a.S + 63) / 64), dim3(64), 0, st>>>(a, packed);
  else {
    const int Spad = (a.S + 63) / 64 * 64;
    k_group_replay_lean<<<dim3(Spad / 64), dim3(64), 0, st>>>(a, packed, thr_g, succ_g, Spad);
    k_group_scatter<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(a.n_slots, a.S, Spad, a.rowmap, thr_g, succ_g, a.gate, a.thr_after);
  }
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
