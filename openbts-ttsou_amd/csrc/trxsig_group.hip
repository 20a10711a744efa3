// trxsig_group.hip -- the kernels of the Transceiver group (include/trxsig_trxgroup.h, csrc/trxsig_trxgroup.cpp):
//   k_group_expand : (slot, ARFCN) -> row of its correlator class, and the row's burst offset / length;
//   k_group_pack   : the stateless detectors' answers gathered into (slot, ARFCN) order for the replay;
//   k_group_replay : class Transceiver's receive state machine (Transceiver/Transceiver.cpp:288-376) for S ARFCNs, a
//                    lane per ARFCN walking its bursts in time order (the adaptive threshold: the serial part);
//   k_group_cache  : the per-timeslot channel / DFE cache of the equalising leg, a lane per (ARFCN, timeslot);
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
// (and forms SNRestimate, :340, a double division, for the bursts that estimate).  The equalising leg's per-timeslot channel
// cache is NOT in this chain (round 4): it needs the verdicts only and runs after it, a lane per (ARFCN, timeslot): k_group_cache.
// A step touches only this lane's state; the inputs of a group are loaded ahead of the 16 steps before them.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kReplayDepth = 16;                            // two frames: a group starts on the call's first timeslot number
constexpr int kExpLds = 64;                                 // exp(-k), k = 0 .. 63, sits in LDS (false detections come in runs)
enum { RV_SUCC = TRXSIG_F_DETECT, RV_CONS = 0x08, RV_MARK = 0x10, RV_PASS = 0x20, RV_EVT = 0x40, RV_TSC = 0x80 };   // the verdict byte: the gate's value, + "energy gate open", + (for k_group_cache) "a normal burst's slot"

// One burst's step of the machine: (thr, prev_false, dcur) -> the same after the burst; returns the verdict byte.
// exp_s: exp(-k), k = 0 .. kExpLds-1 in LDS; exp_tab: the whole table in memory (trxsig_group.h).
__device__ __forceinline__ int replay_step(double &thr, int &prev_false, int &dcur, int code, float avg, int fn, const double *exp_s,
                                           const double *__restrict__ exp_tab) {
  const bool act = (code & RP_ACT) != 0;                   // OFF / IDLE slots never reach the state (:288-291)
  const bool det = (code & RP_DET) != 0;
  const float thrF = (float)thr;
  const bool pass = act && (avg > thrF * thrF);
  const bool succ = pass && det, fail = pass && !det;
  const bool qdec = act && !pass && dcur > 50;             // ((double)d > 50 of an integer d)
  double t1 = thr - 1.0;                                   // mEnergyThreshold -= 1.0F; floor 0 (:338-339, 368-369)
  t1 = t1 < 0.0 ? 0.0 : t1;
  double tn_ = succ ? t1 : (qdec ? thr - 10.0 : thr);
  if (__any(fail)) {                                       // exp(-framesElapsed) (:355, 374)
    const int d = dcur;
    const bool near = (unsigned)d < (unsigned)kExpLds;
    double e = exp_s[near ? d : 0];
    if (__any(fail && !near)) {
      const int k = d < -TRXG_EXP_LO ? -TRXG_EXP_LO : (d > TRXG_EXP_HI ? TRXG_EXP_HI : d);
      const double eg = exp_tab[k + TRXG_EXP_LO];
      e = near ? e : eg;
    }
    tn_ = fail ? thr + 10.0 * e : tn_;                     // 10.0F*exp(...): float * double
  }
  thr = tn_;
  const bool mark = fail || qdec;                          // prevFalseDetectionTime = this burst's time
  prev_false = mark ? fn : prev_false;
  dcur = mark ? 0 : dcur;
  // RV_CONS: the step looked at the frame difference to prevFalseDetectionTime (the quiet test or exp(-frames)); RV_MARK: it re-based it
  return (succ ? RV_SUCC : 0) | (pass ? RV_PASS : 0) | (((act && !pass) || fail) ? RV_CONS : 0) | (mark ? RV_MARK : 0);
}

__global__ __launch_bounds__(64) void k_group_replay(TrxGroupReplay a, const float4 *__restrict__ packed, double *__restrict__ thr_g,
                                                     uint8_t *__restrict__ verdict_g, int Spad) {
  __shared__ double exp_s[kExpLds];
  // a latency chain of a few waves that usually runs BESIDE a kernel filling every SIMD (the demodulator, on the group's side
  // stream): ask the instruction arbiter for the highest wave priority, or every step waits behind eight other waves' turns
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  exp_s[lane] = a.exp_tab[TRXG_EXP_LO + lane];
  const int col = blockIdx.x * 64 + lane;
  const bool mine = col < a.S;
  const int sc = mine ? col : a.S - 1;                      // (a spare lane shadows the last ARFCN; its state is not stored)
  __syncthreads();                                          // exp_s complete (the only barrier; every lane reaches it)
  double thr = a.state[sc].thr;
  int prev_false = a.state[sc].prev_false_fn;
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
      const size_t g = (size_t)(in ? t : 0) * a.S + sc;
      const float2 w = *reinterpret_cast<const float2 *>(packed + g);   // {code, avgPwr}; the amplitude is k_group_scatter's business
      v[i] = in ? w : make_float2(0.0f, 0.0f);
    }
  };
  fetch(0, cur);
  for (int t0 = 0; t0 < a.n_slots; t0 += kReplayDepth) {
    float2 nxt[kReplayDepth];
    fetch(t0 + kReplayDepth, nxt);
    double *__restrict__ thr_row = thr_g + (size_t)t0 * Spad;
    uint8_t *__restrict__ v_row = verdict_g + (size_t)t0 * Spad;
#pragma unroll
    for (int i = 0; i < kReplayDepth; i++) {
      int fn = fnA + ((a.tn0 + i) >> 3);                     // (uniform)
      fn -= fn >= kHyperframe ? kHyperframe : 0;
      if (((a.tn0 + i) & 7) == 0) {                          // a new frame (uniform branch)
        dcur += 1;
        dcur -= dcur >= half ? kHyperframe : 0;
      }
      const int v = replay_step(thr, prev_false, dcur, __float_as_int(cur[i].x), cur[i].y, fn, exp_s, a.exp_tab);
      thr_row[(size_t)i * Spad + col] = thr;
      v_row[(size_t)i * Spad + col] = (uint8_t)((v & (RV_SUCC | RV_PASS)) | ((__float_as_int(cur[i].x) & RP_TSC) ? RV_TSC : 0));
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

// ---------------------------------------------------------------------------------------------------------------------
// k_group_replay_seg<K>: the same machine for a LONG call, parallel in time (round 4).  The chain above is one dependent step per
// timeslot -- 1,000 slots replay in ~100 us however few ARFCNs there are.  Here an ARFCN's slots are cut into K segments, a
// lane per (ARFCN, segment); the segments are replayed SIDE BY SIDE from assumed start states (first: the call's own start
// state), then a lane per ARFCN walks the segment boundaries from the true start and VALIDATES each segment in turn:
//   a segment's recorded outputs and end state stand if it started from exactly the threshold (the double, bit for bit) the walk
//   arrives with, and from the same prevFalseDetectionTime -- or never looked at the clock before re-basing it itself (RV_CONS /
//   RV_MARK: then its outputs do not depend on the incoming prevFalseDetectionTime, and it hands that value on unchanged if it
//   never re-based it).
// The first segment that fails the test is replayed again from the state the walk arrived with (which is the TRUE state: everything
// before it is validated), later segments from the walk's best knowledge; so the validated prefix grows by at least one segment
// per round, the loop ends after at most K rounds with the serial result value for value, and after one or two when the state
// forgets its past inside a segment -- the threshold on its floor of 0, a false detection re-basing the clock -- the steady
// state of a running cell.  Workgroup = 256 / K ARFCNs x K segments, everything about the boundaries in LDS.
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_group_replay_seg(TrxGroupReplay a, const float4 *__restrict__ packed, double *__restrict__ thr_g,
                                                          uint8_t *__restrict__ verdict_g, int Spad, int Ls) {
  constexpr int A = 256 / K;
  enum { SF_CONS = 1, SF_MARK = 2 };                         // looked at the incoming clock before re-basing it / re-based it
  __shared__ double exp_s[kExpLds];
  __shared__ long long n_thr[K][A], u_thr[K][A], e_thr[K][A];   // next start / start last run from / end of the last run (as bits)
  __shared__ int n_pf[K][A], u_pf[K][A], e_pf[K][A];
  __shared__ uint8_t need[K][A], sflag[K][A];
  __shared__ int all_done[2];                                // (by round parity: a slow reader of one never meets the next round's writer)
  __builtin_amdgcn_s_setprio(3);                            // (as k_group_replay: a few latency-bound waves beside a machine-filling kernel)
  const int tid = threadIdx.x;
  if (tid < kExpLds) exp_s[tid] = a.exp_tab[TRXG_EXP_LO + tid];
  const int ai = tid % A, j = tid / A;
  const int col = blockIdx.x * A + ai;
  const bool mine = col < a.S;
  const int sc = mine ? col : a.S - 1;
  const long long s0_thr = __double_as_longlong(a.state[sc].thr);
  const int s0_pf = a.state[sc].prev_false_fn;
  n_thr[j][ai] = s0_thr; n_pf[j][ai] = s0_pf; need[j][ai] = 1;
  __syncthreads();
  const int ts = j * Ls, te = (ts + Ls < a.n_slots) ? ts + Ls : a.n_slots;   // this lane's slots [ts, te)
  constexpr int half = kHyperframe / 2;
#ifdef TRX_REPLAY_DEBUG
  long long tk0 = wall_clock64(), t_run = 0, t_walk = 0;
#endif
  for (int round = 0;; round++) {
#ifdef TRX_REPLAY_DEBUG
    const long long tr0 = wall_clock64();
#endif
    if (need[j][ai]) {
      const long long s_thr = n_thr[j][ai];
      const int s_pf = n_pf[j][ai];
      double thr = __longlong_as_double(s_thr);
      int prev_false = s_pf;
      // Ls is a multiple of 8 (the launcher's choice), so every segment starts on the call's first timeslot number: the position
      // of the frame boundary inside a group of eight steps (i0) is the same for every lane -- the step's control flow is uniform
      const int i0 = (8 - (a.tn0 & 7)) & 7;                  // step i of a group of eight has timeslot number 0 iff i == i0
      int fn = a.fn0 + ((a.tn0 + ts) >> 3);
      fn %= kHyperframe;
      const int dcur0 = fn_delta(fn, prev_false);
      // the step with timeslot number 0 moves fn and the frame difference on by one BEFORE the burst is looked at; a segment that
      // starts on such a step starts one behind
      fn -= i0 == 0 ? 1 : 0;
      fn += fn < 0 ? kHyperframe : 0;
      int dcur = dcur0 - (i0 == 0 ? 1 : 0);
      int seen = 0;                                          // RV_CONS before the first RV_MARK; RV_MARK
      // the inputs of the next eight slots are in flight while these eight are replayed; slots past the segment's (the call's)
      // end read as inactive (code 0: the state does not move) and store nothing
      float2 w[8], wn[8];
      auto fetch = [&](int t0, float2 (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int t = t0 + i;
          const bool in = t < te;
          const float2 x = *reinterpret_cast<const float2 *>(packed + (size_t)(in ? t : 0) * a.S + sc);
          v[i] = in ? x : make_float2(0.0f, 0.0f);
        }
      };
      fetch(ts, w);
      double *pt = thr_g + (size_t)ts * Spad + col;
      uint8_t *pv = verdict_g + (size_t)ts * Spad + col;
#pragma unroll 1
      for (int t0 = ts; t0 < ts + Ls; t0 += 8) {             // (uniform trip count; Ls % 8 == 0)
        fetch(t0 + 8, wn);
        const int n_valid = mine ? te - t0 : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (i == i0) {                                     // a new frame (uniform)
            fn += 1; fn -= fn >= kHyperframe ? kHyperframe : 0;
            dcur += 1; dcur -= dcur >= half ? kHyperframe : 0;
          }
          const int v = replay_step(thr, prev_false, dcur, __float_as_int(w[i].x), w[i].y, fn, exp_s, a.exp_tab);
          seen |= ((seen & RV_MARK) ? 0 : (v & RV_CONS)) | (v & RV_MARK);
          if (i < n_valid) {
            pt[(size_t)i * Spad] = thr;
            pv[(size_t)i * Spad] = (uint8_t)((v & (RV_SUCC | RV_PASS)) | ((__float_as_int(w[i].x) & RP_TSC) ? RV_TSC : 0));
          }
        }
        pt += (size_t)8 * Spad; pv += (size_t)8 * Spad;
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = wn[i];
      }
      u_thr[j][ai] = s_thr; u_pf[j][ai] = s_pf;
      e_thr[j][ai] = __double_as_longlong(thr); e_pf[j][ai] = prev_false;
      sflag[j][ai] = (uint8_t)(((seen & RV_CONS) ? SF_CONS : 0) | ((seen & RV_MARK) ? SF_MARK : 0));
    }
    const int par = round & 1;
    if (tid == 0) all_done[par] = 1;
    __syncthreads();
#ifdef TRX_REPLAY_DEBUG
    const long long tr1 = wall_clock64();
    t_run += tr1 - tr0;
#endif
    if (j == 0) {                                            // the walk: a lane per ARFCN over its K boundaries
      long long w_thr = s0_thr;
      int w_pf = s0_pf;
      bool prefix = true;                                    // everything before jj is validated: (w_thr, w_pf) is the TRUE state
      for (int jj = 0; jj < K; jj++) {
        const int fl = sflag[jj][ai];
        const bool ok = u_thr[jj][ai] == w_thr && (u_pf[jj][ai] == w_pf || !(fl & SF_CONS));
        need[jj][ai] = ok ? 0 : 1;
        if (!ok) { n_thr[jj][ai] = w_thr; n_pf[jj][ai] = w_pf; prefix = false; }
        // what the walk knows of the state behind segment jj: exact while `prefix`, else the best guess for the segments after it
        w_thr = e_thr[jj][ai];
        w_pf = (fl & SF_MARK) ? e_pf[jj][ai] : w_pf;
      }
      if (!prefix) all_done[par] = 0;
      else if (mine) { a.state[col].thr = __longlong_as_double(w_thr); a.state[col].prev_false_fn = w_pf; }
    }
    __syncthreads();
#ifdef TRX_REPLAY_DEBUG
    t_walk += wall_clock64() - tr1;
    if ((all_done[par] || round > K + 1) && tid == 0) printf("replay_seg<%d> block %d: %d rounds (n_slots %d, Ls %d) total %lld run %lld walk %lld ticks (100 MHz)\n", K, blockIdx.x, round + 1, a.n_slots, Ls, wall_clock64() - tk0, t_run, t_walk);
    if (!all_done[par] && j == 0 && round >= 3) {
      int first_bad = -1;
      for (int jj = 0; jj < K; jj++) if (need[jj][ai] && first_bad < 0) first_bad = jj;
      printf("  round %d block %d arfcn %d first invalid segment %d: used thr %.17g pf %d  flags %d\n", round, blockIdx.x, col, first_bad,
             __longlong_as_double(u_thr[first_bad < 0 ? 0 : first_bad][ai]), u_pf[first_bad < 0 ? 0 : first_bad][ai], sflag[first_bad < 0 ? 0 : first_bad][ai]);
    }
#endif
    if (all_done[par]) return;
    if (round > K + 1) {                                          // (at most K rounds by construction; the bound only makes the exit unconditional --
      if (tid == 0 && a.err) atomicOr(a.err, 1);                  //  and if it is ever hit, the host hears of it instead of getting unvalidated thresholds)
      return;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_group_replay_wave (round 5): the same machine, the same validated segments -- but a WAVE per (ARFCN, segment of 32 or 64 timeslots), its
// lanes the segment's slots, and the serial part visits only the slots at which the state CAN move.  What round 4's debug timing said
// of k_group_replay_seg: a slot-step is ~60 dependent instructions behind two mask reads, ~0.3 us, whether or not anything happens
// in it -- and in a running cell almost nothing does: the threshold sits on its floor of 0, every active burst passes the energy
// gate, a detected burst takes 1 from a threshold that is 0 already.  Here everything about a segment that does not depend on the
// state is one instruction across the lanes (the slots' codes and powers: one load each; which slots start a frame: a constant
// mask), energyDetect's decision for all the slots against the CURRENT threshold is one compare and a ballot, and the slots that can
// change the state under it are a mask:
//     a burst the correlator missed behind an open gate (the threshold rises, the clock is re-based),
//     an active burst under the threshold once 50 quiet frames CAN have passed (re-bases the clock, lowers the threshold; before that it
//       only LOOKS at the clock, which the boundary walk wants to know -- found after the run, a lane per slot -- and costs no visit),
//     a detected burst while the threshold is not 0 (the threshold falls by one).
// Of those the last kind needs no visit either: c detected bursts in a row take max(thr - c, 0) from the threshold, exactly (see the loop),
// so every lane forms the threshold its slot meets from the number of detected bursts ahead of it, and a turn of the loop takes the whole
// run of slots up to the first one that does something else.  The wave jumps from one such slot to the next (find-first-bit), the frame
// difference to prevFalseDetectionTime at a slot is a population count of the frame-start mask, and all of it is wave-uniform: scalar
// branches, no execution masks.  A slot's
// threshold-after is handed to the lanes from the slot of the change onwards; its verdict is formed afterwards, a lane per slot,
// from the threshold the slot before it left (the same float arithmetic as the step's).  Segments, assumed start states, the
// boundary walk and the proof that it ends with the serial result are k_group_replay_seg's, word for word; the walk is done by
// every wave for itself (K <= 16 boundaries from LDS, uniform), two buffers by round parity, one barrier a round.
// Workgroup = one ARFCN, K = ceil(n_slots / seg) <= 16 waves; seg = 32 while sixteen segments of 32 cover the call (the kernel lasts as long as
// its busiest wave, and a busy ARFCN's wave visits most of its slots), else 64.  The kernel gathers the detectors' answers through the row
// map itself (k_group_pack is not launched) and leaves the rows' gate and threshold (nor k_group_scatter).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kWaveSegs = 16;                               // waves per workgroup: calls of up to 64 * 16 = 1,024 timeslots
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(64 * kWaveSegs) void k_group_replay_wave(TrxGroupReplay a, double *__restrict__ thr_g, uint8_t *__restrict__ verdict_g, int Spad,
                                                                      int K, int seg) {
  enum { SF_CONS = 1, SF_MARK = 2 };
  __shared__ long long b_uthr[2][kWaveSegs], b_ethr[2][kWaveSegs];
  __shared__ int b_upf[2][kWaveSegs], b_epf[2][kWaveSegs], b_fl[2][kWaveSegs];
  __builtin_amdgcn_s_setprio(3);
  typedef unsigned long long u64;
  const int lane = threadIdx.x & 63;
  const int j = uni(threadIdx.x >> 6);                      // this wave's segment
  const int col = blockIdx.x;                               // the ARFCN
  const int ts = j * seg, t = ts + lane;                     // (seg = 64, or 32: the upper half of the lanes idles -- see the launcher)
  const bool in = lane < seg && t < a.n_slots;
  // the slot's row and, through it, the stateless detectors' answers (k_group_pack's gather, done here: one launch less)
  const int row = in ? a.rowmap[(size_t)t * a.S + col] : -1;
  int code = 0;
  float avg = 0.0f;
  if (row >= 0) {
    const int f = a.flags[row];
    avg = a.avgpwr[row];
    code = ((f & TRXSIG_F_BADLEN) ? 0 : RP_ACT) | ((f & TRXSIG_F_DETECT) ? RP_DET : 0) | (row < a.n_tsc_rows ? RP_TSC : 0);
  }
  const bool act_l = (code & RP_ACT) != 0;
  const u64 act_m = __builtin_amdgcn_ballot_w64(act_l);
  const u64 det_m = __builtin_amdgcn_ballot_w64((code & RP_DET) != 0);
  const long long s0_thr = uni64(__double_as_longlong(a.state[col].thr));
  const int s0_pf = uni(a.state[col].prev_false_fn);
  if (a.ev_list && blockIdx.x == 0 && threadIdx.x < TRXG_CLASS_RACH)   // k_group_cache_wave lists the estimating bursts behind these counts
    a.ev_list[a.class_base[threadIdx.x] + threadIdx.x] = 0;
  // slots that start a frame: the frame number and the frame difference move on by one BEFORE such a burst is looked at
  const int i0 = (8 - (a.tn0 & 7)) & 7;
  const u64 frame_m = (0x0101010101010101ull << i0) & (seg == 64 ? ~0ull : 0xffffffffull);
  constexpr int half = kHyperframe / 2;
  int fn_seg = (a.fn0 + ((a.tn0 + ts) >> 3)) % kHyperframe; // the frame of the segment's first slot ...
  const int fn_first = fn_seg;
  fn_seg -= i0 == 0 ? 1 : 0;                                // ... and one behind it where that slot itself starts the frame
  fn_seg += fn_seg < 0 ? kHyperframe : 0;
  // this wave's last run: the state it started from, the state it ended with, what it saw
  long long u_thr = 0, e_thr = 0;
  int u_pf = 0, e_pf = 0, fl = 0;
  long long n_thr = s0_thr;
  int n_pf = s0_pf;
  bool need = true;
  double o_thr = 0.0;                                        // this lane's slot: the threshold after it, and its verdict
  int o_v = 0;
#ifdef TRX_REPLAY_DEBUG
  const long long dk0 = wall_clock64();
  long long d_run = 0, d_load = 0;
  int d_events = 0, d_runs = 0;
  if (act_m || true) d_load = wall_clock64() - dk0;
#endif
  for (int round = 0;; round++) {
#ifdef TRX_REPLAY_DEBUG
    const long long dr0 = wall_clock64();
    d_runs += need ? 1 : 0;
#endif
    if (need) {
      double thr = __longlong_as_double(n_thr);
      const double thr0 = thr;                              // the threshold the segment's first slot meets
      int pf = n_pf;
      int seen;
      o_thr = thr;
      // frame difference at slot i = wrap(dbase + frames started in (mark, i]); before any re-basing: FNDelta(first frame, pf) (- 1, see above)
      int dbase = fn_delta(fn_first, pf) - (i0 == 0 ? 1 : 0);
      int nbm = 0;
      int fm = 64;                                          // the first slot that re-based the clock
      u64 todo = ~0ull;                                     // slots not yet passed
      for (;;) {
        // A turn takes a whole RUN of slots: up to the next slot that does something else, the only thing that moves the state is a
        // detected burst behind an open gate taking 1 from the threshold (floor 0) -- and c such steps from thr are max(thr - c, 0) in ONE
        // subtraction, exactly: x - 1 is exact for x >= 1 (and negative, hence floored, below), and thr - c is a multiple of thr's last
        // place no larger than thr.  So every lane forms the threshold its slot meets from the number of detected bursts ahead of it, and
        // energyDetect's decision from that; the run ends at the first slot that breaks the hypothesis -- a false detection, a detected
        // burst UNDER its threshold (it was counted as a step and is none), a burst under the threshold where 50 quiet frames can have
        // passed -- which is then handled on its own.
        const u64 dm = act_m & det_m & todo;
        const int c_l = __builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0));   // detected bursts ahead of this lane's slot
        double thr_l = thr - (double)c_l;
        thr_l = c_l > 0 ? (thr_l < 0.0 ? 0.0 : thr_l) : thr;
        const float thrF = (float)thr_l;
        const u64 pass_m = __builtin_amdgcn_ballot_w64(avg > thrF * thrF) & act_m;
        const u64 under_m = act_m & ~pass_m;
        const bool may_quiet = dbase + (seg >> 3) - nbm > 50;
        // (the one subtraction stands for the steps while they are exact: below 2^52.  A threshold can be anything -- 10 exp(-d) of a clock
        // that runs AHEAD of the bursts, d = -35 .. -39, lands between 2^53 and 2^59, where x - 1 rounds and c of them are not x - c: there
        // every detected burst ends the run and takes its step on its own)
        const bool exact = __builtin_amdgcn_ballot_w64(thr < 4503599627370496.0) != 0;
        const u64 brk_m = ((det_m & under_m) | (pass_m & ~det_m) | (may_quiet ? under_m : 0ull) | (exact ? 0ull : dm)) & todo;
        const int i = brk_m ? __builtin_ctzll(brk_m) : 64;
        // the run's slots: the threshold after each (a detected burst's own step included)
        const double t1 = thr_l - 1.0;
        const double after_l = (act_l && (code & RP_DET)) ? (t1 < 0.0 ? 0.0 : t1) : thr_l;
        const u64 run_m = todo & (i < 64 ? (1ull << i) - 1 : ~0ull);
        o_thr = ((run_m >> lane) & 1) ? after_l : o_thr;
        if (i == 64) {                                      // nothing breaks the run: the state behind the segment is the last lane's
          thr = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(after_l) >> 32), 63) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(after_l), 63));
          break;
        }
        thr = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(thr_l) >> 32), i) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(thr_l), i));   // the threshold slot i meets
        const u64 bit = 1ull << i;
        if (det_m & pass_m & bit) {                         // (only where the run could not take it: mEnergyThreshold -= 1.0F; floor 0, :338-339, 368-369)
          const double t = thr - 1.0;
          thr = t < 0.0 ? 0.0 : t;
        } else if (!(det_m & bit) || may_quiet) {           // (a detected burst under its threshold only looks at the clock unless it is quiet)
          const int nb = __builtin_popcountll(frame_m & ((bit << 1) - 1));   // frames started in slots 0 .. i
          int d = dbase + nb - nbm;
          d -= d >= half ? kHyperframe : 0;
          bool mark = true;
          if (pass_m & bit) {                               // a false detection: + 10.0F*exp(-framesElapsed) (:355, 374)
            const int k = d < -TRXG_EXP_LO ? -TRXG_EXP_LO : (d > TRXG_EXP_HI ? TRXG_EXP_HI : d);
            thr = thr + 10.0 * a.exp_tab[k + TRXG_EXP_LO];
          } else if (d > 50) thr = thr - 10.0;              // under the threshold, 50 quiet frames: 10 off it (:300-304)
          else mark = false;
          if (mark) {                                       // prevFalseDetectionTime = this burst's time
            int fn = fn_seg + nb;
            fn -= fn >= kHyperframe ? kHyperframe : 0;
            pf = fn; dbase = 0; nbm = nb;
            fm = fm < i ? fm : i;
          }
        }
#ifdef TRX_REPLAY_DEBUG
        d_events++;
#endif
        o_thr = lane >= i ? thr : o_thr;
        todo = ~((bit << 1) - 1);                           // (i = 63: nothing left)
        if (todo == 0) break;
      }
      // each slot's verdict from the threshold the slot before it left (energyDetect's decision and the correlator's answer behind it: the
      // walk's own float arithmetic); and what the boundary walk wants to know of the run: did a slot LOOK at the clock (a burst under
      // the threshold, a false detection) before or when the first one re-based it; did one re-base it
      double prev = __shfl_up(o_thr, 1);
      prev = lane == 0 ? thr0 : prev;
      const float pF = (float)prev;
      const bool pass_l = act_l && (avg > pF * pF);
      o_v = pass_l ? (RV_PASS | ((code & RP_DET) ? RV_SUCC : 0)) : 0;
      const u64 looked_m = act_m & ~(__builtin_amdgcn_ballot_w64(pass_l) & det_m);
      const u64 first_m = fm < 64 ? ((2ull << fm) - 1) : ~0ull;
      seen = ((looked_m & first_m) != 0 ? SF_CONS : 0) | (fm < 64 ? SF_MARK : 0);
      u_thr = n_thr; u_pf = n_pf;
      e_thr = uni64(__double_as_longlong(thr)); e_pf = pf; fl = seen;
    }
#ifdef TRX_REPLAY_DEBUG
    d_run += wall_clock64() - dr0;
#endif
    const int par = round & 1;
    if (lane == 0) { b_uthr[par][j] = u_thr; b_upf[par][j] = u_pf; b_ethr[par][j] = e_thr; b_epf[par][j] = e_pf; b_fl[par][j] = fl; }
    __syncthreads();
    long long w_thr = s0_thr;
    int w_pf = s0_pf;
    bool prefix = true;
    for (int jj = 0; jj < K; jj++) {                         // the walk over the boundaries (every wave for itself; uniform)
      const int f = uni(b_fl[par][jj]);
      const bool ok = uni64(b_uthr[par][jj]) == w_thr && (uni(b_upf[par][jj]) == w_pf || !(f & SF_CONS));
      if (jj == j) { need = !ok; n_thr = w_thr; n_pf = w_pf; }
      prefix = prefix && ok;
      w_thr = uni64(b_ethr[par][jj]);
      w_pf = (f & SF_MARK) ? uni(b_epf[par][jj]) : w_pf;
    }
#ifdef TRX_REPLAY_DEBUG
    if ((prefix || round > K + 1) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 77))
      printf("replay_wave block %d wave %d: %d rounds, %d runs, %d events, act %d; ticks (100 MHz): load %lld runs %lld total %lld\n", blockIdx.x, j, round + 1, d_runs, d_events,
             (int)__builtin_popcountll(act_m), d_load, d_run, wall_clock64() - dk0);
#endif
    if (prefix) {
      if (threadIdx.x == 0) { a.state[col].thr = __longlong_as_double(w_thr); a.state[col].prev_false_fn = w_pf; }
      break;
    }
    if (round > K + 1) {                                    // (at most K rounds by construction; as k_group_replay_seg: the exit is unconditional and the host hears of it)
      if (threadIdx.x == 0 && a.err) atomicOr(a.err, 1);
      break;
    }
  }
  if (in) {
    const int v = o_v;
    thr_g[(size_t)t * Spad + col] = o_thr;
    verdict_g[(size_t)t * Spad + col] = (uint8_t)(v | ((code & RP_TSC) ? RV_TSC : 0));
    if (row >= 0) {                                         // ... and the row's own results (the other forms leave this to k_group_scatter / k_group_cache)
      a.gate[row] = (uint8_t)(v & RV_SUCC);
      a.thr_after[row] = o_thr;
    }
  }
}

// The per-timeslot channel cache of the equalising leg (:313-325, 341-349, 357, 370).  What it needs from the threshold
// recurrence is only each burst's verdict (energy gate open? correlator detected?), and a timeslot's cache entry is touched by
// that timeslot's bursts alone -- so it is NOT part of the serial chain: a lane per (ARFCN, timeslot) walks its own bursts, one
// per frame (an eighth of the steps, eight times the lanes, integer work only), after k_group_replay has left the verdicts in
// (slot, ARFCN) order.  Per burst: the entry is stale when 50 frames have passed since its estimate or it is empty (:317); a
// detected normal burst behind a stale entry estimates the channel (RV_EVT; its row's taps become the entry); tix = the
// tap-table entry that equalises the burst; a missed normal burst or a detected access burst drops the entry (:357, :370).
__global__ __launch_bounds__(256) void k_group_cache(TrxGroupReplay a, const uint8_t *__restrict__ verdict_g, const double *__restrict__ thr_g, int Spad,
                                                     int rows_done) {
  const int id = blockIdx.x * 256 + threadIdx.x;            // timeslot-major: neighbouring lanes are neighbouring ARFCNs
  if (id >= 8 * a.S) return;
  const int tn = id / a.S, col = id - tn * a.S;
  const int S8 = a.S * 8;
  int est = a.state[col].est_fn[tn], src = a.state[col].tap_src[tn];
  const int t_first = (tn - a.tn0) & 7;                     // the call's first slot with this timeslot number
  int fn = a.fn0 + ((a.tn0 + t_first) >> 3);
  fn -= fn >= kHyperframe ? kHyperframe : 0;
#ifdef TRX_REPLAY_DEBUG
  const long long ck0 = wall_clock64();
  int c_evt = 0;
#endif
  // What the walk needs of a burst is its verdict byte (which says "normal burst" too: RV_TSC) and its row: kFr frames' worth are
  // loaded TOGETHER, before the walk looks at any (round 4 had eight in flight and the next eight behind them: the walk is short,
  // so it waited 2 us for memory every eight steps -- 20 us for 58 frames, profiles/r05_replay_probe.txt).  The threshold and the
  // amplitude are only needed where a burst estimates (a few steps of a call): fetched there.
  constexpr int kFr = 64;
  for (int tb = t_first; tb < a.n_slots; tb += 8 * kFr) {
    int ver[kFr], row[kFr];
    // (unconditional loads: a slot past the call's end reads the chunk's first cell again and is not looked at)
    const uint8_t *pv = verdict_g + (size_t)tb * Spad + col;
    const int32_t *pr = a.rowmap + (size_t)tb * a.S + col;
#pragma unroll
    for (int i = 0; i < kFr; i++) {
      const bool in = tb + 8 * i < a.n_slots;
      ver[i] = pv[in ? (size_t)(8 * i) * Spad : 0];
      row[i] = pr[in ? (size_t)(8 * i) * a.S : 0];
    }
    // every value is taken delivery of HERE: the walk below stores as it goes, stores and loads share one counter, and a wait for a
    // load left pending across the walk's branches becomes a wait for the previous step's stores (0.35 us a step: what round 4's walk
    // spent its 20 us on)
#pragma unroll
    for (int i = 0; i < kFr; i++) asm volatile("" ::"v"(ver[i]), "v"(row[i]));
    if (!rows_done) {                                       // (the replay kernel did not know the rows: the threshold after each burst moves there now)
      const double *pt = thr_g + (size_t)tb * Spad + col;
#pragma unroll
      for (int i0 = 0; i0 < kFr; i0 += 16) {
        double th[16];
#pragma unroll
        for (int k = 0; k < 16; k++) th[k] = pt[(tb + 8 * (i0 + k) < a.n_slots) ? (size_t)(8 * (i0 + k)) * Spad : 0];
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("" ::"v"(th[k]));
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (tb + 8 * (i0 + k) < a.n_slots && row[i0 + k] >= 0) a.thr_after[row[i0 + k]] = th[k];
      }
    }
#pragma unroll
    for (int i = 0; i < kFr; i++) {
      const int t = tb + 8 * i;
      if (t < a.n_slots) {                                  // (uniform)
        const int v = ver[i];
        const bool is_tsc = (v & RV_TSC) != 0;
        const bool pass = (v & RV_PASS) != 0, succ = (v & RV_SUCC) != 0, fail = pass && !succ;
        const bool stale = pass && is_tsc && (fn_delta(fn, est) > 50 || src < 0);   // ((double)d > 50 of an integer d)
        int sr = stale ? -1 : src;
        const bool evt = succ && stale;                      // this burst estimates the channel
#ifdef TRX_REPLAY_DEBUG
        c_evt += __any(evt) ? 1 : 0;
#endif
        sr = evt ? S8 + row[i] : sr;
        est = evt ? fn : est;
        const int tix = (succ && is_tsc) ? sr : 0;
        sr = ((fail && is_tsc) || (succ && !is_tsc)) ? -1 : sr;
        src = sr;
        // ... and the cell's row: the gate, the estimation event, the tap index, SNRestimate = |amp|^2 / (thr^2 + 1) in double with the
        // threshold AFTER its decrement (:340)
        const int rw = row[i];
        if (rw >= 0) {
          a.gate[rw] = (uint8_t)(v & RV_SUCC);
          a.ev[rw] = evt ? 1 : 0;
          a.tap_ix[rw] = tix;
          if (evt) {
            const trx_c32 am = a.amp[rw];
            const double th = thr_g[(size_t)t * Spad + col];
            const float n2 = am.i * am.i + am.r * am.r;       // Complex::norm2 (Complex.h:119)
            a.snr[rw] = (float)((double)n2 / (th * th + 1.0));
          }
        }
      }
      fn += 1; fn -= fn >= kHyperframe ? kHyperframe : 0;   // the next frame's slot with this timeslot number
    }
  }
  a.state[col].est_fn[tn] = est;
  a.state[col].tap_src[tn] = src;
#ifdef TRX_REPLAY_DEBUG
  if (id == 0) printf("group_cache: %d slots, steps with an estimating lane %d, ticks (100 MHz) %lld\n", a.n_slots, c_evt, wall_clock64() - ck0);
#endif
}

// k_group_cache_wave (round 5): the same cache walk, a WAVE per (ARFCN, timeslot) with the timeslot's bursts -- one a frame -- in its lanes.
// k_group_cache above is bound by its instruction count (~60 dependent instructions a burst on a lone wave: 19 us for 58 frames,
// profiles/r05_replay_probe.txt) although next to nothing happens in it: the entry changes only where a burst ESTIMATES (the first
// detected normal burst behind an empty or 50-frame-old entry) or DROPS it (a missed normal burst, a detected access burst).  Those
// are masks over the lanes: which bursts are detected normal bursts, which drop the entry, which are more than 50 frames past the
// current estimate (a compare against est, a lane per frame).  The next estimating burst is a find-first-bit -- behind the next
// drop if one comes first -- and everything between two of them takes the entry as it stands: tap index, event flag and SNRestimate
// leave a lane per burst, all estimating bursts' divisions side by side.
__global__ __launch_bounds__(1024) void k_group_cache_wave(TrxGroupReplay a, const uint8_t *__restrict__ verdict_g, const double *__restrict__ thr_g, int Spad,
                                                           int rows_done) {
  typedef unsigned long long u64;
  __shared__ int l_cnt[TRXG_CLASS_RACH], l_base[TRXG_CLASS_RACH];   // this workgroup's estimating bursts per class, and where its share of the class's list starts
  const int lane = threadIdx.x & 63;
  const int idr = uni(blockIdx.x * 16 + (threadIdx.x >> 6)); // (ARFCN, timeslot), timeslot-major as k_group_cache
  const bool mine = idr < 8 * a.S;                           // (a spare wave shadows the last pair and stores nothing: the workgroup's barriers are everyone's)
  const int id = mine ? idr : 8 * a.S - 1;
  const int tn = id / a.S, col = id - tn * a.S;
  const int S8 = a.S * 8;
  int est = uni(a.state[col].est_fn[tn]), src = uni(a.state[col].tap_src[tn]);
  const int t_first = (tn - a.tn0) & 7;                     // the call's first slot with this timeslot number
  int fn0 = a.fn0 + ((a.tn0 + t_first) >> 3);               // ... and its frame
  fn0 -= fn0 >= kHyperframe ? kHyperframe : 0;
  const int turns = (a.n_slots + 511) / 512;                 // 64 frames a turn; the same number of turns for every wave (a turn may hold nothing of a late timeslot)
  for (int c = 0; c < turns; c++) {
    const int t = t_first + 512 * c + 8 * lane;
    const bool in = t < a.n_slots;
    const int v = in ? verdict_g[(size_t)t * Spad + col] : 0;
    const int row = (in && mine) ? a.rowmap[(size_t)t * a.S + col] : -1;
    if (threadIdx.x < TRXG_CLASS_RACH) l_cnt[threadIdx.x] = 0;
    int fn = fn0 + lane;
    fn -= fn >= kHyperframe ? kHyperframe : 0;
    const bool c_l = (v & (RV_TSC | RV_PASS | RV_SUCC)) == (RV_TSC | RV_PASS | RV_SUCC);   // a detected normal burst behind an open gate
    const bool drop_l = (v & (RV_TSC | RV_PASS | RV_SUCC)) == (RV_TSC | RV_PASS) || (v & (RV_TSC | RV_SUCC)) == RV_SUCC;   // (:357, :370)
    const u64 c_m = __builtin_amdgcn_ballot_w64(c_l), drop_m = __builtin_amdgcn_ballot_w64(drop_l);
    int src_l = src;                                        // the entry as this lane's burst finds it (after its own estimate)
    u64 evt_m = 0, todo = ~0ull;
    for (;;) {
      // the next burst that estimates: with an entry, the first detected normal burst more than 50 frames past its estimate -- or the
      // first one behind the next drop, if that comes earlier; without one, the first detected normal burst
      u64 cand = c_m & todo;
      if (src >= 0) {
        const u64 old_m = __builtin_amdgcn_ballot_w64(fn_delta(fn, est) > 50);   // ((double)d > 50 of an integer d)
        const u64 dr = drop_m & todo;
        const u64 behind = dr ? ~((2ull << __builtin_ctzll(dr)) - 1) : 0ull;     // (a drop in lane 63: nothing behind it)
        cand &= old_m | behind;
      }
      if (cand == 0) break;
      const int i = __builtin_ctzll(cand);
      src = S8 + __builtin_amdgcn_readlane(row, i);
      est = __builtin_amdgcn_readlane(fn, i);
      evt_m |= 1ull << i;
      src_l = lane >= i ? src : src_l;
      todo = ~((2ull << i) - 1);
    }
    // the entry behind the turn's last burst: dropped if a drop follows the last estimate
    const u64 after = evt_m ? ~((2ull << (63 - __builtin_clzll(evt_m))) - 1) : ~0ull;
    src = (drop_m & after) ? -1 : src;
    const bool evt = row >= 0 && ((evt_m >> lane) & 1);
    int k = 0, lp = 0;
    __syncthreads();                                        // l_cnt is zero
    if (row >= 0) {
      const double th = thr_g[(size_t)t * Spad + col];
      if (!rows_done) {
        a.gate[row] = (uint8_t)(v & RV_SUCC);
        a.thr_after[row] = th;
      }
      a.ev[row] = evt ? 1 : 0;
      a.tap_ix[row] = c_l ? src_l : 0;
      if (evt) {                                            // SNRestimate = |amp|^2 / (thr^2 + 1) in double, the threshold AFTER its decrement (:340)
        const trx_c32 am = a.amp[row];
        const float n2 = am.i * am.i + am.r * am.r;         // Complex::norm2 (Complex.h:119)
        a.snr[row] = (float)((double)n2 / (th * th + 1.0));
#pragma unroll
        for (int q = 1; q < TRXG_CLASS_RACH; q++) k += row >= a.class_base[q] ? 1 : 0;
        lp = atomicAdd(&l_cnt[k], 1);
      }
    }
    // ... and the estimating bursts go on their class's list for the channel estimate (k_eq_list's job; the order of a list decides which
    // wave estimates which burst, nothing else): counted per workgroup in LDS, ONE addition to the class's count per workgroup and class
    // (an addition per burst -- a thousand to one address -- cost 14 us)
    __syncthreads();
    if (a.ev_list && threadIdx.x < TRXG_CLASS_RACH && l_cnt[threadIdx.x] > 0)
      l_base[threadIdx.x] = atomicAdd(a.ev_list + a.class_base[threadIdx.x] + threadIdx.x, l_cnt[threadIdx.x]);
    __syncthreads();
    if (a.ev_list && evt) a.ev_list[a.class_base[k] + k + 1 + l_base[k] + lp] = row - a.class_base[k];
    fn0 += 64;
    fn0 -= fn0 >= kHyperframe ? kHyperframe : 0;
  }
  if (lane == 0 && mine) {
    a.state[col].est_fn[tn] = est;
    a.state[col].tap_src[tn] = src;
  }
}

// (slot, ARFCN) order -> rows on the demodulating leg: gate and the threshold after the burst (the equalising leg: k_group_cache)
__global__ __launch_bounds__(256) void k_group_scatter(TrxGroupReplay a, int Spad, const double *__restrict__ thr_g,
                                                       const uint8_t *__restrict__ verdict_g) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)a.n_slots * a.S) return;
  const int row = a.rowmap[g];
  if (row < 0) return;
  const int t = (int)(g / a.S), s = (int)(g - (long long)t * a.S);
  const size_t q = (size_t)t * Spad + s;
  a.gate[row] = (uint8_t)(verdict_g[q] & RV_SUCC);
  a.thr_after[row] = thr_g[q];
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

// the wave form (k_group_replay_wave) for calls of up to 1,024 timeslots; otherwise, and under TRXSIG_TUNE_GROUP_REPLAY = 1, the forms that
// step through every slot
int trx_group_replay_form(int n_slots) { return (n_slots <= 64 * kWaveSegs && trx_knob(TRX_KNOB_GROUP_REPLAY) == 0) ? 0 : 1; }
static bool replay_wave_form(const TrxGroupReplay &a) { return a.form == 0; }

hipError_t trx_launch_group_pack(hipStream_t st, const TrxGroupReplay &a, float4 *packed) {
  const long long n = (long long)a.n_slots * a.S;
  if (n <= 0) return hipSuccess;
  if (replay_wave_form(a)) return hipSuccess;               // (k_group_replay_wave gathers for itself)
  k_group_pack<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, a.n_tsc_rows, a.rowmap, a.flags, a.avgpwr, a.amp, packed);
  return hipGetLastError();
}

hipError_t trx_launch_group_replay(hipStream_t st, const TrxGroupReplay &a, float4 *packed, double *thr_g, uint8_t *verdict_g, int32_t *tix_g,
                                   TrxProfiler *prof) {
  const long long n = (long long)a.n_slots * a.S;
  if (n <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_GROUP, st);
  const int Spad = (a.S + 63) / 64 * 64;
  // the wave form (k_group_replay_wave) for calls of up to 1,024 timeslots; otherwise (and under TRXSIG_TUNE_GROUP_REPLAY = 1)
  // long calls replay parallel in time, short ones one step after the other (the forms that step through every slot; segment length: a
  // multiple of eight timeslots, so that every segment starts on the same timeslot number)
  const bool wave = replay_wave_form(a);
  if (wave) {
    // segments of 32 timeslots while sixteen of them cover the call: a busy ARFCN's wave visits most of its slots, one after the other,
    // and the kernel lasts as long as the busiest wave (profiles/r05_replay_probe.txt)
    const int seg = a.n_slots <= 32 * kWaveSegs ? 32 : 64;
    const int K = (a.n_slots + seg - 1) / seg;
    k_group_replay_wave<<<dim3(a.S), dim3(64 * K), 0, st>>>(a, thr_g, verdict_g, Spad, K, seg);
  } else if (a.n_slots >= 384) k_group_replay_seg<16><<<dim3((a.S + 15) / 16), dim3(256), 0, st>>>(a, packed, thr_g, verdict_g, Spad, ((a.n_slots + 15) / 16 + 7) / 8 * 8);
  else if (a.n_slots >= 128) k_group_replay_seg<8><<<dim3((a.S + 31) / 32), dim3(256), 0, st>>>(a, packed, thr_g, verdict_g, Spad, ((a.n_slots + 7) / 8 + 7) / 8 * 8);
  else k_group_replay<<<dim3(Spad / 64), dim3(64), 0, st>>>(a, packed, thr_g, verdict_g, Spad);
  // the equalising leg's cache walk visits every (slot, ARFCN) cell once and leaves the rows' results itself; on the other leg the wave form
  // has left them already, the other forms scatter
  (void)tix_g;
  if (a.equalize && wave) k_group_cache_wave<<<dim3((8 * a.S + 15) / 16), dim3(1024), 0, st>>>(a, verdict_g, thr_g, Spad, 1);
  else if (a.equalize) k_group_cache<<<dim3((8 * a.S + 255) / 256), dim3(256), 0, st>>>(a, verdict_g, thr_g, Spad, 0);
  else if (!wave) k_group_scatter<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(a, Spad, thr_g, verdict_g);
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
