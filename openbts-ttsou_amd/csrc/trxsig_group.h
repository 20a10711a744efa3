// trxsig_group.h -- internal: what the host side of the Transceiver group (trxsig_trxgroup.cpp) and its kernels
// (trxsig_group.hip) share.  The group is S `Transceiver` objects (Transceiver/Transceiver.cpp, one per ARFCN) whose
// per-burst state machine -- mEnergyThreshold, prevFalseDetectionTime, the per-timeslot channel / DFE cache -- is
// replayed ON THE DEVICE, a lane per ARFCN in burst order, between the stateless batch detectors and the batch
// demodulator / equaliser, so that one call serves n_slots x S bursts without a host round trip in the middle.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "trxsig_launch.h"

// Per-ARFCN receive state of class Transceiver (Transceiver.h:95-116), device resident.
struct TrxGroupArfcn {
  double thr;                // mEnergyThreshold (Transceiver.cpp:88: 250.0)
  int32_t prev_false_fn;     // prevFalseDetectionTime.FN() (only the frame number enters Time::operator-, GSMCommon.h:414-417)
  int32_t pad;
  int32_t est_fn[8];         // channelEstimateTime[ts].FN()
  int32_t tap_src[8];        // -1: channelResponse[ts] == NULL; else the entry of the tap table that holds DFEForward[ts],
                             // DFEFeedback[ts] and chanRespOffset[ts]
};

// exp(-k) for the integer frame differences Transceiver.cpp:355,374 can form, tabulated on the host with the host's libm
// (the function the reference calls): entry k + TRXG_EXP_LO, k clipped to [-TRXG_EXP_LO, TRXG_EXP_HI].  exp(710) = +inf
// and exp(-746) = 0 in double, so the clipped ends are the values of everything beyond them.
#define TRXG_EXP_LO 710
#define TRXG_EXP_HI 746
#define TRXG_EXP_N (TRXG_EXP_LO + TRXG_EXP_HI + 1)

// Row classes: rows (bursts that reach a correlator) are grouped by class, class k = TSC k for k < 8, 8 = RACH.
#define TRXG_NCLASS 9
#define TRXG_CLASS_RACH 8

struct TrxGroupExpand {
  int S, n_slots, tn0, sps, fixed_len, G;                  // G: segment columns per slot
  long long slot_stride, arfcn_stride, base;               // burst (t, a) starts at sample base + t*slot_stride + a*arfcn_stride
  int rx_nb;                                               // > 0: the bursts are a receive front end's (trxsig_rxgen.h); `off` gets
                                                           // the row's burst index there, a*rx_nb + t, instead of a sample offset
  const int32_t *src_off, *src_len;                        // non-NULL: the bursts are listed -- burst t of ARFCN a is entry a*src_nb + t
  int src_nb;                                              // (a receive front end's trxsig_rxfe_pop); strides / fixed_len unused then
  const uint16_t *gid;                                     // [8][S]: which segment column ARFCN a belongs to on timeslot tn
  const int32_t *pos;                                      // [8][S]: its place inside that segment
  const int32_t *seg_base;                                 // [n_slots][G]: first row of the segment, -1 = no correlator (OFF / IDLE)
  int32_t *rowmap;                                         // [n_slots][S] -> row or -1
  int32_t *off, *len;                                      // [rows]
};
hipError_t trx_launch_group_expand(hipStream_t st, const TrxGroupExpand &a);

struct TrxGroupReplay {
  int S, n_slots, fn0, tn0, equalize, n_tsc_rows;          // rows < n_tsc_rows are normal bursts, the rest access bursts
  int form;                                                // trx_group_replay_form(n_slots), decided ONCE per call: 0 = k_group_replay_wave (gathers the detectors'
                                                           // answers itself and leaves the rows' gate / threshold: no k_group_pack, no k_group_scatter), 1 = the stepping forms
  const int32_t *rowmap;
  const uint8_t *flags; const trx_c32 *amp; const float *avgpwr;   // the stateless detectors' answers (energy gate off)
  const double *exp_tab;
  TrxGroupArfcn *state;
  uint8_t *gate;                                           // TRXSIG_F_DETECT where pullRadioVector returns a SoftVector
  uint8_t *ev;                                             // 1 where the burst's channel is estimated (Transceiver.cpp:341-349)
  int32_t *tap_ix;                                         // tap-table entry the burst is equalised with
  float *snr;                                              // SNRestimate[ts] (:340) of an estimating burst
  double *thr_after;                                       // mEnergyThreshold after the burst
  int32_t *ev_list;                                        // (wave form, equalising leg; else NULL) the estimating bursts LISTED per class for the channel estimate:
  int class_base[TRXG_NCLASS + 1];                         // class k's count at ev_list[class_base[k] + k], its rows (relative to class_base[k]) behind it
  int *err;                                                // device word: bit 0 = a time-parallel replay left its loop at the round bound without a
                                                           // validated result (never observed: the proof says at most K rounds); trxsig_trxgroup_collect reports it
};
// packed: scratch of n_slots * S float4 (the detectors' answers in (slot, ARFCN) order)
// thr_g / verdict_g (and tix_g on the equalising leg, else NULL): trx_group_replay_scratch(S, n_slots) entries each, the replay's
// (slot, ARFCN)-ordered outputs before k_group_scatter moves them to the rows
size_t trx_group_replay_scratch(int S, int n_slots);
int trx_group_replay_form(int n_slots);
// the gather first, on its own: on a large call it runs on the context's stream BEFORE the fork, so that the side stream's first
// kernel is the replay itself and starts together with the demodulator -- once that kernel has filled the machine, the replay's
// few workgroups wait for it to drain (measured: 80 us instead of 15)
hipError_t trx_launch_group_pack(hipStream_t st, const TrxGroupReplay &a, float4 *packed);
hipError_t trx_launch_group_replay(hipStream_t st, const TrxGroupReplay &a, float4 *packed, double *thr_g, uint8_t *verdict_g, int32_t *tix_g,
                                   TrxProfiler *prof);

// toa_eq[row] = TOA - chanRespOffset[ts] (Transceiver.cpp:393) for the gated normal-burst rows
hipError_t trx_launch_group_toa_eq(hipStream_t st, int n_rows, const uint8_t *gate, const float *toa, const int32_t *tap_ix,
                                   const float *chan_off_tab, float *toa_eq);
// end of a call: the taps estimated in this batch that are still a slot's current ones move into the slot's cache entry
hipError_t trx_launch_group_commit(hipStream_t st, int S, TrxGroupArfcn *state, trx_c32 *w_tab, trx_c32 *b_tab, float *chan_off_tab);

// ---- transmit half (trxsig_grouptx.hip): addRadioVector / pushRadioVector (Transceiver.cpp:100-113, 138-181) for S ARFCNs ----
// A burst is kept as its PAYLOAD -- 148 bits, one per byte, then its gain pow(10, -RSSI/10) as a float: TRXG_PAYLOAD_WORDS
// 32-bit words -- in a per-ARFCN pool; the queue and the filler table hold payload slots (-1 = the dummy burst every filler
// entry starts as, Transceiver.cpp:66-75).  Arrays indexed [..][S] keep a lane-per-ARFCN walk coalesced.
#define TRXG_PAYLOAD_WORDS 38
struct TrxGroupTx {
  int S, qcap, npool;              // ARFCNs; queue capacity and payload slots per ARFCN (npool >= qcap + 102*8)
  int32_t *q_fn, *q_key;           // [qcap][S]: the priority queue (trxsig_txq.h), key = tn | slot << 3
  int32_t *q_n;                    // [S] its size
  int16_t *free_stack;             // [npool][S] free payload slots
  int32_t *free_n;                 // [S]
  int16_t *filler;                 // [102][8][S]: fillerTable[FN % modulus][TN] (Transceiver.h:79)
  uint8_t *fmod;                   // [8][S]: fillerModulus[TN] (setModulus, :183-204)
  uint32_t *pool;                  // [S][npool][TRXG_PAYLOAD_WORDS]
  const uint32_t *dummy;           // [TRXG_PAYLOAD_WORDS]: gDummyBurst, gain 1
  uint32_t *status;                // [S] bit 0: a burst was dropped because the queue / pool was full
};
// n new bursts as they arrived: dgram n x 154 bytes (TN, FN big-endian, RSSI, 148 bits one per byte), arfcn[n] the ARFCN each came
// for (all checked by the host); gain_tab26[q + 12] = (float)pow(10, q), q = -12 .. 13.  Two launches: ARRIVE -- parsing and the
// per-ARFCN sort (arrival order kept), nothing of the queues' state in it, so it may run on another stream than the queues' --
// leaves its lists in a_lf / a_lk (trx_group_tx_arrive_ints(S, n, &t) ints each) and a_tot (t ints); INGEST enters them in the
// queues and copies the payloads.  ref_fn: a frame number near the datagrams' (the first one's); far != 0: some datagram lies
// 2^17 frames or more from it (trxsig_txq_lds.h's packed entries cannot say it: the kernel works on the arrays in memory -- same
// results, slower).
size_t trx_group_tx_arrive_ints(int S, int n, size_t *tot_ints);
hipError_t trx_launch_group_tx_arrive(hipStream_t st, int S, int n, const uint8_t *dgram, const int32_t *arfcn, int32_t *a_lf, int32_t *a_lk,
                                      int32_t *a_tot);
hipError_t trx_launch_group_tx_ingest(hipStream_t st, const TrxGroupTx &x, int n, const uint8_t *dgram, const int32_t *a_lf, const int32_t *a_lk,
                                      const int32_t *a_tot, const float *gain_tab26, int ref_fn, int far);
// pushRadioVector for n_slots timeslots from (fn0, tn0) on every ARFCN: bits_out [S][n_slots][148], gain_out [S][n_slots], fq_out
// [S][n_slots] (1 = the burst came from the queue)
hipError_t trx_launch_group_tx_push(hipStream_t st, const TrxGroupTx &x, int fn0, int tn0, int n_slots, uint8_t *bits_out, float *gain_out,
                                    uint8_t *fq_out);
// the add call's ingest and the push that follows it in ONE launch (the queues go to LDS and back once); far_add != 0: some datagram
// of the add call lies TRXQ_PK_WIN frames or more from fn0
hipError_t trx_launch_group_tx_both(hipStream_t st, const TrxGroupTx &x, int n, const uint8_t *dgram, const int32_t *a_lf, const int32_t *a_lk,
                                    const int32_t *a_tot, const float *gain_tab26, int far_add, int fn0, int tn0, int n_slots, uint8_t *bits_out,
                                    float *gain_out, uint8_t *fq_out);
