/* trxsig_trxgroup.h -- S `Transceiver` objects (one per ARFCN: TRXManager/TRXManager.cpp:44-54 makes one per carrier,
 * apps/OpenBTS.cpp:66) served by ONE GPU context at batch speed: the receive side of Transceiver/Transceiver.cpp:207-410
 * (expectedCorrType + pullRadioVector) for n_slots consecutive timeslots x S ARFCNs per call, and (end of this header) the
 * transmit side :100-181 (addRadioVector / pushRadioVector: priority queue, stale dump, filler table).
 *
 * Why it exists: include/trxsig_transceiver.h answers pullRadioVector one burst per call (a PCIe round trip and three to
 * six tiny launches each: 38-94 us, slower than the reference on one CPU core).  The batch detectors of trxsig.h are
 * fast but stateless, while pullRadioVector is not: mEnergyThreshold, prevFalseDetectionTime and the per-timeslot
 * channel / DFE cache carry from burst to burst of an ARFCN (SURVEY 8a' item 14).  The group keeps that state ON THE
 * DEVICE and replays it there:
 *
 *   1. the host classifies every (slot, ARFCN) with expectedCorrType (:207-269) and groups the bursts that reach a
 *      correlator into rows by class -- training sequence 0..7 (each ARFCN has its own mTSC) and access bursts;
 *   2. one stateless detector launch per class in use, energy gate off: energyDetect's avgPwr, analyzeTrafficBurst /
 *      detectRACHBurst's (detected, amplitude, TOA) for every row;
 *   3. k_group_replay, a lane per ARFCN walking its bursts in time order: energyDetect's decision against the adaptive
 *      threshold (:298-306), the threshold updates (:303, 338-339, 355-356, 367-375, in the reference's double
 *      arithmetic; exp() from a table filled by the host's libm), the channel cache's "estimate now?" (:313-325), which
 *      taps equalise which burst, SNRestimate (:340);
 *   4. channel estimate + designDFE for the rows the replay marked (:341-349), equalizeBurst with the slot's taps
 *      (:391-396) or demodulateBurst (:385-388) for every row that comes back as a SoftVector;
 *   5. the slots' cache entries are refreshed from this call's estimates.
 *   Everything is enqueued on the context's stream; nothing synchronises until the caller collects.
 *
 * Results equal S independent trxsig_trx objects fed the same bursts one by one, and oracle/transceiver_model.py
 * (tests/test_gpu_trxgroup.py): soft bits, RSSI, timing offset and the threshold after every burst (exact double).
 * Thread safety: one caller at a time per group (the reference serialises a Transceiver with mLock); the context's
 * other entry points may be used between calls.
 */
#ifndef TRXSIG_TRXGROUP_H
#define TRXSIG_TRXGROUP_H

#include "trxsig_frontend.h"
#include "trxsig_transceiver.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trxsig_trxgroup trxsig_trxgroup;

/* n_arfcn Transceivers on `ctx` (which the group borrows: destroy the group first).  tsc_leg: TRXSIG_TSCLEG_EQUALIZE
 * (Transceiver/Transceiver.cpp as written; needs a context with sps == 1) or TRXSIG_TSCLEG_DEMOD (see
 * trxsig_trx_set_tsc_leg; any sps).  Every ARFCN starts as Transceiver::Transceiver leaves it (:58-92): all slots
 * NONE, TSC 0, threshold 250.0, prevFalseDetectionTime = channelEstimateTime[] = (start_fn, start_tn). */
int trxsig_trxgroup_create(trxsig_trxgroup **out, trxsig_ctx *ctx, int n_arfcn, int tsc_leg, int start_fn, int start_tn);
void trxsig_trxgroup_destroy(trxsig_trxgroup *g);
int trxsig_trxgroup_arfcns(const trxsig_trxgroup *g);

/* driveControl (:439-580) of ARFCN `arfcn`: as trxsig_trx_control.  SETSLOT / SETTSC take effect at the next pull. */
int trxsig_trxgroup_control(trxsig_trxgroup *g, int arfcn, const char *command, char *response, int response_cap);
int trxsig_trxgroup_expected_corr_type(const trxsig_trxgroup *g, int arfcn, int tn, int fn);

/* What one pull leaves behind, device resident, owned by the group, valid until its next pull.  A "row" is a burst that
 * reached a correlator; rows are grouped by class (TSC 0..7 in turn, then access bursts), in (slot, ARFCN) order inside
 * a class. */
typedef struct {
  int n_slots, n_arfcn, n_rows;
  const int32_t *d_row;        /* [n_slots][n_arfcn]: the burst's row, -1 where the slot is OFF / IDLE (NULL comes back)  */
  const uint8_t *d_valid;      /* [n_rows] TRXSIG_F_DETECT where pullRadioVector returns a SoftVector, else 0              */
  const uint8_t *d_flags;      /* [n_rows] the stateless detector's TRXSIG_F_* (energy gate off)                           */
  const trxsig_c32 *d_amp;     /* [n_rows] amplitude                                                                        */
  const float *d_toa;          /* [n_rows] TOA in samples                                                                   */
  const float *d_avgpwr;       /* [n_rows] energyDetect's avgPwr                                                            */
  const double *d_threshold;   /* [n_rows] mEnergyThreshold after the burst                                                 */
  const float *d_soft;         /* [n_rows][soft_stride]: the SoftVector's first 148 values WHERE d_valid IS SET.  Elsewhere
                                * unspecified: zeros, or -- demodulating leg, large calls, where the state machine replays
                                * beside the demodulator -- the demodulated burst of a row the stateless detector flagged
                                * and the machine then did not accept (energy gate).  d_valid decides, as the return
                                * value of pullRadioVector does.                                                          */
  int soft_stride;
} trxsig_trxgroup_result;

/* pullRadioVector for n_slots consecutive timeslots starting at (fn, tn), every ARFCN: the burst of slot t (time
 * (fn, tn) + t timeslots) and ARFCN a starts at d_samples + t*slot_stride + a*arfcn_stride (complex samples, device)
 * and has (156 + (TN % 4 == 0)) * sps samples (radioInterface.cpp:370-378), or burst_len samples if burst_len > 0.
 * Offsets must stay below 2^31 samples.  Asynchronous: d_samples must stay unchanged until the stream has run. */
int trxsig_trxgroup_pull(trxsig_trxgroup *g, const trxsig_c32 *d_samples, int64_t slot_stride, int64_t arfcn_stride,
                         int burst_len, int fn, int tn, int n_slots, trxsig_trxgroup_result *res);

/* The same with the bursts coming straight from the radio: RadioInterface::pullBuffer + driveReceiveRadio
 * (radioInterface.cpp:197-273, 359-401) + pullRadioVector in one call.  `fe` is a receive front end with one stream per
 * ARFCN on the group's context (trxsig_frontend.h; sps 4, used through its fused calls only); d_iq / n_chunks as
 * trxsig_rxfe_push_detect_demod_normal.  The detectors -- midamble and access-burst alike, by each slot's expectedCorrType --
 * compute their samples from the int16 chunks; the resampled stream never exists in memory.  The push completes *n_slots
 * timeslots (possibly 0); the first one is at time (fn, the front end's current TN) -- the GSM clock is the caller's
 * (radioInterface.cpp:364-366).
 * That fused form is the 260 : 96 resampler (sps 4) feeding the TRXSIG_TSCLEG_DEMOD leg.  A group on the equalising leg, or at
 * another sps -- the reference's own configuration is sps 1 with the equaliser -- takes the same call through the resampled
 * stream instead: trxsig_rxfe_push + trxsig_rxfe_pop + trxsig_trxgroup_pull_bursts on what the pop lists (such a front end is
 * then a push / pop one).  Same results where both routes exist. */
int trxsig_trxgroup_pull_rxfe(trxsig_trxgroup *g, trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, int fn, int *n_slots,
                              trxsig_trxgroup_result *res);

/* trxsig_trxgroup_pull on LISTED bursts: burst t of ARFCN a is entry a*n_per_arfcn + t of d_offset / d_length (samples into
 * d_samples) -- the layout trxsig_rxfe_pop hands out, 157-156-156-156 lengths included; (fn, tn) = the time of burst 0. */
int trxsig_trxgroup_pull_bursts(trxsig_trxgroup *g, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length,
                                int n_per_arfcn, int fn, int tn, trxsig_trxgroup_result *res);

/* What the caller of pullRadioVector sees, on the host, for the last pull (synchronises the stream): entry t*n_arfcn + a
 *   h_valid   1 where a SoftVector came back
 *   h_soft    [n_slots*n_arfcn][148] its soft bits (untouched where h_valid is 0); may be NULL
 *   h_rssi    (int) floor(20 log10(9450 / |amp|))   (:400)        h_timing  (int) round(TOA * 256 / sps)   (:402)
 *   h_threshold  mEnergyThreshold after the burst (NaN where no correlator ran); may be NULL */
int trxsig_trxgroup_collect(trxsig_trxgroup *g, uint8_t *h_valid, float *h_soft, int *h_rssi, int *h_timing, double *h_threshold);

/* trxsig_trxgroup_pull on host samples (copied to a device buffer of the group first; PCIe-inclusive), same layout;
 * h_samples holds (n_slots-1)*slot_stride + (n_arfcn-1)*arfcn_stride + (burst_len ? burst_len : 157*sps) samples. */
int trxsig_trxgroup_pull_host(trxsig_trxgroup *g, const trxsig_c32 *h_samples, int64_t slot_stride, int64_t arfcn_stride,
                              int burst_len, int fn, int tn, int n_slots);

/* Pipelined mode (off by default; demodulating leg): a large pull (the ones that replay the state machine on the group's side
 * stream, see trxsig_trxgroup_pull) RETURNS WITHOUT JOINING that stream, so the machine of call i replays while call i+1's
 * detectors run -- the replay is a latency chain of two waves per 128 ARFCNs and otherwise ends every call with the GPU idle.
 * What that changes for the caller: d_flags, d_amp, d_toa, d_avgpwr and d_soft of a result are ordered on the context's stream
 * as always; d_valid, d_threshold and the group's own state are complete on that stream only after trxsig_trxgroup_sync (or
 * trxsig_trxgroup_collect / _energy_threshold, which call it, or any later pull that is not pipelined).  Results live in one of
 * two workspace sets: a result stays valid until the SECOND pull after its own.  Values are the same in either mode. */
int trxsig_trxgroup_set_pipelined(trxsig_trxgroup *g, int on);
/* Which pulls replay the state machine on the group's SIDE stream, beside the demodulator (demodulating leg): those with at
 * least `rows` (slot, ARFCN) rows; 0 = never (the default since round 4: a long call's replay is parallel in time and one stream
 * is faster).  An implementation choice for A/B measurements and for pipelined mode; same values either way. */
int trxsig_trxgroup_set_beside_rows(trxsig_trxgroup *g, int rows);
/* Which pulls on a fused front end (trxsig_trxgroup_pull_fused) detect their access bursts BESIDE their normal bursts: those with at
 * least `rows` rows that hold both kinds (the access-burst class goes first on the context's stream, the normal-burst classes
 * follow on the group's side stream and are joined before the state machine; two cross-stream hops of ~10 us are what it costs).
 * 0 = never, the default: measured slower than the classes in series (DESIGN "What else was tried").  An implementation choice for
 * A/B measurements; same values either way (tested). */
int trxsig_trxgroup_set_split_rows(trxsig_trxgroup *g, int rows);
/* the context's stream waits for every replay still in flight on the side stream (no host wait) */
int trxsig_trxgroup_sync(trxsig_trxgroup *g);
/* mEnergyThreshold of one ARFCN now (synchronises) */
int trxsig_trxgroup_energy_threshold(trxsig_trxgroup *g, int arfcn, double *thr);

/* ---- the transmit half: addRadioVector (:100-113) / pushRadioVector (:138-181) for every ARFCN of the group -------------------
 * The transmit priority queue (mTransmitPriorityQueue), the stale-burst dump and the filler table fillerTable[FN % modulus][TN]
 * (modulus 26 / 51 / 102 by the slot's channel combination, setModulus :183-204) live ON THE DEVICE, per ARFCN; a burst is kept
 * as its 148 bits and its gain pow(10, -RSSI/10) (integer division, as the reference writes it), never as modulated samples --
 * modulateBurst + scaleVector of the same bits and gain give the same samples each time, so what reaches the radio is decided
 * here and FORMED by the transmit back end's one kernel (trxsig_txbe, fused: bits -> modulate -> resample -> gain -> int16).
 * Bursts with equal timestamps leave the queue in the order they would leave the reference's std::priority_queue
 * (csrc/trxsig_txq.h reproduces its moves; tests/test_txqueue_order.py).  Capacity: 256 queued bursts per ARFCN; a burst that
 * does not fit is dropped and reported by trxsig_trxgroup_tx_queue_size. */

/* driveTransmitPriorityQueue's parse + addRadioVector for n datagrams in arrival order: datagram i (154 bytes, host memory:
 * TN, FN big-endian, RSSI, 148 bits -- Transceiver.cpp:596-620, TRXManager.cpp:173-200) belongs to ARFCN h_arfcn[i] (the data
 * socket it arrived on).  A timeslot > 7, a frame number >= gHyperframe or an unknown ARFCN refuses the whole call
 * (TRXSIG_EINVAL, nothing queued; cf. trxsig_trx_decode_tx_datagram).  Asynchronous; the host buffers are consumed at return. */
int trxsig_trxgroup_add_bursts(trxsig_trxgroup *g, const uint8_t *h_datagrams, const int32_t *h_arfcn, int n);
/* The same without the host copy (round 5): the group lends a PINNED block to receive into -- recvfrom() straight into
 * (*h_datagrams)[154 i], the socket's ARFCN into (*h_arfcn)[i], i < n_max -- and trxsig_trxgroup_add_staged(g, n) adds its first n
 * datagrams: the host only checks the headers (the refusal rule above), the block goes up in one DMA as it arrived, and parsing
 * (TN, big-endian FN, RSSI -> pow(10, -RSSI/10) with the integer division), the per-ARFCN sort that keeps the arrival order, the
 * queue insertion and the payload copies are the device's: a kernel behind the upload (k_group_tx_arrive), and the queues' kernel --
 * whose launch is left to the call that next needs the queues: the trxsig_trxgroup_push that follows takes the insertions into its own
 * launch (the usual order of a transmit loop; results do not depend on it), another add or trxsig_trxgroup_tx_queue_size launch them first.  Three blocks take turns: after add_staged the
 * pointers are the DMA's; ask again for the next batch (the call waits, if it must, for the upload and the ingest that last used
 * that block's set: that is where a host is held back when the device is more than a batch behind).
 * trxsig_trxgroup_add_bursts is this with a copy into the block first.
 * Streams: the uploads and the queue's kernels (this call's ingest, trxsig_trxgroup_push's walk) run on streams of the group's
 * own, in call order, beside whatever the context's stream is doing (the transmit back end of the batch before); everything
 * that hands results to the caller (trxsig_trxgroup_push, _push_txbe, _tx_queue_size) orders them on the context's stream. */
int trxsig_trxgroup_tx_staging(trxsig_trxgroup *g, int n_max, uint8_t **h_datagrams, int32_t **h_arfcn);
int trxsig_trxgroup_add_staged(trxsig_trxgroup *g, int n);

/* pushRadioVector(nowTime) for n_slots consecutive timeslots from (fn, tn), every ARFCN, in time order per ARFCN: stale bursts
 * (earlier than the slot) move to the filler table at THEIR time's entry, a burst for exactly the slot replaces the slot's
 * filler entry and goes out, otherwise the filler entry goes out.  What goes out, device resident, owned by the group, ordered on
 * the context's stream, valid until its next push (work enqueued on the context's stream BEFORE that push may still read it: two
 * output sets alternate): *d_bits [n_arfcn][n_slots][148] (one bit per byte), *d_gain [n_arfcn][n_slots], *d_from_queue
 * [n_arfcn][n_slots] (1: from the queue, 0: filler) -- the layout trxsig_txbe_push_bursts / trxsig_modulate_batch take. */
int trxsig_trxgroup_push(trxsig_trxgroup *g, int fn, int tn, int n_slots, const uint8_t **d_bits, const float **d_gain,
                         const uint8_t **d_from_queue);
/* the same, handed straight to a transmit back end with one stream per ARFCN (guard symbols 8 + (TN % 4 == 0), :105):
 * n_slots <= the back end's max_bursts; trxsig_txbe_pop then yields the int16 stream for the radio */
int trxsig_trxgroup_push_txbe(trxsig_trxgroup *g, trxsig_txbe *be, int fn, int tn, int n_slots);
/* bursts waiting in ARFCN `arfcn`'s queue (synchronises); *dropped (may be NULL) = 1 if that ARFCN ever lost a burst to a full queue */
int trxsig_trxgroup_tx_queue_size(trxsig_trxgroup *g, int arfcn, int *dropped);

#ifdef __cplusplus
}
#endif
#endif /* TRXSIG_TRXGROUP_H */
