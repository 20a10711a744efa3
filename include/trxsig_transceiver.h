/* trxsig_transceiver.h -- host side of the burst path: the per-ARFCN orchestration that the reference keeps
 * in class Transceiver (Transceiver/Transceiver.{h,cpp}) around the sigProcLib calls, rebuilt on top of
 * libtrxsig's GPU entry points:
 *
 *   - pullRadioVector (Transceiver.cpp:271-410): expectedCorrType schedule (:207-269), energyDetect against the
 *     adaptive mEnergyThreshold (init 250.0, -10 after 50 quiet frames, -1 on success with floor 0,
 *     +10*exp(-frames) on a false detection), TSC leg = analyzeTrafficBurst + per-timeslot channel / DFE
 *     cache (re-estimated on the first burst of a slot and after 50 frames) + equalizeBurst, RACH leg =
 *     detectRACHBurst + demodulateBurst, RSSI and timing offset;
 *   - addRadioVector / pushRadioVector (:100-113, :138-181): modulate + scale by pow(10,-RSSI/10), the
 *     transmit priority queue, the filler table [FN % modulus][TN] (modulus 26/51/102 by channel
 *     combination, :183-204) pre-loaded with the modulated dummy burst;
 *   - the UDP wire formats (:582-639, :641-677; TRXManager/README.TRXManager) and the control commands
 *     (:439-580) as pure functions on byte buffers / strings (a socket loop only has to move them);
 *   - createLPF (sigProcLib.cpp:1102-1150).
 *
 * Every number comes from the GPU kernels or from the reference's own host arithmetic (double where the
 * reference uses double); one call = one burst, as in the reference, so each call is a PCIe round trip --
 * this is the drop-in form, not the fast one (see INTEGRATION.md for the batched service).
 * State is per object (one ARFCN); the object is not thread-safe (the reference serialises with mLock).
 */
#ifndef TRXSIG_TRANSCEIVER_H
#define TRXSIG_TRANSCEIVER_H

#include "trxsig.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trxsig_trx trxsig_trx;

/* Transceiver::CorrType (Transceiver.h:73-78) and ChannelCombination (:82-92) */
enum { TRXSIG_CORR_OFF = 0, TRXSIG_CORR_TSC = 1, TRXSIG_CORR_RACH = 2, TRXSIG_CORR_IDLE = 3 };
enum { TRXSIG_CHAN_NONE = 0, TRXSIG_CHAN_I, TRXSIG_CHAN_II, TRXSIG_CHAN_III, TRXSIG_CHAN_IV, TRXSIG_CHAN_V,
       TRXSIG_CHAN_VI, TRXSIG_CHAN_VII, TRXSIG_CHAN_LOOPBACK };
#define TRXSIG_RX_DATAGRAM_BYTES 158    /* gSlotLen + 10 (Transceiver.cpp:658, 674) */
#define TRXSIG_TX_DATAGRAM_BYTES 154    /* gSlotLen + 1 + 4 + 1 (:590) */

/* Transceiver::Transceiver (:40-92): samples per symbol, start time.  The TSC leg equalises
 * (equalizeBurst "assumes symbol-rate sampling"), so it needs sps == 1; RACH and TX work at any sps. */
int trxsig_trx_create(trxsig_trx **out, int device, int sps, int start_fn, int start_tn);
/* How the TSC leg of pullRadioVector ends.  TRXSIG_TSCLEG_EQUALIZE (the default): Transceiver/Transceiver.cpp:313-349,
 * 391-396 -- per-timeslot channel estimate + designDFE, every detected burst through equalizeBurst (sps == 1).
 * TRXSIG_TSCLEG_DEMOD: analyzeTrafficBurst + demodulateBurst and no channel cache, which is what
 * Transceiver52M/Transceiver.cpp:272, 322, 382 does while its mMaxExpectedDelay <= 1 ("needDFE" false); any sps.  The
 * sigProcLib calls are Transceiver/sigProcLib.cpp's in both (full-window analyzeTrafficBurst, energyDetect over 20*sps
 * consecutive samples) and so is the slot schedule (expectedCorrType :207-269). */
enum { TRXSIG_TSCLEG_EQUALIZE = 0, TRXSIG_TSCLEG_DEMOD = 1 };
int trxsig_trx_set_tsc_leg(trxsig_trx *t, int leg);
void trxsig_trx_destroy(trxsig_trx *t);
const char *trxsig_trx_last_error(const trxsig_trx *t);
trxsig_ctx *trxsig_trx_context(trxsig_trx *t);          /* the underlying library context */

/* driveControl (:439-580): one NUL-terminated "CMD ..." in, the "RSP ..." out (empty string where the
 * reference sends nothing).  Returns the response length, < 0 on a buffer that is too small.
 * RXTUNE / TXTUNE always tune successfully (there is no radio behind this object). */
int trxsig_trx_control(trxsig_trx *t, const char *command, char *response, int response_cap);

/* expectedCorrType (:207-269) */
int trxsig_trx_expected_corr_type(const trxsig_trx *t, int tn, int fn);

/* pullRadioVector for the burst that the receive FIFO would deliver: n complex samples at time (fn, tn).
 * Returns 1 and fills h_soft (n/sps values; *n_soft), *rssi, *timing_offset when a SoftVector comes back,
 * 0 when the reference returns NULL (slot off/idle, energy gate, no detection), < 0 on error. */
int trxsig_trx_pull_radio_vector(trxsig_trx *t, const trxsig_c32 *h_burst, int n, int tn, int fn, float *h_soft,
                                 int *n_soft, int *rssi, int *timing_offset);

/* driveReceiveFIFO's serialisation (:655-674): 158 bytes */
int trxsig_trx_encode_rx_datagram(int tn, int fn, int rssi, int timing_offset, const float *soft, int n_soft,
                                  uint8_t out[TRXSIG_RX_DATAGRAM_BYTES]);
/* driveTransmitPriorityQueue's parse (:585-632): 154 bytes in; bits as they arrive (one per byte).
 * Deviation: a frame number outside [0, gHyperframe = 2048*26*51) is rejected (TRXSIG_EINVAL) -- the reference
 * accepts any 32-bit value and later indexes its filler table with it.  The entry points below that take (fn, tn)
 * reject such an fn the same way. */
int trxsig_trx_decode_tx_datagram(const uint8_t *in, int len, int *tn, int *fn, int *rssi, uint8_t bits[148]);

/* addRadioVector (:100-113): modulate, scale, queue */
int trxsig_trx_add_radio_vector(trxsig_trx *t, const uint8_t bits[148], int rssi, int tn, int fn);
/* pushRadioVector (:138-181) for time (fn, tn): stale bursts go to the filler table, then either the queued
 * burst for exactly this time or the filler entry is what reaches the transmit FIFO; it is copied to h_out
 * (*n_out samples, <= 157*sps).  *from_queue = 1 when it came from the queue. */
int trxsig_trx_push_radio_vector(trxsig_trx *t, int tn, int fn, trxsig_c32 *h_out, int *n_out, int *from_queue);

/* introspection (tests, monitoring) */
double trxsig_trx_energy_threshold(const trxsig_trx *t);
int trxsig_trx_filler_modulus(const trxsig_trx *t, int tn);
int trxsig_trx_queue_size(const trxsig_trx *t);

/* ---- driveTransmitFIFO's deadline clock and latency controller (:679-729) and writeClockInterface (:733-746) as a pure state
 * machine (no device, no socket: a service loop feeds it the radio clock and moves its answers).  GSM::Time arithmetic as the
 * reference's class does it (GSM/GSMCommon.h:327-455: operator+(Time), incTN, decTN, the wrap-aware comparisons).
 *   mTransmitDeadlineClock = the burst that must be pushed into the transmit FIFO now; mTransmitLatency = how far ahead of the
 *   radio clock it runs (runTransceiver.cpp:53: GSM::Time(2,0)); after an under-run the latency grows by one frame (at most
 *   once per 10 frames), after 216 frames without one it shrinks by a timeslot while above GSM::Time(1,1). */
typedef struct {
  int deadline_fn, deadline_tn;        /* mTransmitDeadlineClock */
  int latency_fn, latency_tn;          /* mTransmitLatency */
  int latency_update_fn, latency_update_tn;   /* mLatencyUpdateTime */
  int last_clock_fn, last_clock_tn;    /* mLastClockUpdateTime */
} trxsig_txclock;
/* Transceiver::Transceiver (:50-57): all three clocks start at the start time */
void trxsig_txclock_init(trxsig_txclock *c, int start_fn, int start_tn, int latency_fn, int latency_tn);
/* One pass of driveTransmitFIFO's loop `while (radioClock + latency > deadline) { ...; pushRadioVector(deadline); deadline.incTN(); }`
 * with the radio clock at (radio_fn, radio_tn).  *underrun = RadioInterface::isUnderrun()'s flag, read and cleared where the
 * reference reads it (first iteration).  Returns how many timeslots to push, starting at (*push_fn, *push_tn) = the deadline
 * before the call; at most max_slots per call (call again for the rest; 0 = nothing due). */
int trxsig_txclock_advance(trxsig_txclock *c, int radio_fn, int radio_tn, int *underrun, int max_slots, int *push_fn, int *push_tn);
/* "periodically update GSM core clock" (:617-618): 1 when mTransmitDeadlineClock > mLastClockUpdateTime + GSM::Time(216,0) */
int trxsig_txclock_indication_due(const trxsig_txclock *c);
/* writeClockInterface (:733-746): "IND CLOCK <deadline FN + 20>" into msg (NUL-terminated; send strlen + 1 bytes), and
 * mLastClockUpdateTime = mTransmitDeadlineClock.  Returns the string length, < 0 when cap is too small. */
int trxsig_txclock_indication(trxsig_txclock *c, char *msg, int cap);

/* createLPF (sigProcLib.cpp:1102-1150): the reference loads one of two coefficient tables (651 receive /
 * 961 send; the cutoff argument is ignored) and normalises it to gainDC / sum(taps), the sum in double.
 * The tables are reference data and are passed in by the caller. */
int trxsig_create_lpf_host(const float *raw_taps, int len, float gain_dc, float *out);

#ifdef __cplusplus
}
#endif
#endif
