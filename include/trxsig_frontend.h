/* trxsig_frontend.h -- the RadioInterface's sample plumbing on the device, in C (C-ABI of libtrxsig):
 *   trxsig_rxfe   RadioInterface::pullBuffer + driveReceiveRadio (Transceiver/radioInterface.cpp:197-273, 359-401) for S
 *                 independent ARFCN streams on one GPU: int16 I/Q chunks of OUTCHUNK = 864 samples at 400 kS/s ->
 *                 unUSRPifyVector (I/Q swapped unless told otherwise, :91-116) -> [192-sample history | chunk] ->
 *                 polyphaseResampleVector(P = 65*sps, Q = 96, rcvLPF) (:244-246) with the first INHISTORY = 130*sps
 *                 outputs dropped (:249-252) -> the stream's receive buffer -> bursts of (156 + (TN % 4 == 0)) * sps samples
 *                 (157-156-156-156, :370-394) handed out as offsets into that buffer, ready for the batch detectors of
 *                 trxsig.h.  One fused kernel per push (conversion, filter and placement; window and taps staged in LDS).
 *   trxsig_txbe   RadioInterface::driveTransmitRadio + pushBuffer (:123-194, 337-357): modulated bursts are appended to
 *                 the stream's send buffer; whenever it holds whole chunks of INCHUNK = 585*sps samples,
 *                 [INHISTORY history | chunks] -> polyphaseResampleVector(P = 96, Q = 65*sps, sendLPF) (:141-144) ->
 *                 scaleVector(gain = 13500) (:148) -> USRPifyVector (:74-89), first OUTHISTORY = 192 outputs dropped (:165).
 *                 One kernel per pop that modulates from the queued bits, filters, scales and packs int16 (trxsig_txbe_set_fused).
 * All buffers are device memory owned by the object (per-stream linear buffers; no reallocation per chunk); sample values
 * are the reference's bit for bit (tests/test_gpu_config4.py, tests/test_gpu_txchain.py).  The USRP itself (the source /
 * sink of the int16 samples) and the GSM clock are the caller's (component 5 is out of scope).
 * Every function returns TRXSIG_OK (0) or a negative TRXSIG_E* code; trxsig_last_error(ctx) has the text. */
#ifndef TRXSIG_FRONTEND_H
#define TRXSIG_FRONTEND_H

#include "trxsig.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TRXSIG_OUTRATE 96                      /* radioInterface.h:37 */
#define TRXSIG_OUTCHUNK (9 * TRXSIG_OUTRATE)   /* :41  864 */
#define TRXSIG_OUTHISTORY (2 * TRXSIG_OUTRATE) /* :39  192 */

typedef struct trxsig_rxfe trxsig_rxfe;
typedef struct trxsig_txbe trxsig_txbe;

/* h_lpf: the L (normally 961, createLPF(cutoff, 961, 65*sps): radioInterface.cpp:230-234) normalised taps, host memory.
 * max_chunks: the most chunks one push may carry (sizes the buffers).  start_tn: TN of the first burst cut.
 * Burst offsets are 32-bit: n_streams * (157*sps + max_chunks * 585*sps) samples must stay below 2^31 (TRXSIG_EINVAL
 * otherwise -- use several front ends). */
int trxsig_rxfe_create(trxsig_rxfe **out, trxsig_ctx *ctx, int n_streams, int max_chunks, const float *h_lpf, int L,
                       int swap_iq, int start_tn);
void trxsig_rxfe_destroy(trxsig_rxfe *fe);
/* d_iq: int16 I/Q pairs, [n_streams][n_chunks * 864][2] (device).  Every chunk is filtered behind the 192 samples
 * that precede it in its stream (the previous push's tail for the first one), exactly as pullBuffer does chunk by chunk.
 * Bursts handed out by an earlier trxsig_rxfe_pop stay valid until this call. */
int trxsig_rxfe_push(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks);
/* Cuts every stream's buffer into as many bursts as it holds ("while (rcvSz > burst size)", :375): *n_bursts per stream,
 * burst j of stream s is entry s * *n_bursts + j of *d_offset / *d_length (samples, into *d_samples).  h_tn[j] (cap_tn
 * entries at least *n_bursts; may be NULL) = the burst's TN (the same schedule on every stream).  The pointers belong to
 * the object and stay valid until the next push. */
int trxsig_rxfe_pop(trxsig_rxfe *fe, const trxsig_c32 **d_samples, const int32_t **d_offset, const int32_t **d_length,
                    int32_t *h_tn, int cap_tn, int *n_bursts);
int trxsig_rxfe_pending(const trxsig_rxfe *fe);   /* samples per stream not yet cut into bursts */
/* trxsig_rxfe_push + trxsig_rxfe_pop + trxsig_detect_demod_normal_batch in one call, with the resampled stream never written to
 * memory: the detect and demodulate kernels compute the samples of their bursts from the int16 chunks (four multiply-adds
 * each; the unfused chain writes 300 MB of complex float32 per 60 K bursts and reads it back 1.4 times).  Same results bit
 * for bit.  Needs sps == 4, a filter of at most 4*260 taps and nsoft <= 148; a front end is used either through this call or
 * through push / pop, not both.  *n_bursts per stream are completed by this push; burst j of stream s is entry
 * s * *n_bursts + j of every output array.  cap_tn = the bursts per stream the output arrays (and h_tn, if given) have room
 * for, i.e. they hold n_streams * cap_tn entries; (628 + n_chunks * 2340) / 624 always suffices.  A push that completes more
 * is refused with TRXSIG_EINVAL before anything is launched or the front end's state changes.  d_iq is read by kernels on the
 * context's stream: keep it unchanged until they have run. */
int trxsig_rxfe_push_detect_demod_normal(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, int tsc, float detect_thresh,
                                         float energy_thresh, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr,
                                         float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride, int32_t *h_tn, int cap_tn,
                                         int *n_bursts);

/* ---- multi-ARFCN channeliser (no such component exists in the reference, which runs one radio and one
 * Transceiver per ARFCN) ----------------------------------------------------------------------------------------------------
 * One WIDEBAND int16 stream at rate_factor x 400 kS/s carries n_carriers ARFCNs.  For every carrier the stream is mixed to
 * baseband with the reference's frequencyShift (sigProcLib.h:149-153: z[n] = x[n] * expjLookup(phase), table trig) and brought
 * to sps samples per symbol by its polyphaseResampleVector(P = 65*sps, Q = 96*rate_factor, h_lpf) behind a history of
 * 192*rate_factor samples, exactly as RadioInterface::pullBuffer does for one carrier (radioInterface.cpp:230-259) -- fused in
 * one kernel: the wideband window is staged once per (carrier, tile), mixed on the way into LDS, filtered from there.  The
 * result lands in the receive buffers of an ordinary front end with n_wide_streams * n_carriers streams (stream w*n_carriers
 * + k = carrier k of wideband stream w): trxsig_rxfe_pop and the batch detectors / the Transceiver group take it from there.
 * Mixer phase: the reference's frequencyShift forms the phase of sample n as a running float sum (`phase += freq` n times),
 * which is sequential over the whole stream and loses a ulp of an ever larger number at every step (after one 2.16 ms chunk
 * at 3.2 MS/s the sum is ~2,700 rad and has drifted by milliradians).  The channeliser keeps frequencyShift's arithmetic --
 * z[n] = x[n] * expjLookup(phase[n]) with the reference's table trig -- and forms the phase of raw sample n (counted from the
 * stream's first sample) directly: phase[n] = (float)(t - 2 pi floor(t / 2 pi)), t = (double) n * (double) freq, every step an
 * IEEE operation; that is frequencyShift called on the one-sample vector {x[n]} with startPhase = phase[n].  With that
 * convention the output equals the reference's two primitives applied per carrier bit for bit (tests/test_gpu_channeliser.py,
 * which also measures the distance to the running-sum form); parity is pinned on the reference's primitives, there is no
 * reference component to compare the whole with.
 * h_carrier_freq[k]: radians per wideband sample (negative of the carrier's offset from the stream's centre, to bring it to 0).
 * h_lpf: L taps of a low-pass at the P-times-interpolated rate (DC gain P), e.g. a Kaiser design with its cutoff at the
 * 200 kHz channel edge. */
int trxsig_rxfe_create_wideband(trxsig_rxfe **out, trxsig_ctx *ctx, int n_wide_streams, int n_carriers, const float *h_carrier_freq,
                                int rate_factor, int max_chunks, const float *h_lpf, int L, int swap_iq, int start_tn);
/* d_iq: int16 I/Q pairs [n_wide_streams][n_chunks * 864 * rate_factor][2] (device): n_chunks chunks of 2.16 ms each */
int trxsig_rxfe_push_wideband(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks);
/* The SHARED-FILTER form of the channeliser (round 4; off by default -- the per-carrier form above is the one pinned on the
 * reference's primitives).  When every carrier lies on the grid of sixteenths of the wideband rate (theta_c = 2 pi k / 16 rad per
 * sample: with rate_factor 8 that is every 200 kHz) the mixer has period 16 and the C per-carrier filters collapse into ONE pass
 * over the raw samples: sixteen partial sums of real taps on raw samples per output instant, shared by all carriers, then sixteen
 * complex multiply-adds per carrier from registers (csrc/trxsig_chan.hip).  The same sum in another order, with exact cos / sin
 * instead of the reference's table trig and fused multiply-adds: equal to the per-carrier form to ~1e-6 of the signal's scale
 * (tests/test_gpu_channeliser.py grades it at 1e-4, identical hard bits), 5-10x faster.  Needs 1, 2, 4, 8 or 16 carriers on that
 * grid and at most 32 taps per output (L <= 32 P); TRXSIG_EINVAL otherwise.  on = 0 returns to the per-carrier form. */
int trxsig_rxfe_set_shared_filter(trxsig_rxfe *fe, int on);

/* h_lpf: the L (normally 651, createLPF(cutoff, 651, 96): radioInterface.cpp:134-138) normalised taps.  max_bursts: the
 * most bursts per stream one push may carry. */
int trxsig_txbe_create(trxsig_txbe **out, trxsig_ctx *ctx, int n_streams, int max_bursts, const float *h_lpf, int L, float gain);
/* How the back end works inside (before the first push; results are the same int16 values either way).  1, the default:
 * FUSED -- a pushed burst leaves only its 148 bits and its gain in a ring; trxsig_txbe_pop runs ONE kernel that computes the
 * modulated samples its filter taps meet from those bits (modulateBurst's three non-zero terms, addRadioVector's scaling),
 * resamples, applies the 13500 gain and packs int16: the complex float32 send buffer of radioInterface.cpp:123-194 never
 * exists in memory.  0: modulate into a send buffer at push, resample it at pop (two kernels, 5 KB per burst written and
 * read back). */
int trxsig_txbe_set_fused(trxsig_txbe *be, int fused);
void trxsig_txbe_destroy(trxsig_txbe *be);
/* modulateBurst (+ addRadioVector's power scaling when d_gain != NULL) of n_bursts bursts per stream, appended to the send
 * buffers: d_bits [n_streams][n_bursts][148] (one bit per byte), h_guard[n_bursts] guard symbols per burst (host; the same
 * schedule on every stream: 8 + (TN % 4 == 0), Transceiver.cpp:105), d_gain [n_streams][n_bursts] or NULL. */
int trxsig_txbe_push_bursts(trxsig_txbe *be, const uint8_t *d_bits, const int32_t *h_guard, const float *d_gain, int n_bursts);
/* Would trxsig_txbe_push_bursts accept this push now?  TRXSIG_OK, or the error that call would return (text in
 * trxsig_last_error) -- nothing is changed either way.  For a caller that must not have consumed its bursts when the push is
 * refused (trxsig_trxgroup_push_txbe).  trxsig_txbe_streams: the n_streams the back end was created with. */
int trxsig_txbe_can_push(trxsig_txbe *be, const int32_t *h_guard, int n_bursts);
int trxsig_txbe_streams(const trxsig_txbe *be);
/* pushBuffer for every stream: *n_samples int16 I/Q pairs per stream at *d_iq + s * *stream_stride pairs (device; valid
 * until the next pop), 0 while less than one chunk is buffered. */
int trxsig_txbe_pop(trxsig_txbe *be, const int16_t **d_iq, int64_t *stream_stride, int *n_samples);
int trxsig_txbe_pending(const trxsig_txbe *be);   /* modulated samples per stream waiting for a whole chunk */

#ifdef __cplusplus
}
#endif
#endif /* TRXSIG_FRONTEND_H */
