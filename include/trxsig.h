/*
 * trxsig.h -- C-ABI of libtrxsig: the MI355X (gfx950) burst-processing path of the
 * OpenBTS software transceiver.
 *
 * This is the drop-in boundary for the reference's `sigProcLib.h` seam
 * (Transceiver/sigProcLib.h:101-384) as it is driven by
 * Transceiver::pullRadioVector / addRadioVector (Transceiver/Transceiver.cpp:271-410,
 * 100-113) and RadioInterface::pullBuffer / pushBuffer (Transceiver/radioInterface.cpp:
 * 197-273, 123-194).  Plain pointers and sizes only; no C++ or torch types.  Every entry
 * point names the reference interface it replaces.
 *
 * Conventions
 *  - complex samples are interleaved float32 {re, im} (the layout of the reference's
 *    Complex<float>, Transceiver/Complex.h:39-44, inside Vector<complex>).
 *  - "d_" pointers are DEVICE pointers on the context's GPU; "h_" pointers are host.
 *  - a batch is B bursts packed in one sample array: burst b occupies
 *    samples[offset[b] .. offset[b]+length[b]).  offset[b] >= 0 (even offsets, i.e. 16-byte
 *    aligned bursts, take the wide-load path); length[b] a multiple of sps with
 *    92*sps <= length[b] <= 157*sps (the reference's own limits: sigProcLib.cpp:215-216,
 *    951, 1045-1050).
 *  - all work is enqueued on the context's HIP stream (trxsig_set_stream) and is
 *    asynchronous; the library never synchronises unless the entry point's comment says so.
 *  - return value: 0 on success, negative TRXSIG_E* on error; no exceptions cross the ABI.
 *  - results are bit-identical to the reference's float32 arithmetic (see DESIGN.md,
 *    "Numerical contract"), except where an entry point's comment states a tolerance.
 *  - There is NO CPU fallback: without a gfx950 device trxsig_create fails.
 */
#ifndef TRXSIG_H
#define TRXSIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: TRXSIG_K_COUNT grew (trxsig_profile_collect writes TRXSIG_K_COUNT entries: a host built against an older header must
 * not be run against this library -- compare trxsig_abi_version() with TRXSIG_ABI_VERSION at start-up, or use
 * trxsig_profile_collect_n, which takes the caller's capacity). */
#define TRXSIG_ABI_VERSION 2

typedef struct trxsig_ctx trxsig_ctx;
typedef struct { float re, im; } trxsig_c32;

enum {
  TRXSIG_OK = 0,
  TRXSIG_EINVAL = -1,   /* bad argument */
  TRXSIG_ENODEV = -2,   /* no usable gfx950 device / HIP runtime error at create */
  TRXSIG_EHIP = -3,     /* HIP runtime error (see trxsig_last_error) */
  TRXSIG_ENOMEM = -4
};

/* per-burst status byte written by the detect entry points */
enum {
  TRXSIG_F_ENERGY = 1,   /* energyDetect() passed           (sigProcLib.cpp:916-932)      */
  TRXSIG_F_DETECT = 2,   /* correlator peak/valley test passed (sigProcLib.cpp:913,1035)   */
  TRXSIG_F_BADLEN = 128  /* length[b]/offset[b] violates the conventions above; burst skipped */
};

/* ---- library set-up: sigProcLibSetup / sigProcLibDestroy (sigProcLib.h:110-113) -------------
 * trxsig_create = generateGSMPulse(2,sps) + sigProcLibSetup(sps) + generateRACHSequence +
 * generateMidamble(0..7) (the calls of Transceiver.cpp:62-64, 424, 553), built on the host and
 * uploaded once.  device = HIP device ordinal.  sps in {1,2,4}.
 * trxsig_destroy on a context that front ends, back ends or Transceiver groups (trxsig_frontend.h, trxsig_trxgroup.h) still live
 * on takes effect when the last of them has been destroyed: such an object never outlives the context it was created on. */
int  trxsig_abi_version(void);
int  trxsig_create(trxsig_ctx **out, int device, int sps);
void trxsig_destroy(trxsig_ctx *ctx);
int  trxsig_sps(const trxsig_ctx *ctx);
int  trxsig_device(const trxsig_ctx *ctx);
/* front ends, back ends and Transceiver groups currently alive on the context (each keeps it alive past trxsig_destroy);
 * a create call that fails leaves this count where it was */
int  trxsig_live_children(const trxsig_ctx *ctx);
/* stream = hipStream_t (NULL = the device's null stream).  One context per calling thread. */
int  trxsig_set_stream(trxsig_ctx *ctx, void *hip_stream);
/* hipStreamSynchronize on the context's stream */
int  trxsig_synchronize(trxsig_ctx *ctx);
void *trxsig_get_stream(trxsig_ctx *ctx);      /* the hipStream_t the context enqueues on */
int  trxsig_get_device(trxsig_ctx *ctx);
const char *trxsig_last_error(const trxsig_ctx *ctx);
/* pre-size the internal device workspace for batches up to max_bursts (allocates; call outside
 * any timed or graph-captured region).  The batch entry points grow it on demand otherwise. */
int  trxsig_reserve(trxsig_ctx *ctx, int max_bursts);

/* ---- constant tables ------------------------------------------------------------------------
 * The constant tables (trig lookup, GMSK rotation, pulse, 8 midambles, RACH sequence, sinc grid)
 * live in ONE device blob so that a multi-GPU job can build them on rank 0 and broadcast them
 * (RCCL ncclBroadcast over xGMI) instead of rebuilding per rank: SURVEY 8e.
 *   trxsig_tables_bytes   size of the blob
 *   trxsig_tables_device  device pointer of this context's blob
 *   trxsig_create_from_tables  build a context on `device` from a blob that is already in that
 *                         device's memory (d_blob); copies it, validates header + checksum.
 *   trxsig_tables_export  copy the blob to host memory (synchronises)
 *   trxsig_tables_build_host  build the blob into host memory without touching any device
 *                         (what rank 0 uploads; also lets the table construction be tested on CPU) */
size_t trxsig_tables_bytes(int sps);
int    trxsig_tables_build_host(int sps, void *h_buf, size_t cap);
void  *trxsig_tables_device(trxsig_ctx *ctx);
int    trxsig_create_from_tables(trxsig_ctx **out, int device, const void *d_blob, size_t bytes);
int    trxsig_tables_export(trxsig_ctx *ctx, void *h_buf, size_t cap);

/* host-side view of the tables (for the C++ facade and tests); pointers valid for ctx lifetime */
typedef struct {
  int sps;
  const float *cos_table;            /* 1025: cosTable  (sigProcLib.cpp:39,207-212) */
  const float *sin_table;            /* 1025: sinTable                              */
  const trxsig_c32 *gmsk_rotation;   /* 157*sps: GMSKRotation (sigProcLib.cpp:214-225) */
  const trxsig_c32 *gmsk_reverse;    /* 157*sps: GMSKReverseRotation                   */
  const float *gsm_pulse;            /* 2*sps+1: generateGSMPulse(2,sps) (sigProcLib.cpp:411-430) */
  const trxsig_c32 *midamble[8];     /* 16*sps each: gMidambles[t]->sequence (sigProcLib.cpp:779-828) */
  float midamble_toa[8];
  trxsig_c32 midamble_gain[8];
  const trxsig_c32 *rach;            /* 41*sps: gRACHSequence->sequence (sigProcLib.cpp:830-857) */
  float rach_toa;
  trxsig_c32 rach_gain;
} trxsig_tables_view;
int trxsig_tables_view_get(const trxsig_ctx *ctx, trxsig_tables_view *out);

/* ---- RX hot path ----------------------------------------------------------------------------
 * trxsig_detect_demod_normal_batch
 *   For every burst: energyDetect(burst, 20*sps, energy_thresh) -> analyzeTrafficBurst(burst, tsc,
 *   detect_thresh, sps, &amp, &TOA) -> on detection demodulateBurst(burst, pulse, sps, amp, TOA):
 *   the TSC leg of Transceiver::pullRadioVector (Transceiver.cpp:298, 327-335, 385-388 /
 *   Transceiver52M/Transceiver.cpp:382-389) with the sequential threshold state machine left to
 *   the caller (SURVEY 8a' item 14).  Replaces sigProcLib.h:246-249, 277-285, 316-320.
 *
 *   d_flags[b]   status byte (TRXSIG_F_*)
 *   d_amp[b]     amplitude estimate (0 when the reference returns "bogus result")
 *   d_toa[b]     time of arrival in samples
 *   d_avgpwr[b]  energyDetect's avgPwr                                   (may be NULL)
 *   d_soft       B x soft_stride floats; the first nsoft (<= min(157, length/sps)) soft bits of
 *                demodulateBurst for detected bursts, zeros otherwise
 *   d_hard       B x soft_stride bytes, SoftVector::bit() = soft > 0.5F (BitVector.h:415-420)
 *                                                                        (may be NULL)
 *   Bursts whose energy test fails are reported undetected (amp = 0, TOA = 0), as the reference
 *   does not run the correlator for them (Transceiver.cpp:298-306).  energy_thresh < 0 disables
 *   the energy gate (every burst is analysed; TRXSIG_F_ENERGY is set). */
int trxsig_detect_demod_normal_batch(trxsig_ctx *ctx,
                                     const trxsig_c32 *d_samples, const int32_t *d_offset,
                                     const int32_t *d_length, int B,
                                     int tsc, float detect_thresh, float energy_thresh,
                                     uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                     float *d_avgpwr,
                                     float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride);

/* trxsig_detect_demod_rach_batch: same for access bursts: detectRACHBurst(burst, detect_thresh,
 *   sps, &amp, &TOA) + demodulateBurst (Transceiver.cpp:362-366, 385-388; sigProcLib.h:263-267).
 *   The 41*sps-tap correlation is evaluated over every lag exactly as the reference does
 *   (same terms, same order), so flags/amp/TOA/soft/hard are all bit-identical. */
int trxsig_detect_demod_rach_batch(trxsig_ctx *ctx,
                                   const trxsig_c32 *d_samples, const int32_t *d_offset,
                                   const int32_t *d_length, int B,
                                   float detect_thresh, float energy_thresh,
                                   uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                   float *d_avgpwr,
                                   float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride);


/* trxsig_demodulate_batch: demodulateBurst alone with caller-supplied amp/TOA
 *   (sigProcLib.h:316-320; used for RACH after detect and by TRANSMIT_LOGGING,
 *   Transceiver.cpp:115-136).  d_enable[b]==0 skips burst b (zeros out); NULL = all. */
int trxsig_demodulate_batch(trxsig_ctx *ctx,
                            const trxsig_c32 *d_samples, const int32_t *d_offset,
                            const int32_t *d_length, int B,
                            const trxsig_c32 *d_amp, const float *d_toa, const uint8_t *d_enable,
                            float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride);

/* trxsig_set_soft_mode: how demodulateBurst's soft bits (sigProcLib.cpp:1056-1097) are computed behind the three calls above
 *   (and behind the Transceiver group's demodulating leg on a complex float32 stream).  Flags, amplitude, TOA and avgPwr are the
 *   reference's values bit for bit in either mode, and so is every HARD bit (SoftVector::bit, soft > 0.5F).
 *   TRXSIG_SOFT_EXACT (the default): every soft bit equals the reference's float32 value (IEEE ==): scaleVector, delayVector's
 *     21 taps, the reverse rotation and the slicer operation by operation, in the reference's order.
 *   TRXSIG_SOFT_TOLERANCE: the accuracy the reference's users are promised elsewhere ("within 1e-4 on soft symbols") spent on
 *     speed -- 1/amp is applied to the 148 outputs instead of the 625 samples and the delay filter accumulates with fused
 *     multiply-adds (csrc/trxsig_demod.h, fused_demod_tol: ~200 instead of ~530 VALU instructions per burst).  Guaranteed
 *     |soft - reference soft| <= 7.4e-5 on the [0, 1] scale of a soft bit (derivation there; measured <= 1.5e-6); a burst for
 *     which the guarantee cannot be given (a soft symbol too close to the slicer's 0.5 for the hard bit to be certain, a NaN or
 *     infinity anywhere, max|sample| * |1/amp| > 8, a TOA off peakDetect's 1/512 grid, an odd burst geometry) is computed by
 *     the exact code inside the same launch and comes out IEEE-equal.  nsoft > 148 always takes the exact code.
 *   Takes effect from the next call on (host-side switch, no synchronisation). */
enum { TRXSIG_SOFT_EXACT = 0, TRXSIG_SOFT_TOLERANCE = 1 };
int trxsig_set_soft_mode(trxsig_ctx *ctx, int mode);
int trxsig_get_soft_mode(const trxsig_ctx *ctx);

/* ---- TX path: modulateBurst (sigProcLib.h:171-174) as called by Transceiver::addRadioVector
 *   (Transceiver.cpp:100-113): bits -> GMSK-rotated impulses -> pulse shaping, then scaleVector by
 *   a real gain.  d_bits: B x 148 bytes (only bit 0 is used, BitVector.cpp:54-63); d_guard[b] =
 *   guard period in symbols (8 or 9); output burst b at d_out[d_out_offset[b]..] with
 *   sps*(148+guard) samples; d_gain may be NULL (no scaling pass, as generateMidamble's use). */
int trxsig_modulate_batch(trxsig_ctx *ctx, const uint8_t *d_bits, const int32_t *d_guard,
                          const float *d_gain, int B,
                          trxsig_c32 *d_out, const int32_t *d_out_offset);

/* The two halves of trxsig_equalize_normal_batch (declared further down) on their own, for callers that keep the reference's per-timeslot cache
 * (Transceiver.cpp:317-349: channel estimate + DFE design only on the first burst of a slot or after 50
 * frames; every burst is then equalised with the cached taps):
 *   trxsig_estimate_dfe_batch: analyzeTrafficBurst(requestChannel) + scaleVector(chan, 1/amp) + designDFE, no
 *     energy gate; snr_thresh = the threshold in SNR = |amp|^2/(thr^2+1) (the reference uses mEnergyThreshold
 *     after its "-= 1" update, :338-340), or snr_value > 0 = the SNR estimate itself for every burst (a caller
 *     that forms it in the reference's double arithmetic).  d_chan_off = chanRespOffset, d_w: B x 7, d_b: B x 5.
 *   trxsig_equalize_taps_batch: scaleVector(burst, 1/amp) + equalizeBurst(burst, toa_eq, w, b) with the taps of
 *     burst b at d_w + 7b, d_b + 5b; d_enable[b] & TRXSIG_F_DETECT selects the bursts to process (zeros otherwise). */
int trxsig_estimate_dfe_batch(trxsig_ctx *ctx, const trxsig_c32 *d_samples, const int32_t *d_offset,
                              const int32_t *d_length, int B, int tsc, float detect_thresh, float snr_thresh,
                              float snr_value, int variant52m, int max_toa, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                              float *d_chan_off, trxsig_c32 *d_w, trxsig_c32 *d_b);
int trxsig_equalize_taps_batch(trxsig_ctx *ctx, const trxsig_c32 *d_samples, const int32_t *d_offset,
                               const int32_t *d_length, int B, const trxsig_c32 *d_amp, const float *d_toa_eq,
                               const uint8_t *d_enable, const trxsig_c32 *d_w, const trxsig_c32 *d_b,
                               float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride);
/* The same steps as the reference's free functions hand them to each other (sigProcLib.h:277-285, 372-384), for the
 * source-compatible facade:
 *   trxsig_channel_estimate_batch: analyzeTrafficBurst(..., requestChannel = true): flags/amp/TOA as the detect calls
 *     (no energy gate), d_chan_off = channelResponseOffset, d_chan = channelResponse (B x 6 taps, already divided by
 *     the midamble gain, sigProcLib.cpp:1024-1025; zeros for bursts that were not detected).
 *   trxsig_design_dfe_batch: designDFE(channelResponse, SNRestimate, Nf = 7, &w, &b) (sigProcLib.cpp:1246-1340) for B
 *     channel estimates; d_amp != NULL applies scaleVector(channelResponse, 1/amp) first (Transceiver.cpp:346),
 *     NULL = the caller has done that.  d_w: B x 7, d_b: B x 5. */
int trxsig_channel_estimate_batch(trxsig_ctx *ctx, const trxsig_c32 *d_samples, const int32_t *d_offset,
                                  const int32_t *d_length, int B, int tsc, float detect_thresh, int variant52m, int max_toa,
                                  uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_chan_off, trxsig_c32 *d_chan);
int trxsig_design_dfe_batch(trxsig_ctx *ctx, const trxsig_c32 *d_chan, const trxsig_c32 *d_amp, const float *d_snr, int B,
                            trxsig_c32 *d_w, trxsig_c32 *d_b);
/* single-burst host-buffer forms (one PCIe round trip each; what include/sigProcLib_trx.h calls):
 * trxsig_equalize_taps_host = scaleVector(burst, 1/amp) + equalizeBurst(burst, toa_eq, 1, w, b) (sigProcLib.h:372-384);
 * pass amp = {1, 0} for a burst that is scaled already. */
int trxsig_channel_estimate_host(trxsig_ctx *ctx, const trxsig_c32 *h_samples, int n, int tsc, float detect_thresh,
                                 int variant52m, int max_toa, uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa,
                                 float *h_chan_off, trxsig_c32 h_chan[6]);
int trxsig_design_dfe_host(trxsig_ctx *ctx, const trxsig_c32 h_chan[6], float snr, trxsig_c32 h_w[7], trxsig_c32 h_b[5]);
int trxsig_equalize_taps_host(trxsig_ctx *ctx, const trxsig_c32 *h_samples, int n, trxsig_c32 amp, float toa_eq,
                              const trxsig_c32 h_w[7], const trxsig_c32 h_b[5], float *h_soft, int nsoft);

/* ---- rate conversion: polyphaseResampleVector (sigProcLib.h:352-354) -------------------------
 * S independent streams (one per ARFCN).  Stream s: input d_in + s*in_stride (n_in samples),
 * output d_out + s*out_stride, ceil(n_in*P/Q) samples each, exactly the reference's indexing
 * (sigProcLib.cpp:1171-1205).  d_lpf: L real taps on the device (createLPF output). */
int trxsig_resample_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, int n_in, int64_t in_stride,
                          int S, int P, int Q, const float *d_lpf, int L,
                          trxsig_c32 *d_out, int64_t out_stride);
int trxsig_resample_out_len(int n_in, int P, int Q);
/* one stream, host buffers (one PCIe round trip): returns the number of output samples, < 0 on error */
int trxsig_resample_host(trxsig_ctx *ctx, const trxsig_c32 *h_in, int n_in, int P, int Q, const float *h_lpf, int L,
                         trxsig_c32 *h_out, int out_capacity);

/* int16 I/Q <-> float: RadioInterface::unUSRPifyVector / USRPifyVector
 *   (radioInterface.cpp:74-116).  swap_iq = 1 reproduces the non-SWLOOPBACK I/Q flip on RX. */
int trxsig_unpack_int16(trxsig_ctx *ctx, const int16_t *d_iq, int64_t n_samples, int swap_iq,
                        trxsig_c32 *d_out);
int trxsig_pack_int16(trxsig_ctx *ctx, const trxsig_c32 *d_in, int64_t n_samples, int16_t *d_iq);
/* scaleVector(x, gain) + USRPifyVector in one pass: the tail of RadioInterface::pushBuffer
 *   (radioInterface.cpp:149-152; the reference uses gain = 13500.0). */
int trxsig_pack_int16_scaled(trxsig_ctx *ctx, const trxsig_c32 *d_in, int64_t n_samples, float gain, int16_t *d_iq);
/* IEEE binary16 I/Q pairs (the sample storage format of BASELINE config 5) -> float.  The reference has
 * no such format (its radio side is int16, radioInterface.cpp:74-116); widening is exact, so results
 * downstream equal the float pipeline's on the same values.  d_iq: 2*n_samples binary16 values. */
int trxsig_unpack_half(trxsig_ctx *ctx, const uint16_t *d_iq, int64_t n_samples, trxsig_c32 *d_out);

/* ---- equaliser: analyzeTrafficBurst(requestChannel) + designDFE + equalizeBurst ---------------
 *   (sigProcLib.h:277-285, 366-370, 382-386; Transceiver.cpp:327-349, 391-396; the windowed
 *   Transceiver52M form with maxTOA when variant52m != 0: Transceiver52M/sigProcLib.cpp:966-1076).
 *   sps must be 1 for equalizeBurst ("Assumes symbol-rate sampling", sigProcLib.cpp:1342).
 *   Per burst: detect + channel estimate; scale channel by 1/amp; SNR = |amp|^2/(thr^2+1);
 *   designDFE(chan, SNR, 7); equalizeBurst(burst/amp, TOA-chanOffset, w, b).  thr is the
 *   energy_thresh argument (the reference uses its adaptive mEnergyThreshold, whose sequential
 *   update stays with the caller: SURVEY 8a' item 14); energy_thresh < 0 disables the gate and
 *   uses thr = 0.  d_w: B x 7, d_b: B x 5 complex, written for detected bursts (may be NULL).
 *   max_toa (52M variant only): 0..17. */
int trxsig_equalize_normal_batch(trxsig_ctx *ctx,
                                 const trxsig_c32 *d_samples, const int32_t *d_offset,
                                 const int32_t *d_length, int B,
                                 int tsc, float detect_thresh, float energy_thresh,
                                 int variant52m, int max_toa,
                                 uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                 trxsig_c32 *d_w, trxsig_c32 *d_b,
                                 float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride);
/* The same with the bursts stored as fp16 I/Q pairs (BASELINE config 5): sample_format TRXSIG_SAMPLES_F16 = d_samples is
 * an array of half-precision {re, im} pairs (4 bytes per sample; d_offset / d_length still count samples),
 * TRXSIG_SAMPLES_C32 = complex float32 as above.  The kernels read the fp16 words themselves and widen in registers (exact),
 * so the results equal the float32 call on the same values bit for bit -- there is no float32 copy of the batch in HBM. */
enum { TRXSIG_SAMPLES_C32 = 0, TRXSIG_SAMPLES_F16 = 1 };
int trxsig_equalize_normal_batch_fmt(trxsig_ctx *ctx, const void *d_samples, int sample_format, const int32_t *d_offset,
                                     const int32_t *d_length, int B, int tsc, float detect_thresh, float energy_thresh,
                                     int variant52m, int max_toa, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                                     trxsig_c32 *d_w, trxsig_c32 *d_b, float *d_soft, uint8_t *d_hard, int nsoft,
                                     int soft_stride);
/* trxsig_equalize_taps_batch (cached DFE taps, Transceiver.cpp:391-396) on either sample storage */
int trxsig_equalize_taps_batch_fmt(trxsig_ctx *ctx, const void *d_samples, int sample_format, const int32_t *d_offset,
                                   const int32_t *d_length, int B, const trxsig_c32 *d_amp, const float *d_toa_eq,
                                   const uint8_t *d_enable, const trxsig_c32 *d_w, const trxsig_c32 *d_b, float *d_soft,
                                   uint8_t *d_hard, int nsoft, int soft_stride);

/* ---- convenience: host-buffer single-call wrappers (copy in, run, copy out, synchronise).
 *   These exist so Transceiver::pullRadioVector can keep calling one burst at a time; they are
 *   PCIe-inclusive and are never what bench.py times. */
int trxsig_detect_demod_normal_host(trxsig_ctx *ctx, const trxsig_c32 *h_samples,
                                    const int32_t *h_offset, const int32_t *h_length, int B,
                                    int tsc, float detect_thresh, float energy_thresh,
                                    uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa,
                                    float *h_avgpwr, float *h_soft, int nsoft, int soft_stride);
int trxsig_detect_demod_rach_host(trxsig_ctx *ctx, const trxsig_c32 *h_samples,
                                  const int32_t *h_offset, const int32_t *h_length, int B,
                                  float detect_thresh, float energy_thresh,
                                  uint8_t *h_flags, trxsig_c32 *h_amp, float *h_toa,
                                  float *h_avgpwr, float *h_soft, int nsoft, int soft_stride);
/* one burst, caller-supplied amp/TOA: demodulateBurst (sigProcLib.h:316-320).  TRXSIG_EINVAL for a burst outside the
 * accepted geometry (92..157 symbols, a multiple of sps samples) or |TOA| > 4096 / NaN -- the batch form writes zeros for
 * such a burst, the one-burst form refuses it (the reference itself handles any length). */
int trxsig_demodulate_host(trxsig_ctx *ctx, const trxsig_c32 *h_samples, int n_samples,
                           trxsig_c32 amp, float toa, float *h_soft, int nsoft);
int trxsig_modulate_host(trxsig_ctx *ctx, const uint8_t *h_bits, const int32_t *h_guard,
                         const float *h_gain, int B, trxsig_c32 *h_out, const int32_t *h_out_offset,
                         int64_t out_samples);

/* ---- L1 FEC soft decode: the consumer of the soft bits (next row after the burst path) ---------
 * SoftVector::decode with ViterbiR2O4 (rate 1/2, order 4, deferral 24; CommonLibs/BitVector.cpp:290-524),
 * the Parity shift registers (CommonLibs/BitVector.h:39-112) and the decoder flows of GSM/GSML1FEC.cpp.
 * d_soft is the soft-bit array the detect/demod calls produce: burst b at d_soft + b*soft_stride, 148
 * values in [0,1].  wire_quantise != 0 applies what the UDP hop does to them in the reference chain --
 * byte = (char)round(v*255.0) (Transceiver.cpp:669), v' = byte/256.0F (TRXManager.cpp:231) -- so the
 * result is what GSM::XCCHL1Decoder / RACHL1Decoder would have produced behind TRXManager.
 *
 * XCCH (SACCH/SDCCH/BCCH..., GSML1FEC.cpp:584-653): block k = bursts 4k..4k+3 in arrival order (the "B"
 *   index of GSM 05.03 4.1.4); e-bits = burst[3..59] and [88..144]; deinterleave, decode 456 -> 228,
 *   invert the 40 parity bits, Fire-code syndrome.  d_frames: 23 octets per block = d[] after LSB8MSB,
 *   packed MSB first (the L2 frame); d_ok[k] = 1 iff the syndrome is zero ("good frame").
 * RACH (GSML1FEC.cpp:475-514): one access burst per entry, e = burst[49..84]; decode 36 -> 18;
 *   d_tail_ok = the four tail bits are zero; d_bsic = the BSIC the parity word encodes (the caller
 *   compares it with its own, :493); d_ra = the 8-bit RA.
 * TCH/FACCH full rate (GSML1FEC.cpp:1030-1163): n_bursts consecutive bursts of one traffic channel; block m
 *   (m < n_bursts/4 - 1) spans bursts 4m..4m+7 through the diagonal deinterleaver.  d_tch: 33 octets per
 *   block = d[260] in GSM 05.03 order packed MSB first (the fixed g610BitOrder permutation into the
 *   vocoder frame is left to the caller); d_tch_good = class-1a parity and tail bits check (decodeTCH's
 *   `good` for a frame that is not stolen); d_stolen = the Hl stealing flag of the block's last burst;
 *   d_facch / d_facch_ok (optional, both or neither) = the XCCH decode of the same block, which the
 *   reference runs when the frame is stolen.  Bad-frame substitution (GSM 06.11, uses random()) stays
 *   with the caller.
 * Generic: n_blocks independent SoftVector::decode runs, n_soft (even, <= 1024) values in, n_soft/2 bits out
 *   (one per byte). */
int trxsig_fec_xcch_decode_batch(trxsig_ctx *ctx, const float *d_soft, int soft_stride, int n_blocks,
                                 int wire_quantise, uint8_t *d_frames, uint8_t *d_ok);
int trxsig_fec_rach_decode_batch(trxsig_ctx *ctx, const float *d_soft, int soft_stride, int n_bursts,
                                 int wire_quantise, uint8_t *d_tail_ok, uint8_t *d_bsic, uint8_t *d_ra);
/* XCCH L1 encode, the transmit-side mirror (XCCHL1Encoder::sendFrame/encode/interleave/transmit,
 * GSML1FEC.cpp:772-845): n_blocks L2 frames of 23 octets -> 4*n_blocks normal bursts of 148 bits, one bit per
 * byte, ready for trxsig_modulate_batch: e-bits at 3..59 / 88..144, zero tails, both stealing flags set, the
 * training sequence `tsc` at 61..86. */
int trxsig_fec_xcch_encode_batch(trxsig_ctx *ctx, const uint8_t *d_frames, int n_blocks, int tsc, uint8_t *d_bits);
int trxsig_fec_tch_decode_batch(trxsig_ctx *ctx, const float *d_soft, int soft_stride, int n_bursts,
                                int wire_quantise, uint8_t *d_tch, uint8_t *d_tch_good, uint8_t *d_facch,
                                uint8_t *d_facch_ok, uint8_t *d_stolen);
int trxsig_fec_viterbi_batch(trxsig_ctx *ctx, const float *d_soft, int n_soft, int64_t in_stride, int n_blocks,
                             uint8_t *d_bits, int64_t out_stride);

/* ---- the free-standing vector primitives of sigProcLib.h (csrc/trxsig_prim.hip) -------------------------------
 * On the burst path these only run fused into the burst kernels above; the stand-alone forms complete the
 * sigProcLib.h surface (convolve :126, correlate :162, vectorSlicer :168, delayVector :180, interpolatePoint :198,
 * peakDetect :208, scaleVector :215, decimateVector :304 of Transceiver/sigProcLib.h; GMSKRotate / GMSKReverseRotate
 * are sigProcLib.cpp:232-264) and are what config 1's call sequence (Transceiver/sigProcLibTest.cpp) runs through.
 * Batch forms: B independent vectors packed in one device array (d_off / d_len in samples); max_len = the largest
 * d_len (sizes the launch).  Values are the reference's, bit for bit: every sum in the reference's order with its
 * skip / break rules.  Host forms: one vector, pageable host buffers, one PCIe round trip.
 * span: ConvType of sigProcLib.h:41-48 (+ CUSTOM of Transceiver52M/sigProcLib.h:47 with cust_start / cust_len).
 * flags: bit 0 = a is real-only, bit 1 = b is real-only (signalVector::isRealOnly: the four arithmetic forms of
 * sigProcLib.cpp:326-365), bit 2 = b has ABSSYM symmetry (:369-398; convolve only); correlate != 0: b is used reversed and conjugated (sigProcLib.cpp:474-503). */
enum { TRXSIG_FULL_SPAN = 0, TRXSIG_OVERLAP_ONLY = 1, TRXSIG_START_ONLY = 2, TRXSIG_WITH_TAIL = 3, TRXSIG_NO_DELAY = 4,
       TRXSIG_CUSTOM = 5 };
int trxsig_convolve_out_len(int La, int Lb, int span, int cust_len);   /* < 0: unknown span */
/* one filter d_b (Lb taps) for all B vectors; out vector i (trxsig_convolve_out_len(d_a_len[i], ...) samples) at d_out_off[i] */
int trxsig_convolve_batch(trxsig_ctx *ctx, const trxsig_c32 *d_a, const int32_t *d_a_off, const int32_t *d_a_len, int B,
                          int max_len, const trxsig_c32 *d_b, int Lb, int span, int flags, int correlate, int cust_start,
                          int cust_len, trxsig_c32 *d_out, const int32_t *d_out_off);
int trxsig_convolve_host(trxsig_ctx *ctx, const trxsig_c32 *h_a, int La, const trxsig_c32 *h_b, int Lb, int span, int flags,
                         int correlate, int cust_start, int cust_len, trxsig_c32 *h_out, int out_cap);   /* returns the length */
/* A delay or an index beyond +-TRXSIG_MAX_INDEX (2^24: a float there has no fractional bits left), an infinity or a NaN is
 * one the reference's table-sinc range reduction (sigProcLib.cpp:163-188, reached from :573-616 and :639-659) cannot reduce
 * -- its `arg -= 1` loop no longer changes arg, the call never returns.  The host forms answer TRXSIG_EINVAL before any
 * launch; the batch forms (the values are on the device) write zeros for such a vector / point.  The device-side reduction
 * itself is loop-free (csrc/trxsig_dev.h, dev_range_reduce), so no argument can keep a wave running. */
#define TRXSIG_MAX_INDEX 16777216.0f
/* delayVector(x, delay): d_out has d_in's layout and must not overlap it */
int trxsig_delay_vector_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                              const float *d_delay, int real_only, trxsig_c32 *d_out);
int trxsig_delay_vector_host(trxsig_ctx *ctx, trxsig_c32 *h_x /* in place */, int n, float delay, int real_only);
/* interpolatePoint(x_i, d_ix[i]) -> d_out[i] */
int trxsig_interpolate_point_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                                   const float *d_ix, int real_only, trxsig_c32 *d_out);
int trxsig_interpolate_point_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, float ix, int real_only, trxsig_c32 *h_out);
/* peakDetect(x_i, &d_index[i], &d_avgpwr[i]) -> d_peak[i]; d_index / d_avgpwr may be NULL */
int trxsig_peak_detect_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                             trxsig_c32 *d_peak, float *d_index, float *d_avgpwr);
int trxsig_peak_detect_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, trxsig_c32 *h_peak, float *h_index, float *h_avgpwr);
/* energyDetect(x_i, window, thresh, &d_avgpwr[i]) -> d_ok[i] (sigProcLib.h:255-258) for any window length; sample_step 1
 * (Transceiver/) or 4 (the Transceiver52M variant, Transceiver52M/sigProcLib.cpp:946-963).  d_avgpwr / d_ok may be NULL. */
int trxsig_energy_detect_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                               unsigned window, int sample_step, float thresh, float *d_avgpwr, uint8_t *d_ok);
int trxsig_energy_detect_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, unsigned window, int sample_step, float thresh,
                              float *h_avgpwr);   /* returns 1 / 0 (the reference's bool), < 0 on error */
/* in place: scaleVector(x_i, d_scale[i]); GMSKRotate / GMSKReverseRotate (elements past the 157*sps table entries are
 * left as they are -- the reference reads past its table there); vectorSlicer */
int trxsig_scale_vector_batch(trxsig_ctx *ctx, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                              const trxsig_c32 *d_scale, int real_only);
int trxsig_gmsk_rotate_batch(trxsig_ctx *ctx, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                             int reverse, int real_only);
int trxsig_vector_slicer_batch(trxsig_ctx *ctx, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len);
/* decimateVector(x_i, factor): d_len[i] / factor samples at d_out_off[i] (d_len[i] must be a multiple of factor: the
 * reference writes past its allocation otherwise, sigProcLib.cpp:1045-1050) */
int trxsig_decimate_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                          int factor, trxsig_c32 *d_out, const int32_t *d_out_off);
/* ---- the rest of sigProcLib.h: the functions no caller on the burst path uses, for a complete surface -------------------
 * dB (sigProcLib.h:102; sigProcLib.cpp:88-114) and dBinv (:105; :117-144): the reference's piecewise-linear float
 *   approximations, on the host (scalar functions of one float).  sinc (:177; :567-571) against the context's trig table.
 * gaussianNoise (:188-190; :618-637): Box-Muller on the C library's rand(), two draws per sample in the reference's
 *   order (more after a zero draw) -- the caller's srand() seed decides the values, as with the reference.  Host.
 * vectorNorm2 / vectorPower (:108, 111; :146-160): the powers summed in index order; d_norm2 / d_power may be NULL.
 * frequencyShift (:149-153; :432-471): y[k] = x[k] * expjLookup(phase_k) (real-only x: expjLookup(phase_k) * x[k].real()),
 *   phase_0 = start_phase, phase_{k+1} = phase_k + freq in float; d_final_phase[i] (optional) = the phase after the last
 *   sample.  d_out may be d_in.  The reference's range reduction is a subtract-one loop that never ends for a float too large
 *   to change by 1: phases beyond +-25000 rad (|start| + n |freq|) are refused by the host forms (TRXSIG_EINVAL); the batch
 *   form bounds the loop instead and its values beyond that range are unspecified.
 * addVector (:184-185; :746-758): x[k] += y[k] over the shorter of the two, in place.
 * offsetVector (:225-226; :760-777): x[k] += offset (real-only x: x[k] = offset + x[k].real()), in place.
 * resampleVector (:352-354; :1213-1243) AS THE REFERENCE BEHAVES: its loop never advances the output iterator, so every
 *   interpolated value lands in element 0 and the other ceil(n * exp_factor) - 1 elements stay zero.  exp_factor >= 1 (NULL in
 *   the reference otherwise: TRXSIG_EINVAL); out vector i at d_out_off[i], trxsig_resample_linear_out_len(n, f) samples.
 * convolve's ABSSYM form (:369-398): trxsig_convolve_batch / _host with flags bit 2 set (b is an ABSSYM filter: half its
 *   taps are used, each on a[t-j] and on a[t-Lb+j]); the reference's reads beyond a's end (its fourth arm has no upper
 *   bound) count as zero. */
float trxsig_db(float x);
float trxsig_dbinv(float x);
int trxsig_sinc_host(const trxsig_ctx *ctx, float x, float *out);
int trxsig_gaussian_noise_host(int length, float variance, trxsig_c32 mean, trxsig_c32 *h_out);
int trxsig_vector_norm2_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B, float *d_norm2,
                              float *d_power);
int trxsig_vector_norm2_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, float *norm2, float *power);
int trxsig_frequency_shift_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                                 const float *d_freq, const float *d_start_phase, int real_only, trxsig_c32 *d_out, float *d_final_phase);
int trxsig_frequency_shift_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, float freq, float start_phase, int real_only,
                                trxsig_c32 *h_out, float *final_phase);
int trxsig_add_vector_batch(trxsig_ctx *ctx, trxsig_c32 *d_x, const int32_t *d_xoff, const int32_t *d_xlen, const trxsig_c32 *d_y,
                            const int32_t *d_yoff, const int32_t *d_ylen, int B, int max_len);
int trxsig_add_vector_host(trxsig_ctx *ctx, trxsig_c32 *h_x, int nx, const trxsig_c32 *h_y, int ny);
int trxsig_offset_vector_batch(trxsig_ctx *ctx, trxsig_c32 *d_x, const int32_t *d_off, const int32_t *d_len, int B, int max_len,
                               const trxsig_c32 *d_offset, int real_only);
int trxsig_resample_linear_out_len(int n, float exp_factor);   /* < 0: exp_factor < 1 */
int trxsig_resample_linear_batch(trxsig_ctx *ctx, const trxsig_c32 *d_in, const int32_t *d_off, const int32_t *d_len, int B,
                                 float exp_factor, const trxsig_c32 *d_end_point, trxsig_c32 *d_out, const int32_t *d_out_off);
int trxsig_resample_linear_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, float exp_factor, trxsig_c32 end_point, trxsig_c32 *h_out,
                                int out_cap);   /* returns the length */
/* one vector, in place on the host buffer; op: 0 scaleVector(scale), 1 GMSKRotate, 2 GMSKReverseRotate, 3 vectorSlicer,
 * 4 offsetVector(scale = the offset) */
int trxsig_elementwise_host(trxsig_ctx *ctx, int op, trxsig_c32 *h_x, int n, trxsig_c32 scale, int real_only);
int trxsig_decimate_host(trxsig_ctx *ctx, const trxsig_c32 *h_x, int n, int factor, trxsig_c32 *h_out);   /* returns n / factor */

/* ---- measurement helpers (HIP events on the context's stream; used by bench.py) ---------------
 * trxsig_timer_*: one start/stop event pair around whatever the caller enqueues in between.
 * trxsig_profile_*: when enabled, every kernel launch made by the library is bracketed by its own
 *   event pair on the context's stream; trxsig_profile_collect synchronises, adds up the elapsed
 *   time and launch count per kernel and resets.  kernel ids: TRXSIG_K_*. */
int trxsig_timer_start(trxsig_ctx *ctx);
int trxsig_timer_stop(trxsig_ctx *ctx, float *elapsed_ms);   /* synchronises on the stop event */
enum { TRXSIG_K_TSC_CORR = 0, TRXSIG_K_TSC_PEAK = 1, TRXSIG_K_DEMOD = 2, TRXSIG_K_RACH_CORR = 3,
       TRXSIG_K_RACH_PEAK = 4, TRXSIG_K_MODULATE = 5, TRXSIG_K_RESAMPLE = 6, TRXSIG_K_EQUALIZE = 7,
       TRXSIG_K_CONVERT = 8, TRXSIG_K_NORMAL_FUSED = 9, TRXSIG_K_FEC = 10, TRXSIG_K_NORMAL_CHAIN = 11,
       TRXSIG_K_EQ_DELAY = 12, TRXSIG_K_EQ_DFE = 13, TRXSIG_K_GROUP = 14, TRXSIG_K_COUNT = 15 };
/* TRXSIG_K_EQUALIZE = k_eq_detect / k_design_dfe; TRXSIG_K_GROUP = the Transceiver group's replay (trxsig_trxgroup.h) */
const char *trxsig_kernel_name(int kernel_id);
int trxsig_profile_enable(trxsig_ctx *ctx, int on);
int trxsig_profile_collect(trxsig_ctx *ctx, float total_ms[TRXSIG_K_COUNT], int launches[TRXSIG_K_COUNT]);
/* the same for a caller that states how many entries its arrays hold (at most `cap` are written); returns the library's
 * kernel count, so a caller built against another header version can tell */
int trxsig_profile_collect_n(trxsig_ctx *ctx, int cap, float *total_ms, int *launches);
int trxsig_kernel_count(void);
/* Implementation choice for A/B measurements (tuning build: libtrxsig_tune.so, `make -C csrc tune`); results are bit-identical
 * whatever is selected.
 *   TRXSIG_TUNE_NORMAL_PATH (trxsig_detect_demod_normal_batch; initial value: env TRXSIG_TSC_VARIANT, else 0):
 *     0 = three kernels: correlate, peak, demodulate (the default and the fastest measured);
 *     1 = one fused kernel, a wave per burst;   2 = fused, two bursts per wave;
 *     3 = fused, four bursts per wave (k_normal_quad);   4 = k_normal_quad's detection half, then k_demod.
 *     5 = ONE launch whose detect workgroups hand over to its demodulate workgroups (k_normal_chain,
 *     trxsig_chain.hip).
 *     The fused kernels and the chain need nsoft <= 148 (the chain also nsoft > 0); otherwise the call takes path 0.
 *   TRXSIG_TUNE_CHAIN_LAG: path 5, tiles of 16 bursts (per stream of every 8th tile) between a tile's detect
 *     workgroup and its demodulate workgroups in launch order.  TRXSIG_TUNE_CHAIN_SPIN: polls before a demodulate
 *     wave gives up waiting for its burst's detection (then the call is reported as failed at the library's next
 *     entry and path 0 is used from there on; 0 makes every wait that is not satisfied at once fail -- tests).
 *     Path 5 is for A/B measurements only: its outputs are valid only after trxsig_synchronize (or the next library
 *     call on the context) has returned TRXSIG_OK -- a caller that waits on the stream by other means would read the
 *     soft bits of a batch whose hand-over timed out without seeing the error; its forward progress also rests on
 *     workgroups starting in launch order, which HIP does not promise.
 *   TRXSIG_TUNE_RACH_PATH (initial value: env TRXSIG_RACH_VARIANT, else 2): 0 = exact correlation at every
 *     lag (k_rach_corr + k_rach_peak), 1 = approximate-then-exact in one kernel, a wave per burst (k_rach_fast),
 *     2 = the same with peakDetect's bisection and the tail in their own kernel, two lanes per burst
 *     (k_rach_front + k_rach_peak2; bursts too close to the threshold to call from approximate valley powers
 *     are handed back to k_rach_fast).  All three give the reference's results.
 *   TRXSIG_TUNE_DEMOD_BESIDE (both libraries; default 0): 1 = trxsig_detect_demod_normal_batch (path 0) detects on the context's
 *     stream and DEMODULATES ON A SIDE STREAM of the context, from its own copy of (flags, amp, TOA): the call returns with
 *     d_flags / d_amp / d_toa / d_avgpwr ordered on the context's stream as always, while d_soft / d_hard -- and the READS of
 *     d_samples / d_offset / d_length -- complete only behind trxsig_synchronize (or the next call that is not of this kind).
 *     The next call's correlator (VALU-bound) then runs beside this call's demodulator (HBM-bound): a throughput lever for a
 *     caller that pipelines batches (bench.py reports it as a side field, never as `value`); same results.
 *   TRXSIG_TUNE_BESIDE_DET_CUS (both libraries; default 0; only meaningful with TRXSIG_TUNE_DEMOD_BESIDE = 1): x > 0 = the two
 *     streams of that arrangement PARTITION the compute units (hipExtStreamCreateWithCUMask): the detectors (k_tsc_corr,
 *     k_tsc_peak2) run on a stream of the library masked to x CUs, the demodulator on one masked to the other CUs, so that the
 *     VALU-bound and the HBM-bound kernel of neighbouring calls co-run without contending for the same CUs' wave slots and LDS.
 *     Then d_flags / d_amp / d_toa / d_avgpwr too are complete only behind trxsig_synchronize.  TRXSIG_TUNE_CU_LAYOUT: 0 = the
 *     detect set is mask bits 0..x-1, 1 = every eighth-of-the-mask takes x/8 of them (tools/cu_split.py measures both).
 *     TRXSIG_TUNE_BESIDE_PRIORITY (without masks): the side stream's priority, 0 = normal, 1 = the device's highest, 2 = its lowest.
 *   TRXSIG_TUNE_GENERIC_TAPS: 1 = midamble correlators without the "exactly +-1 tap component" form (which
 *     is only taken when the actual taps have that shape).  Default 0.
 *   TRXSIG_TUNE_SPECULATIVE_PEAK: path 0's peakDetect kernel.  0 = two lanes per burst (early and late point of
 *     each bisection step side by side, sinc table in LDS, window in registers; k_tsc_peak2, the default),
 *     1 = eight lanes per burst with the bisection speculated two levels at a time (k_tsc_peak8; 25 us per
 *     64 K bursts, LDS bandwidth), 2 = a lane per burst, the reference's serial loop (k_tsc_peak; 18 us,
 *     k_tsc_peak2 15 us).  All three are bit-identical.
 *   LIBRARY-WIDE knobs (both libraries; they act on every context of the process and are read by the launchers -- the library
 *   reads no environment variable on a launch path; all give the same results):
 *   TRXSIG_TUNE_EQ_TAIL: 1 = scaleVector + delayVector + equalizeBurst in one kernel (k_eq_dfe4, the default), 2 = k_eq_delay +
 *     k_eq_dfe2 through the scratch rows.  TRXSIG_TUNE_EQ_DENSE: marked bursts per call above which the Transceiver group's channel
 *     estimate runs a lane per burst instead of a wave per burst (default 4096).  TRXSIG_TUNE_RXRES_WPB: windows per workgroup of the
 *     receive resampler (0 = chosen from the launch size); TRXSIG_TUNE_RXRES_ROWS: 1 = its tap rows in visiting order (default),
 *     0 = in branch order.  TRXSIG_TUNE_CHAN_TPW: tiles per workgroup of the shared-filter channeliser (0 = chosen from the launch size).
 *   TRXSIG_TUNE_GROUP_REPLAY: the Transceiver group's receive state machine (trxsig_trxgroup.h).  0 = a wave per ARFCN and 64-timeslot
 *     segment that visits only the timeslots at which the state can move (calls of up to 1,024 timeslots; the default), 1 = a lane per
 *     ARFCN (and segment) stepping through every timeslot (round 4's kernels, also what longer calls take). */
enum { TRXSIG_TUNE_NORMAL_PATH = 0, TRXSIG_TUNE_RACH_PATH = 1, TRXSIG_TUNE_GENERIC_TAPS = 2, TRXSIG_TUNE_SPECULATIVE_PEAK = 3,
       TRXSIG_TUNE_CHAIN_LAG = 4, TRXSIG_TUNE_CHAIN_SPIN = 5, TRXSIG_TUNE_DEMOD_BESIDE = 7, TRXSIG_TUNE_BESIDE_DET_CUS = 8,
       TRXSIG_TUNE_CU_LAYOUT = 9, TRXSIG_TUNE_BESIDE_PRIORITY = 11, TRXSIG_TUNE_EQ_TAIL = 12, TRXSIG_TUNE_EQ_DENSE = 13,
       TRXSIG_TUNE_RXRES_WPB = 14, TRXSIG_TUNE_RXRES_ROWS = 15, TRXSIG_TUNE_CHAN_TPW = 16, TRXSIG_TUNE_GROUP_REPLAY = 17 };
int trxsig_set_tuning(trxsig_ctx *ctx, int key, int value);
/* 1 in libtrxsig_tune.so (every implementation above selectable), 0 in the product library libtrxsig.so, which carries the
 * defaults only (normal path 0 with the two-lane peak kernel, RACH paths 1 and 2) and answers TRXSIG_EINVAL to the rest. */
int trxsig_tuning_build(void);
/* validate a host copy of the table blob (magic, version, size, checksum): 0 if valid */
int trxsig_tables_validate_host(const void *h_blob, size_t bytes);
/* Multi-GPU set-up for a host that is not Python (one process or thread per GPU): the ONE collective of the
 * path, an RCCL broadcast of the table blob over xGMI.  Every rank calls it with its ncclComm_t and a device buffer of
 * trxsig_tables_bytes() bytes -- on `root` that buffer holds the blob (trxsig_tables_device(ctx) of a context made with
 * trxsig_create, or an upload of trxsig_tables_build_host), on the others it receives it; then the others call
 * trxsig_create_from_tables, which validates magic / version / checksum.  Equivalent to
 *   ncclBroadcast(d_blob, d_blob, bytes, ncclUint8, root, comm, stream);
 * librccl.so is loaded at the first call (libtrxsig does not link it).  Stream-ordered: synchronise `hip_stream` before
 * trxsig_create_from_tables. */
int trxsig_tables_broadcast(void *nccl_comm, void *d_blob, size_t bytes, int root, void *hip_stream);
/* The error bar of the access-burst detector's approximate correlation pass for this table blob (host computation,
 * no GPU): *bound x sqrt(sum |x[n]|^2 over the burst) bounds |approximate - reference| correlation amplitude at any
 * lag (derivation: csrc/trxsig_rach.hip); *seq_norm (optional) = the 2-norm of the RACH sequence, for scale.  The
 * detector only uses the approximate values to decide what to recompute with the reference's arithmetic
 * (detectRACHBurst, Transceiver/sigProcLib.cpp:860-914), so the bound affects speed, never results. */
int trxsig_tables_rach_error_bound(const void *h_blob, size_t bytes, float *bound, float *seq_norm);

#ifdef __cplusplus
}
#endif
#endif /* TRXSIG_H */
