// trxsig_launch.h -- internal launcher prototypes.  The kernels live in trxsig_normal.hip (default normal-burst
// path), trxsig_fused.hip (single-launch alternates), trxsig_rach.hip, trxsig_eq.hip, trxsig_tx.hip, trxsig_fec.hip;
// shared device code in trxsig_dev.h, trxsig_corr.h, trxsig_bisect.h, trxsig_demod.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "trxsig.h"
#include "trxsig_tables.h"

// optional per-kernel event bracketing (trxsig_profile_*): called by the launchers around each kernel
struct TrxProfiler {
  virtual void begin(int kernel_id, hipStream_t st) = 0;
  virtual void end(int kernel_id, hipStream_t st) = 0;
  virtual ~TrxProfiler() {}
};

// number of complex slots per burst in the detect->peak record (SoA, [slot][Bpad])
int trx_rec_slots(int sps);

// dT: device tables, hT: the host copy (the midamble taps travel as a kernel argument)
hipError_t trx_launch_copy_verdict(hipStream_t st, const uint8_t *flags, const trx_c32 *amp, const float *toa, int B, uint8_t *flags_out,
                                   trx_c32 *amp_out, float *toa_out);
hipError_t trx_launch_tsc_detect(hipStream_t st, int sps, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples,
                                 const int32_t *off, const int32_t *len, int B, int tsc,
                                 float detect_thresh, float energy_thresh, trx_c32 *rec, int Bpad,
                                 uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr,
                                 int variant /* bit 0: generic taps (no tap-class specialisation); bit 1: 8-lanes-per-burst speculative peak kernel; bit 2: lane-per-burst peak kernel (default: 2 lanes per burst) */,
                                 TrxProfiler *prof);

// the whole normal-burst leg in one kernel (k_normal_fused); lanes_per_burst = 64 or 32, nsoft <= 148
hipError_t trx_launch_normal_fused(hipStream_t st, int sps, int lanes_per_burst, const TrxTables *dT, const TrxTables *hT,
                                   const trx_c32 *samples, const int32_t *off, const int32_t *len, int B, int tsc,
                                   float detect_thresh, float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa,
                                   float *avgpwr, float *soft, uint8_t *hard, int nsoft, int stride,
                                   int generic_taps /* 1: no tap-class specialisation */, TrxProfiler *prof, int soft_tolerance = 0);

// the whole normal-burst leg in ONE launch with an in-launch hand-over (trxsig_chain.hip); nsoft 1..148; det: 16 bytes
// per burst, tag words clear on entry (left clear on exit); status: host-visible word raised when a wait runs out
size_t trx_chain_ws_bytes(int bursts);
hipError_t trx_launch_normal_chain(hipStream_t st, int sps, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples,
                                   const int32_t *off, const int32_t *len, int B, int tsc, float detect_thresh,
                                   float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr, float *soft,
                                   uint8_t *hard, int nsoft, int stride, void *det, unsigned *status, int lag,
                                   unsigned spin_limit, int generic_taps, TrxProfiler *prof, int dbg = 0, int soft_tolerance = 0);

// free-standing vector primitives of sigProcLib.h (trxsig_prim.hip); op: 0 scaleVector, 1 GMSKRotate, 2 GMSKReverseRotate,
// 3 vectorSlicer, 4 offsetVector (scale[v] = the offset)
int trx_convolve_out_len(int La, int Lb, int span, int cust_len);
hipError_t trx_launch_convolve(hipStream_t st, const trx_c32 *a, const int32_t *a_off, const int32_t *a_len, int B, int max_out,
                               const trx_c32 *b, int Lb, int span, int flags, int correlate, int cust_start, int cust_len,
                               trx_c32 *out, const int32_t *out_off);
hipError_t trx_launch_delay_vector(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off, const int32_t *len,
                                   int B, const float *delay, int real_only, trx_c32 *out);
hipError_t trx_launch_interpolate_point(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off,
                                        const int32_t *len, int B, const float *ix, int real_only, trx_c32 *out);
hipError_t trx_launch_peak_detect(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off, const int32_t *len,
                                  int B, trx_c32 *peak, float *index, float *avgpwr);
hipError_t trx_launch_energy_detect(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, unsigned window,
                                    int step, float thresh, float *avgpwr, uint8_t *ok);
hipError_t trx_launch_elementwise(hipStream_t st, int op, const TrxTables *dT, trx_c32 *x, const int32_t *off, const int32_t *len,
                                  int B, int max_len, const trx_c32 *scale, int real_only);
hipError_t trx_launch_decimate(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, int max_len,
                               int factor, trx_c32 *out, const int32_t *out_off);

// the rest of sigProcLib.h's surface (trxsig_prim.hip): vectorNorm2 / vectorPower, frequencyShift, addVector, resampleVector
hipError_t trx_launch_vector_norm2(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, float *norm2_out,
                                   float *power_out);
float trx_frequency_shift_max_phase(void);
hipError_t trx_launch_frequency_shift(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off, const int32_t *len,
                                      int B, const float *freq, const float *start, int real_only, trx_c32 *out, float *final_phase);
hipError_t trx_launch_add_vector(hipStream_t st, trx_c32 *x, const int32_t *xoff, const int32_t *xlen, const trx_c32 *y,
                                 const int32_t *yoff, const int32_t *ylen, int B, int max_len);
hipError_t trx_launch_resample_linear(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, float exp_factor,
                                      const trx_c32 *end_point, trx_c32 *out, const int32_t *out_off);

// RACH detect: ws = workspace of trx_rach_rec_floats(sps) * Bpad floats
int trx_rach_rec_floats(int sps);
hipError_t trx_launch_rach_detect(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                                  const int32_t *off, const int32_t *len, int B, float detect_thresh,
                                  float energy_thresh, float *ws, int Bpad, uint8_t *flags, trx_c32 *amp,
                                  float *toa, float *avgpwr, TrxProfiler *prof);

// bound on |approximate - reference| correlation amplitude per unit of sqrt(burst energy) (error model in trxsig_rach.hip)
float trx_rach_amp_err(const TrxTables *hT);
// same results, approximate-then-exact single kernel (k_rach_fast)
hipError_t trx_launch_rach_fast(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                                const int32_t *off, const int32_t *len, int B, float detect_thresh,
                                float energy_thresh, float amp_err /* trx_rach_amp_err(host tables) */,
                                float *ws /* trx_rach_rec_floats() floats per burst */, int Bpad,
                                int split /* 1: k_rach_front + k_rach_peak2 + hand-over, 0: k_rach_fast alone */,
                                uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr, TrxProfiler *prof);

// need_mask != 0: burst enabled iff (flags[b] & need_mask) == need_mask; need_mask == 0: iff flags[b] != 0;
// flags == NULL: every burst enabled.
// Library-wide implementation knobs (A/B measurements and the tests that force a route): set through trxsig_set_tuning, read by the
// launchers -- the library reads NO environment variable on a launch path.  Defaults in trxsig_api.cpp.
enum { TRX_KNOB_EQ_TAIL = 0,      // 1 = k_eq_dfe4 (scale + delay + DFE in one kernel, default), 2 = k_eq_delay + k_eq_dfe2 through the scratch rows
       TRX_KNOB_EQ_DENSE = 1,     // marked bursts above which the lane-per-burst k_eq_detect takes over from the wave-per-burst estimate (4096)
       TRX_KNOB_RXRES_WPB = 2,    // k_rx_resample: windows per workgroup; 0 = chosen from the launch size
       TRX_KNOB_RXRES_ROWS = 3,   // k_rx_resample: 1 = tap rows in visiting order (default), 0 = in branch order
       TRX_KNOB_CHAN_TPW = 4,     // k_channelise16: tiles per workgroup; 0 = chosen from the launch size
       TRX_KNOB_GROUP_REPLAY = 5, // the Transceiver group's state machine: 0 = a wave per 64-slot segment visiting the slots that move the state
                                  // (k_group_replay_wave, default), 1 = a lane per ARFCN (and segment) stepping through every slot (round 4)
       TRX_KNOB_COUNT = 6 };
int trx_knob(int id);
void trx_knob_set(int id, int value);

hipError_t trx_launch_demod(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                            const int32_t *off, const int32_t *len, int B, const trx_c32 *amp,
                            const float *toa, const uint8_t *flags, int need_mask, float *soft,
                            uint8_t *hard, int nsoft, int stride, TrxProfiler *prof, int soft_tolerance = 0);
// (soft_tolerance: TRXSIG_SOFT_TOLERANCE -- the rearranged demodulator of trxsig_demod.h; hard bits exact, soft bits within 7.4e-5)

hipError_t trx_launch_modulate(hipStream_t st, int sps, const TrxTables *dT, const uint8_t *bits, const int32_t *guard,
                               const float *gain, int B, trx_c32 *out, const int32_t *out_off, TrxProfiler *prof);
hipError_t trx_launch_resample(hipStream_t st, const trx_c32 *in, int n, long long in_stride, int S, int P, int Q,
                               const float *lpf, int L, trx_c32 *out, long long out_stride, int nout,
                               TrxProfiler *prof);
// the tiled resampler in full (trxsig_tx.hip): int16 I/Q in with history windows (RadioInterface::pullBuffer) or int16 out
// with gain (pushBuffer); see k_resample.  OB is filled in by the launcher.
struct TrxResampleArgs {
  const void *in; long long in_stride;                     // stream s at in + s*in_stride (elements: trx_c32 or int16 pairs)
  const short2 *hist; int hist_len;                        // int16 input: the hist_len samples before each stream's start
  int n, win_step, swap;                                   // samples per window, start-to-start distance of windows, I/Q swap
  const float *lpf; int L, P, Q;
  int o_skip, n_out;                                       // outputs [o_skip, n_out) of every window are produced
  void *out; long long out_stride, out_win_step;           // window w of stream s writes at out + s*out_stride + w*out_win_step
  float gain;                                              // int16 output
  int OB, xcap, taps_lds, tap_pitch, tap_g, row_inv;       // filled in by the launcher
  // int16 input, wideband (the channeliser): output stream s is carrier s % mix_carriers of raw stream s / mix_carriers, mixed down
  // on the way into LDS: z[n] = x[n] * expjLookup(phase[n]) (frequencyShift's arithmetic, sigProcLib.cpp:459) with the phase of
  // raw sample n formed directly, phase[n] = (float)(t - 2 pi floor(t / 2 pi)), t = (double) n * (double) freq, instead of by
  // the reference's running float sum (trxsig_frontend.h says why).
  // mix_freq: [mix_carriers] floats (device); mix_n0: global index of raw sample 0 of this launch's `in`; mix_tables: the
  // context's tables (trig lookup)
  const float *mix_freq; int mix_carriers; long long mix_n0; const TrxTables *mix_tables;
  // input computed from burst bits (the fused transmit back end): in = the bit ring [S][in_stride slots][148], tx_gain the gains
  // [S][in_stride]; tx_start[m] / tx_meta[m] (m < tx_n, device, ascending): where burst m starts in the window and
  // slot | guard << 16 | has_gain << 20
  const TrxTables *tx_tables; const float *tx_gain; const int32_t *tx_start, *tx_meta; int tx_n, tx_sps;
};
hipError_t trx_launch_resample_ex(hipStream_t st, TrxResampleArgs a, int S, int n_windows, bool in_i16, bool out_i16,
                                  TrxProfiler *prof, bool in_bits = false);
// the channeliser's shared-filter form (trxsig_chan.hip): C carriers on a grid of sixteenths of the wideband rate in one pass
// binmap: four bits per carrier c, its bin k_c (theta_c = 2 pi k_c / 16): with four carriers or more the kernel takes the bins from one FFT
hipError_t trx_launch_channelise16(hipStream_t st, TrxResampleArgs a, int S_wide, int C, int n_windows, const float2 *tw, TrxProfiler *prof,
                                   unsigned long long binmap);
hipError_t trx_launch_tx_ring_store(hipStream_t st, const uint8_t *bits, const float *gain, int S, int nb, int head, int cap, uint8_t *ring,
                                    float *gring);
hipError_t trx_launch_burst_index(hipStream_t st, int S, int nb, long long stride, int rd, int tn0, int sps, int32_t *off,
                                  int32_t *len);
// The receive front end fused into the normal-burst detectors (trxsig_rxgen.h, sps = 4, 65*4 : 96 with at most four taps per
// output): the bursts are described by where they start in the stream of resampled samples, and the kernels compute those
// samples from the raw int16 stream as they need them.
struct TrxRxGen {
  const short2 *raw; long long raw_stride;                 // this push: stream s at raw + s*raw_stride, K*864 samples each
  const short2 *keep;                                      // [S][1056]: history + last chunk of the previous push (its window)
  const float4 *tpb;                                       // [65] taps of branch 4*(24 n mod 65) in slot n: lpf[br + 260 k], k = 0..3
  int K, swap, skipD;                                      // chunks in this push; I/Q swap; INHISTORY outputs skipped + (L-1)/2/Q
  int w0, w1;                                              // the last w0 (w1) outputs of a chunk have tap 0 (tap 1) beyond the window's end
  int tail, tn0, nb;                                       // uncut resampled samples before this push; TN of burst 0; bursts per stream
  const int32_t *sel;                                      // optional: launch index b stands for burst sel[b] (= s*nb + j); NULL: b itself
};
hipError_t trx_launch_rx_normal(hipStream_t st, const TrxTables *dT, const TrxTables *hT, const TrxRxGen &gen, int B, int tsc,
                                float detect_thresh, float energy_thresh, trx_c32 *rec, int Bpad, uint8_t *flags, trx_c32 *amp,
                                float *toa, float *avgpwr, float *soft, uint8_t *hard, int nsoft, int stride, int generic_taps,
                                TrxProfiler *prof, int soft_tolerance = 0);
// the other legs on bursts computed from the raw stream (the Transceiver group's fused front end): detectRACHBurst
// (k_rach_front_rx + k_rach_peak2 + the hand-over in k_rach_fast_rx; ws as trx_launch_rach_fast) and demodulateBurst alone
// with caller-supplied amplitude / TOA / enable flags (k_demod_rx)
hipError_t trx_launch_rx_rach(hipStream_t st, const TrxTables *dT, const TrxRxGen &gen, const int32_t *len /* samples per burst */, int B,
                              float detect_thresh, float energy_thresh, float amp_err, float *ws, int Bpad, uint8_t *flags, trx_c32 *amp,
                              float *toa, float *avgpwr, TrxProfiler *prof);
hipError_t trx_launch_rx_demod(hipStream_t st, const TrxTables *dT, const TrxRxGen &gen, int B, const trx_c32 *amp, const float *toa,
                               const uint8_t *flags, int need_mask, float *soft, uint8_t *hard, int nsoft, int stride, TrxProfiler *prof,
                               int soft_tolerance = 0);
// pack: 0 = int16 I/Q -> complex float (swap: I/Q flipped), 1 = complex float -> int16 I/Q, 2 = fp16 I/Q -> complex float
hipError_t trx_launch_convert(hipStream_t st, int pack, const void *in, long long n, int swap, void *out,
                              TrxProfiler *prof, float gain = 1.0f /* pack == 1: scaleVector before the cast */);

// sps = 1 equaliser path; xd: B x xstride complex scratch (xstride >= 157), toa_eq: B floats scratch,
// w: B x 7, bq: B x 5 complex
hipError_t trx_launch_equalize(hipStream_t st, const TrxTables *dT, const void *samples, int fmt /* TRXSIG_SAMPLES_* */, const int32_t *off,
                               const int32_t *len, int B, int tsc, float detect_thresh, float energy_thresh,
                               int variant52m, int max_toa, uint8_t *flags, trx_c32 *amp, float *toa, float *toa_eq,
                               trx_c32 *w, trx_c32 *bq, trx_c32 *xd, int xstride, float *soft, uint8_t *hard,
                               int nsoft, int stride, TrxProfiler *prof, bool geom52 = false /* trx_eq52_geometry(host tables, tsc) */);
// analyzeTrafficBurst's 52M window (ref52:983-1000) has the geometry k_eq_detect52 is written for when maxTOA = 4 AND the
// training sequence's expectedTOAPeak is 20 (true of all eight sequences of the tables this library builds; checked, not assumed)
inline bool trx_eq52_geometry(const TrxTables *hT, int tsc) {
  return hT && tsc >= 0 && tsc < 8 && (unsigned)round((double)((hT->mid_toa[tsc] + 5.0f) + (float)(size_t)((16 - 1) / 2))) == 20u;
}


// L1 FEC soft decode (trxsig_fec.hip).  mode 0: generic SoftVector::decode of nblk blocks of n soft values ->
// nout bits (one byte each) at out0 + b*out_stride; mode 1: XCCH (four bursts per block, in_stride = floats per
// burst): out0 = 23 octets per block, out1 = parity ok; mode 2: RACH (one burst per block): out0 = tail ok,
// out1 = BSIC, out2 = RA.  wire != 0: the UDP hop's 8-bit quantisation of the soft values first.
hipError_t trx_launch_fec(hipStream_t st, int mode, const float *soft, long long in_stride, int n, int nout, int nblk,
                          int wire, uint8_t *out0, uint8_t *out1, uint8_t *out2, long long out_stride, TrxProfiler *prof,
                          int ilv8 = 0 /* mode 1 through the 8-burst TCH deinterleaver (FACCH); mode 3 = TCH: out0 = 33
                                          octets of d[260], out1 = good, out2 = stolen */);

// the halves of trx_launch_equalize (see trxsig_eq.hip): channel estimate + designDFE with an explicit SNR
// threshold and no energy gate; equalizeBurst with caller-supplied taps (flags: DETECT bit = burst enabled)
hipError_t trx_launch_estimate_dfe(hipStream_t st, const TrxTables *dT, const void *samples, int fmt, const int32_t *off,
                                   const int32_t *len, int B, int tsc, float detect_thresh, float snr_thresh,
                                   float snr_value, int variant52m, int max_toa, uint8_t *flags, trx_c32 *amp, float *toa,
                                   float *toa_eq, float *chan_off, trx_c32 *w, trx_c32 *bq, trx_c32 *chan /* B x 6 or NULL */,
                                   TrxProfiler *prof, const uint8_t *enable = nullptr /* only bursts with enable[b] != 0; nothing
                                   is written for the others */, const float *snr_in = nullptr /* SNR estimate per burst */,
                                   bool geom52 = false, int32_t *work = nullptr /* B + 1 ints of scratch: lets a marked subset (enable) of
                                   the Transceiver/ variant be listed and estimated a wave per burst (k_eq_estimate_wave) */,
                                   bool listed = false /* work holds the list already: the count, then the marked bursts in any order */);
// designDFE(chan, snr, 7) alone; amp != NULL: scaleVector(chan, 1/amp) first
hipError_t trx_launch_design_dfe(hipStream_t st, const trx_c32 *chan, const trx_c32 *amp, const float *snr, int B, trx_c32 *w,
                                 trx_c32 *bq, TrxProfiler *prof);
hipError_t trx_launch_equalize_taps(hipStream_t st, const TrxTables *dT, const void *samples, int fmt, const int32_t *off,
                                    const int32_t *len, int B, const trx_c32 *amp, const float *toa_eq,
                                    const uint8_t *flags, const trx_c32 *w, const trx_c32 *bq, trx_c32 *xd, int xstride,
                                    float *soft, uint8_t *hard, int nsoft, int stride, TrxProfiler *prof,
                                    const int32_t *tap_ix = nullptr /* burst b uses the taps at index tap_ix[b] */);

// XCCH L1 encode: nblk L2 frames (23 octets each) -> 4*nblk bursts of 148 bits (one per byte); tsc_bits: the 26
// training-sequence bits (device)
hipError_t trx_launch_fec_xcch_encode(hipStream_t st, const uint8_t *frames, int nblk, const uint8_t *tsc_bits, uint8_t *bits,
                                      TrxProfiler *prof);
