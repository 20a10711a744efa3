// trxsig_eq.hip -- the equaliser path (sps = 1): channel estimate, designDFE, equalizeBurst.
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#include <cstdlib>

#include "trxsig_bisect.h"
#include "trxsig_demod.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Equaliser path (sps = 1; "Assumes symbol-rate sampling", sigProcLib.cpp:1342): the TSC leg of
// pullRadioVector with a channel estimate and a decision-feedback equaliser
// (Transceiver.cpp:298-349, 391-396; Transceiver52M/Transceiver.cpp for the windowed variant).
//
//   k_eq_detect : one lane per burst.  energyDetect, analyzeTrafficBurst(requestChannel) --
//                 correlation (full 36-lag window, or the 52M CUSTOM span of 2*maxTOA+1 lags),
//                 peakDetect, valley test, delayVector on the correlation, 6-tap channel pick --
//                 then scaleVector(chan, 1/amp), SNR and designDFE(chan, SNR, 7).  Everything here
//                 is tiny and strictly sequential per burst, so bursts are the parallel axis.
//   k_eq_delay  : 16 lanes per burst.  delayVector(burst/amp, -(TOA - chanOffset)).
//   k_eq_dfe    : one lane per burst.  7-tap feed-forward FIR + the 156-step decision-feedback
//                 recursion of equalizeBurst (:1352-1384) and the slicer.
// ---------------------------------------------------------------------------------------------
#define EQ_NC 36            /* max correlation lags kept per burst */
static constexpr int kEqWaveMax = 2048;   // k_eq_estimate_wave: waves launched; a call of at most this many bursts takes it without a list
// more marked bursts than this: a lane per burst (k_eq_detect) is the faster arrangement again (TRXSIG_EQ_DENSE: the tests force either route)
static int eq_dense() { return trx_knob(TRX_KNOB_EQ_DENSE); }   // (trxsig_set_tuning(TRXSIG_TUNE_EQ_DENSE): a test switches it between cases)
// A/B (tuning build only): TRXSIG_EQ_DETECT_GENERIC=1 keeps the padded kernel also where the fixed-geometry one applies (maxTOA 4)
static int eq_detect_generic() {
#ifdef TRX_TUNING_BUILD
  static const int v = std::getenv("TRXSIG_EQ_DETECT_GENERIC") ? std::atoi(std::getenv("TRXSIG_EQ_DETECT_GENERIC")) : 0;
  return v;
#else
  return 0;
#endif
}
// instantiations: k_eq_detect52 for config 5's geometry (52M window, maxTOA 4, expectedTOAPeak 20: `geom52`, which the caller derives
// from the host's copy of the tables -- trx_eq52_geometry, trxsig_launch.h); else k_eq_detect for the 52M window with maxTOA <= 5 (11
// lags, 26 window samples; maxTOA 4 without zero pads) and for everything else (the classic 36-lag window, wide 52M windows).
// TRXSIG_EQ_DETECT_GENERIC=1 (A/B measurements): 1 = never k_eq_detect52, 2 = nor the pad-free instantiation.
#define EQ_DETECT_LAUNCH_T(SMP, ...)                                                                        \
  if (variant52m && max_toa == 4 && eq_detect_generic() < 2) k_eq_detect<9, 26, SMP, 4><<<dim3((B + 63) / 64), dim3(64), 0, st>>>(__VA_ARGS__); \
  else if (variant52m && max_toa <= 5) k_eq_detect<12, 26, SMP><<<dim3((B + 63) / 64), dim3(64), 0, st>>>(__VA_ARGS__); \
  else k_eq_detect<EQ_NC, 52, SMP><<<dim3((B + 63) / 64), dim3(64), 0, st>>>(__VA_ARGS__)
#define EQ_DETECT_LAUNCH(...)                                                                               \
  do { if (fmt == TRXSIG_SAMPLES_F16) { EQ_DETECT_LAUNCH_T(SmpF16, __VA_ARGS__); } else { EQ_DETECT_LAUNCH_T(SmpC32, __VA_ARGS__); } } while (0)
#define EQ_DELAY_LAUNCH(...)                                                                                \
  do {                                                                                                      \
    const dim3 g_((B + 15) / 16), b_(256);                                                                  \
    if (fmt == TRXSIG_SAMPLES_F16) k_eq_delay<SmpF16><<<g_, b_, 0, st>>>(__VA_ARGS__);                      \
    else k_eq_delay<SmpC32><<<g_, b_, 0, st>>>(__VA_ARGS__);                                                \
  } while (0)

// ---------------------------------------------------------------------------------------------
// k_eq_delay: equalizeBurst's first step (sigProcLib.cpp:1352-1356 after Transceiver.cpp:346): scaleVector(burst, 1/amp)
//   then delayVector(-(TOA - chanOffset)) (:573-616) at one sample per symbol, every sample kept (the equaliser wants the
//   whole burst).  SIXTEEN lanes per burst, four bursts per wave: a wave per burst (k_demod<1,RAW>) spent more
//   instructions on its bookkeeping than on the 157 x 21 products and kept a third of its lanes idle -- 600 instructions
//   per burst, issue-bound at 44 us per 65,536 bursts.  Here a lane owns TEN CONSECUTIVE outputs, whose 21-tap windows
//   overlap: 30 LDS reads feed 210 multiply-adds (the burst sits in LDS, scaled, between zero pads that stand for
//   "taps outside the vector are skipped", :584-590: +-0 products).  Results go back through LDS so that the stores are
//   coalesced.  Same terms in the same order as delayVector: j ascending over the 21 real taps; the taps come from the
//   sinc grid when the fraction lies on it (always after peakDetect), else from the table sinc as the reference computes them.
// ---------------------------------------------------------------------------------------------
template <typename SMP>
__global__ __launch_bounds__(256) void k_eq_delay(const TrxTables *__restrict__ T, const void *__restrict__ samples,
                                                  const int32_t *__restrict__ offset, const int32_t *__restrict__ length, int B,
                                                  const cx *__restrict__ amp_in, const float *__restrict__ toa_in,
                                                  const uint8_t *__restrict__ flags, int need_mask, cx *__restrict__ xd,
                                                  int xstride) {
  constexpr int PADL = 16, ROW = 208, OPL = 10;            // 16 lanes x OPL outputs >= 157; windows span [PADL - 10, PADL + 169]
  static_assert(PADL >= 10 && PADL + 16 * OPL - 1 + 10 < ROW, "every window stays inside the row");
  __shared__ __attribute__((aligned(16))) cx rows[16][ROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hl = lane & 15, slot = wave * 4 + (lane >> 4);
  const int b = blockIdx.x * 16 + slot;
  const bool live = b < B;
  int off = 0, N = 0;
  cx amp = mk(1.0f, 0.0f);
  float toa = 0.0f;
  uint8_t fl = 0;
  if (live) { off = offset[b]; N = length[b]; amp = amp_in[b]; toa = toa_in[b]; if (flags) fl = flags[b]; }
  bool enabled = live && (off >= 0) && (N >= 92) && (N <= 157) && (fabsf(toa) <= 4096.0f);   // k_demod's gate (also rejects NaN/inf)
  if (flags) enabled = enabled && (need_mask ? ((fl & need_mask) == need_mask) : (fl != 0));
  if (!enabled) N = 0;
  cx *S = rows[slot];
  // ---- the burst's loads first: sample n = hl + 16 i ----
  typename SMP::raw_t v[OPL];
#pragma unroll
  for (int i = 0; i < OPL; i++) {
    const int n = hl + 16 * i;
    v[i] = n < N ? SMP::ldraw(samples, (long long)off + n) : SMP::zero();
  }
  const cx inv = cdiv(mk(1.0f, 0.0f), amp);                // ((complex)1.0)/channel (Transceiver.cpp:346 / :1066)
  const float delay = -toa;                                // delayVector bookkeeping (:577-582)
  const int io = (int)floorf(delay);
  const float frac = delay - (float)io;
  const bool filt = fabs((double)frac) > 1e-2;
  float tp[21];
  {
    const float f512 = frac * 512.0f;
    const int f = (int)f512;
    const bool grid = f < 512 && (float)f == f512;          // (frac can round to exactly 1.0 for a tiny negative delay: off the grid)
    const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[f & 511]);
    float g[24];
#pragma unroll
    for (int q = 0; q < 6; q++) { const float4 r4 = row[q]; g[4 * q] = r4.x; g[4 * q + 1] = r4.y; g[4 * q + 2] = r4.z; g[4 * q + 3] = r4.w; }
#pragma unroll
    for (int j = 0; j < 21; j++) tp[j] = g[j];
    if (__any(!grid)) {
      // off the grid (never after peakDetect): tap j = sinc(pi*((j - 10) - frac)) (:588), computed by lanes j and j - 16 of the burst
      const float tv0 = dev_sinc(T->sinT, TRX_PI_F * ((float)(hl - 10) - frac));
      const float tv1 = dev_sinc(T->sinT, TRX_PI_F * ((float)(hl + 6) - frac));
      const int first = lane & ~15;
#pragma unroll
      for (int j = 0; j < 21; j++) {
        const float tj = j < 16 ? __shfl(tv0, first + j, 64) : __shfl(tv1, first + j - 16, 64);
        tp[j] = grid ? g[j] : tj;
      }
    }
  }
  // ---- the scaled burst between zeros, already moved by the integer delay: sample n at S[PADL + n + io] (what falls
  //      outside the row lies outside every window below) ----
#pragma unroll
  for (int i = 0; i < ROW / 16; i++) S[hl + 16 * i] = mk(0, 0);
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < OPL; i++) {
    const int n = hl + 16 * i, p = PADL + n + io;
    if (n < N && p >= 0 && p < ROW) S[p] = cmul(SMP::widen(v[i]), inv);   // scaleVector (:713-723)
  }
  wave_lds_fence();
  // ---- outputs m = OPL*hl + i: shifted[m] = filtered[m - io] inside [0, N), else 0 (:597-613); tap j of output m meets
  //      position PADL + m + 10 - j = w[i + 20 - j] ----
  cx y[OPL];
  {
    // packed float32 pairs (one v_pk_mul_f32 + one v_pk_add_f32 per complex-by-real multiply-add, each half rounded on its own:
    // the same values as cmulr + cadd in half the instructions -- this kernel is bound by what its few waves can issue)
    const cx *W = S + (PADL + OPL * hl - 10);
    v2f w[OPL + 20], tp2[21];
#pragma unroll
    for (int k = 0; k < OPL + 20; k++) w[k] = pk(W[k]);
#pragma unroll
    for (int j = 0; j < 21; j++) { tp2[j].x = tp[j]; tp2[j].y = tp[j]; }
#pragma unroll
    for (int i = 0; i < OPL; i++) {
      v2f acc = pk(mk(0, 0));
#pragma unroll
      for (int j = 0; j < 21; j++) acc = pk_cadd(acc, pk_mul(w[i + 20 - j], tp2[j]));   // convolve(..., NO_DELAY), j ascending (:590)
      const int t = OPL * hl + i - io;
      const v2f r = filt ? acc : w[i + 10];
      y[i] = (t >= 0 && t < N) ? mk(r.x, r.y) : mk(0, 0);
    }
  }
  // ---- through LDS again, so that a store instruction writes 16 consecutive samples of each burst ----
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < OPL; i++) S[OPL * hl + i] = y[i];
  wave_lds_fence();
  // The whole row is written: the delayed burst's N samples, zeros from N to the row's end (and a row of zeros for a burst the
  // gate refused).  The equaliser's feed-forward sums then meet a zero SAMPLE wherever the reference skips a term beyond the
  // burst -- +0 + (+-0) = +0, the same value -- and carry no range check per tap (k_eq_dfe2's producer).
  cx *out = xd + (size_t)(live ? b : 0) * xstride;
#pragma unroll
  for (int i = 0; i < OPL; i++) {
    const int m = hl + 16 * i;
    if (live && m < xstride) out[m] = m < N ? S[m] : mk(0, 0);
  }
}

// designDFE(channelResponse, SNRestimate, Nf = 7, ...) (sigProcLib.cpp:1246-1340), nu = 5: fully unrolled in registers.
// chan: the six channel taps (already scaled by 1/amp, Transceiver.cpp:346); w: feed-forward, bq: feedback taps.
__device__ __forceinline__ void design_dfe7(const cx (&chan)[6], float snr, cx (&w)[7], cx (&bq)[5]) {
  // Complex arithmetic on packed float32 pairs (pk_cmul / pk_cadd / pk_csub of trxsig_dev.h: Complex.h's products and sums, each
  // rounded on its own); norms, divisions and square roots stay scalar.  A lane per channel estimate: what bounds it is the
  // number of instructions one wave can issue.
  constexpr int Nf = 7, nu = 5;
  auto conj2 = [](v2f z) { v2f r; r.x = z.x; r.y = -z.y; return r; };
  auto nrm = [](v2f z) { return z.y * z.y + z.x * z.x; };    // Complex::norm2 (Complex.h:119)
  v2f G0[Nf], G1[Nf];
#pragma unroll
  for (int k = 0; k < Nf; k++) { G0[k] = pk(mk(0, 0)); G1[k] = pk(mk(0, 0)); }
  G0[0] = pk(mk((float)(1.0 / (double)sqrtf(snr)), 0.0f));  // :1261
#pragma unroll
  for (int j = 0; j <= nu; j++) G1[j] = pk(mk(chan[j].r, -chan[j].i));
  v2f Lu[Nf - 1][Nf - 1];                                   // L[i][j], i < j <= Nf-1, stored at [i][j-i-1]
  v2f Lfb[nu];                                              // L[Nf-1][Nf .. Nf+nu-1]
  float d = 0.0f;
#pragma unroll
  for (int i = 0; i < Nf; i++) {
    d = nrm(G0[0]) + nrm(G1[0]);                            // :1272
    const v2f g0c = conj2(G0[0]), g1c = conj2(G1[0]);
#pragma unroll
    for (int k = 1; k < Nf; k++) {                          // *Lptr = (G0[k]*conj(G0[0]) + G1[k]*conj(G1[0]))/d (:1277)
      const int col = i + k;
      const bool need = (i < Nf - 1) ? (col <= Nf - 1) : (col >= Nf && col < Nf + nu);
      if (need) {
        const v2f tt = pk_cadd(pk_cmul(G0[k], g0c), pk_cmul(G1[k], g1c));
        v2f v; v.x = tt.x / d; v.y = tt.y / d;
        if (i < Nf - 1) Lu[i][k - 1] = v; else Lfb[col - Nf] = v;
      }
    }
    v2f kk;                                                 // G1[0] / G0[0] = G1[0] * G0[0].inv() (:1282; Complex.h:85, 154-160)
    {
      const float n = nrm(G0[0]);
      v2f inv; inv.x = G0[0].x / n; inv.y = -G0[0].y / n;
      kk = pk_cmul(G1[0], inv);
    }
    if (i != Nf - 1) {
      v2f G0n[Nf], G1n[Nf];
      const v2f kc = conj2(kk);
      v2f km; km.x = kk.x * -1.0f; km.y = kk.y * -1.0f;      // k * -1
#pragma unroll
      for (int q = 0; q < Nf; q++) G0n[q] = pk_cadd(pk_cmul(G1[q], kc), G0[q]);      // :1285-1287
#pragma unroll
      for (int q = 0; q < Nf; q++) G1n[q] = pk_cadd(pk_cmul(G0[q], km), G1[q]);      // :1289-1291
#pragma unroll
      for (int q = 0; q < Nf - 1; q++) G1n[q] = G1n[q + 1];                           // delayVector(G1new,-1) (:1292)
      G1n[Nf - 1] = pk(mk(0, 0));
      const v2f sc = pk(mk((float)(1.0 / (double)sqrtf((float)(1.0 + (double)nrm(kk)))), 0.0f));   // :1294-1295
#pragma unroll
      for (int q = 0; q < Nf; q++) { G0[q] = pk_cmul(G0n[q], sc); G1[q] = pk_cmul(G1n[q], sc); }
    }
  }
#pragma unroll
  for (int j = 0; j < nu; j++) {                            // :1301-1304: * -1, conj
    const v2f t1 = pk_cmul(Lfb[j], pk(mk(-1.0f, 0.0f)));
    bq[j] = mk(t1.x, -t1.y);
  }
  v2f v[Nf];
  v[Nf - 1] = pk(mk(1.0f, 0.0f));
#pragma unroll
  for (int k = Nf - 2; k >= 0; k--) {                       // :1310-1319
    v2f vk = pk(mk(0, 0));
#pragma unroll
    for (int j = k + 1; j < Nf; j++) vk = pk_csub(vk, pk_cmul(v[j], Lu[k][j - k - 1]));
    v[k] = vk;
  }
#pragma unroll
  for (int i = 0; i < Nf; i++) {                            // :1323-1335
    v2f wi = pk(mk(0, 0));
    const int endPt = (nu < (Nf - 1 - i)) ? nu : (Nf - 1 - i);
#pragma unroll
    for (int k = 0; k < Nf; k++)
      if (k < endPt + 1) wi = pk_cadd(wi, pk_cmul(v[i + k < Nf ? i + k : Nf - 1], pk(mk(chan[k < 6 ? k : 5].r, -chan[k < 6 ? k : 5].i))));
    w[i] = mk(wi.x / d, wi.y / d);
  }
}

// FIXT > 0 (round 4): the 52M windowed geometry with maxTOA = FIXT KNOWN AT COMPILE TIME (2 FIXT + 1 lags from lag 20 - FIXT on, a
// 26-sample window): every tap index is a constant, so the zero pads -- 20 of the 58 LDS rows -- are not needed (a term that would meet
// a pad is simply not formed: it would add +-0), and fp16 samples are parked as stored (4 bytes).  18 KB instead of 29 per 64 bursts:
// EIGHT one-wave workgroups per CU instead of five -- the kernel's waves wait on each other's instruction latencies, and what was
// missing was a second wave per SIMD, which the LDS, not the batch size, was denying.
template <int NCMAX, int NXMAX, typename SMP, int FIXT = 0>  // correlation lags / window samples kept per burst; sample storage
__global__ __launch_bounds__(64) void k_eq_detect(const TrxTables *__restrict__ T, const void *__restrict__ samples,
                                                  const int32_t *__restrict__ offset,
                                                  const int32_t *__restrict__ length, int B, int tsc,
                                                  float detect_thresh, float energy_thresh, int variant52m,
                                                  int max_toa, uint8_t *__restrict__ flags,
                                                  cx *__restrict__ amp_out, float *__restrict__ toa_out,
                                                  float *__restrict__ toa_eq, cx *__restrict__ w_out,
                                                  cx *__restrict__ b_out, float snr_thresh, float snr_value,
                                                  float *__restrict__ chan_off_out, cx *__restrict__ chan_out,
                                                  const uint8_t *__restrict__ enable, const float *__restrict__ snr_in,
                                                  const int32_t *__restrict__ dense_gate, int dense_min) {
  // dense_gate (optional): the number of marked bursts (k_eq_list); this kernel serves the call only when they are MANY
  // (> dense_min: the wave-per-burst kernel launched before it has then left them alone), else it ends at once.
  if (dense_gate && *dense_gate <= dense_min) return;
  // enable (optional): only bursts with enable[b] != 0 are processed, nothing is written for the others (the Transceiver
  // group estimates the channel of the few bursts its replay marks, trxsig_group.hip); snr_in (optional): the SNR
  // estimate per burst.  snr_value > 0: the SNR estimate itself (the Transceiver facade forms it on the host in the reference's
  // double arithmetic, Transceiver.cpp:340); else snr_thresh >= 0: the threshold that enters
  // SNR = |amp|^2/(thr^2+1); else energy_thresh.  chan_off_out (optional): chanRespOffset (:343).
  // LDS (one wave per workgroup, a column per lane), rows of 64 complex:
  //   xp  rows [0, XROWS): the correlation window between zero pads (row PADF + a = window sample a; zeros from La on);
  //   cp  the correlation between zero pads: logical row CPAD + i = lag i.  Its front pad IS xp's last CPAD rows (window
  //       samples until the correlation is done, zeroed after it; only delayVector reads the pads);
  //   loc (peakDetect's 26 rows) and, after it, shf (the delayed correlation) reuse xp's first rows.
  // The pads turn the reference's "skip the taps that fall outside" (:480-498, :584-590) into products with zero samples
  // (+-0: adding them never changes a value), so the inner loops carry no bounds checks.  The same storage first serves as
  // the staging area of the burst loads.  29 KB for the 52M window: five workgroups per CU (four would be every one of the
  // 1024 workgroups of a 65,536-burst launch resident only if the dispatcher balanced them perfectly).
  constexpr bool FIX = FIXT > 0;
  constexpr int F_NC = 2 * FIXT + 1, F_START = 20 - FIXT;   // lags kept; window sample under lag 0's last tap (expectedTOAPeak 20, ref52:990-1000)
  static_assert(!FIX || (FIXT >= 3 && FIXT <= 5 && NCMAX == F_NC && NXMAX == 26), "the fixed 52M geometry: maxTOA 3..5, span 5");
  constexpr int PADF = FIX ? 0 : 10, XROWS = PADF + NXMAX, CPAD = FIX ? 0 : 10, CROWS = NCMAX + CPAD;    // cp body + back pad
  constexpr int NSLOT = 20 + NXMAX, PITCH = NSLOT | 1, NPASS = (NSLOT + 63) / 64;
  static_assert(XROWS - CPAD >= 26 && XROWS - CPAD >= NCMAX, "loc / shf must stay clear of cp's front pad");
  constexpr bool RAWST = FIX && sizeof(typename SMP::raw_t) == 4;   // park the samples as stored (fp16 pairs), widen on the way back
  constexpr size_t kWork = sizeof(cx) * 64 * (XROWS + CROWS), kStage = (RAWST ? sizeof(unsigned) : sizeof(float) * 2) * 64 * PITCH;
  __shared__ __attribute__((aligned(16))) char lds_raw[kWork > kStage ? kWork : kStage];
  cx (*xp)[64] = reinterpret_cast<cx (*)[64]>(lds_raw);
  cx (*cp)[64] = xp + (XROWS - CPAD);
  cx (*shf)[64] = xp;
  float *st_re = reinterpret_cast<float *>(lds_raw), *st_im = st_re + 64 * PITCH;
  typename SMP::raw_t *st_raw = reinterpret_cast<typename SMP::raw_t *>(lds_raw);
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < B && (!enable || enable[b < B ? b : 0] != 0);
  if (enable && !__any(live)) return;                      // (wave-uniform: the workgroup is one wave)
#ifdef TRX_EQ_PROBE                                        // clock64() stamps come back through toa_out (tools/eq_probe.py)
  long long pt_[8] = {0};
  int pk_ = 0;
#if TRX_EQ_PROBE == 2                                       // the staging block in detail instead of the phases
#define TRX_STAMP()
#define TRX_STAMP2() pt_[pk_++] = clock64()
#else
#define TRX_STAMP() pt_[pk_++] = clock64()
#define TRX_STAMP2()
#endif
#else
#define TRX_STAMP()
#define TRX_STAMP2()
#endif
#if defined(TRX_EQ_PROBE) && TRX_EQ_PROBE == 2
  pt_[pk_++] = clock64();
#else
  TRX_STAMP();
#endif
  const int off = live ? offset[b] : 0, N = live ? length[b] : 0;
  uint8_t fl = 0;
  cx amp = mk(0, 0);
  float toa = 0.0f;
  const bool good = live && (off >= 0) && (N >= 92) && (N <= 157);
  // ---- window geometry (pure arithmetic, the same for every burst but for the length check) ----
  int ncorr, winStart, La, startIndex;
  unsigned maxTOA = (unsigned)max_toa;
  bool winOk;
  if (!variant52m) {
    ncorr = 36; winStart = 56; La = 36; startIndex = 7;    // NO_DELAY, Lb = 16 (:951-955, 295-300)
    winOk = ncorr <= NCMAX && La <= NXMAX;                 // (the launcher picks an instantiation that fits)
  } else {                                                 // ref52:983-1000
    if (maxTOA < 3) maxTOA = 3;
    unsigned spanTOA = maxTOA;
    if (spanTOA < 5) spanTOA = 5;
    winStart = (int)(66 - spanTOA);
    La = (int)(16 + 2 * spanTOA);
    ncorr = (int)(2 * maxTOA + 1);
    const unsigned expectedTOAPeak = (unsigned)round((double)((T->mid_toa[tsc] + 5.0f) + (float)(size_t)((16 - 1) / 2)));
    startIndex = (int)(expectedTOAPeak - maxTOA);
    winOk = !(ncorr > NCMAX || La > NXMAX || winStart < 0 || winStart + La > N);
  }

  // ---- the burst loads, coalesced: a lane per burst would touch 64 different lines with every load instruction (46
  //      instructions x 64 lines per wave: 28 % of the kernel went there).  Instead the wave fetches burst k's 20 energy
  //      samples and its correlation window with ONE instruction (a lane per sample: ~8 lines), all 64 bursts' loads in
  //      flight together, parks them burst-major in LDS (odd pitch: conflict-free both ways) and every lane then reads
  //      its own burst's samples back. ----
  cx ev[20], wv[NXMAX];
  {
    const int step = variant52m ? 4 : 1;
    // No load is predicated per burst (a mask per load cost ~12 instructions with VALU -> SALU -> exec round trips: the 64
    // loads took 12 k cycles to ISSUE): a burst that is not good reads the wave's first good burst instead, and an index
    // beyond a burst's end (only possible when its window does not fit, which is refused below) is clamped to its last
    // sample -- what such loads return is never used.
    const unsigned long long goodm = __ballot(good);
    TRX_STAMP2();                                          // (2: 1) offsets and lengths are here
    const int safe_off = __builtin_amdgcn_readlane(off, goodm ? (int)__builtin_ctzll(goodm) : 0);
    int offv = good ? off : safe_off, nm1v = good ? N - 1 : 0;
    // (readlane below takes these from EVERY lane, also from lanes the branch around the loads masks off: pin them here, or
    // the compiler sinks their computation under that mask and the masked lanes' registers hold whatever was there)
    asm volatile("" : "+v"(offv), "+v"(nm1v));
#pragma unroll
    for (int p = 0; p < NPASS; p++) {
      const int slot = lane + 64 * p;
      const bool is_win = slot >= 20;
      const bool slot_ok = slot < 20 + La && slot < NSLOT;
      int idx = is_win ? winStart + (slot - 20) : slot * step;
      idx = idx < 0 ? 0 : idx;
      typename SMP::raw_t v[64];                          // as stored; widened only once every load has been issued
#pragma unroll
      for (int k = 0; k < 64; k++) v[k] = SMP::zero();
      if (slot_ok && goodm) {                              // (one mask for all 64 loads: it does not depend on the burst)
#pragma unroll
        for (int k = 0; k < 64; k++) {
          const int off_k = __builtin_amdgcn_readlane(offv, k), nm1_k = __builtin_amdgcn_readlane(nm1v, k);
          v[k] = SMP::ldraw(samples, (long long)off_k + (idx < nm1_k ? idx : nm1_k));
        }
      }
      TRX_STAMP2();                                        // (2: 2) loads issued
      if (slot < NSLOT) {
        if (RAWST) {
#pragma unroll
          for (int k = 0; k < 64; k++) st_raw[k * PITCH + slot] = v[k];
        } else {
#pragma unroll
          for (int k = 0; k < 64; k++) { const cx f = SMP::widen(v[k]); st_re[k * PITCH + slot] = f.r; st_im[k * PITCH + slot] = f.i; }
        }
      }
    }
    wave_lds_fence();
    TRX_STAMP2();                                          // (2: 3) loads complete, parked in LDS
    if (RAWST) {
#pragma unroll
      for (int i = 0; i < 20; i++) ev[i] = SMP::widen(st_raw[lane * PITCH + i]);
#pragma unroll
      for (int a = 0; a < NXMAX; a++) wv[a] = SMP::widen(st_raw[lane * PITCH + 20 + a]);   // zeros past La
    } else {
#pragma unroll
      for (int i = 0; i < 20; i++) ev[i] = mk(st_re[lane * PITCH + i], st_im[lane * PITCH + i]);
#pragma unroll
      for (int a = 0; a < NXMAX; a++) wv[a] = mk(st_re[lane * PITCH + 20 + a], st_im[lane * PITCH + 20 + a]);   // zeros past La
    }
    wave_lds_fence();                                      // the staging area is dead: xp / cp take its place
    TRX_STAMP2();                                          // (2: 4) read back
  }
  if (!live) return;                                       // no barriers below: each lane owns its columns
  if (!good) {
    flags[b] = TRXSIG_F_BADLEN; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f;
    return;
  }
  // ---- energyDetect (:916-932; the 52M variant strides by 4, ref52:946-963) ----
  {
    float energy = 0.0f;
#pragma unroll
    for (int i = 0; i < 20; i++) energy += norm2(ev[i]);
    const bool ok = energy_thresh < 0.0f || energy / (float)20u > energy_thresh * energy_thresh;
    if (!ok) { flags[b] = 0; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f; return; }
    fl = TRXSIG_F_ENERGY;
  }
  TRX_STAMP();                                             // 1: energy
  // ---- correlation ----
  if (!winOk) {
    flags[b] = TRXSIG_F_BADLEN; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f;
    return;
  }
  // the fixed geometry is what the launcher promised (else: the checked loops below, which need no pads either)
  const bool fixed_ok = FIX && variant52m && maxTOA == (unsigned)FIXT && startIndex == F_START && ncorr == F_NC && La == NXMAX;
#pragma unroll
  for (int a = 0; a < PADF; a++) xp[a][lane] = mk(0, 0);
#pragma unroll
  for (int a = 0; a < NXMAX; a++) xp[PADF + a][lane] = wv[a];   // the lane's LDS column (zeros from La on)
#pragma unroll
  for (int a = 0; a < CPAD; a++) cp[CPAD + ncorr + a][lane] = mk(0, 0);
  v2f ctap[16];                                              // (packed float32 pairs from here on: see trxsig_dev.h)
#pragma unroll
  for (int j = 0; j < 16; j++) ctap[j] = pk(T->mid_ctap[tsc][15 - j]);
  // every tap index t - j of every lag lies in [-PADF, NXMAX): no checks (always so for the geometries above)
  const bool padded = !FIX && startIndex - 15 >= -PADF && startIndex + ncorr - 1 < NXMAX;
  if (FIX && fixed_ok) {
    // every index a constant: straight from the window's registers (F_START + i - j lies in [F_START - 15, F_START + F_NC - 1] = inside the window)
#pragma unroll
    for (int i = 0; i < F_NC; i++) {
      v2f sum = pk(mk(0, 0));
#pragma unroll
      for (int j = 0; j < 16; j++) sum = pk_cadd(sum, pk_cmul(pk(wv[(F_START + i - j >= 0 && F_START + i - j < NXMAX) ? F_START + i - j : 0]), ctap[j]));
      cp[i][lane] = unpk(sum);
    }
  } else if (padded) {
    for (int i = 0; i < ncorr; i++) {
      const cx (*row)[64] = xp + (PADF + startIndex + i);
      v2f sum = pk(mk(0, 0));
#pragma unroll
      for (int j = 0; j < 16; j++) sum = pk_cadd(sum, pk_cmul(pk(row[-j][lane]), ctap[j]));   // tmp[j] = conj(mid[15-j]) (:480-498), j ascending
      cp[CPAD + i][lane] = unpk(sum);
    }
  } else {
    for (int i = 0; i < ncorr; i++) {
      const int t = startIndex + i;
      v2f sum = pk(mk(0, 0));
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int ai = t - j;
        if (ai >= 0 && ai < La) sum = pk_cadd(sum, pk_cmul(pk(xp[PADF + ai][lane]), ctap[j]));
      }
      cp[CPAD + i][lane] = unpk(sum);
    }
  }
#pragma unroll
  for (int a = 0; a < CPAD; a++) cp[a][lane] = mk(0, 0);    // the window is dead: its last rows become cp's front pad
  static_assert(!FIX || F_START - 15 >= 0, "the fixed geometry's taps stay inside the window");

  TRX_STAMP();                                             // 2: correlation
  // ---- peakDetect (:663-711) ----
  float maxP = 0.0f, maxIndex = -1.0f;
  for (int i = 0; i < ncorr; i++) {
    const float p = norm2(cp[CPAD + i][lane]);
    if (p > maxP) { maxP = p; maxIndex = (float)i; }
  }
  {
    // the bisection itself as k_tsc_peak runs it (peak_bisect: both candidate sinc rows of the next step prefetched
    // with 16-byte loads while this step computes; the eq_interp form made two dependent gather round trips per step).
    // loc = corr[M-12 .. M+11] with zeros where interpolatePoint skips (lag < 0, lag > n-2); the window's LDS is free now.
    cx (*loc)[64] = xp;
    const int M = (int)maxIndex;
#pragma unroll
    for (int j = 0; j < 24; j++) {
      const int lag = M - 12 + j;
      loc[j][lane] = (lag >= 0 && lag <= ncorr - 2) ? cp[CPAD + (lag < 0 ? 0 : (lag > NCMAX - 1 ? NCMAX - 1 : lag))][lane] : mk(0, 0);
    }
    loc[24][lane] = mk(0, 0);
    loc[25][lane] = mk(0, 0);
    float peakIx;
    amp = peak_bisect<64>(T->sinc_grid, loc, lane, M, &peakIx);
    toa = peakIx;
  }

  TRX_STAMP();                                             // 3: peakDetect
  // ---- analyzeTrafficBurst tail (:961-1035) ----
  bool detected = false;
  float chanOff = 0.0f;
  cx chan[6];
  if ((toa < 0.0f) || (toa > (float)ncorr)) {
    amp = mk(0, 0);
  } else {
    const int p = (int)rintf(toa);
    float valley = 0.0f;
    int numRms = 0;
    for (int i = 2; i <= 5; i++) {
      if (p - i >= 0) { valley += norm2(cp[CPAD + p - i][lane]); numRms++; }
      if (p + i < ncorr) { valley += norm2(cp[CPAD + p + i][lane]); numRms++; }
    }
    if (numRms < 2) {
      amp = mk(0, 0);
    } else {
      const float RMS = (float)((double)sqrtf(valley / (float)numRms) + 0.00001);
      const float peakToMean = sqrtf(norm2(amp)) / RMS;
      amp = cdiv(amp, T->mid_gain[tsc]);
      float TOAoffset;
      if (!variant52m) {
        toa = toa - T->mid_toa[tsc];
        toa = toa - 10.0f;
        TOAoffset = T->mid_toa[tsc] + 10.0f;
      } else {
        toa = toa - (float)maxTOA;
        TOAoffset = (float)maxTOA;
      }
      detected = peakToMean > detect_thresh;
      if (detected) {
        // delayVector(corr, -TOA) (:573-616) on the lane's column
        const float delay = -toa;
        const int io = (int)floorf(delay);
        const float frac = delay - (float)io;
        const cx (*src)[64] = cp + CPAD;
        if (fabs((double)frac) > 1e-2) {
          v2f row[12];                                     // two consecutive taps per register pair
          {
            const float4 *r4 = reinterpret_cast<const float4 *>(T->sinc_grid[(int)(frac * 512.0f) & 511]);
#pragma unroll
            for (int q = 0; q < 6; q++) { const float4 v4 = r4[q]; row[2 * q].x = v4.x; row[2 * q].y = v4.y; row[2 * q + 1].x = v4.z; row[2 * q + 1].y = v4.w; }
          }
          if (FIX) {
            // no pads: a tap that would meet one is not formed (it would add +-0 to a sum that starts at +0: the same value).  With the
            // promised geometry every index is a constant and the correlation comes from registers; else the checks are live.
            v2f cr[NCMAX];
#pragma unroll
            for (int i = 0; i < NCMAX; i++) cr[i] = pk(cp[i][lane]);
#pragma unroll
            for (int t = 0; t < NCMAX; t++) {
              v2f sum = pk(mk(0, 0));
#pragma unroll
              for (int j = 0; j < 21; j++) {
                const int q = t + 10 - j;
                if (q >= 0 && q < NCMAX && q < ncorr)
                  sum = (j & 1) ? pk_cadd(sum, pk_mul_tap<1>(cr[q < 0 ? 0 : (q >= NCMAX ? NCMAX - 1 : q)], row[j >> 1]))
                                : pk_cadd(sum, pk_mul_tap<0>(cr[q < 0 ? 0 : (q >= NCMAX ? NCMAX - 1 : q)], row[j >> 1]));
              }
              if (t < ncorr) shf[t][lane] = unpk(sum);
            }
          } else {
            for (int t = 0; t < ncorr; t++) {              // taps t + 10 - j outside [0, ncorr) meet cp's zero pads
              const cx (*crow)[64] = cp + (CPAD + t + 10);
              v2f sum = pk(mk(0, 0));
#pragma unroll
              for (int j = 0; j < 21; j++)
                sum = (j & 1) ? pk_cadd(sum, pk_mul_tap<1>(pk(crow[-j][lane]), row[j >> 1])) : pk_cadd(sum, pk_mul_tap<0>(pk(crow[-j][lane]), row[j >> 1]));
              shf[t][lane] = unpk(sum);
            }
          }
          src = shf;
        }
        // integer shift folded into the reads: w[k] = src[k - io] inside [0,n), else 0
        auto wv = [&](int k) {
          const int q = k - io;
          return (q >= 0 && q < ncorr) ? src[q][lane] : mk(0, 0);
        };
        float maxEnergy = -1.0f;
        int maxI = -1;
        for (int i = 0; i < 7; i++) {                      // :1012-1021
          const float st = TOAoffset + (float)(i - 5);
          if (st + (float)6u > (float)(unsigned)ncorr) continue;
          if (st < 0.0f) continue;
          const int s0 = (int)floorf(st);
          float energy = 0.0f;
          for (int k = 0; k < 6; k++) energy += norm2(wv(s0 + k));
          if ((double)energy > 0.95 * (double)maxEnergy) { maxI = i; maxEnergy = energy; }
        }
        const int s0 = (int)floorf(TOAoffset + (float)(maxI - 5));
        const cx ginv = cdiv(mk(1.0f, 0.0f), T->mid_gain[tsc]);
#pragma unroll
        for (int k = 0; k < 6; k++) chan[k] = cmul(wv(s0 + k), ginv);   // :1024-1025
        chanOff = (float)(5 - maxI);                       // :1029
      }
    }
  }
  fl |= detected ? TRXSIG_F_DETECT : 0;
  flags[b] = fl;
  amp_out[b] = amp;
  toa_out[b] = toa;
  toa_eq[b] = toa - chanOff;                               // equalizeBurst(..., TOA - chanRespOffset, ...)
  if (chan_off_out) chan_off_out[b] = chanOff;
  if (chan_out) {                                          // analyzeTrafficBurst's channelResponse (:1024-1025), zeros if not detected
#pragma unroll
    for (int k = 0; k < 6; k++) chan_out[(size_t)b * 6 + k] = detected ? chan[k] : mk(0, 0);
  }
  if (!detected) return;
  TRX_STAMP();                                             // 4: tail, delayVector, channel pick

  // ---- Transceiver.cpp:341-347: SNR, scaleVector(chan, 1/amp), designDFE(chan, SNR, 7) (:1246-1340) ----
  const float thr = snr_thresh >= 0.0f ? snr_thresh : (energy_thresh < 0.0f ? 0.0f : energy_thresh);
  const float snr = snr_in ? snr_in[b] : (snr_value > 0.0f ? snr_value : (float)((double)norm2(amp) / ((double)(thr * thr) + 1.0)));
  const cx ainv = cdiv(mk(1.0f, 0.0f), amp);
#pragma unroll
  for (int k = 0; k < 6; k++) chan[k] = cmul(chan[k], ainv);

  cx w7[7], bq[5];
  design_dfe7(chan, snr, w7, bq);
#pragma unroll
  for (int i = 0; i < 7; i++) w_out[(size_t)b * 7 + i] = w7[i];
  constexpr int nu = 5;
#pragma unroll
  for (int j = 0; j < nu; j++) b_out[(size_t)b * nu + j] = bq[j];
#ifdef TRX_EQ_PROBE
  pt_[5] = clock64();                                      // 5: designDFE + taps
  {
    long long v_ = 0;
    for (int k = 1; k < 8; k++) if ((b & 7) == k) v_ = pt_[k] - pt_[0];
    if ((b & 7) == 6) v_ = pt_[0] & 0xFFFFFF;              // absolute start / end (24 bits): the launch's spread over time
    if ((b & 7) == 7) v_ = pt_[5] & 0xFFFFFF;
    toa_out[b] = (float)v_;
  }
#endif
#undef TRX_STAMP
#undef TRX_STAMP2
}

// ---------------------------------------------------------------------------------------------
// k_eq_detect52 (round 4): k_eq_detect's job for the ONE geometry config 5 runs -- the 52M window with maxTOA = 4 (nine lags from
// lag 16 on, a 26-sample window from sample 61, expectedTOAPeak = 20: ref52:983-1000; the launcher checks the last on the host's
// copy of the tables) -- with 256 threads = four waves, still a lane per burst, around ONE copy of the sinc table in LDS.
// What the general kernel spends and this one does not (tools/eq_probe.py: 67 k cycles per wave, 23 k of them in peakDetect):
//   * the bisection's sinc rows came from L2 with twelve 16-byte gathers per step, every lane another row (64 lines per
//     instruction: the texture path, shared by the CU's four waves, was the bound).  Here they are ten 4-byte LDS reads;
//   * with nine lags interpolatePoint's loop (:646-657) runs over the SAME eight lags i = 0..7 at every point the bisection
//     visits (start = max(0, floor(ix) - 10) = 0 and end = min(floor(ix) + 11, 8) = 8 for floor(ix) in [-3, 9], which
//     M - 2 <= floor(ix) <= M + 1, -1 <= M <= 8 guarantees), tap i - floor(ix) + 10 in [1, 20]: eight multiply-adds from
//     registers per point in place of twenty-one (thirteen of them products with the zeros that stood for skipped terms);
//     early, late and final point of a step share floor(early) after step 0, hence one run of ten taps serves all three;
//   * the correlation never leaves registers for the argmax, the bisection and delayVector; valley, channel pick and the
//     integer shift read their dynamic lags from the lane's LDS column with all reads in flight at once.
// Same terms in the same order as k_eq_detect (and the reference): the results are the same values.
// LDS: 48 KB table + per wave max(staging area, 18 rows of 64 complex): 96 KB (fp16 storage) / 144 KB (complex float) -- one
// workgroup per CU, as many waves per SIMD as the 64-thread form has at 65,536 bursts.
// ---------------------------------------------------------------------------------------------
template <typename SMP>
struct EqDetect52 {
  static constexpr int MT = 4, NC = 2 * MT + 1, NX = 26, START = 20 - MT, WIN0 = 61;
  static constexpr bool RAWST = sizeof(typename SMP::raw_t) == 4;
  // fp16 storage: a burst's first NCHUNK 16-byte pieces (samples 0 .. 4 NCHUNK - 1: the energy samples 0, 4, .., 76 and the window
  // 61 .. 86) are parked as stored; complex float storage: the 46 samples themselves, widened
  static constexpr int NCHUNK = (WIN0 + NX + 3) / 4;
  static constexpr int NSLOT = RAWST ? 4 * NCHUNK : 20 + NX, PITCH = NSLOT | 1;
  static constexpr size_t kStage = (RAWST ? sizeof(unsigned) : sizeof(float) * 2) * 64 * PITCH;
  static constexpr size_t kCols = sizeof(cx) * 64 * (2 * NC);
  static constexpr size_t kSlice = ((kStage > kCols ? kStage : kCols) + 15) & ~(size_t)15;
  static constexpr size_t kLds = sizeof(SincLds) + 4 * kSlice;
};

template <typename SMP>
__global__ __launch_bounds__(256) void k_eq_detect52(const TrxTables *__restrict__ T, const void *__restrict__ samples,
                                                     const int32_t *__restrict__ offset, const int32_t *__restrict__ length, int B, int tsc,
                                                     float detect_thresh, float energy_thresh, uint8_t *__restrict__ flags,
                                                     cx *__restrict__ amp_out, float *__restrict__ toa_out, float *__restrict__ toa_eq,
                                                     cx *__restrict__ w_out, cx *__restrict__ b_out, float snr_thresh, float snr_value,
                                                     float *__restrict__ chan_off_out, cx *__restrict__ chan_out,
                                                     const uint8_t *__restrict__ enable, const float *__restrict__ snr_in) {
  typedef EqDetect52<SMP> G;
  constexpr int NC = G::NC, NX = G::NX, PITCH = G::PITCH, NSLOT = G::NSLOT;
  extern __shared__ __attribute__((aligned(16))) char lds52[];
  SincLds &stab = *reinterpret_cast<SincLds *>(lds52);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char *mine = lds52 + sizeof(SincLds) + (size_t)wave * G::kSlice;
  cx (*cp)[64] = reinterpret_cast<cx (*)[64]>(mine);        // the correlation, a column per lane; shf: its delayed copy
  cx (*shf)[64] = cp + NC;
  float *st_re = reinterpret_cast<float *>(mine), *st_im = st_re + 64 * PITCH;
  typename SMP::raw_t *st_raw = reinterpret_cast<typename SMP::raw_t *>(mine);
  const int b = blockIdx.x * 256 + tid;
  const bool live = b < B && (!enable || enable[b < B ? b : 0] != 0);
  if (enable && !__syncthreads_or(live)) return;            // (workgroup-uniform)
#ifdef TRX_EQ_PROBE
  long long pt_[8] = {0};
  int pk_ = 0;
#if TRX_EQ_PROBE == 2                                       // the staging block in detail instead of the phases
#define TRX_STAMP()
#define TRX_STAMP2() pt_[pk_++] = clock64()
#else
#define TRX_STAMP() pt_[pk_++] = clock64()
#define TRX_STAMP2()
#endif
#else
#define TRX_STAMP()
#define TRX_STAMP2()
#endif
#if defined(TRX_EQ_PROBE) && TRX_EQ_PROBE == 2
  pt_[pk_++] = clock64();
#else
  TRX_STAMP();
#endif
  const int off = live ? offset[b] : 0, N = live ? length[b] : 0;
  // (loaded here, with the offsets: a load issued behind the result stores further down would make the wave wait for those stores to land --
  //  the counter of outstanding memory operations does not tell loads from stores -- before designDFE could start)
  const float snr_pre = (snr_in && live) ? snr_in[b] : 0.0f;
  uint8_t fl = 0;
  cx amp = mk(0, 0);
  float toa = 0.0f;
  const bool good = live && (off >= 0) && (N >= 92) && (N <= 157);
  const bool winOk = G::WIN0 + NX <= N;                     // ref52:993-1000: the window must lie inside the burst
  // ---- the burst loads, coalesced through LDS exactly as k_eq_detect does them (see there) ----
  cx ev[20], wv[NX];
  {
    const unsigned long long goodm = __ballot(good);
    TRX_STAMP2();                                          // (2: 1) offsets and lengths are here
    const int safe_off = __builtin_amdgcn_readlane(off, goodm ? (int)__builtin_ctzll(goodm) : 0);
    int offv = good ? off : safe_off, nm1v = good ? N - 1 : 0;
    asm volatile("" : "+v"(offv), "+v"(nm1v));
    if constexpr (G::RAWST) {
      // fp16 storage: TWO bursts per load instruction, a lane per 16-byte piece (lanes 0..21 burst 2 j, lanes 32..53 burst 2 j + 1):
      // the same lines as a lane per sample would touch, in half the load instructions, four samples per lane.  (A good burst has
      // at least 92 samples: pieces 0 .. NCHUNK - 1 = samples 0..87 lie inside it.)
      static_assert(G::NCHUNK <= 32 && 4 * G::NCHUNK <= 92, "two bursts per wave instruction, inside the shortest burst");
      struct __attribute__((packed, aligned(4))) Piece { unsigned q[4]; };
      const int c = lane & 31, half = lane >> 5;
      const bool piece_ok = c < G::NCHUNK;
      Piece v[32];
#pragma unroll
      for (int j = 0; j < 32; j++) v[j] = Piece{{0u, 0u, 0u, 0u}};
      int off_k[32];                                       // (every lane takes part in the exchange: before the branch)
#pragma unroll
      for (int j = 0; j < 32; j++) off_k[j] = __shfl(offv, 2 * j + half, 64);
      if (piece_ok && goodm) {
#pragma unroll
        for (int j = 0; j < 32; j++)
          v[j] = *reinterpret_cast<const Piece *>(reinterpret_cast<const unsigned *>(samples) + ((long long)off_k[j] + 4 * c));
      }
      TRX_STAMP2();                                        // (2: 2) loads issued
      {                                                    // the table's loads queue behind the bursts': nothing waits for them alone
        float4 tv[3072 / 256];
        sinc_lds_issue<256>(T, tid, tv);
        sinc_lds_store<256>(stab, tid, tv);
      }
      if (piece_ok) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
#pragma unroll
          for (int q = 0; q < 4; q++) st_raw[(2 * j + half) * PITCH + 4 * c + q] = v[j].q[q];
        }
      }
      wave_lds_fence();
      TRX_STAMP2();                                        // (2: 3) loads complete, table and bursts parked
#pragma unroll
      for (int i = 0; i < 20; i++) ev[i] = SMP::widen(st_raw[lane * PITCH + 4 * i]);   // energyDetect strides by 4 (ref52:946-963)
#pragma unroll
      for (int a = 0; a < NX; a++) wv[a] = SMP::widen(st_raw[lane * PITCH + G::WIN0 + a]);
    } else {
      const int slot = lane;
      const bool slot_ok = slot < NSLOT;
      const int idx = slot >= 20 ? G::WIN0 + (slot - 20) : slot * 4;   // energyDetect strides by 4 (ref52:946-963)
      typename SMP::raw_t v[64];
#pragma unroll
      for (int k = 0; k < 64; k++) v[k] = SMP::zero();
      if (slot_ok && goodm) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
          const int off_k = __builtin_amdgcn_readlane(offv, k), nm1_k = __builtin_amdgcn_readlane(nm1v, k);
          v[k] = SMP::ldraw(samples, (long long)off_k + (idx < nm1_k ? idx : nm1_k));
        }
      }
      TRX_STAMP2();                                        // (2: 2) loads issued
      {
        float4 tv[3072 / 256];
        sinc_lds_issue<256>(T, tid, tv);
        sinc_lds_store<256>(stab, tid, tv);
      }
      if (slot_ok) {
#pragma unroll
        for (int k = 0; k < 64; k++) { const cx f = SMP::widen(v[k]); st_re[k * PITCH + slot] = f.r; st_im[k * PITCH + slot] = f.i; }
      }
      wave_lds_fence();
      TRX_STAMP2();                                        // (2: 3) loads complete, table and bursts parked
#pragma unroll
      for (int i = 0; i < 20; i++) ev[i] = mk(st_re[lane * PITCH + i], st_im[lane * PITCH + i]);
#pragma unroll
      for (int a = 0; a < NX; a++) wv[a] = mk(st_re[lane * PITCH + 20 + a], st_im[lane * PITCH + 20 + a]);
    }
    wave_lds_fence();                                      // the staging area is dead: cp / shf take its place
  }
  TRX_STAMP2();                                            // (2: 4) read back
  __syncthreads();                                         // the table is whole; no barrier below (each lane owns its columns)
  TRX_STAMP2();                                            // (2: 5) barrier
  if (!live) return;
  if (!good || !winOk) {
    // (energyDetect comes first in the reference: a short burst that fails it reports 0, not BADLEN -- as k_eq_detect)
    bool e_ok = true;
    if (good) {
      float energy = 0.0f;
#pragma unroll
      for (int i = 0; i < 20; i++) energy += norm2(ev[i]);
      e_ok = energy_thresh < 0.0f || energy / (float)20u > energy_thresh * energy_thresh;
    }
    flags[b] = e_ok ? TRXSIG_F_BADLEN : 0; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f;
    return;
  }
  // ---- energyDetect ----
  {
    float energy = 0.0f;
#pragma unroll
    for (int i = 0; i < 20; i++) energy += norm2(ev[i]);
    const bool ok = energy_thresh < 0.0f || energy / (float)20u > energy_thresh * energy_thresh;
    if (!ok) { flags[b] = 0; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f; return; }
    fl = TRXSIG_F_ENERGY;
  }
  TRX_STAMP();                                             // 1: energy
  // ---- correlation (:480-498): lag i ends on window sample START + i; tmp[j] = conj(mid[15 - j]), j ascending ----
  v2f cr[NC];
  {
    v2f ctap[16];
#pragma unroll
    for (int j = 0; j < 16; j++) ctap[j] = pk(T->mid_ctap[tsc][15 - j]);
#pragma unroll
    for (int i = 0; i < NC; i++) {
      v2f sum = pk(mk(0, 0));
#pragma unroll
      for (int j = 0; j < 16; j++) sum = pk_cadd(sum, pk_cmul(pk(wv[G::START + i - j]), ctap[j]));
      cr[i] = sum;
      cp[i][lane] = unpk(sum);
    }
  }
  static_assert(G::START - 15 >= 0 && G::START + NC - 1 < NX, "every tap of every lag lies inside the window");
  TRX_STAMP();                                             // 2: correlation
  // ---- peakDetect (:663-711) ----
  float maxP = 0.0f, maxIndex = -1.0f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const float p = norm2(unpk(cr[i]));
    if (p > maxP) { maxP = p; maxIndex = (float)i; }
  }
  int e = 0;
  {
    const int M = (int)maxIndex;
    // the ten taps row[f][col0 .. col0 + 9], col0 = 8 - floor(early): late point = taps 0..7, final point 1..8, early point 2..9
    v2f t[5];
    auto load_taps = [&](int f, int col0) {
      const float *rw = &stab.row[f][col0];
#pragma unroll
      for (int k = 0; k < 5; k++) { t[k].x = rw[2 * k]; t[k].y = rw[2 * k + 1]; }
    };
    auto point = [&](auto sh) {                            // interpolatePoint: sum over lags 0..7 of corr[i] * tap[sh + i], i ascending
      constexpr int SH = decltype(sh)::value;
      v2f pt = pk(mk(0, 0));
#pragma unroll
      for (int i = 0; i < NC - 1; i++)
        pt = ((SH + i) & 1) ? pk_cadd(pt, pk_mul_tap<1>(cr[i], t[(SH + i) >> 1])) : pk_cadd(pt, pk_mul_tap<0>(cr[i], t[(SH + i) >> 1]));
      return unpk(pt);
    };
    bool active = true;
    auto decide = [&](int inc) {                           // :690-697
      const float ne = norm2(point(std::integral_constant<int, 2>())), nl = norm2(point(std::integral_constant<int, 0>()));
      if (active) {
        if (ne < nl) e += inc;
        else if (ne > nl) e -= inc;
        else active = false;                               // "else break" (:695)
      }
    };
    load_taps(0, 9 - M);                                   // early = M - 1: floor = M - 1
    decide(256);
    const int col0 = e < 0 ? 10 - M : 9 - M;               // floor(early) = M - 2 once the first step went down, else M - 1: it stays
#pragma unroll 1
    for (int inc = 128; inc >= 1; inc >>= 1) {             // increments 2^-2 .. 2^-9
      load_taps(e & 511, col0);
      decide(inc);
    }
    load_taps(e & 511, col0);                              // early + 1 has the same fractional part
    amp = point(std::integral_constant<int, 1>());
    toa = ((float)(M - 1) + (float)e * 0.001953125f) + 1.0f;   // exact: the reference's +-2^-k steps are exact too
  }
  TRX_STAMP();                                             // 3: peakDetect
  // ---- analyzeTrafficBurst's tail (:961-1035, ref52) ----
  bool detected = false;
  float chanOff = 0.0f;
  cx chan[6];
  if ((toa < 0.0f) || (toa > (float)NC)) {
    amp = mk(0, 0);
  } else {
    const int p = (int)rintf(toa);
    cx vlo[4], vhi[4];
#pragma unroll
    for (int i = 2; i <= 5; i++) {                          // all eight reads in flight
      const int a = p - i, c = p + i;
      vlo[i - 2] = cp[a < 0 ? 0 : (a > NC - 1 ? NC - 1 : a)][lane];
      vhi[i - 2] = cp[c < 0 ? 0 : (c > NC - 1 ? NC - 1 : c)][lane];
    }
    float valley = 0.0f;
    int numRms = 0;
#pragma unroll
    for (int i = 2; i <= 5; i++) {
      if (p - i >= 0) { valley += norm2(vlo[i - 2]); numRms++; }
      if (p + i < NC) { valley += norm2(vhi[i - 2]); numRms++; }
    }
    if (numRms < 2) {
      amp = mk(0, 0);
    } else {
      const float RMS = (float)((double)sqrtf(valley / (float)numRms) + 0.00001);
      const float peakToMean = sqrtf(norm2(amp)) / RMS;
      amp = cdiv(amp, T->mid_gain[tsc]);
      toa = toa - (float)(unsigned)G::MT;
      const float TOAoffset = (float)(unsigned)G::MT;
      detected = peakToMean > detect_thresh;
      if (detected) {
        // delayVector(corr, -TOA) (:573-616): a tap that would meet a lag outside [0, NC) is not formed (it would add +-0)
        const float delay = -toa;
        const int io = (int)floorf(delay);
        const float frac = delay - (float)io;
        const cx (*src)[64] = cp;
        if (fabs((double)frac) > 1e-2) {
          v2f row[12];
          {
            const float4 *r4 = reinterpret_cast<const float4 *>(stab.row[(int)(frac * 512.0f) & 511]);
#pragma unroll
            for (int q = 0; q < 6; q++) { const float4 v4 = r4[q]; row[2 * q].x = v4.x; row[2 * q].y = v4.y; row[2 * q + 1].x = v4.z; row[2 * q + 1].y = v4.w; }
          }
#pragma unroll
          for (int tq = 0; tq < NC; tq++) {
            v2f sum = pk(mk(0, 0));
#pragma unroll
            for (int j = 0; j < 21; j++) {
              const int q = tq + 10 - j;
              if (q >= 0 && q < NC)
                sum = (j & 1) ? pk_cadd(sum, pk_mul_tap<1>(cr[q < 0 ? 0 : (q >= NC ? NC - 1 : q)], row[j >> 1]))
                              : pk_cadd(sum, pk_mul_tap<0>(cr[q < 0 ? 0 : (q >= NC ? NC - 1 : q)], row[j >> 1]));
            }
            shf[tq][lane] = unpk(sum);
          }
          src = shf;
        }
        // integer shift folded into the reads: w[k] = src[k - io] inside [0, NC), else 0
        auto wdyn = [&](int k) {
          const int q = k - io;
          return (q >= 0 && q < NC) ? src[q][lane] : mk(0, 0);
        };
        cx wk[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
          const int q = k - io;
          const cx r = src[q < 0 ? 0 : (q > NC - 1 ? NC - 1 : q)][lane];
          wk[k] = (q >= 0 && q < NC) ? r : mk(0, 0);
        }
        float nk[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) nk[k] = norm2(wk[k]);
        // :1012-1021 with TOAoffset = 4 and nine lags: windows i = 1..4 start on lag i - 1 (i = 0: st < 0; i = 5, 6: st + 6 > 9)
        float maxEnergy = -1.0f;
        int maxI = -1;
#pragma unroll
        for (int i = 1; i <= 4; i++) {
          float energy = 0.0f;
#pragma unroll
          for (int k = 0; k < 6; k++) energy += nk[i - 1 + k];
          if ((double)energy > 0.95 * (double)maxEnergy) { maxI = i; maxEnergy = energy; }
        }
        static_assert(G::MT == 4 && NC == 9, "the window list above");
        const cx ginv = cdiv(mk(1.0f, 0.0f), T->mid_gain[tsc]);
        // (dynamic reads of the lane's column again: a select chain over wk[] makes the compiler put wk[] in scratch.  maxI = -1 --
        //  no window taken, energies that do not compare: NaN -- reads from TOAoffset - 6 on like the reference)
        const int s0 = (int)floorf(TOAoffset + (float)(maxI - 5));
#pragma unroll
        for (int k = 0; k < 6; k++) chan[k] = cmul(wdyn(s0 + k), ginv);   // :1024-1025
        chanOff = (float)(5 - maxI);                       // :1029
      }
    }
  }
  fl |= detected ? TRXSIG_F_DETECT : 0;
  flags[b] = fl;
  amp_out[b] = amp;
  toa_out[b] = toa;
  toa_eq[b] = toa - chanOff;
  if (chan_off_out) chan_off_out[b] = chanOff;
  if (chan_out) {
#pragma unroll
    for (int k = 0; k < 6; k++) chan_out[(size_t)b * 6 + k] = detected ? chan[k] : mk(0, 0);
  }
  if (!detected) return;
  TRX_STAMP();                                             // 4: tail, delayVector, channel pick
  // ---- Transceiver.cpp:341-347: SNR, scaleVector(chan, 1/amp), designDFE(chan, SNR, 7) ----
  const float thr = snr_thresh >= 0.0f ? snr_thresh : (energy_thresh < 0.0f ? 0.0f : energy_thresh);
  const float snr = snr_in ? snr_pre : (snr_value > 0.0f ? snr_value : (float)((double)norm2(amp) / ((double)(thr * thr) + 1.0)));
  const cx ainv = cdiv(mk(1.0f, 0.0f), amp);
#pragma unroll
  for (int k = 0; k < 6; k++) chan[k] = cmul(chan[k], ainv);
  cx w7[7], bq[5];
  design_dfe7(chan, snr, w7, bq);
#pragma unroll
  for (int i = 0; i < 7; i++) w_out[(size_t)b * 7 + i] = w7[i];
#pragma unroll
  for (int j = 0; j < 5; j++) b_out[(size_t)b * 5 + j] = bq[j];
#ifdef TRX_EQ_PROBE
  pt_[5] = clock64();
  {
    long long v_ = 0;
    for (int k = 1; k < 8; k++) if ((b & 7) == k) v_ = pt_[k] - pt_[0];
    if ((b & 7) == 6) v_ = pt_[0] & 0xFFFFFF;
    if ((b & 7) == 7) v_ = pt_[5] & 0xFFFFFF;
    toa_out[b] = (float)v_;
  }
#endif
#undef TRX_STAMP
#undef TRX_STAMP2
}

// designDFE(Nf = 7, nu = 5) with the lanes of a wave (design_dfe7 above is the same computation in one lane): lane q holds G0[q] and
// G1[q]; every iteration of the recursion (:1270-1297) updates the seven columns side by side, element 0 is broadcast, G1's delay by
// one (:1292) is a shift between neighbouring lanes.  Lane k's column quotients L[i][i + k] go to LDS for the back-substitution
// (:1310-1319), which is serial and done by every lane alike; lane i then forms w[i] (:1323-1335).  Every element sees the operations
// design_dfe7 applies to it, in the same order: the same values.  Results: lane i < 7 returns w[i], lane j + 1 (j < 5) returns b[j].
// Lu_s: 36 complex, v_s: 8 complex of the wave's LDS.
__device__ __forceinline__ void design_dfe7_lanes(const cx (&chan)[6], float snr, int lane, cx *Lu_s, cx *v_s, cx &w_mine, cx &b_mine) {
  constexpr int Nf = 7, nu = 5;
  auto conj2 = [](v2f z) { v2f r; r.x = z.x; r.y = -z.y; return r; };
  auto nrm = [](v2f z) { return z.y * z.y + z.x * z.x; };    // Complex::norm2 (Complex.h:119)
  auto bcast0 = [](v2f z) {
    v2f r;
    r.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z.x), 0));
    r.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z.y), 0));
    return r;
  };
  const int q = lane < Nf ? lane : Nf - 1;                   // (lanes beyond the seventh shadow lane 6: their values are never used)
  v2f G0 = pk(mk(0, 0)), G1 = pk(mk(0, 0));
  if (q == 0) G0 = pk(mk((float)(1.0 / (double)sqrtf(snr)), 0.0f));   // :1261
#pragma unroll
  for (int j = 0; j <= nu; j++)
    if (q == j) G1 = pk(mk(chan[j].r, -chan[j].i));
  v2f lfb = pk(mk(0, 0));
  float d = 0.0f;
#pragma unroll
  for (int i = 0; i < Nf; i++) {
    const v2f G00 = bcast0(G0), G10 = bcast0(G1);
    d = nrm(G00) + nrm(G10);                                 // :1272
    const v2f g0c = conj2(G00), g1c = conj2(G10);
    {                                                        // *Lptr = (G0[k]*conj(G0[0]) + G1[k]*conj(G1[0]))/d (:1277), k = this lane
      const v2f tt = pk_cadd(pk_cmul(G0, g0c), pk_cmul(G1, g1c));
      v2f v; v.x = tt.x / d; v.y = tt.y / d;
      const int col = i + q;
      if (i < Nf - 1) { if (q >= 1 && col <= Nf - 1 && lane < Nf) Lu_s[i * 6 + (q - 1)] = unpk(v); }
      else if (q >= 1 && col >= Nf && col < Nf + nu) lfb = v;
    }
    v2f kk;                                                  // G1[0] / G0[0] (:1282)
    {
      const float n = nrm(G00);
      v2f inv; inv.x = G00.x / n; inv.y = -G00.y / n;
      kk = pk_cmul(G10, inv);
    }
    if (i != Nf - 1) {
      const v2f kc = conj2(kk);
      v2f km; km.x = kk.x * -1.0f; km.y = kk.y * -1.0f;
      const v2f G0n = pk_cadd(pk_cmul(G1, kc), G0);          // :1285-1287
      v2f G1n = pk_cadd(pk_cmul(G0, km), G1);                // :1289-1291
      v2f sh;                                                // delayVector(G1new, -1) (:1292): lane q takes lane q + 1's, the last a zero
      sh.x = __shfl_down(G1n.x, 1, 64); sh.y = __shfl_down(G1n.y, 1, 64);
      G1n = (lane < Nf - 1) ? sh : pk(mk(0, 0));
      const v2f sc = pk(mk((float)(1.0 / (double)sqrtf((float)(1.0 + (double)nrm(kk)))), 0.0f));   // :1294-1295
      G0 = pk_cmul(G0n, sc);
      G1 = pk_cmul(G1n, sc);
    }
  }
  {                                                          // :1301-1304: * -1, conj
    const v2f t1 = pk_cmul(lfb, pk(mk(-1.0f, 0.0f)));
    b_mine = mk(t1.x, -t1.y);
  }
  wave_lds_fence();
  v2f v[Nf];
  v[Nf - 1] = pk(mk(1.0f, 0.0f));
#pragma unroll
  for (int k = Nf - 2; k >= 0; k--) {                        // :1310-1319
    v2f vk = pk(mk(0, 0));
#pragma unroll
    for (int j = k + 1; j < Nf; j++) vk = pk_csub(vk, pk_cmul(v[j], pk(Lu_s[k * 6 + (j - k - 1)])));
    v[k] = vk;
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < Nf; k++) v_s[k] = unpk(v[k]);
  }
  wave_lds_fence();
  {                                                          // :1323-1335, i = this lane
    v2f wi = pk(mk(0, 0));
    const int endPt = (nu < (Nf - 1 - q)) ? nu : (Nf - 1 - q);
#pragma unroll
    for (int k = 0; k < Nf - 1; k++)
      if (k < endPt + 1) wi = pk_cadd(wi, pk_cmul(pk(v_s[q + k < Nf ? q + k : Nf - 1]), pk(mk(chan[k].r, -chan[k].i))));
    w_mine = mk(wi.x / d, wi.y / d);
  }
}

// ---------------------------------------------------------------------------------------------
// k_eq_list + k_eq_estimate_wave (round 4): analyzeTrafficBurst(requestChannel) + designDFE for FEW bursts, a WAVE per burst.
// The Transceiver asks for a channel estimate once per timeslot and 51 frames (Transceiver.cpp:313-325): of the group's rows some 2 %
// are marked, one or two per wave of a lane-per-burst kernel -- k_eq_detect<36, 52> then runs 936 waves at one or two active lanes
// each and takes one wave's full latency (46 us with VALU 0 % busy, profiles/r04_pmc_config4_reference_chain.txt), and the one-ARFCN
// object's one-burst call (trxsig_estimate_dfe_batch, B = 1) pays the same.  Here the marked bursts are listed first (k_eq_list: one
// workgroup, a prefix sum over the flags -- the list comes out in the same order every run) and each listed burst gets a wave:
//   lanes 0..35 the 36 lags of the correlation (sigProcLib.cpp:480-498; the Transceiver/ variant: whole 36-lag window from sample 56);
//   the argmax by a wave reduction with the reference's first-maximum rule; peakDetect's bisection SPECULATED in two super-steps
//   (trxsig_bisect.h: every node of the next five / four levels evaluated at once, the decisions replayed along the reference's path);
//   fused_tail (valley, peak-to-mean, TOA bookkeeping); lanes 0..35 the outputs of delayVector on the correlation; lanes 0..6 the
//   channel-pick windows; designDFE with a lane per column of its generator recursion (design_dfe7_lanes).
// Each value is formed by the same operations in the same order as in k_eq_detect (and the reference): same results.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_eq_list(const uint8_t *__restrict__ enable, int B, int32_t *__restrict__ list, int32_t *__restrict__ count) {
  // thread t owns bursts t, t + 1024, t + 2048, ...: a wave's loads are 64 consecutive bytes, 32 of a thread's in flight at a time
  // (unconditional, index clamped: a load under a branch is waited for at the join).  The first 64 flags of a thread are kept as a
  // bit mask, so a call of up to 65,536 bursts reads its flags once.  The list holds thread 0's bursts first, then thread 1's, ...:
  // not sorted, but the same on every run.
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int K = (B + 1023) / 1024;
  int n = 0;
  unsigned long long m = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    uint8_t v[32];
#pragma unroll
    for (int u = 0; u < 32; u++) { const int i = (k0 + u) * 1024 + t; v[u] = enable[i < B ? i : B - 1]; }
#pragma unroll
    for (int u = 0; u < 32; u++) {
      const bool on = ((k0 + u) * 1024 + t < B) && v[u] != 0;
      n += on;
      if (k0 + u < 64) m |= (unsigned long long)on << ((k0 + u) & 63);
    }
  }
  int incl = n;                                             // inclusive prefix sum: inside the wave, then over the sixteen waves
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) before += w < wave ? wsum[w] : 0;
  int pos = before + incl - n;
  while (m) {
    const int k = (int)__builtin_ctzll(m);
    m &= m - 1;
    list[pos++] = k * 1024 + t;
  }
  for (int k = 64; k < K; k++) {                            // (calls beyond 65,536 bursts)
    const int i = k * 1024 + t;
    if (i < B && enable[i] != 0) list[pos++] = i;
  }
  if (t == 1023) *count = before + incl;
}

template <typename SMP>
__global__ __launch_bounds__(256) void k_eq_estimate_wave(const TrxTables *__restrict__ T, const void *__restrict__ samples,
                                                          const int32_t *__restrict__ offset, const int32_t *__restrict__ length, int B, int tsc,
                                                          float detect_thresh, uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                          float *__restrict__ toa_out, float *__restrict__ toa_eq, cx *__restrict__ w_out,
                                                          cx *__restrict__ b_out, float snr_thresh, float snr_value,
                                                          float *__restrict__ chan_off_out, cx *__restrict__ chan_out,
                                                          const float *__restrict__ snr_in, const int32_t *__restrict__ list,
                                                          const int32_t *__restrict__ count, int dense_min) {
  constexpr int NL = 36, FRONT = 8, BACK = 8, PADC = 13, W0 = 56, START = 7;   // :951-955, 295-300: NO_DELAY correlation, Lb = 16
  static_assert(START - 15 >= -FRONT && START + NL - 1 < NL + BACK, "every tap of every lag meets a sample or a zero pad");
  __shared__ cx Ws[4][FRONT + NL + BACK];
  __shared__ cx Cs[4][PADC + NL + PADC];
  __shared__ cx locs[4][26];
  __shared__ cx shfs[4][NL];
  __shared__ __attribute__((aligned(16))) float Vs[4][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cx *W = Ws[wave], *Cc = Cs[wave], *loc = locs[wave], *shf = shfs[wave];
  float *V = Vs[wave];
  const int n_waves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
  const int n = list ? *count : B;
  if (list && n > dense_min) return;                         // many marked bursts: k_eq_detect, launched next, takes the call
  const cx gain = T->mid_gain[tsc];
  const float mid_toa = T->mid_toa[tsc];
  cx ctap[16];
#pragma unroll
  for (int j = 0; j < 16; j++) ctap[j] = T->mid_ctap[tsc][15 - j];   // tmp[j] = conj(mid[15 - j]) (:480-498)
  for (int it = w0; it < n; it += n_waves) {
    const int b = __builtin_amdgcn_readfirstlane(list ? list[it] : it);
    const int off = offset[b], N = length[b];
    if (!((off >= 0) && (N >= 92) && (N <= 157))) {
      if (lane == 0) { flags[b] = TRXSIG_F_BADLEN; amp_out[b] = mk(0, 0); toa_out[b] = 0.0f; toa_eq[b] = 0.0f; }
      continue;
    }
    // ---- the window between zero pads; the correlation's pads ----
    if (lane < FRONT) W[lane] = mk(0, 0);
    if (lane < BACK) W[FRONT + NL + lane] = mk(0, 0);
    if (lane < PADC) { Cc[lane] = mk(0, 0); Cc[PADC + NL + lane] = mk(0, 0); }
    if (lane < NL) W[FRONT + lane] = SMP::ld(samples, (long long)off + W0 + lane);
    wave_lds_fence();
    // ---- correlation: lag i = lane ends on window sample START + i; j ascending ----
    float P = 0.0f;
    int Tm = -1;
    if (lane < NL) {
      cx acc = mk(0, 0);
      const cx *wp = W + (FRONT + START + lane);
#pragma unroll
      for (int j = 0; j < 16; j++) acc = cadd(acc, cmul(wp[-j], ctap[j]));
      Cc[PADC + lane] = acc;
      const float pw = norm2(acc);
      if (pw > 0.0f) { P = pw; Tm = lane; }                 // "if (p > maxP)" from maxP = 0 (:675): a lag without power is never the maximum
    }
    // the first super-step's points do not depend on the data (early starts at M - 1): fetch its sinc rows under the reduction
    const int relA = kFusedRel5.v[lane];
    const int eA = (relA >> 2) * 16;
    float rowA[24];
    fused_row(T, eA, rowA);
    wave_argmax(P, Tm);                                     // larger power wins, equal power: the smaller lag (the first maximum)
    const int M = Tm;
    wave_lds_fence();
    if (lane < 26) {                                        // lags M-12 .. M+11 as interpolatePoint sees them (never the last sample, :646)
      const int lag = M - 12 + lane;
      loc[lane] = (lane >= 24 || lag < 0 || lag > NL - 2) ? mk(0, 0) : Cc[PADC + lag];
    }
    wave_lds_fence();
    // ---- peakDetect's bisection, speculated (k_normal_fused's arrangement for 64 lanes per burst) ----
    int e = 0;                                              // early = M-1 + e/512
    bool active = true;
    cx peak = mk(0, 0);
    {
      const cx ptA = fused_point(loc, eA, relA & 3, rowA);                  // levels 1-5: +-256 .. +-16
      fused_decide<64, 5, false>(ptA, lane, 256, e, active, peak);
      const int relB = kFusedRel4F.v[lane], eB = e + (relB >> 2);           // levels 6-9: +-8 .. +-1, and the finals
      float rowB[24];
      fused_row(T, eB, rowB);
      const cx ptB = fused_point(loc, eB, relB & 3, rowB);
      fused_decide<64, 4, true>(ptB, lane, 8, e, active, peak);
    }
    if (!active) {                                          // the reference left its loop on equal powers (:695): interpolatePoint(early + 1) where it stopped
      float srow[24];
      fused_row(T, e, srow);
      peak = fused_point(loc, e, 1, srow);
    }
    cx amp;
    float toa;
    bool detected, energy_ok;
    fused_tail<1, 64>([&](int lag) { return norm2(Cc[PADC + lag]); }, V, lane, M, e, peak, true, 0.0f, cinv(gain), mid_toa, detect_thresh,
                      -1.0f, amp, toa, detected, energy_ok);
    float chanOff = 0.0f;
    cx chan[6];
#pragma unroll
    for (int k = 0; k < 6; k++) chan[k] = mk(0, 0);
    if (detected) {                                         // (wave-uniform)
      const float TOAoffset = mid_toa + 10.0f;
      // delayVector(corr, -TOA) (:573-616): output t = lane; taps t + 10 - j outside [0, 36) meet the zero pads
      const float delay = -toa;
      const int io = (int)floorf(delay);
      const float frac = delay - (float)io;
      const cx *src = Cc + PADC;
      if (fabs((double)frac) > 1e-2) {
        const float *row = T->sinc_grid[(int)(frac * 512.0f) & 511];
        if (lane < NL) {
          cx sum = mk(0, 0);
          const cx *cp = Cc + (PADC + lane + 10);
#pragma unroll
          for (int j = 0; j < 21; j++) sum = cadd(sum, cmulr(cp[-j], row[j]));
          shf[lane] = sum;
        }
        src = shf;
        wave_lds_fence();
      }
      auto wv = [&](int k) {                                // the integer shift folded into the reads
        const int q = k - io;
        return (q >= 0 && q < NL) ? src[q] : mk(0, 0);
      };
      // :1012-1021: window i = lane
      float energy = 0.0f;
      bool valid = false;
      if (lane < 7) {
        const float st = TOAoffset + (float)(lane - 5);
        valid = !(st + (float)6u > (float)(unsigned)NL) && !(st < 0.0f);
        const int s0 = (int)floorf(st);
        for (int k = 0; k < 6; k++) energy += norm2(wv(s0 + k));
      }
      float maxEnergy = -1.0f;
      int maxI = -1;
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const float en = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(energy), i));
        const bool ok = __builtin_amdgcn_readlane((int)valid, i) != 0;
        if (ok && (double)en > 0.95 * (double)maxEnergy) { maxI = i; maxEnergy = en; }
      }
      const int s0 = (int)floorf(TOAoffset + (float)(maxI - 5));
      const cx ginv = cdiv(mk(1.0f, 0.0f), gain);
#pragma unroll
      for (int k = 0; k < 6; k++) chan[k] = cmul(wv(s0 + k), ginv);   // :1024-1025
      chanOff = (float)(5 - maxI);                           // :1029
    }
    if (lane == 0) {
      flags[b] = (uint8_t)(TRXSIG_F_ENERGY | (detected ? TRXSIG_F_DETECT : 0));
      amp_out[b] = amp;
      toa_out[b] = toa;
      toa_eq[b] = toa - chanOff;
      if (chan_off_out) chan_off_out[b] = chanOff;
    }
    if (chan_out && lane == 0) {                           // zeros if not detected
#pragma unroll
      for (int k = 0; k < 6; k++) chan_out[(size_t)b * 6 + k] = chan[k];
    }
    if (detected) {
      // Transceiver.cpp:341-347: SNR, scaleVector(chan, 1/amp), designDFE(chan, SNR, 7)
      const float thr = snr_thresh >= 0.0f ? snr_thresh : 0.0f;
      const float snr = snr_in ? snr_in[b] : (snr_value > 0.0f ? snr_value : (float)((double)norm2(amp) / ((double)(thr * thr) + 1.0)));
      const cx ainv = cdiv(mk(1.0f, 0.0f), amp);
#pragma unroll
      for (int k = 0; k < 6; k++) chan[k] = cmul(chan[k], ainv);
      cx w_mine, b_mine;
      design_dfe7_lanes(chan, snr, lane, shf, loc, w_mine, b_mine);   // (shf, loc: dead by now)
      if (lane < 7) w_out[(size_t)b * 7 + lane] = w_mine;
      if (lane >= 1 && lane < 6) b_out[(size_t)b * 5 + (lane - 1)] = b_mine;
    }
    wave_lds_fence();                                       // the next burst reuses the wave's LDS
  }
}

// The burst's row of xd was written by k_eq_delay only if that kernel accepted the burst (k_demod's gate: DETECT flag,
// 92..157 samples, |TOA| <= 4096 and not NaN).  The equaliser must apply the same gate, or it would equalise whatever an
// earlier call left in the row and hand back plausible-looking soft bits.
__device__ __forceinline__ bool eq_enabled(uint8_t fl, int N, float toa_eq) {
  return (fl & TRXSIG_F_DETECT) && N >= 92 && N <= 157 && (fabsf(toa_eq) <= 4096.0f);
}

// equalizeBurst after its delayVector: xd = delayed, scaled burst (B x xstride complex)
__global__ __launch_bounds__(64) void k_eq_dfe(const TrxTables *__restrict__ T, const cx *__restrict__ xd, int xstride,
                                               const int32_t *__restrict__ length, int B,
                                               const uint8_t *__restrict__ flags, const float *__restrict__ toa_eq,
                                               const cx *__restrict__ w_in,
                                               const cx *__restrict__ b_in, float *__restrict__ soft,
                                               uint8_t *__restrict__ hard, int nsoft, int stride,
                                               const int32_t *__restrict__ tap_ix) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float *sb = soft + (size_t)b * stride;
  uint8_t *hb = hard ? hard + (size_t)b * stride : nullptr;
  if (!eq_enabled(flags[b], length[b], toa_eq[b])) {
    for (int m = 0; m < nsoft; m++) { sb[m] = 0.0f; if (hb) hb[m] = 0; }
    return;
  }
  const int N = length[b];
  const cx *x = xd + (size_t)b * xstride;
  const size_t tb = tap_ix ? (size_t)tap_ix[b] : (size_t)b;
  cx w[7], bq[5], hist[5], win[7];
#pragma unroll
  for (int j = 0; j < 7; j++) w[j] = w_in[tb * 7 + j];
#pragma unroll
  for (int j = 0; j < 5; j++) { bq[j] = b_in[tb * 5 + j]; hist[j] = mk(0, 0); }
  // win[j] = x[k + 6 - j] (zero outside the burst); FULL_SPAN keeps [6, 6+N) (:1352-1356)
#pragma unroll
  for (int j = 0; j < 7; j++) win[j] = (6 - j < N) ? x[6 - j] : mk(0, 0);
  const int nout = nsoft < N ? nsoft : N;
  for (int k = 0; k < nout; k++) {
    cx d = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 7; j++) {                          // convolve general branch: sum += a[t-j]*b[j], t = k+6
      const int ai = k + 6 - j;
      if (ai >= 0 && ai < N) d = cadd(d, cmul(win[j], w[j]));
    }
#pragma unroll
    for (int j = 0; j < 5; j++)                            // feedback over past decisions (:1370-1374)
      if (k - 1 - j >= 0) d = cadd(d, cmul(bq[j], hist[j]));
    d = cmul(d, T->rev[k]);                                // :1375
    const float re = d.r;
    const cx dec = mk((re > 0.0f) ? 1.0f : -1.0f, 0.0f);   // :1378
    const cx fbv = cmul(dec, T->rot[k]);                   // :1380
#pragma unroll
    for (int j = 4; j > 0; j--) hist[j] = hist[j - 1];
    hist[0] = fbv;
    float sv = (float)(0.5 * (double)(re + 1.0F));         // vectorSlicer (:513-515)
    if (sv > 1.0f) sv = 1.0f;
    if (sv < 0.0f) sv = 0.0f;
    sb[k] = sv;
    if (hb) hb[k] = sv > 0.5F;
#pragma unroll
    for (int j = 6; j > 0; j--) win[j] = win[j - 1];
    win[0] = (k + 7 < N) ? x[k + 7] : mk(0, 0);
  }
  for (int m = nout; m < nsoft; m++) { sb[m] = 0.0f; if (hb) hb[m] = 0; }
}


// ---------------------------------------------------------------------------------------------
// k_eq_dfe2: equalizeBurst (:1352-1384) for 64 bursts per workgroup with TWO waves: lane l of the producer wave and
//   lane l of the consumer wave share burst l.
//   * The feed-forward FIR does not depend on the decisions: the PRODUCER computes ff[k] = sum_j x[k+6-j] w[j]
//     (the reference's terms in the reference's order) a tile of 16 symbols ahead and hands it over through LDS.
//   * The CONSUMER runs the serial part only: d = ((((ff + b0 h0) + b1 h1) + ...) -- the same accumulator the
//     reference continues after its feed-forward terms -- reverse rotation, decision, feedback, slicer
//     (66 instead of 125 VALU per symbol on the critical path).
//   * Neither touches global memory per symbol: the delayed burst comes in and the soft bits go out tile-wise, with
//     lanes spread along k (4 rows x 16 symbols per instruction = whole cache lines) instead of 64 rows x one symbol.
//   (k_eq_dfe: a lane per burst doing everything, one uncoalesced 8-byte load and 4-byte store per symbol with one
//   step of latency cover: 110 us per 64 K bursts.)
// ---------------------------------------------------------------------------------------------
// One symbol of equalizeBurst's decision-feedback loop (:1370-1384) on the feed-forward sum ff: the feedback terms continue the
// same accumulator in the reference's order, reverse rotation, decision, the rotated decision goes into the history, slicer.
// (Round 3 also ran this step on packed float32 pairs -- pk_cmul / pk_cadd of trxsig_dev.h, 31 instead of 66 instructions, the
// same values: k_eq_dfe2 57.2 -> 60.1 us.  The recursion is bound by the latency of its dependent chain, not by the
// consumer wave's instruction count.)
__device__ __forceinline__ float dfe_step(int k, int nout, cx ff, cx rv, cx rt, const v2f (&bq)[5], v2f (&hist)[5]) {
  // Branch-free: a burst shorter than the tile keeps stepping (its history is never looked at again) and only the result is
  // masked.  Packed float32 pairs (pk_cmul / pk_cadd, trxsig_dev.h: the reference's products and sums, each rounded on its
  // own): a complex multiply-add is 4 instructions instead of 8.
  v2f d = pk(ff);                                           // the feed-forward terms, already summed in order
#pragma unroll
  for (int j = 0; j < 5; j++)                               // feedback over past decisions (:1370-1374)
    if (k - 1 - j >= 0) d = pk_cadd(d, pk_cmul(bq[j], hist[j]));
  d = pk_cmul(d, pk(rv));                                   // :1375
  const float re = d.x;
  const cx dec = mk((re > 0.0f) ? 1.0f : -1.0f, 0.0f);      // :1378
  const v2f fbv = pk_cmul(pk(dec), pk(rt));                 // :1380
#pragma unroll
  for (int j = 4; j > 0; j--) hist[j] = hist[j - 1];
  hist[0] = fbv;
  float sv = (re + 1.0F) * 0.5F;                            // vectorSlicer (:513-515): (float)(0.5*(double)(re + 1.0F)); re + 1.0F is 0 or
  //                                                           at least 2^-24 in magnitude, so halving it in float is exact too
  if (sv > 1.0f) sv = 1.0f;
  if (sv < 0.0f) sv = 0.0f;
  return k < nout ? sv : 0.0f;
}

#define EQ_TK 16             /* symbols per tile */
#define EQ_NT 10             /* tiles: 160 >= 157 symbols */
__global__ __launch_bounds__(128) void k_eq_dfe2(const TrxTables *__restrict__ T, const cx *__restrict__ xd, int xstride,
                                                 const int32_t *__restrict__ length, int B,
                                                 const uint8_t *__restrict__ flags, const float *__restrict__ toa_eq,
                                                 const cx *__restrict__ w_in,
                                                 const cx *__restrict__ b_in, float *__restrict__ soft,
                                                 uint8_t *__restrict__ hard, int nsoft, int stride,
                                                 const int32_t *__restrict__ tap_ix) {
  // tap_ix (optional): burst b is equalised with the taps at w_in + 7 tap_ix[b], b_in + 5 tap_ix[b] (the per-timeslot cache
  // of Transceiver.cpp:317-349: many bursts share one estimate); NULL = its own taps at index b.
  __shared__ cx xt[64][EQ_TK + 1];                          // the producer's own staging of the delayed burst
  __shared__ cx fft[2][64][EQ_TK + 1];                      // producer -> consumer
  __shared__ float sft[2][64][EQ_TK + 1];                   // consumer -> producer (soft bits on their way out)
  const int lane = threadIdx.x & 63;
  const bool producer = threadIdx.x >= 64;                  // wave-uniform
  const int b0 = blockIdx.x * 64;
  const int b = b0 + lane;
  const int bb = b < B ? b : B - 1;
  const int N = length[bb];
  const bool det = b < B && eq_enabled(flags[bb], N, toa_eq[bb]);
  const int nout = det ? (nsoft < N ? nsoft : N) : 0;       // symbols this burst really produces (zeros beyond)
  const size_t tb = (tap_ix && det) ? (size_t)tap_ix[bb] : (tap_ix ? (size_t)0 : (size_t)bb);
  // Barrier u (u = 0..9): ff tile u is ready and soft tile u-1 is complete; barrier 10: soft tile 9 is complete.
  // Both waves execute exactly eleven barriers.
#ifdef TRX_DFE_PROBE
  long long pw_ = 0, pt0_ = clock64();
#define DFE_SYNC() do { const long long a_ = clock64(); __syncthreads(); pw_ += clock64() - a_; } while (0)
#else
#define DFE_SYNC() __syncthreads()
#endif
  if (producer) {
    const int kc = lane & 15, r0 = lane >> 4;               // tile traffic: this lane moves column kc of rows r0 + 4 i
    const cx *x = xd + (size_t)bb * xstride;
    v2f w[7], win[6];
#pragma unroll
    for (int j = 0; j < 7; j++) w[j] = pk(w_in[tb * 7 + j]);
#pragma unroll
    for (int m = 0; m < 6; m++) win[m] = pk(x[5 - m]);       // win[m] = x[16 u + 5 - m] (the row holds zeros from N on: k_eq_delay)
    // sample tile u: a = 16 u + 6 + c, c = 0..15 (output k = 16 u + i needs x[k + 6 - j]: FULL_SPAN keeps [6, 6+N), :1352-1356)
    auto load_tile = [&](int u, cx (&v)[16]) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int rb = b0 + r0 + 4 * i, a = EQ_TK * u + 6 + kc;
        v[i] = (rb < B && a < xstride) ? xd[(size_t)rb * xstride + a] : mk(0, 0);
      }
    };
    auto write_out = [&](int u) {                           // soft tile u, lanes along k
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int r = r0 + 4 * i, k = EQ_TK * u + kc, rb = b0 + r;
        if (rb < B && k < nsoft) {
          const float sv = sft[u & 1][r][kc];
          soft[(size_t)rb * stride + k] = sv;
          if (hard) hard[(size_t)rb * stride + k] = sv > 0.5F;
        }
      }
    };
    cx v[16];
    load_tile(0, v);
    for (int u = 0; u < EQ_NT; u++) {
      wave_lds_fence();                                     // everybody has taken its row of the previous tile
#pragma unroll
      for (int i = 0; i < 16; i++) xt[r0 + 4 * i][kc] = v[i];
      wave_lds_fence();
      if (u + 1 < EQ_NT) load_tile(u + 1, v);               // the next tile's loads fly during this tile's arithmetic
      v2f xa[16];
#pragma unroll
      for (int i = 0; i < 16; i++) xa[i] = pk(xt[lane][i]);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        v2f d = pk(mk(0, 0));
#pragma unroll
        for (int j = 0; j < 7; j++) {                       // convolve general branch: sum += a[t-j]*b[j], t = k+6
          const v2f xv = (i - j >= 0) ? xa[(i - j >= 0) ? i - j : 0] : win[(j - i - 1 < 6) ? j - i - 1 : 5];
          // (a term beyond the burst is skipped in the reference; here it is added as the product with a zero SAMPLE instead --
          // k_eq_delay wrote zeros from N to the end of the row, the tile loader zeros beyond it --: the sum starts at +0 and
          // +0 + (+-0) = +0, so the value is the same, with no branch and no range check; k + 6 - j >= 0 always)
          d = pk_cadd(d, pk_cmul(xv, w[j]));
        }
        fft[u & 1][lane][i] = mk(d.x, d.y);
      }
#pragma unroll
      for (int m = 0; m < 6; m++) win[m] = xa[15 - m];
      DFE_SYNC();                                           // barrier u
      if (u >= 1) write_out(u - 1);
    }
    DFE_SYNC();                                             // barrier 10
    write_out(EQ_NT - 1);
#ifdef TRX_DFE_PROBE
    if (lane == 0 && b0 < B) { soft[(size_t)b0 * stride + 0] = (float)pw_; soft[(size_t)b0 * stride + 1] = (float)(clock64() - pt0_); }
    __syncthreads();
#endif
  } else {
    v2f bq[5], hist[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { bq[j] = pk(b_in[tb * 5 + j]); hist[j] = pk(mk(0, 0)); }
    for (int u = 0; u < EQ_NT; u++) {
      DFE_SYNC();                                           // barrier u
      // the tile's operands up front: feed-forward sums (LDS), rotation factors (uniform -> scalar loads)
      cx ffv[EQ_TK], rv[EQ_TK], rt[EQ_TK];
#pragma unroll
      for (int i = 0; i < EQ_TK; i++) {
        ffv[i] = fft[u & 1][lane][i];
        rv[i] = T->rev[EQ_TK * u + i];                      // (rev/rot hold 157 * 4 entries: in range for k < 160)
        rt[i] = T->rot[EQ_TK * u + i];
      }
#pragma unroll
      for (int i = 0; i < EQ_TK; i++) sft[u & 1][lane][i] = dfe_step(EQ_TK * u + i, nout, ffv[i], rv[i], rt[i], bq, hist);
    }
    DFE_SYNC();                                             // barrier 10
#ifdef TRX_DFE_PROBE
    __syncthreads();
    if (lane == 0 && b0 < B) { soft[(size_t)b0 * stride + 2] = (float)pw_; soft[(size_t)b0 * stride + 3] = (float)(clock64() - pt0_); }
#endif
  }
}
#undef DFE_SYNC

#ifdef TRX_TUNING_BUILD   /* k_eq_dfe3: the single-kernel equaliser tail, measured slower; tuning library only */
// ---------------------------------------------------------------------------------------------
// k_eq_dfe3 (round 3, A/B only): scaleVector + equalizeBurst in ONE kernel -- k_eq_delay's job done inside k_eq_dfe2's workgroup, so
//   the delayed burst (82 MB written and 108 MB read back per 65,536 bursts) never visits HBM.  64 bursts per workgroup,
//   FOUR waves, lane l of every wave belongs to burst l; time runs in tiles of 16 symbols and the roles are pipelined:
//     wave 2, loader    raw samples of tile s+2, lanes along k (a row's 16 samples are one 32/64-byte piece), widened, scaled by
//                       1/amp (scaleVector) and parked in a 64-entry ring per burst ALREADY SHIFTED by the burst's integer
//                       delay: sample a sits at position p = a + floor(delay), so no address below depends on the TOA and
//                       any TOA is served (samples outside the burst are zeros: "taps outside the vector are skipped");
//     waves 0, 1        delayVector's 21-tap fractional filter (k_eq_delay's arithmetic: j ascending, real taps from the sinc
//                       grid or the table sinc) for tile s, eight outputs each per burst, from the ring to a tile buffer;
//     wave 2            the feed-forward FIR of tile s-1 (k_eq_dfe2's producer) from that buffer;
//     wave 3            the decision-feedback recursion of tile s-2 (k_eq_dfe2's consumer);
//     wave 2            soft bits of tile s-3 out, lanes along k.
//   One workgroup barrier per step, 14 steps.  Same terms in the same order as k_eq_delay + k_eq_dfe2: value-identical
//   (TRXSIG_EQ_DFE_VARIANT=3 selects it: tests/test_gpu_equalize.py passes with it; measured slower than the two kernels,
//   see launch_eq_tail).
//   Delayed sample m of tile u sits at tile index m - (16 u + 6): the feed-forward sum of output k = 16 u + i meets delayed
//   samples k .. k + 6 (FULL_SPAN keeps [6, 6 + N), :1352-1356), i.e. tile u's entries i .. and the previous tile's last six.
// ---------------------------------------------------------------------------------------------
#define EQ3_RING 64
template <typename SMP>
__global__ __launch_bounds__(256) void k_eq_dfe3(const TrxTables *__restrict__ T, const void *__restrict__ samples,
                                                 const int32_t *__restrict__ offset, const int32_t *__restrict__ length, int B,
                                                 const cx *__restrict__ amp_in, const float *__restrict__ toa_eq,
                                                 const uint8_t *__restrict__ flags, const cx *__restrict__ w_in,
                                                 const cx *__restrict__ b_in, const int32_t *__restrict__ tap_ix,
                                                 float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft, int stride) {
  __shared__ cx ring[64][EQ3_RING + 1];                     // [burst][p & 63] (odd pitch: a lane per burst reads conflict-free)
  __shared__ cx dl[2][64][EQ_TK + 1];                       // delay waves -> feed-forward wave
  __shared__ cx fft[2][64][EQ_TK + 1];                      // feed-forward wave -> consumer
  __shared__ float sft[2][64][EQ_TK + 1];                   // consumer -> feed-forward wave (soft bits on their way out)
  __shared__ int r_off[64], r_n[64], r_io[64];
  __shared__ cx r_inv[64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int b0 = blockIdx.x * 64;
  const int b = b0 + lane;
  const int bb = b < B ? b : B - 1;
  const int N = length[bb];
  const int off = offset[bb];
  const float te = toa_eq[bb];
  const bool det = b < B && off >= 0 && eq_enabled(flags[bb], N, te);
  const int nout = det ? (nsoft < N ? nsoft : N) : 0;       // symbols this burst really produces (zeros beyond)
  const size_t tb = (tap_ix && det) ? (size_t)tap_ix[bb] : (tap_ix ? (size_t)0 : (size_t)bb);
  // delayVector bookkeeping (:577-582).  A burst that is not equalised (flag, length, |TOA| > 4096, NaN) gets delay 0: its TOA may
  // be anything, and the table sinc's subtract-one range reduction does not come back from a huge argument
  const float delay = det ? -te : 0.0f;
  const int io = (int)floorf(delay);
  const float frac = delay - (float)io;
  if (wave == 2) {                                          // what the loader needs per row
    r_off[lane] = off; r_n[lane] = det ? N : 0; r_io[lane] = io;
    r_inv[lane] = cdiv(mk(1.0f, 0.0f), amp_in[bb]);         // ((complex)1.0)/amp (Transceiver.cpp:391)
  }
  const int kc = lane & 15, r0 = lane >> 4;                 // tile traffic: this lane moves column kc of rows r0 + 4 i
  // raw tile v: positions p = 16 v + kc, i.e. samples a = p - io of each row
  auto load_raw = [&](int v, typename SMP::raw_t (&rv)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = r0 + 4 * i;
      const int a = EQ_TK * v + kc - r_io[r];
      rv[i] = (a >= 0 && a < r_n[r]) ? SMP::ldraw(samples, (long long)r_off[r] + a) : SMP::zero();
    }
  };
  auto park_raw = [&](int v, const typename SMP::raw_t (&rv)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = r0 + 4 * i;
      ring[r][(EQ_TK * v + kc) & (EQ3_RING - 1)] = cmul(SMP::widen(rv[i]), r_inv[r]);   // scaleVector (:713-723)
    }
  };
  if (wave == 2) {
    wave_lds_fence();                                       // the row tables are this wave's own writes
    typename SMP::raw_t rv[16];
    load_raw(-1, rv); park_raw(-1, rv);
    load_raw(0, rv); park_raw(0, rv);
  }
  __syncthreads();

  if (wave < 2) {
    // ---- delayVector (:573-616), eight outputs of every tile per wave ----
    const bool filt = fabs((double)frac) > 1e-2;
    float tp[21];
    {
      const float f512 = frac * 512.0f;
      const int f = (int)f512;
      const bool grid = f < 512 && (float)f == f512;
      const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[f & 511]);
      float g[24];
#pragma unroll
      for (int q = 0; q < 6; q++) { const float4 r4 = row[q]; g[4 * q] = r4.x; g[4 * q + 1] = r4.y; g[4 * q + 2] = r4.z; g[4 * q + 3] = r4.w; }
#pragma unroll
      for (int j = 0; j < 21; j++) tp[j] = g[j];
      if (__any(!grid)) {                                   // off the grid (never after peakDetect): sinc(pi*((j - 10) - frac)) (:588)
#pragma unroll
        for (int j = 0; j < 21; j++) {
          const float tj = dev_sinc(T->sinT, TRX_PI_F * ((float)(j - 10) - frac));
          tp[j] = grid ? g[j] : tj;
        }
      }
    }
    const int h8 = 8 * wave;
    for (int s = -1; s <= EQ_NT + 2; s++) {
      if (s <= EQ_NT - 1) {
        const int m0 = EQ_TK * s + 6 + h8;                  // first delayed sample of this wave's half tile
        cx w[28];
#pragma unroll
        for (int q = 0; q < 28; q++) w[q] = ring[lane][(m0 - 10 + q) & (EQ3_RING - 1)];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          cx acc = mk(0, 0);
#pragma unroll
          for (int j = 0; j < 21; j++) acc = cadd(acc, cmulr(w[i + 20 - j], tp[j]));   // convolve(..., NO_DELAY), j ascending (:590)
          const int t = m0 + i - io;
          const cx r = filt ? acc : w[i + 10];
          dl[s & 1][lane][h8 + i] = (t >= 0 && t < N && det) ? r : mk(0, 0);           // shifted[m] inside [0, N), else 0 (:597-613)
        }
      }
      __syncthreads();
    }
  } else if (wave == 2) {
    // ---- loader, feed-forward FIR, soft bits out ----
    cx w[7], win[6];
#pragma unroll
    for (int j = 0; j < 7; j++) w[j] = w_in[tb * 7 + j];
#pragma unroll
    for (int m = 0; m < 6; m++) win[m] = mk(0, 0);
    auto write_out = [&](int u) {                           // soft tile u, lanes along k
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int r = r0 + 4 * i, k = EQ_TK * u + kc, rb = b0 + r;
        if (rb < B && k < nsoft) {
          const float sv = sft[u & 1][r][kc];
          soft[(size_t)rb * stride + k] = sv;
          if (hard) hard[(size_t)rb * stride + k] = sv > 0.5F;
        }
      }
    };
    for (int s = -1; s <= EQ_NT + 2; s++) {
      typename SMP::raw_t rv[16];
      const bool do_load = s + 2 <= EQ_NT + 1;              // (ring tiles up to NT + 1: the last delay step reads into it)
      if (do_load) load_raw(s + 2, rv);                     // in flight under the arithmetic below
      else {
#pragma unroll
        for (int i = 0; i < 16; i++) rv[i] = SMP::zero();
      }
      const int u = s - 1;
      if (u >= -1 && u <= EQ_NT - 1) {
        cx xa[16];
#pragma unroll
        for (int i = 0; i < 16; i++) xa[i] = dl[u & 1][lane][i];
        if (u >= 0) {
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int k = EQ_TK * u + i;
            cx d = mk(0, 0);
#pragma unroll
            for (int j = 0; j < 7; j++) {                   // convolve general branch: sum += a[t-j]*b[j], t = k+6
              const int ai = k + 6 - j;
              const cx xv = (i - j >= 0) ? xa[(i - j >= 0) ? i - j : 0] : win[(j - i - 1 < 6) ? j - i - 1 : 5];
              if (ai >= 0 && ai < N) d = cadd(d, cmul(xv, w[j]));
            }
            fft[u & 1][lane][i] = d;
          }
        }
#pragma unroll
        for (int m = 0; m < 6; m++) win[m] = xa[15 - m];    // delayed samples 16 u + 21 - m: the next tile's x[k + 6 - j], j > i
      }
      if (do_load) park_raw(s + 2, rv);
      if (s - 3 >= 0 && s - 3 <= EQ_NT - 1) write_out(s - 3);
      __syncthreads();
    }
  } else {
    // ---- the decision-feedback recursion (k_eq_dfe2's consumer) ----
    v2f bq[5], hist[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { bq[j] = pk(b_in[tb * 5 + j]); hist[j] = pk(mk(0, 0)); }
    for (int s = -1; s <= EQ_NT + 2; s++) {
      const int u = s - 2;
      if (u >= 0 && u <= EQ_NT - 1) {
        cx ffv[EQ_TK], rv[EQ_TK], rt[EQ_TK];
#pragma unroll
        for (int i = 0; i < EQ_TK; i++) {
          ffv[i] = fft[u & 1][lane][i];
          rv[i] = T->rev[EQ_TK * u + i];                    // (rev/rot hold 157 * 4 entries: in range for k < 160)
          rt[i] = T->rot[EQ_TK * u + i];
        }
#pragma unroll
        for (int i = 0; i < EQ_TK; i++) sft[u & 1][lane][i] = dfe_step(EQ_TK * u + i, nout, ffv[i], rv[i], rt[i], bq, hist);
      }
      __syncthreads();
    }
  }
}

#endif  // TRX_TUNING_BUILD

// ---------------------------------------------------------------------------------------------
// k_eq_dfe4 (round 4): scaleVector + delayVector + equalizeBurst in ONE kernel -- what k_eq_delay and k_eq_dfe2 do through the
//   160-sample scratch row of every burst (84 MB written, 107 MB read back per 65,536 bursts), with that row never leaving the chip.
//   k_eq_dfe3 (round 3, tuning library) had the same aim and lost: its sample ring and 16-symbol tiles took 78 KB of LDS, two
//   workgroups per CU, two generations.  Here the delay waves keep their 28-sample windows in REGISTERS, fed by 16-byte loads of the
//   lane's own burst (a lane per burst in every wave: no transposition, no ring), and the tiles are EIGHT symbols: 23 KB of LDS,
//   all 1,024 workgroups of a 65,536-burst call resident at four waves per SIMD (28 KB with the tile buffer in three copies).
//   64 bursts per workgroup, four waves, lane l of every wave = burst l; time runs in steps of eight symbols:
//     waves 2, 3  delayVector (:573-616; k_eq_delay's arithmetic: scaleVector(1/amp) on the way in, 21 real taps from the sinc grid or
//                 the table sinc, j ascending) -- tile u (delayed samples 8 u + 6 .. 8 u + 13) is wave 2 + (u & 1)'s, over steps
//                 u - 2 and u - 1, four outputs in each; tile -1 (samples 0 .. 5: the first window of the feed-forward filter) too;
//     wave 1      the feed-forward FIR of tile u in step u (k_eq_dfe2's producer), soft bits of tile u - 2 out (lanes along k);
//     wave 0      the decision-feedback recursion of tile u - 1 (k_eq_dfe2's consumer).
//   One workgroup barrier per step, 25 steps.  Same terms in the same order as k_eq_delay + k_eq_dfe2: the same values.
// ---------------------------------------------------------------------------------------------
#define EQ4_TK 8
#ifndef TRX_D4_EXP
#define TRX_D4_EXP 0          /* 1, 2, 3: timing experiments (wrong results): the delay / feed-forward / recursion role with most of its arithmetic left out */
#endif
#define EQ4_NT 20             /* 160 >= 157 symbols */
#ifdef TRX_D4_PROBE
#define D4_BARRIER() do { const long long a_ = clock64(); d4_work += a_ - d4_t; __syncthreads(); d4_t = clock64(); d4_wait += d4_t - a_; } while (0)
#else
#define D4_BARRIER() __syncthreads()
#endif
template <typename SMP>
__global__ __launch_bounds__(256, 4) void k_eq_dfe4(const TrxTables *__restrict__ T, const void *__restrict__ samples,
                                                    const int32_t *__restrict__ offset, const int32_t *__restrict__ length, int B,
                                                    const cx *__restrict__ amp_in, const float *__restrict__ toa_eq,
                                                    const uint8_t *__restrict__ flags, const cx *__restrict__ w_in,
                                                    const cx *__restrict__ b_in, const int32_t *__restrict__ tap_ix,
                                                    float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft, int stride) {
  constexpr int TK = EQ4_TK, NT = EQ4_NT, S0 = -3, S1 = NT + 1;   // steps S0 .. S1
  __shared__ cx xt[3][64][TK + 1];                          // delay waves -> feed-forward wave (tile u in xt[u mod 3]: written over steps u - 2, u - 1, read in step u)
  __shared__ cx fft[2][64][TK + 1];                         // feed-forward wave -> consumer
  __shared__ float sft[2][64][TK + 1];                      // consumer -> feed-forward wave (soft bits on their way out)
  __shared__ __attribute__((aligned(8))) float tapl[2][64][22];   // a delay wave's 21 taps per lane (registers are what the delay waves are short of)
  const int lane = threadIdx.x & 63;
#ifdef TRX_D4_PROBE                                         // tools/dfe4_probe.py: per role, cycles between barriers (work) and at them (wait)
  long long d4_t = clock64(), d4_work = 0, d4_wait = 0;
  const long long d4_t0 = d4_t;
#endif
  // (the roles rotate from workgroup to workgroup: a SIMD then hosts one wave of each role instead of four of a kind, and what it has to
  //  issue per step is the roles' average, not the heaviest role's)
  // (workgroups 256 apart tend to share a CU -- the dispatcher deals them out round-robin over 8 XCDs x 32 CUs --: those get different rotations)
  const int wave = __builtin_amdgcn_readfirstlane((((int)threadIdx.x >> 6) + ((int)blockIdx.x >> 8) + (int)blockIdx.x) & 3);
  const int b0 = blockIdx.x * 64;
  const int b = b0 + lane;
  const int bb = b < B ? b : B - 1;
  const int N = length[bb];
  const float te = toa_eq[bb];
  const bool det = b < B && eq_enabled(flags[bb], N, te);  // k_eq_dfe2's gate
  if (wave >= 2) {
    // ---- delayVector: this wave's tiles u = d - 2, d, d + 2, ... (d = wave - 2; wave 3 starts with tile -1) ----
    const int d = wave - 2;
    const int off = offset[bb];
    const int Nd = (det && off >= 0) ? N : 0;              // k_eq_delay's gate: a refused burst is a row of zeros
    const long long base = Nd > 0 ? (long long)off : 0;
    const v2f inv = pk(cdiv(mk(1.0f, 0.0f), amp_in[bb]));  // ((complex)1.0)/amp (Transceiver.cpp:391)
    const float delay = det ? -te : 0.0f;                  // (a burst that is not equalised may carry any TOA: see k_eq_dfe3)
    const int io = (int)floorf(delay);
    const float frac = delay - (float)io;
    const bool filt = fabs((double)frac) > 1e-2;
    v2f *const tl = reinterpret_cast<v2f *>(tapl[d][lane]);   // taps 2 q, 2 q + 1 at tl[q]
    {
      const float f512 = frac * 512.0f;
      const int f = (int)f512;
      const bool grid = f < 512 && (float)f == f512;       // (frac can round to exactly 1.0 for a tiny negative delay: off the grid)
      const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[f & 511]);
      float g[24];
#pragma unroll
      for (int q = 0; q < 6; q++) { const float4 r4 = row[q]; g[4 * q] = r4.x; g[4 * q + 1] = r4.y; g[4 * q + 2] = r4.z; g[4 * q + 3] = r4.w; }
      if (__any(!grid)) {                                  // off the grid (never after peakDetect): sinc(pi*((j - 10) - frac)) (:588)
#pragma unroll
        for (int j = 0; j < 21; j++) {
          const float tj = dev_sinc(T->sinT, TRX_PI_F * ((float)(j - 10) - frac));
          g[j] = grid ? g[j] : tj;
        }
      }
#pragma unroll
      for (int q = 0; q < 11; q++) { v2f t2; t2.x = g[2 * q]; t2.y = g[2 * q + 1]; tl[q] = t2; }
    }
    wave_lds_fence();
    // The window: a RING of 32 scaled samples in registers, sample n at w[(n - n0) & 31] (n0 = the first own tile's first sample).  Own
    // tile number k (u = u_first + 2 k) reads samples n0 + 16 k .. n0 + 16 k + 27 -- tap j of output c meets ring entry
    // (16 k + c + 20 - j) & 31, a compile-time index once k's parity is one (the loop below runs two own tiles per round) -- and the
    // sixteen samples the next own tile adds replace the sixteen this one read first.  No entry is ever moved.
    v2f w[32];
    constexpr int PER = 16 / (16 / SMP::kBytes);           // 16-byte loads that bring sixteen samples (4 at fp16, 8 at complex float)
    constexpr int SPL = 16 / SMP::kBytes;                  // samples per load
    struct __attribute__((packed, aligned(4))) Piece { typename SMP::raw_t q[SPL]; };
    // sixteen samples n .. n + 15 in two moves: the loads (into rv, as stored) are issued a step before they are needed ...
    typename SMP::raw_t rv[16];
    bool rv_inside = false;                                // (wave-uniform) every lane's sixteen lie inside its burst: no masks when they are taken
    auto issue16 = [&](int n) {
      const bool inside = n >= 0 && n + 15 < Nd;
      rv_inside = __all(inside || Nd == 0);
      if (rv_inside) {                                     // (a refused burst reads burst 0's samples: its outputs are zeros whatever they are)
        Piece pc[PER];
        const long long a0 = base + (Nd > 0 ? n : 0);
#pragma unroll
        for (int q = 0; q < PER; q++) pc[q] = *reinterpret_cast<const Piece *>(reinterpret_cast<const typename SMP::raw_t *>(samples) + (a0 + SPL * q));
#pragma unroll
        for (int q = 0; q < PER; q++) {
#pragma unroll
          for (int e = 0; e < SPL; e++) rv[SPL * q + e] = pc[q].q[e];
        }
      } else {                                              // a burst's edge in some lane: sample by sample, index clamped
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int nn = n + k;
          const int nc = nn < 0 ? 0 : (nn >= Nd ? (Nd > 0 ? Nd - 1 : 0) : nn);
          rv[k] = SMP::ldraw(samples, base + nc);
        }
      }
    };
    // ... and scaled into ring entries R0 .. R0 + 15 (zeros outside [0, Nd)) once the entries' old samples have been read
    auto take16 = [&](int n, auto r0_) {
      constexpr int R0 = decltype(r0_)::value;
      if (rv_inside) {
#pragma unroll
        for (int k = 0; k < 16; k++) w[R0 + k] = pk_cmul(pk(SMP::widen(rv[k])), inv);   // scaleVector (:713-723)
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int nn = n + k;
          const v2f sc = pk_cmul(pk(SMP::widen(rv[k])), inv);
          w[R0 + k] = (nn >= 0 && nn < Nd) ? sc : pk(mk(0, 0));
        }
      }
    };
    const int u_first = d == 1 ? -1 : 0;
    const int n_own = d == 1 ? (NT + 1) / 2 + 1 : NT / 2;  // own tiles: -1, 1, .., NT - 1 / 0, 2, .., NT - 2
    const int n0 = TK * u_first + 6 - io - 10;
    issue16(n0); take16(n0, std::integral_constant<int, 0>());
    issue16(n0 + 16); take16(n0 + 16, std::integral_constant<int, 16>());
    // outputs C0 .. C0 + NC - 1 of own tile u (ring phase PH) into its tile buffer.  (Five in the tile's first step, three in its second, which
    // also takes the next samples in: the two delay waves are in opposite halves of their tiles, and the longer half sets the workgroup's step.)
    constexpr int NC1 = 5;
    auto outputs = [&](int u, auto ph_, auto c0_, auto nc_) {
      constexpr int PH = decltype(ph_)::value, C0 = decltype(c0_)::value, NC = decltype(nc_)::value;
      const int m0 = TK * u + 6;
      cx *row = xt[((u % 3) + 3) % 3][lane];
      v2f acc[NC];
#pragma unroll
      for (int c = 0; c < NC; c++) acc[c] = pk(mk(0, 0));
#pragma unroll
      for (int q = 0; q < (TRX_D4_EXP == 1 ? 1 : 11); q++) {   // convolve(..., NO_DELAY), j ascending (:590): the outputs side by side, tap pair by tap pair
        const v2f tq = tl[q];
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = pk_cadd(acc[c], pk_mul_tap<0>(w[(PH + C0 + c + 20 - 2 * q) & 31], tq));
        if (2 * q + 1 <= 20) {
#pragma unroll
          for (int c = 0; c < NC; c++) acc[c] = pk_cadd(acc[c], pk_mul_tap<1>(w[(PH + C0 + c + 19 - 2 * q) & 31], tq));
        }
        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (the tap reads four pairs at a time: all eleven up front cost the registers they were moved to LDS for)
      }
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const int m = m0 + C0 + c, t = m - io;
        const v2f r = filt ? acc[c] : w[(PH + C0 + c + 10) & 31];
        row[C0 + c] = (t >= 0 && t < Nd && m < Nd) ? unpk(r) : mk(0, 0);   // shifted[m] inside [0, N), else 0 (:597-613); the row holds zeros from N on
      }
    };
    // one own tile = two steps.  k: its number; the samples of own tile k + 1 beyond this one's: n0 + 16 k + 32 .. + 47, into the ring's half
    // this tile read first
    auto own_tile = [&](int k, auto ph_) {
      constexpr int PH = decltype(ph_)::value;
      const int u = u_first + 2 * k;
      outputs(u, ph_, std::integral_constant<int, 0>(), std::integral_constant<int, NC1>());
      const bool more = k + 1 < n_own;
      if (more) issue16(n0 + 16 * k + 32);                 // (after the sums: sixteen more live registers they have no room for) a step to land
      D4_BARRIER();
      outputs(u, ph_, std::integral_constant<int, NC1>(), std::integral_constant<int, TK - NC1>());
      if (more) take16(n0 + 16 * k + 32, std::integral_constant<int, PH>());
      D4_BARRIER();
    };
    constexpr int STEPS = S1 - S0 + 1;
    const int lead = d == 1 ? 0 : 1;                       // wave 3's tile -1 starts in step S0, wave 2's tile 0 a step later
    for (int i = 0; i < lead; i++) D4_BARRIER();
    for (int k = 0; k < n_own; k += 2) {
      own_tile(k, std::integral_constant<int, 0>());
      if (k + 1 < n_own) own_tile(k + 1, std::integral_constant<int, 16>());
    }
    for (int i = lead + 2 * n_own; i < STEPS; i++) D4_BARRIER();
  } else if (wave == 1) {
    // ---- the feed-forward FIR (k_eq_dfe2's producer) and the soft bits' way out ----
    const size_t tb = (tap_ix && det) ? (size_t)tap_ix[bb] : (tap_ix ? (size_t)0 : (size_t)bb);
    v2f wf[7], win[6];
#pragma unroll
    for (int j = 0; j < 7; j++) wf[j] = pk(w_in[tb * 7 + j]);
#pragma unroll
    for (int m = 0; m < 6; m++) win[m] = pk(mk(0, 0));
    const int kc = lane & (TK - 1), r0 = lane / TK;          // tile traffic: this lane moves column kc of rows r0 + (64 / TK) i
    auto write_out = [&](int u) {                           // soft tile u, lanes along k
#pragma unroll
      for (int i = 0; i < TK; i++) {
        const int r = r0 + (64 / TK) * i, k = TK * u + kc, rb = b0 + r;
        if (rb < B && k < nsoft) {
          const float sv = sft[u & 1][r][kc];
          soft[(size_t)rb * stride + k] = sv;
          if (hard) hard[(size_t)rb * stride + k] = sv > 0.5F;
        }
      }
    };
    for (int s = S0; s <= S1; s++) {
      const int u = s;
      if (u >= -1 && u <= NT - 1) {
        v2f xa[TK];
#pragma unroll
        for (int i = 0; i < TK; i++) xa[i] = pk(xt[((u % 3) + 3) % 3][lane][i]);
        if (u >= 0) {
#pragma unroll
          for (int i = 0; i < TK; i++) {
            v2f dsum = pk(mk(0, 0));
#pragma unroll
            for (int j = 0; j < (TRX_D4_EXP == 2 ? 1 : 7); j++) {   // convolve general branch: sum += a[t-j]*b[j], t = k+6 (zero samples beyond the burst: see k_eq_dfe2)
              const v2f xv = (i - j >= 0) ? xa[(i - j >= 0) ? i - j : 0] : win[(j - i - 1 < 6) ? j - i - 1 : 5];
              dsum = pk_cadd(dsum, pk_cmul(xv, wf[j]));
            }
            fft[u & 1][lane][i] = unpk(dsum);
          }
        }
#pragma unroll
        for (int m = 0; m < 6; m++) win[m] = xa[TK - 1 - m];  // delayed samples 8 u + 13 - m: the next tile's x[k + 6 - j], j > i
      }
      if (s - 2 >= 0 && s - 2 <= NT - 1) write_out(s - 2);
      D4_BARRIER();
    }
  } else {
    // ---- the decision-feedback recursion (k_eq_dfe2's consumer) ----
    const int nout = det ? (nsoft < N ? nsoft : N) : 0;     // symbols this burst really produces (zeros beyond)
    const size_t tb = (tap_ix && det) ? (size_t)tap_ix[bb] : (tap_ix ? (size_t)0 : (size_t)bb);
    v2f bq[5], hist[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { bq[j] = pk(b_in[tb * 5 + j]); hist[j] = pk(mk(0, 0)); }
    for (int s = S0; s <= S1; s++) {
      const int u = s - 1;
      if (u >= 0 && u <= NT - 1) {
        cx ffv[TK], rv[TK], rt[TK];
#pragma unroll
        for (int i = 0; i < TK; i++) {
          ffv[i] = fft[u & 1][lane][i];
          rv[i] = T->rev[TK * u + i];                       // (rev/rot hold 157 * 4 entries: in range for k < 160)
          rt[i] = T->rot[TK * u + i];
        }
#pragma unroll
        for (int i = 0; i < (TRX_D4_EXP == 3 ? 1 : TK); i++) sft[u & 1][lane][i] = dfe_step(TK * u + i, nout, ffv[i], rv[i], rt[i], bq, hist);
      }
      D4_BARRIER();
    }
  }
#ifdef TRX_D4_PROBE
  // (the probe build hands the stamps back through the soft bits of the workgroup's first four bursts: row b0 + role = {work, wait, total})
  if (lane == 0 && b0 + wave < B) {
    float *o = soft + (size_t)(b0 + wave) * stride;
    o[0] = (float)d4_work; o[1] = (float)d4_wait; o[2] = (float)(clock64() - d4_t0); o[3] = (float)wave;
  }
#endif
}

#undef D4_BARRIER

// TRXSIG_EQ_DFE_VARIANT=1 (environment, A/B): the lane-per-burst k_eq_dfe instead of the producer/consumer k_eq_dfe2
void launch_eq_dfe(hipStream_t st, const TrxTables *dT, const cx *xd, int xstride, const int32_t *len, int B, const uint8_t *flags,
                   const float *toa_eq, const cx *w, const cx *bq, float *soft, uint8_t *hard, int nsoft, int stride,
                   const int32_t *tap_ix = nullptr) {
#ifdef TRX_TUNING_BUILD
  static const bool legacy = std::getenv("TRXSIG_EQ_DFE_VARIANT") && std::atoi(std::getenv("TRXSIG_EQ_DFE_VARIANT")) == 1;   // (2: k_eq_dfe2)
#else
  constexpr bool legacy = false;
#endif
  if (legacy)
    k_eq_dfe<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, xd, xstride, len, B, flags, toa_eq, w, bq, soft, hard, nsoft, stride, tap_ix);
  else
    k_eq_dfe2<<<dim3((B + 63) / 64), dim3(128), 0, st>>>(dT, xd, xstride, len, B, flags, toa_eq, w, bq, soft, hard, nsoft, stride, tap_ix);
}

// designDFE on its own (a lane per channel estimate): chan B x 6 (as analyzeTrafficBurst returns it, i.e. before the
// 1/amp scaling -- pass amp = NULL if the caller has scaled it already), snr B floats
__global__ __launch_bounds__(64) void k_design_dfe(const cx *__restrict__ chan_in, const cx *__restrict__ amp, const float *__restrict__ snr,
                                                   int B, cx *__restrict__ w_out, cx *__restrict__ b_out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  cx chan[6], w[7], bq[5];
#pragma unroll
  for (int k = 0; k < 6; k++) chan[k] = chan_in[(size_t)b * 6 + k];
  if (amp) {
    const cx ainv = cdiv(mk(1.0f, 0.0f), amp[b]);          // scaleVector(chan, 1/amp) (Transceiver.cpp:346)
#pragma unroll
    for (int k = 0; k < 6; k++) chan[k] = cmul(chan[k], ainv);
  }
  design_dfe7(chan, snr[b], w, bq);
#pragma unroll
  for (int i = 0; i < 7; i++) w_out[(size_t)b * 7 + i] = w[i];
#pragma unroll
  for (int j = 0; j < 5; j++) b_out[(size_t)b * 5 + j] = bq[j];
}

}  // namespace

// scaleVector + equalizeBurst: k_eq_delay + k_eq_dfe2 through the xd scratch (default; TRXSIG_EQ_DFE_VARIANT=1: the lane-per-burst
// k_eq_dfe) or, TRXSIG_EQ_DFE_VARIANT=3 in the tuning library, the single fused kernel k_eq_dfe3 -- value-identical and, as measured, slower: 160 us
// against 37 + 57 per 65,536 bursts.  Its 78 KB of LDS leave two workgroups per CU, so the 1,024 workgroups run in two
// generations, and each generation lasts as long as the decision-feedback recursion of its 157 symbols (a latency chain,
// ~0.33 us per symbol) however well the other roles hide under it; the two-kernel form has all 1,024 consumer waves in flight
// at once.
#ifdef TRX_TUNING_BUILD
static int eq_dfe_variant() {
  static const int v = std::getenv("TRXSIG_EQ_DFE_VARIANT") ? std::atoi(std::getenv("TRXSIG_EQ_DFE_VARIANT")) : 0;
  return v;
}
#endif
// trxsig_set_tuning(TRXSIG_TUNE_EQ_TAIL, 2) (A/B and the tests): k_eq_delay + k_eq_dfe2 through the scratch rows instead of the fused k_eq_dfe4
static bool eq_tail_fused() { return trx_knob(TRX_KNOB_EQ_TAIL) != 2; }
static void launch_eq_tail(hipStream_t st, const TrxTables *dT, const void *samples, int fmt, const int32_t *off, const int32_t *len, int B,
                           const trx_c32 *amp, const float *toa_eq, const uint8_t *flags, const trx_c32 *w, const trx_c32 *bq, trx_c32 *xd,
                           int xstride, float *soft, uint8_t *hard, int nsoft, int stride, const int32_t *tap_ix, TrxProfiler *prof) {
#ifdef TRX_TUNING_BUILD
  if (eq_dfe_variant() == 3) {
    if (prof) prof->begin(TRXSIG_K_EQ_DFE, st);
    const dim3 g((B + 63) / 64), blk(256);
    if (fmt == TRXSIG_SAMPLES_F16) k_eq_dfe3<SmpF16><<<g, blk, 0, st>>>(dT, samples, off, len, B, amp, toa_eq, flags, w, bq, tap_ix, soft, hard, nsoft, stride);
    else k_eq_dfe3<SmpC32><<<g, blk, 0, st>>>(dT, samples, off, len, B, amp, toa_eq, flags, w, bq, tap_ix, soft, hard, nsoft, stride);
    if (prof) prof->end(TRXSIG_K_EQ_DFE, st);
    return;
  }
#endif
  if (eq_tail_fused()) {                                    // one kernel: the delayed burst stays on the chip
    if (prof) prof->begin(TRXSIG_K_EQ_DFE, st);
    const dim3 g((B + 63) / 64), blk(256);
    if (fmt == TRXSIG_SAMPLES_F16) k_eq_dfe4<SmpF16><<<g, blk, 0, st>>>(dT, samples, off, len, B, amp, toa_eq, flags, w, bq, tap_ix, soft, hard, nsoft, stride);
    else k_eq_dfe4<SmpC32><<<g, blk, 0, st>>>(dT, samples, off, len, B, amp, toa_eq, flags, w, bq, tap_ix, soft, hard, nsoft, stride);
    if (prof) prof->end(TRXSIG_K_EQ_DFE, st);
    return;
  }
  if (prof) prof->begin(TRXSIG_K_EQ_DELAY, st);
  EQ_DELAY_LAUNCH(dT, samples, off, len, B, amp, toa_eq, flags, TRXSIG_F_DETECT, xd, xstride);
  if (prof) { prof->end(TRXSIG_K_EQ_DELAY, st); prof->begin(TRXSIG_K_EQ_DFE, st); }
  launch_eq_dfe(st, dT, xd, xstride, len, B, flags, toa_eq, w, bq, soft, hard, nsoft, stride, tap_ix);
  if (prof) prof->end(TRXSIG_K_EQ_DFE, st);
}

// k_eq_detect52 needs more LDS than a kernel gets by default: raise the limit once per instantiation
template <typename SMP, typename... A>
static hipError_t launch_eq_detect52(hipStream_t st, int B, A... a) {
  static bool raised[64] = {};                              // per device (a process may drive several)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!raised[dev]) {                                       // (idempotent: two threads racing here both set the same value)
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_eq_detect52<SMP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)EqDetect52<SMP>::kLds);
    if (attr != hipSuccess) return attr;
    raised[dev] = true;
  }
  k_eq_detect52<SMP><<<dim3((B + 255) / 256), dim3(256), EqDetect52<SMP>::kLds, st>>>(a...);
  return hipSuccess;
}
#define EQ_DETECT52_LAUNCH(...)                                                                             \
  do {                                                                                                      \
    const hipError_t e52_ = fmt == TRXSIG_SAMPLES_F16 ? launch_eq_detect52<SmpF16>(st, B, __VA_ARGS__)      \
                                                      : launch_eq_detect52<SmpC32>(st, B, __VA_ARGS__);     \
    if (e52_ != hipSuccess) { if (prof) prof->end(TRXSIG_K_EQUALIZE, st); return e52_; }                    \
  } while (0)

hipError_t trx_launch_design_dfe(hipStream_t st, const trx_c32 *chan, const trx_c32 *amp, const float *snr, int B, trx_c32 *w,
                                 trx_c32 *bq, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_EQUALIZE, st);
  k_design_dfe<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(chan, amp, snr, B, w, bq);
  if (prof) prof->end(TRXSIG_K_EQUALIZE, st);
  return hipGetLastError();
}

hipError_t trx_launch_equalize(hipStream_t st, const TrxTables *dT, const void *samples, int fmt, const int32_t *off,
                               const int32_t *len, int B, int tsc, float detect_thresh, float energy_thresh,
                               int variant52m, int max_toa, uint8_t *flags, trx_c32 *amp, float *toa, float *toa_eq,
                               trx_c32 *w, trx_c32 *bq, trx_c32 *xd, int xstride, float *soft, uint8_t *hard,
                               int nsoft, int stride, TrxProfiler *prof, bool geom52) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_EQUALIZE, st);
  if (geom52 && variant52m && max_toa == 4 && eq_detect_generic() == 0)
    EQ_DETECT52_LAUNCH(dT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, (cx *)amp, toa, toa_eq, (cx *)w, (cx *)bq, -1.0f,
                       0.0f, (float *)nullptr, (cx *)nullptr, (const uint8_t *)nullptr, (const float *)nullptr);
  else
    EQ_DETECT_LAUNCH(dT, samples, off, len, B, tsc, detect_thresh, energy_thresh,
                     variant52m, max_toa, flags, amp, toa, toa_eq, w, bq, -1.0f, 0.0f, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  if (prof) prof->end(TRXSIG_K_EQUALIZE, st);
  launch_eq_tail(st, dT, samples, fmt, off, len, B, amp, toa_eq, flags, w, bq, xd, xstride, soft, hard, nsoft, stride, nullptr, prof);
  return hipGetLastError();
}


// the two halves of trx_launch_equalize on their own (the Transceiver facade caches DFE taps per timeslot):
// channel estimate + designDFE only (energy gate off, explicit SNR threshold) ...
hipError_t trx_launch_estimate_dfe(hipStream_t st, const TrxTables *dT, const void *samples, int fmt, const int32_t *off,
                                   const int32_t *len, int B, int tsc, float detect_thresh, float snr_thresh,
                                   float snr_value, int variant52m, int max_toa, uint8_t *flags, trx_c32 *amp, float *toa,
                                   float *toa_eq, float *chan_off, trx_c32 *w, trx_c32 *bq, trx_c32 *chan, TrxProfiler *prof,
                                   const uint8_t *enable, const float *snr_in, bool geom52, int32_t *work, bool listed) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_EQUALIZE, st);
  // few bursts (the Transceiver/ variant): a wave per burst -- a marked subset (listed first; `work`: B + 1 ints) or a small call
  if (!variant52m && eq_detect_generic() == 0 && ((enable && work) || (!enable && B <= kEqWaveMax))) {
    const int32_t *list = nullptr, *count = nullptr;
    if (enable) {
      if (!listed) k_eq_list<<<dim3(1), dim3(1024), 0, st>>>(enable, B, work + 1, work);
      list = work + 1; count = work;
    }
    const int waves = B < kEqWaveMax ? B : kEqWaveMax;
    const dim3 grid((waves + 3) / 4), block(256);
    if (fmt == TRXSIG_SAMPLES_F16)
      k_eq_estimate_wave<SmpF16><<<grid, block, 0, st>>>(dT, samples, off, len, B, tsc, detect_thresh, flags, (cx *)amp, toa, toa_eq, (cx *)w, (cx *)bq,
                                                         snr_thresh, snr_value, chan_off, (cx *)chan, snr_in, list, count, eq_dense());
    else
      k_eq_estimate_wave<SmpC32><<<grid, block, 0, st>>>(dT, samples, off, len, B, tsc, detect_thresh, flags, (cx *)amp, toa, toa_eq, (cx *)w, (cx *)bq,
                                                         snr_thresh, snr_value, chan_off, (cx *)chan, snr_in, list, count, eq_dense());
    if (enable && B > eq_dense())                             // the marked bursts may be many (a cell in bad shape: every miss costs the slot its cache)
      EQ_DETECT_LAUNCH(dT, samples, off, len, B, tsc, detect_thresh, -1.0f, variant52m, max_toa, flags, amp, toa, toa_eq, w, bq, snr_thresh,
                       snr_value, chan_off, chan, enable, snr_in, count, eq_dense());
  } else if (geom52 && variant52m && max_toa == 4 && eq_detect_generic() == 0)
    EQ_DETECT52_LAUNCH(dT, samples, off, len, B, tsc, detect_thresh, -1.0f, flags, (cx *)amp, toa, toa_eq, (cx *)w, (cx *)bq, snr_thresh,
                       snr_value, chan_off, (cx *)chan, enable, snr_in);
  else
    EQ_DETECT_LAUNCH(dT, samples, off, len, B, tsc, detect_thresh, -1.0f, variant52m,
                     max_toa, flags, amp, toa, toa_eq, w, bq, snr_thresh, snr_value, chan_off, chan, enable, snr_in, nullptr, 0);
  if (prof) prof->end(TRXSIG_K_EQUALIZE, st);
  return hipGetLastError();
}
// ... and scaleVector(burst, 1/amp) + equalizeBurst(burst, toa_eq, w, b) with caller-supplied taps (7 + 5 per burst)
hipError_t trx_launch_equalize_taps(hipStream_t st, const TrxTables *dT, const void *samples, int fmt, const int32_t *off,
                                    const int32_t *len, int B, const trx_c32 *amp, const float *toa_eq,
                                    const uint8_t *flags, const trx_c32 *w, const trx_c32 *bq, trx_c32 *xd, int xstride,
                                    float *soft, uint8_t *hard, int nsoft, int stride, TrxProfiler *prof, const int32_t *tap_ix) {
  if (B <= 0) return hipSuccess;
  launch_eq_tail(st, dT, samples, fmt, off, len, B, amp, toa_eq, flags, w, bq, xd, xstride, soft, hard, nsoft, stride, tap_ix, prof);
  return hipGetLastError();
}
