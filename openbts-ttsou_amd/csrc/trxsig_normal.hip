// trxsig_normal.hip -- the default normal-burst path (three launches per batch):
//   k_tsc_corr   : 16 lanes (one DPP row) per burst, 4 bursts per wave.  Midamble correlation over the
//                  36-symbol window with the 16 non-zero taps, energy detect, argmax, and a small
//                  per-burst record of the lags around the peak.
//   k_tsc_peak2  : TWO lanes per burst.  The serial part of the reference (early-late bisection of
//                  peakDetect, valley RMS, threshold): the early and the late point of a step are the
//                  only work that can run side by side.  (k_tsc_peak: a lane per burst, the reference's
//                  loop as written; k_tsc_peak8: eight lanes, speculative.  A/B alternatives.)
//   k_demod      : one wave per burst (trxsig_demod.h).
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#include "trxsig_bisect.h"
#include "trxsig_corr.h"
#include "trxsig_demod.h"
#include "trxsig_rxgen.h"

#ifndef TRX_RXC_WPS
#define TRX_RXC_WPS 1
#endif
#ifndef TRX_RXD_WPS
#define TRX_RXD_WPS 1
#endif

namespace {

template <int SPS, unsigned TAPCLS>
__global__ __launch_bounds__(256, TRX_CORR_WPS) void k_tsc_corr(const TrxTables *__restrict__ T,
                                                  const cx *__restrict__ samples,
                                                  const int32_t *__restrict__ offset,
                                                  const int32_t *__restrict__ length, int B, TapArg taps,
                                                  cx *__restrict__ rec, int Bpad) {
  typedef CorrGeom<SPS> G;
  // one LDS row per burst, owned by the 16 lanes of its DPP row; no workgroup barrier anywhere.
  // The row first holds the zero-padded window, later (same storage) the correlation.
  // The row first holds |x[i]|^2 of the energy window, then the zero-padded window, then the correlation.
  static_assert(8 * G::WPAD >= 4 * G::NE, "the energy norms are staged in the row itself");
  __shared__ __attribute__((aligned(16))) cx rows[16][G::WPAD];
  (void)T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane >> 4, r = lane & 15;
  const int slot = wave * 4 + row;                         // burst slot in this workgroup

  cx tap[16];
#pragma unroll
  for (int k = 0; k < 16; k++) tap[k] = mk(taps.v[2 * k], taps.v[2 * k + 1]);

  // (sixteen-byte sample loads -- lane r the sample PAIRS r, r + 16, ..: 5 + 3 loads instead of 9 + 5, 16-byte LDS writes -- were built
  //  and measured in round 5, same box, three times back to back: 34.2-35.9 against 33.0-34.0 us; not kept, profiles/r05_corr_wide_loads_ab.txt)
  typedef CorrIn<SPS> In;
  In in[TRX_CORR_ROUNDS];
#pragma unroll
  for (int i = 0; i < TRX_CORR_ROUNDS; i++)
    corr_issue<SPS>(in[i], (blockIdx.x * TRX_CORR_ROUNDS + i) * 16 + slot, B, r, samples, offset, length);
#pragma unroll
  for (int i = 0; i < TRX_CORR_ROUNDS; i++) {
    int M;
    float energy;
    corr_round<SPS, true, true, TAPCLS, In>(in[i], rows[slot], reinterpret_cast<float4 *>(rows[slot]), lane, r, tap, rec, Bpad, M, energy);
  }
}


#ifdef TRX_TUNING_BUILD   /* k_tsc_peak (a lane per burst) and k_tsc_peak8 (eight lanes, speculative): A/B alternatives, tuning build only */
// ---------------------------------------------------------------------------------------------
// k_tsc_peak: one lane per burst.  peakDetect's early-late bisection (sigProcLib.cpp:684-701),
//   the bogus-TOA check, the valley RMS, the detection threshold, amp = peak/gain and the TOA
//   bookkeeping of analyzeTrafficBurst (:959-1000, 1035), plus energyDetect's decision (:929-931).
// ---------------------------------------------------------------------------------------------
template <int SPS>
__global__ __launch_bounds__(64) void k_tsc_peak(const TrxTables *__restrict__ T,
                                                 const cx *__restrict__ rec, int Bpad, int B, int tsc,
                                                 float detect_thresh, float energy_thresh,
                                                 uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                 float *__restrict__ toa_out,
                                                 float *__restrict__ avgpwr_out) {
  typedef CorrGeom<SPS> G;
  __shared__ cx loc[26][64];                               // lags M-12 .. M+11 of each lane's burst; [24],[25] zero
  __shared__ float pw[G::NS][64];                          // |corr|^2 of lags M-H .. M+H (valley power)
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < B;
  const int bb = live ? b : B - 1;

  // ---- everything this lane will need from the record, loaded up front (coalesced across lanes) ----
  const cx meta = rec_slot(rec, Bpad, G::NS, bb);
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const bool good = M != -2;
#pragma unroll
  for (int s = 0; s < G::NS; s++) {
    const cx v = rec_slot(rec, Bpad, s, bb);
    pw[s][lane] = norm2(v);
    const int j = s - (G::H - 12);
    if (j >= 0 && j < 24) {
      const int lag = M - 12 + j;
      loc[j][lane] = (lag > G::NL - 2) ? mk(0, 0) : v;     // interpolatePoint never uses the last sample (:646)
    }
  }
  loc[24][lane] = mk(0, 0);
  loc[25][lane] = mk(0, 0);
  // (each lane only ever reads its own column: no barrier needed)

  float peakIx;
  const cx peak = peak_bisect<64>(T->sinc_grid, loc, lane, M, &peakIx);

  // ---- analyzeTrafficBurst tail ----
  float toa = peakIx;
  cx amp = peak;
  bool detected = false;
  // energy_thresh < 0 disables the gate (trxsig.h)
  const bool energy_ok = good && (energy_thresh < 0.0f ||
                                  energy / (float)(unsigned)G::NE > energy_thresh * energy_thresh);
  if (!(toa < 0.0f) && !(toa > (float)G::NL) && good) {
    const int p = (int)rintf(toa);
    float valley = 0.0f;
    int numRms = 0;
#pragma unroll
    for (int i = 2 * SPS; i <= 5 * SPS; i++) {             // :971-980, this order
      const int lo = p - i, hi = p + i;
      int slo = lo - M + G::H, shi = hi - M + G::H;        // 0 .. NS-1 because |p - M| <= 1
      slo = slo < 0 ? 0 : slo; shi = shi > G::NS - 1 ? G::NS - 1 : shi;
      const float vlo = pw[slo][lane], vhi = pw[shi][lane];
      if (lo >= 0) { valley += vlo; numRms++; }
      if (hi < G::NL) { valley += vhi; numRms++; }
    }
    if (numRms < 2) {
      amp = mk(0, 0);
    } else {
      const float RMS = (float)((double)sqrtf(valley / (float)numRms) + 0.00001);   // :989
      const float peakToMean = sqrtf(norm2(amp)) / RMS;   // Complex::abs() via double sqrt == sqrtf
      amp = cdiv(amp, T->mid_gain[tsc]);                   // :997
      toa = toa - T->mid_toa[tsc];                         // :998
      toa = toa - (float)((66 - 56) * SPS);                // :1000
      detected = peakToMean > detect_thresh;
    }
  } else {
    amp = mk(0, 0);                                        // "bogus result" (:964-968); TOA left as is
  }
  if (!energy_ok) { amp = mk(0, 0); toa = 0.0f; detected = false; }   // Transceiver.cpp:298-306

  if (live) {
    uint8_t fl = 0;
    if (!good) fl = TRXSIG_F_BADLEN;
    else fl = (energy_ok ? TRXSIG_F_ENERGY : 0) | (detected ? TRXSIG_F_DETECT : 0);
    flags[b] = fl;
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = good ? energy / (float)(unsigned)G::NE : 0.0f;
  }
}


// ---------------------------------------------------------------------------------------------
// k_tsc_peak8: k_tsc_peak's job (peakDetect's bisection + analyzeTrafficBurst's tail from the
//   detect->peak record) with EIGHT lanes per burst and the bisection speculated two levels at a time
//   (fused_point / fused_decide, see k_normal_fused): 6 dependent point evaluations instead of 10
//   dependent steps of two, eight waves per SIMD instead of one.  The sinc rows of the first three
//   super-steps (nodes on multiples of 16/512) come from an LDS copy; the last three gather from L2.
//   Measured SLOWER than k_tsc_peak (25 vs 18 us per 64 K bursts): 48 lane-evaluations per burst instead
//   of 19, each pulling 21 correlation words and a sinc row through the LDS, make it LDS-bandwidth
//   bound (ablation: neither the L2 gathers nor occupancy matter).  Kept as an A/B option
//   (TRXSIG_TUNE_SPECULATIVE_PEAK).
// ---------------------------------------------------------------------------------------------
template <int SPS>
struct Peak8Geom {
  typedef CorrGeom<SPS> G;
  static constexpr int NV = 2 * (3 * SPS + 1);
  static constexpr int O_PW = 26 * 2;                                   // floats: after loc[26]
  static constexpr int O_V = O_PW + ((G::NS + 3) & ~3);
  static constexpr int STRIDE = O_V + ((NV + 3) & ~3);                  // floats per burst (multiple of 4)
};

template <int SPS>
__global__ __launch_bounds__(256, 8) void k_tsc_peak8(const TrxTables *__restrict__ T, const cx *__restrict__ rec, int Bpad,
                                                   int B, cx gain_inv, float mid_toa, float detect_thresh,
                                                   float energy_thresh, uint8_t *__restrict__ flags,
                                                   cx *__restrict__ amp_out, float *__restrict__ toa_out,
                                                   float *__restrict__ avgpwr_out) {
  typedef CorrGeom<SPS> G;
  typedef Peak8Geom<SPS> P8;
  __shared__ __attribute__((aligned(16))) float stab[32][24];            // sinc rows f = 0, 16, .., 496
  __shared__ __attribute__((aligned(16))) float scratch[32][P8::STRIDE];
  {
    float tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { const int ix = threadIdx.x * 3 + k; tv[k] = T->sinc_grid[16 * (ix / 24)][ix % 24]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { const int ix = threadIdx.x * 3 + k; stab[ix / 24][ix % 24] = tv[k]; }
  }
  const int lane = threadIdx.x & 63;
  const int r = lane & 7;
  const int slot = threadIdx.x >> 3;                       // burst slot in the workgroup
  const int b = blockIdx.x * 32 + slot;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  float *S = scratch[slot];
  cx *loc = reinterpret_cast<cx *>(S);
  float *pw = S + P8::O_PW, *V = S + P8::O_V;

  const cx meta = rec_slot(rec, Bpad, G::NS, bb);
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const bool good = M != -2;
#pragma unroll
  for (int s0 = 0; s0 < G::NS; s0 += 8) {
    const int sl = s0 + r;
    if (sl < G::NS) {
      const cx v = rec_slot(rec, Bpad, sl, bb);
      pw[sl] = norm2(v);
      const int j = sl - (G::H - 12);
      if (j >= 0 && j < 24) loc[j] = (M - 12 + j > G::NL - 2) ? mk(0, 0) : v;   // never the last sample (:646)
    }
  }
  if (r < 2) loc[24 + r] = mk(0, 0);
  __syncthreads();                                         // stab complete (the only barrier); also orders the scratch writes

  int e = 0;                                               // early = M-1 + e/512
  asm volatile("" : "+v"(e));
  bool active = true;
  cx peak = mk(0, 0);
  {
    const int rel2 = kFusedRel2.v[r], rel1 = kFusedRel1.v[r];
#pragma unroll
    for (int st = 0; st < 4; st++) {                       // increments 256,128 | 64,32 | 16,8 | 4,2
      const int inc_last = 128 >> (2 * st);
      const int el = e + (rel2 >> 2) * inc_last;
      float srow[24];
      if (st < 3) {                                        // nodes on multiples of 16/512: the LDS copy
        const float4 *rw = reinterpret_cast<const float4 *>(stab[(el & 511) >> 4]);
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const float4 t4 = rw[q];
          srow[4 * q] = t4.x; srow[4 * q + 1] = t4.y; srow[4 * q + 2] = t4.z; srow[4 * q + 3] = t4.w;
        }
      } else {
        fused_row(T, el, srow);
      }
      const cx pt = fused_point(loc, el, rel2 & 3, srow);
      fused_decide<8, 2, false>(pt, lane, 2 * inc_last, e, active, peak);
    }
    {                                                      // the ninth step: +-1
      float srow[24];
      fused_row(T, e, srow);
      const cx pt = fused_point(loc, e, rel1 & 3, srow);
      fused_decide<8, 1, false>(pt, lane, 1, e, active, peak);
    }
    float srow[24];                                        // interpolatePoint(early + 1) where the loop stopped (:699-700)
    fused_row(T, e, srow);
    peak = fused_point(loc, e, 1, srow);
    asm volatile("" : "+v"(peak.r), "+v"(peak.i));         // (finished here: not to be interleaved with the tail)
  }
  cx amp;
  float toa;
  bool detected, energy_ok;
  fused_tail<SPS, 8>([&](int lag) { const int sl = lag - M + G::H; return (sl < 0 || sl >= G::NS) ? 0.0f : pw[sl]; }, V, r, M, e,
                     peak, good, energy, gain_inv, mid_toa, detect_thresh, energy_thresh, amp, toa, detected, energy_ok);
  if (live && r == 0) {
    uint8_t fl = 0;
    if (!good) fl = TRXSIG_F_BADLEN;
    else fl = (energy_ok ? TRXSIG_F_ENERGY : 0) | (detected ? TRXSIG_F_DETECT : 0);
    flags[b] = fl;
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = good ? energy / (float)(unsigned)G::NE : 0.0f;
  }
}



#endif  // TRX_TUNING_BUILD

// ---------------------------------------------------------------------------------------------
// k_tsc_peak2: k_tsc_peak's job with TWO lanes per burst (pair_bisect, trxsig_bisect.h: early and late point of a
//   step side by side, sinc table in LDS, correlation window in registers) and two waves per SIMD instead of one.
//   (k_tsc_peak gathers both candidate rows from L2 a step ahead: 12 divergent 16-byte gathers per lane and step,
//   which the CU's vector cache serves one lane at a time -- 1.4 k cycles per step with few distinct rows, 3 k with
//   256.)  The valley powers come from registers too (three candidate alignments, selected by rint(toa) - M).
// ---------------------------------------------------------------------------------------------
// (round 4: the kernel's first 12.6 k of 22 k cycles per wave were its loads -- 66 per thread, the table's 48 KB once per 128 bursts
//  among them, profiles/r04_peak_probe.txt.  A workgroup is now 512 threads = 256 bursts round one copy of the table.  Leaving the
//  valley's loads to the even lane alone -- the odd lane's tail is never stored -- was slower: 17.8 against 14.7 us; the pair's two
//  lanes ask for the same address, which the load unit merges, and a load under a branch is waited for at the join.)
#ifndef TRX_PEAK2_THREADS
#define TRX_PEAK2_THREADS 512
#endif
constexpr int kPeak2Threads = TRX_PEAK2_THREADS;
template <int SPS>
__global__ __launch_bounds__(kPeak2Threads) void k_tsc_peak2(const TrxTables *__restrict__ T, const cx *__restrict__ rec, int Bpad,
                                                   int B, cx gain_inv, float mid_toa, float detect_thresh,
                                                   float energy_thresh, uint8_t *__restrict__ flags,
                                                   cx *__restrict__ amp_out, float *__restrict__ toa_out,
                                                   float *__restrict__ avgpwr_out) {
  typedef CorrGeom<SPS> G;
  constexpr int NL = G::NL, NE = G::NE, NP = 3 * SPS + 3;  // NP: valley powers kept per side
  __shared__ __attribute__((aligned(16))) SincLds stab;
  const int tid = threadIdx.x;
  const int h = tid & 1;                                   // 0: early point (and the final one), 1: late point
  const int b = blockIdx.x * (kPeak2Threads / 2) + (tid >> 1);
  const bool live = b < B;
  const int bb = live ? b : B - 1;
#ifdef TRX_PEAK_PROBE                                      // tools/peak_probe.py: clock64() stamps come back through avgpwr
  long long pt_[16] = {0};
  int pk_ = 0;
#define TRX_STAMP() pt_[pk_++] = clock64()
#else
#define TRX_STAMP()
#endif
  TRX_STAMP();

  // ---- loads first: this thread's 12 float4 of the table, then its part of the detect->peak record ----
  float4 tv[3072 / kPeak2Threads];
  sinc_lds_issue<kPeak2Threads>(T, tid, tv);
  const cx meta = rec_slot(rec, Bpad, G::NS, bb);
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const bool good = M != -2;
  const float4 *rec4 = reinterpret_cast<const float4 *>(rec);
  cx q[23];                                                // pair_bisect's window: corr[M - 12 + k + 2h] = record slots H - 12 + 2h + k
  {
    // twelve slot pairs from pair (H - 12) / 2 + h on: slot H - 12 + k + 2h is half (H - 12 + k) & 1 of pair number
    // ((H - 12 + k) >> 1) - ((H - 12) >> 1) of them, whatever h is
    constexpr int S0 = G::H - 12, P0 = S0 >> 1;
    float4 pw[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pw[i] = rec4[(size_t)(P0 + h + i) * Bpad + bb];
#pragma unroll
    for (int k = 0; k < 23; k++) {
      const int ix = k + 2 * h;
      const float4 p4 = pw[((S0 + k) >> 1) - P0];
      const cx v = ((S0 + k) & 1) ? mk(p4.z, p4.w) : mk(p4.x, p4.y);
      q[k] = (M - 12 + ix > NL - 2) ? mk(0, 0) : v;        // interpolatePoint never uses the last sample (:646)
    }
  }
  // corr at M - (5sps+1) + k and M + (2sps-1) + k, k < NP: every lag the valley can touch (|rint(toa) - M| <= 1);
  // in flight with the rest, first needed in the tail
  // (the pair shares them: the even lane loads the side below the peak, the odd lane the side above -- one unconditional load each, the
  //  slot chosen by the lane's parity -- and the powers cross over by DPP when the tail wants them)
  cx vmine_[NP];
  constexpr int VLO = G::H - (5 * SPS + 1), VHI = G::H + (2 * SPS - 1);
  if constexpr ((VLO & 1) == 0 && (VHI & 1) == 0) {         // both sides start on a pair (sps 4): eight 16-byte loads
    float4 pv[(NP + 1) / 2];
#pragma unroll
    for (int i = 0; i < (NP + 1) / 2; i++) pv[i] = rec4[(size_t)(((h ? VHI : VLO) >> 1) + i) * Bpad + bb];
#pragma unroll
    for (int k = 0; k < NP; k++) vmine_[k] = (k & 1) ? mk(pv[k >> 1].z, pv[k >> 1].w) : mk(pv[k >> 1].x, pv[k >> 1].y);
  } else {
#pragma unroll
    for (int k = 0; k < NP; k++) vmine_[k] = rec_slot(rec, Bpad, (h ? VHI : VLO) + k, bb);
  }
  sinc_lds_store<kPeak2Threads>(stab, tid, tv);
  TRX_STAMP();
  __syncthreads();                                         // the only barrier
  TRX_STAMP();

  int e;
  const cx peak = pair_bisect(stab, q, h, e);
  TRX_STAMP();
  // ---- analyzeTrafficBurst's tail (k_tsc_peak's arithmetic) ----
  const float early = (float)(M - 1) + (float)e * 0.001953125f;
  float toa = early + 1.0f;
  cx amp = peak;
  bool detected = false;
  const bool energy_ok = good && (energy_thresh < 0.0f || energy / (float)(unsigned)NE > energy_thresh * energy_thresh);
  const bool sane = !(toa < 0.0f) && !(toa > (float)NL) && good;
  if (sane) {
    const int pk = (int)rintf(toa);
    const int a = pk - M + 1;                              // 0, 1 or 2
    // valley in the reference's order (:971-980): i = 2sps..5sps, (peak - i) then (peak + i); the terms the
    // reference skips (index < 0 or >= n) are +0 in the record, and adding +0 to a sum of non-negative terms
    // changes nothing; numRms is counted arithmetically.
    float plo[NP], phi[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const float mine = norm2(vmine_[k]);
      const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0xB1, 0xf, 0xf, true));   // lane ^ 1
      plo[k] = h ? other : mine; phi[k] = h ? mine : other;
    }
    float valley = 0.0f;
#pragma unroll
    for (int i = 2 * SPS; i <= 5 * SPS; i++) {
      const int t = 5 * SPS - i, u = i - 2 * SPS;
      const float vlo = a == 0 ? plo[t] : (a == 1 ? plo[t + 1] : plo[t + 2]);
      const float vhi = a == 0 ? phi[u] : (a == 1 ? phi[u + 1] : phi[u + 2]);
      valley = valley + vlo;
      valley = valley + vhi;
    }
    int nlo = (pk < 5 * SPS ? pk : 5 * SPS) - 2 * SPS + 1;           // i <= pk
    int nhi = (NL - 1 - pk < 5 * SPS ? NL - 1 - pk : 5 * SPS) - 2 * SPS + 1;   // pk + i <= NL-1
    nlo = nlo < 0 ? 0 : nlo; nhi = nhi < 0 ? 0 : nhi;
    const int numRms = nlo + nhi;
    if (numRms < 2) {
      amp = mk(0, 0);
    } else {
      const float RMS = (float)((double)sqrtf(valley / (float)numRms) + 0.00001);   // :989
      const float peakToMean = sqrtf(norm2(amp)) / RMS;
      amp = cmul(amp, gain_inv);                           // amp / gain = amp * gain.inv() (Complex.h:85), :997
      toa = toa - mid_toa;                                 // :998
      toa = toa - (float)((66 - 56) * SPS);                // :1000
      detected = peakToMean > detect_thresh;
    }
  } else {
    amp = mk(0, 0);                                        // "bogus result" (:964-968); TOA left as is
  }
  if (!energy_ok) { amp = mk(0, 0); toa = 0.0f; detected = false; }   // Transceiver.cpp:298-306

  if (live && h == 0) {
    uint8_t fl = 0;
    if (!good) fl = TRXSIG_F_BADLEN;
    else fl = (energy_ok ? TRXSIG_F_ENERGY : 0) | (detected ? TRXSIG_F_DETECT : 0);
    flags[b] = fl;
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = good ? energy / (float)(unsigned)G::NE : 0.0f;
  }
#ifdef TRX_PEAK_PROBE
  TRX_STAMP();
  if (live && avgpwr_out) {
    long long v = 0;
    for (int k = 1; k < 16; k++) if ((b & 15) == k) v = pt_[k] - pt_[0];
    if (h == 0) avgpwr_out[b] = (float)v;
  }
#endif
#undef TRX_STAMP
}

// ---------------------------------------------------------------------------------------------
// The receive front end fused in (trxsig_rxgen.h; sps = 4): k_tsc_corr_rx / k_demod_rx are k_tsc_corr / k_demod with the
// burst's samples computed from the raw int16 stream instead of loaded: the wave first parks the stretch of raw samples
// its burst depends on (140 for the correlator's two windows, 236 for the whole burst) in LDS as floats, then every lane
// produces the samples it would have loaded -- four multiply-adds each -- and the kernel carries on unchanged.
// ---------------------------------------------------------------------------------------------
template <int SPS, unsigned TAPCLS>
__global__ __launch_bounds__(256, TRX_RXC_WPS) void k_tsc_corr_rx(TrxRxGen a, int B, TapArg taps, cx *__restrict__ rec, int Bpad) {
  static_assert(SPS == 4, "the fused front end is the 260 : 96 resampler");
  typedef CorrGeom<SPS> G;
  constexpr int XCAP = 160;                                // raw samples behind resampled samples [0, 92*SPS) of a burst: <= 140
  static_assert(G::WPAD >= XCAP, "the raw stretch is parked in the burst's own row");
  __shared__ __attribute__((aligned(16))) cx rows[16][G::WPAD];
  __shared__ float4 tpb_s[RXG_NT];
  if (threadIdx.x < RXG_NT) tpb_s[threadIdx.x] = a.tpb[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane >> 4, r = lane & 15;
  const int slot = wave * 4 + row;
  cx tap[16];
#pragma unroll
  for (int k = 0; k < 16; k++) tap[k] = mk(taps.v[2 * k], taps.v[2 * k + 1]);
  const int b = blockIdx.x * 16 + slot;
  CorrIn<SPS> in;
  in.b = b; in.live = b < B; in.good = in.live;
  const RxBurst u = rx_burst(a, in.live ? b : 0);
  const int jlo = rx_index(u.g0, a.skipD).j0 - 3;
  cx *X = rows[slot];
  {
    short2 v[XCAP / 16];
#pragma unroll
    for (int i = 0; i < XCAP / 16; i++) v[i] = rx_raw(a, u.s, jlo + r + 16 * i);   // (a dead row reads burst 0's stretch)
#pragma unroll
    for (int i = 0; i < XCAP / 16; i++) X[r + 16 * i] = rx_widen(a, v[i]);
  }
  wave_lds_fence();
  {
    static_assert(G::NL % 16 == 0 && G::NE % 16 == 0, "every lane owns a sample in every round");
    const int nb = rx_boundary(u.g0);                      // samples from the burst's start to the next chunk boundary
    RxIdx ix = rx_index(u.g0 + 56 * SPS + r, a.skipD);     // window sample q = r + 16 i
#pragma unroll
    for (int i = 0; i < CorrIn<SPS>::NW; i++) {
      in.w[i] = rx_sample(X, tpb_s, jlo, ix, nb - (56 * SPS + r + 16 * i), a.w0, a.w1);
      ix = rx_step<16>(ix);
      if (i % 3 == 2) asm volatile("" : "+v"(in.w[i].r), "+v"(in.w[i].i), "+v"(in.w[i - 1].r), "+v"(in.w[i - 1].i), "+v"(in.w[i - 2].r), "+v"(in.w[i - 2].i) : : "memory");   // (see k_demod_rx)
    }
    ix = rx_index(u.g0 + r, a.skipD);                      // energy-window sample i = r + 16 q
#pragma unroll
    for (int q = 0; q < G::NEQ; q++) {
      in.e[q] = rx_sample(X, tpb_s, jlo, ix, nb - (r + 16 * q), a.w0, a.w1);
      ix = rx_step<16>(ix);
      if (q % 3 == 2) asm volatile("" : "+v"(in.e[q].r), "+v"(in.e[q].i), "+v"(in.e[q - 1].r), "+v"(in.e[q - 1].i), "+v"(in.e[q - 2].r), "+v"(in.e[q - 2].i) : : "memory");
    }
  }
  wave_lds_fence();                                        // the raw stretch is dead: the row becomes corr_round's
  int M;
  float energy;
  corr_round<SPS, true, true, TAPCLS>(in, rows[slot], reinterpret_cast<float4 *>(rows[slot]), lane, r, tap, rec, Bpad, M, energy);
}

template <int SPS, bool TOL = false>
__global__ __launch_bounds__(64 * TRX_DEMOD_WAVES, TRX_RXD_WPS) void k_demod_rx(const TrxTables *__restrict__ T, TrxRxGen a, int B,
                                                                  const cx *__restrict__ amp_in, const float *__restrict__ toa_in,
                                                                  const uint8_t *__restrict__ flags, int need_mask,
                                                                  float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft,
                                                                  int stride) {
  static_assert(SPS == 4, "the fused front end is the 260 : 96 resampler");
  typedef DemodGeom<SPS, 148> G;
  constexpr int XCAP = 256;                                // raw samples behind one burst: <= 236
  static_assert(G::U >= XCAP, "the raw stretch is parked in the burst's staging area");
  __shared__ cx ph[TRX_DEMOD_WAVES][G::U];
  __shared__ float4 tpb_s[RXG_NT];
  if (threadIdx.x < RXG_NT) tpb_s[threadIdx.x] = a.tpb[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * TRX_DEMOD_WAVES + wave;       // wave-uniform
  if (b >= B) return;
  float *sb = soft + (size_t)b * stride;
  uint8_t *hb = hard ? hard + (size_t)b * stride : nullptr;
  const cx amp = amp_in[b];
  const float toa = toa_in[b];
  bool enabled = fabsf(toa) <= 4096.0f;                    // also rejects NaN/inf TOA
  if (flags) enabled = enabled && (need_mask ? ((flags[b] & need_mask) == need_mask) : (flags[b] != 0));
  if (!enabled) {
    for (int m = lane; m < nsoft; m += 64) { sb[m] = 0.0f; if (hb) hb[m] = 0; }
    return;
  }
  const RxBurst u = rx_burst(a, b);
  const int jlo = rx_index(u.g0, a.skipD).j0 - 3;
  cx *X = ph[wave];
  {
    short2 v[XCAP / 64];
#pragma unroll
    for (int i = 0; i < XCAP / 64; i++) v[i] = rx_raw(a, u.s, jlo + lane + 64 * i);
#pragma unroll
    for (int i = 0; i < XCAP / 64; i++) X[lane + 64 * i] = rx_widen(a, v[i]);
  }
  wave_lds_fence();
  // sample n = lane + 64 i of the burst (consecutive lanes = consecutive samples: consecutive tap slots, neighbouring raw samples)
  constexpr int NSL = (157 * SPS + 63) / 64;
  cx sv[NSL];
  {
    const int nb = rx_boundary(u.g0);
    RxIdx ix = rx_index(u.g0 + lane, a.skipD);
#pragma unroll
    for (int i = 0; i < NSL; i++) {
      // (samples past the burst's end -- last round only -- read staged or stale LDS inside this wave's area and are not staged)
      sv[i] = rx_sample(X, tpb_s, jlo, ix, nb - (lane + 64 * i), a.w0, a.w1);
      ix = rx_step<64>(ix);
      // (pins the sample before the next ones' LDS reads are issued: hipcc otherwise issues all thirty reads first -- 152 VGPRs)
      if (i % 2 == 1) asm volatile("" : "+v"(sv[i].r), "+v"(sv[i].i), "+v"(sv[i - 1].r), "+v"(sv[i - 1].i) : : "memory");
    }
  }
  const int N = u.N;
  auto stage = [&](cx *P, cx inv, int lo) {                // scaleVector (:713-723) into the polyphase staging area
    typedef DemodGeom<SPS, 148> D;
    const int u0 = lane + lo;                              // sample n = lane + 64 i sits at position u0 + 64 i: same phase, 16 entries on
    if (u0 >= 0) {
      cx *p0 = P + (u0 % SPS) * D::QLEN + u0 / SPS;
#pragma unroll
      for (int i = 0; i < NSL; i++)
        if (lane + 64 * i < N && u0 + 64 * i < D::U) p0[i * (64 / SPS)] = cmul(sv[i], inv);
    } else {
#pragma unroll
      for (int i = 0; i < NSL; i++) {
        const int uu = u0 + 64 * i;
        if (lane + 64 * i < N && uu >= 0 && uu < D::U) P[(uu % SPS) * D::QLEN + uu / SPS] = cmul(sv[i], inv);
      }
    }
  };
  if (TOL) {                                                // TRXSIG_SOFT_TOLERANCE: the rearranged form unless this burst has to be exact
    float xm = 0.0f;
#pragma unroll
    for (int i = 0; i < NSL; i++)
      if (lane + 64 * i < N) xm = max3_abs(sv[i].r, sv[i].i, xm);
    auto stage_raw = [&](cx *P, int lo) {
      typedef DemodGeom<SPS, 148> D;
      const int u0 = lane + lo;
      if (u0 >= 0) {
        cx *p0 = P + (u0 % SPS) * D::QLEN + u0 / SPS;
#pragma unroll
        for (int i = 0; i < NSL; i++)
          if (lane + 64 * i < N && u0 + 64 * i < D::U) p0[i * (64 / SPS)] = sv[i];
      } else {                                              // (an access burst's delay: some of this lane's samples fall off the front)
#pragma unroll
        for (int i = 0; i < NSL; i++) {
          const int uu = u0 + 64 * i;
          if (lane + 64 * i < N && uu >= 0 && uu < D::U) P[(uu % SPS) * D::QLEN + uu / SPS] = sv[i];
        }
      }
    };
    if (fused_demod_tol_ex<SPS>(T, ph[wave], N, amp, toa, lane, sb, hb, nsoft, xm, stage_raw)) return;
  }
  fused_demod_ex<SPS, 64>(T, ph[wave], N, amp, toa, lane, sb, hb, nsoft, stage, [] {}, nullptr, nullptr);   // (starts with an LDS fence)
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int trx_rec_slots(int sps) {
  switch (sps) {
    case 1: return 2 * CorrGeom<1>::NPAIR;                  // (slot pairs: trxsig_corr.h)
    case 2: return 2 * CorrGeom<2>::NPAIR;
    case 4: return 2 * CorrGeom<4>::NPAIR;
  }
  return 0;
}

template <int S>
static void launch_tsc_detect(hipStream_t st, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples, const int32_t *off,
                              const int32_t *len, int B, int tsc, float detect_thresh, float energy_thresh,
                              trx_c32 *rec, int Bpad, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr,
                              int variant, TrxProfiler *prof) {
  TapArg ta;
  for (int k = 0; k < 16; k++) { ta.v[2 * k] = hT->mid_ctap[tsc][k].r; ta.v[2 * k + 1] = hT->mid_ctap[tsc][k].i; }
  if (prof) prof->begin(TRXSIG_K_TSC_CORR, st);
  const dim3 cgrid((B + 16 * TRX_CORR_ROUNDS - 1) / (16 * TRX_CORR_ROUNDS));
  if (!(variant & 1) && tap_classes(hT, tsc) == TapPattern<S>::value)
    k_tsc_corr<S, TapPattern<S>::value><<<cgrid, dim3(256), 0, st>>>(dT, samples, off, len, B, ta, rec, Bpad);
  else
    k_tsc_corr<S, TRX_TAPS_GENERIC><<<cgrid, dim3(256), 0, st>>>(dT, samples, off, len, B, ta, rec, Bpad);
  if (prof) { prof->end(TRXSIG_K_TSC_CORR, st); prof->begin(TRXSIG_K_TSC_PEAK, st); }
  // gain.inv() (Complex.h:154-160) in the reference's float arithmetic; this file is built with -ffp-contract=off
  const trx_c32 g = hT->mid_gain[tsc];
  const float n = g.i * g.i + g.r * g.r;
  trx_c32 ginv; ginv.r = g.r / n; ginv.i = -g.i / n;
#ifdef TRX_TUNING_BUILD
  if (variant & 4) {
    k_tsc_peak<S><<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, rec, Bpad, B, tsc, detect_thresh, energy_thresh,
                                                            flags, amp, toa, avgpwr);
  } else if (variant & 2) {
    k_tsc_peak8<S><<<dim3((B + 31) / 32), dim3(256), 0, st>>>(dT, rec, Bpad, B, ginv, hT->mid_toa[tsc], detect_thresh,
                                                              energy_thresh, flags, amp, toa, avgpwr);
  } else
#endif
  {
    k_tsc_peak2<S><<<dim3((B + kPeak2Threads / 2 - 1) / (kPeak2Threads / 2)), dim3(kPeak2Threads), 0, st>>>(dT, rec, Bpad, B, ginv, hT->mid_toa[tsc], detect_thresh,
                                                                energy_thresh, flags, amp, toa, avgpwr);
  }
  if (prof) prof->end(TRXSIG_K_TSC_PEAK, st);
}

// the verdict of a batch (flags, amp, TOA) into a private copy: the demodulator of TRXSIG_TUNE_DEMOD_BESIDE reads it while the next
// call's k_tsc_peak2 already overwrites the caller's arrays
namespace {
__global__ __launch_bounds__(256) void k_copy_verdict(const uint8_t *__restrict__ f, const cx *__restrict__ a, const float *__restrict__ t, int B,
                                                      uint8_t *__restrict__ fo, cx *__restrict__ ao, float *__restrict__ to) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  fo[b] = f[b]; ao[b] = a[b]; to[b] = t[b];
}
}  // namespace
hipError_t trx_launch_copy_verdict(hipStream_t st, const uint8_t *flags, const trx_c32 *amp, const float *toa, int B, uint8_t *flags_out,
                                   trx_c32 *amp_out, float *toa_out) {
  if (B <= 0) return hipSuccess;
  k_copy_verdict<<<dim3((B + 255) / 256), dim3(256), 0, st>>>(flags, amp, toa, B, flags_out, amp_out, toa_out);
  return hipGetLastError();
}

hipError_t trx_launch_tsc_detect(hipStream_t st, int sps, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples,
                                 const int32_t *off, const int32_t *len, int B, int tsc,
                                 float detect_thresh, float energy_thresh, trx_c32 *rec, int Bpad,
                                 uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr, int variant,
                                 TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  switch (sps) {
    case 1: launch_tsc_detect<1>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, rec, Bpad, flags, amp, toa, avgpwr, variant, prof); break;
    case 2: launch_tsc_detect<2>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, rec, Bpad, flags, amp, toa, avgpwr, variant, prof); break;
    case 4: launch_tsc_detect<4>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, rec, Bpad, flags, amp, toa, avgpwr, variant, prof); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t trx_launch_rx_normal(hipStream_t st, const TrxTables *dT, const TrxTables *hT, const TrxRxGen &gen, int B, int tsc,
                                float detect_thresh, float energy_thresh, trx_c32 *rec, int Bpad, uint8_t *flags, trx_c32 *amp,
                                float *toa, float *avgpwr, float *soft, uint8_t *hard, int nsoft, int stride, int generic_taps,
                                TrxProfiler *prof, int soft_tolerance) {
  if (B <= 0) return hipSuccess;
  if (nsoft > 148 || gen.nb <= 0) return hipErrorInvalidValue;
  constexpr int S = 4;
  TapArg ta;
  for (int k = 0; k < 16; k++) { ta.v[2 * k] = hT->mid_ctap[tsc][k].r; ta.v[2 * k + 1] = hT->mid_ctap[tsc][k].i; }
  if (prof) prof->begin(TRXSIG_K_TSC_CORR, st);
  const dim3 cgrid((B + 15) / 16);
  if (!generic_taps && tap_classes(hT, tsc) == TapPattern<S>::value)
    k_tsc_corr_rx<S, TapPattern<S>::value><<<cgrid, dim3(256), 0, st>>>(gen, B, ta, rec, Bpad);
  else
    k_tsc_corr_rx<S, TRX_TAPS_GENERIC><<<cgrid, dim3(256), 0, st>>>(gen, B, ta, rec, Bpad);
  if (prof) { prof->end(TRXSIG_K_TSC_CORR, st); prof->begin(TRXSIG_K_TSC_PEAK, st); }
  const trx_c32 g = hT->mid_gain[tsc];                     // gain.inv() (Complex.h:154-160), as launch_tsc_detect
  const float n = g.i * g.i + g.r * g.r;
  trx_c32 ginv; ginv.r = g.r / n; ginv.i = -g.i / n;
  k_tsc_peak2<S><<<dim3((B + kPeak2Threads / 2 - 1) / (kPeak2Threads / 2)), dim3(kPeak2Threads), 0, st>>>(dT, rec, Bpad, B, ginv, hT->mid_toa[tsc], detect_thresh, energy_thresh,
                                                              flags, amp, toa, avgpwr);
  if (prof) prof->end(TRXSIG_K_TSC_PEAK, st);
  if (nsoft > 0) {
    if (prof) prof->begin(TRXSIG_K_DEMOD, st);
    if (soft_tolerance)
      k_demod_rx<S, true><<<dim3((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), dim3(64 * TRX_DEMOD_WAVES), 0, st>>>(
          dT, gen, B, amp, toa, flags, TRXSIG_F_DETECT, soft, hard, nsoft, stride);
    else
      k_demod_rx<S><<<dim3((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), dim3(64 * TRX_DEMOD_WAVES), 0, st>>>(
          dT, gen, B, amp, toa, flags, TRXSIG_F_DETECT, soft, hard, nsoft, stride);
    if (prof) prof->end(TRXSIG_K_DEMOD, st);
  }
  return hipGetLastError();
}

hipError_t trx_launch_rx_demod(hipStream_t st, const TrxTables *dT, const TrxRxGen &gen, int B, const trx_c32 *amp, const float *toa,
                               const uint8_t *flags, int need_mask, float *soft, uint8_t *hard, int nsoft, int stride, TrxProfiler *prof,
                               int soft_tolerance) {
  if (B <= 0 || nsoft <= 0) return hipSuccess;
  if (nsoft > 148 || gen.nb <= 0) return hipErrorInvalidValue;
  if (prof) prof->begin(TRXSIG_K_DEMOD, st);
  if (soft_tolerance)
    k_demod_rx<4, true><<<dim3((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), dim3(64 * TRX_DEMOD_WAVES), 0, st>>>(dT, gen, B, amp, toa, flags,
                                                                                                                  need_mask, soft, hard, nsoft, stride);
  else
    k_demod_rx<4><<<dim3((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), dim3(64 * TRX_DEMOD_WAVES), 0, st>>>(dT, gen, B, amp, toa, flags, need_mask,
                                                                                                            soft, hard, nsoft, stride);
  if (prof) prof->end(TRXSIG_K_DEMOD, st);
  return hipGetLastError();
}

hipError_t trx_launch_demod(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                            const int32_t *off, const int32_t *len, int B, const trx_c32 *amp,
                            const float *toa, const uint8_t *flags, int need_mask, float *soft,
                            uint8_t *hard, int nsoft, int stride, TrxProfiler *prof, int soft_tolerance) {
  if (B <= 0) return hipSuccess;
  const dim3 grid((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), block(64 * TRX_DEMOD_WAVES);
  if (prof) prof->begin(TRXSIG_K_DEMOD, st);
#define TRX_DEMOD_CASE(S)                                                                                          \
  case S:                                                                                                          \
    if (nsoft <= 148 && soft_tolerance)                                                                            \
      k_demod<S, false, 148, SmpC32, true><<<grid, block, 0, st>>>(dT, samples, off, len, B, amp, toa, flags, need_mask, soft, hard, \
                                                     nsoft, stride);                                              \
    else if (nsoft <= 148)                                                                                         \
      k_demod<S, false, 148><<<grid, block, 0, st>>>(dT, samples, off, len, B, amp, toa, flags, need_mask, soft, hard, \
                                                     nsoft, stride);                                              \
    else                                                                                                           \
      k_demod<S, false, 157><<<grid, block, 0, st>>>(dT, samples, off, len, B, amp, toa, flags, need_mask, soft, hard, \
                                                     nsoft, stride);                                              \
    break;
  switch (sps) {
    TRX_DEMOD_CASE(1)
    TRX_DEMOD_CASE(2)
    TRX_DEMOD_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef TRX_DEMOD_CASE
  if (prof) prof->end(TRXSIG_K_DEMOD, st);
  return hipGetLastError();
}

